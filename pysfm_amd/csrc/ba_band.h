// ba_band.h - the reduced camera system as ONE workgroup sees it: k_flatten (band -> dense masked matrix) and
// k_band_solve (left-looking block Cholesky down the band).  gfx950 (MI355X, CDNA4).
#pragma once

#include "ba_device.h"

namespace ba {

#ifndef BA_BAND_TEMPLATES_ONLY      // (the non-template kernel: compiled by ba_band_solve.hip alone)
// --------------------------------------------------------------------------
// solve_motion_normal_eqns, the flatten + mask step (bundle_adjuster.py:290-299):
// A[r,c] = S.transpose(0,2,1,3).reshape(6nco,6nco)[keep[r], keep[c]] from the block band.
// Used when the band is too wide for k_band_solve (dense LU on the GPU instead).
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_flatten(int nco, int hb, int nkeep, const int* __restrict__ keep,
                                                    const double* __restrict__ S,
                                                    const double* __restrict__ b, double* __restrict__ Aout,
                                                    double* __restrict__ rhs) {
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (tid >= (long long)nkeep * nkeep) return;
  const int r = (int)(tid / nkeep), c = (int)(tid % nkeep);
  const int p = keep[r], q = keep[c];
  const int i = p / 6, a = p % 6, j = q / 6, d = q % 6;
  double v = 0.0;
  if (i <= j) {
    if (j - i <= hb) v = S[band_block(i, j, hb + 1) + a * 6 + d];
  } else if (i - j <= hb) {
    v = S[band_block(j, i, hb + 1) + d * 6 + a];
  }
  Aout[tid] = v;
  if (c == 0) rhs[r] = b[p];
}
#endif

// --------------------------------------------------------------------------
// solve_motion_normal_eqns on the device (bundle_adjuster.py:281-312) for a block-banded
// reduced system: S x = b by block Cholesky S = U^T U, forward and backward
// substitution, all in ONE workgroup that slides an LDS window of the last hb block rows
// of U down the band (left-looking):
//   row j:  B[d] = S[j,j+d] - sum_{m=1..hb} U[j-m,j]^T U[j-m,j+d]      (d = 0..hb)
//           U[j,j] = chol(B[0]);  U[j,j+d] = U[j,j]^-T B[d];  y_j likewise from b
//   then    x_j = U[j,j]^-1 (y_j - sum_d U[j,j+d] x_{j+d})  for j = nco-1 .. 0.
// Masked camera parameters (param_mask) become identity rows/columns with zero rhs,
// which deletes them from the system exactly as the reference's row/column deletion
// does and leaves x = 0 there.  S is SPD whenever the reference's LU solve is
// meaningful; a non-positive pivot is reported through *info (caller falls back to
// the dense LU path, which reproduces the reference's LinAlgError semantics).
// LDS: hb*(hb+1)*288 B ring + one row; hb <= kMaxBandSolve.
// --------------------------------------------------------------------------




// Pipelined left-looking schedule (HB = block half-bandwidth, compile-time so that the
// sums over the HB previous rows are fully unrolled and their LDS loads batched):
//   phase A  wavefront 0: chol + panel of row j (the serial critical path)
//            wavefronts 1..7: row j+1 minus the contributions of rows j+1-HB .. j-1
//   phase B  all: row j+1 minus the contribution of row j (which phase A just produced)
// so the O(HB^2) update of the next row hides behind the serial 6x6 factorisation.
template <int HB, bool MASKED>
__global__ __launch_bounds__(kSolveThreads) void k_band_solve(int nco, int ch, const double* __restrict__ S,
                                                              const double* __restrict__ b,
                                                              const unsigned char* __restrict__ mask,
                                                              double* __restrict__ U, double* __restrict__ y,
                                                              double* __restrict__ dinvg, double* __restrict__ x,
                                                              int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int HB1 = HB + 1, ROWLEN = HB1 * 36, NTASK = ROWLEN + 6, HBM = HB > 0 ? HB : 1;
  constexpr int RS = band_ring_stride(HB);            // padded LDS stride of a ring row
  const int tid = threadIdx.x;
  double* ring = sm;                                  // [HBM][RS]       rows j-HB .. j-1 of U (zero = no row)
  double* yring = ring + (size_t)HBM * RS;            // [HBM][6]
  double* xring = yring + HBM * 6;                    // [HBM][6]        (backward pass)
  double* Bbuf = xring + HBM * 6;                     // [2][NTASK]      row being factored / row being built
  double* part = Bbuf + 2 * NTASK;                    // [16][6]         partial sums (backward pass)
  int* bad = reinterpret_cast<int*>(part + 16 * 6);
  double* stage = part + 16 * 6 + 8;                  // [ch][ROWLEN]    chunk of S (forward) / U (backward)
  double* bstage = stage + (size_t)ch * ROWLEN;       // [ch][6]         chunk of b / y
  double* dstage = bstage + (size_t)ch * 6;           // [ch][6]         chunk of 1/diag (backward)
  unsigned char* mstage = reinterpret_cast<unsigned char*>(dstage + (size_t)ch * 6);   // [(ch+HB)*6]
  if (tid == 0) *bad = 0;
  for (int i = tid; i < HBM * RS + HBM * 6; i += kSolveThreads) ring[i] = 0.0;         // ring + yring
  long long t_c0 = 0, t_w0 = 0;
  if (tid == 0) { t_c0 = clock64(); t_w0 = wall_clock64(); }

  // stage `rows` contiguous band rows of S, b and the mask starting at row j0
  auto stage_chunk = [&](int j0) {
    const int rows = min(ch, nco - j0);
    copy_to_lds<kSolveThreads>(stage, S + (size_t)j0 * ROWLEN, rows * ROWLEN, tid);
    for (int i = tid; i < rows * 6; i += kSolveThreads) bstage[i] = b[(size_t)j0 * 6 + i];
    if (MASKED) {
      const int mc = (rows + HB) * 6;
      for (int i = tid; i < mc; i += kSolveThreads) mstage[i] = (j0 * 6 + i < nco * 6) ? mask[j0 * 6 + i] : 1;
    }
  };
  // entry tk of band row j (jj = j - chunk start), masked parameters replaced by identity rows
  auto staged = [&](int jj, int tk) -> double {
    if (tk < ROWLEN) {
      const int d = tk / 36, e = tk % 36, a = e / 6, c = e % 6;
      double v = stage[jj * ROWLEN + tk];
      if (MASKED && (!mstage[jj * 6 + a] || !mstage[(jj + d) * 6 + c])) v = (d == 0 && a == c) ? 1.0 : 0.0;
      return v;
    }
    const int a = tk - ROWLEN;
    return (MASKED && !mstage[jj * 6 + a]) ? 0.0 : bstage[jj * 6 + a];
  };
  // contribution of U row (slot) at distance m to entry tk of the row being built
  auto term = [&](int slot, int m, int tk) -> double {
    const double* row = ring + (size_t)slot * RS;
    double dot = 0.0;
    if (tk < ROWLEN) {
      const int d = tk / 36, e = tk % 36, a = e / 6, c = e % 6;
      const bool ok = m + d <= HB;
      const double* Uj = row + m * 36 + a;                          // U[r, j][:, a]
      const double* Ujd = row + (ok ? m + d : m) * 36 + c;          // U[r, j+d][:, c]
#pragma unroll
      for (int q = 0; q < 6; ++q) dot += Uj[q * 6] * Ujd[q * 6];
      return ok ? dot : 0.0;
    }
    const int a = tk - ROWLEN;
    const double* Uj = row + m * 36 + a;
    const double* yr = yring + slot * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) dot += Uj[q * 6] * yr[q];
    return dot;
  };

  stage_chunk(0);
  lds_barrier();
  for (int tk = tid; tk < NTASK; tk += kSolveThreads) Bbuf[tk] = staged(0, tk);   // row 0 has no predecessors
  int cur = 0;                                        // Bbuf[cur] = row j, Bbuf[1-cur] = row j+1
  int jslot = 0;                                      // j % HB
  int chunk0 = 0;                                     // first row of the staged chunk
  for (int j = 0; j < nco; ++j) {
    if (j + 1 < nco && j + 1 == chunk0 + ch) {        // row j+1 opens the next chunk: stage it now
      lds_barrier();
      if (*bad) {                                     // uniform (every thread reads the same LDS word);
        if (tid == 0) *info = *bad;                   // a failed pivot only produces NaNs until here
        return;
      }
      chunk0 = j + 1;
      stage_chunk(chunk0);
    }
    lds_barrier();                                    // Bbuf[cur], ring rows <= j-1 and the staged chunk are visible
    double* Brow = Bbuf + cur * NTASK;
    double* Bnext = Bbuf + (1 - cur) * NTASK;
    const int nslot = (HB > 0 && jslot + 1 == HB) ? 0 : jslot + 1;     // (j+1) % HB
    if (tid >= 64) {
      // ---- phase A, wavefronts 1..7: row j+1 from S minus the rows j+1-HB .. j-1 (m = 2..HB).
      // A group of G adjacent lanes shares one 3x3 sub-block of one 6x6 block: lane g of
      // the group owns the term m = g + 2 (36 LDS loads feed 54 FMAs, all loads in one
      // batch), then the G partial 3x3 blocks are summed with cross-lane shuffles.
      if (j + 1 < nco) {
        constexpr int NT = HB > 1 ? HB - 1 : 1;                          // number of terms m = 2..HB
        constexpr int G = NT <= 1 ? 1 : NT <= 2 ? 2 : NT <= 4 ? 4 : NT <= 8 ? 8 : NT <= 16 ? 16 : 32;
        const int jj1 = j + 1 - chunk0;
        for (int task = tid - 64; task < HB1 * 4 * G; task += kSolveThreads - 64) {
          const int g = task % G, blk = task / G;
          const int d = blk >> 2, a0 = (blk & 2) ? 3 : 0, c0 = (blk & 1) ? 3 : 0;
          const int m = g + 2;
          double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          if (HB >= 2 && m <= HB && m + d <= HB) {
            int slot = nslot - m;
            if (slot < 0) slot += HB;
            const double* row = ring + (size_t)slot * RS;
            const double* Uj = row + m * 36 + a0;                         // U[r, j+1][q][a0 .. a0+2]
            const double* Ujd = row + (m + d) * 36 + c0;                  // U[r, j+1+d][q][c0 .. c0+2]
            double ua[18], uc[18];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
#pragma unroll
              for (int i = 0; i < 3; ++i) { ua[q * 3 + i] = Uj[q * 6 + i]; uc[q * 3 + i] = Ujd[q * 6 + i]; }
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
#pragma unroll
              for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[i * 3 + k] += ua[q * 3 + i] * uc[q * 3 + k];
              }
            }
          }
#pragma unroll
          for (int i = 0; i < 9; ++i) acc[i] = group_sum<G>(acc[i]);
          if (g == 0) {
            double sv[9];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
              for (int k = 0; k < 3; ++k) sv[i * 3 + k] = staged(jj1, d * 36 + (a0 + i) * 6 + c0 + k);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
              for (int k = 0; k < 3; ++k) Bnext[d * 36 + (a0 + i) * 6 + c0 + k] = sv[i * 3 + k] - acc[i * 3 + k];
            }
          }
        }
        // right-hand side of row j+1: 6 entries, same split over m
        for (int task = tid - 64; task < 6 * G; task += kSolveThreads - 64) {
          const int g = task % G, a = task / G;
          const int m = g + 2;
          double acc = 0.0;
          if (HB >= 2 && m <= HB) {
            int slot = nslot - m;
            if (slot < 0) slot += HB;
            const double* Uj = ring + (size_t)slot * RS + m * 36 + a;
            const double* yr = yring + slot * 6;
#pragma unroll
            for (int q = 0; q < 6; ++q) acc += Uj[q * 6] * yr[q];
          }
          acc = group_sum<G>(acc);
          if (g == 0) Bnext[ROWLEN + a] = staged(jj1, ROWLEN + a) - acc;
        }
      }
    } else {
      // ---- phase A, wavefront 0: U[j,j] = chol(B[0]) (lane c owns column c), then the panel
      //      U[j,j+d] = U[j,j]^-T B[d], y_j = U[j,j]^-T rhs, and the stores of row j
      const int c = tid < 6 ? tid : 5;
      double col[6];
#pragma unroll
      for (int p = 0; p < 6; ++p) col[p] = Brow[p * 6 + c];
      int fail = 0;
      double dinv = 0.0;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const double piv = lane_bcast(col[q], q);
        if (!(piv > 0.0) && !fail) fail = q + 1;
        const double inv = rsqrt_nr(piv);
        const double uqq = piv * inv;
        if (c == q) dinv = inv;
        const double uqc = c == q ? uqq : (c > q ? col[q] * inv : 0.0);
        col[q] = uqc;
#pragma unroll
        for (int p = q + 1; p < 6; ++p) {
          const double uqp = lane_bcast(uqc, p);
          if (p <= c) col[p] -= uqp * uqc;
        }
      }
      if (fail && tid == 0) *bad = 6 * j + fail;
      // every lane needs the factor: Uf[p][q] (p < q) and 1/U[q][q], broadcast from lane q
      double Uf[15], di[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        di[q] = lane_bcast(dinv, q);
#pragma unroll
        for (int p = 0; p < q; ++p) Uf[q * (q - 1) / 2 + p] = lane_bcast(col[p], q);
      }
      if (tid < 6) {                                  // diagonal block of row j (upper triangle, zeros below)
        dinvg[6 * (size_t)j + c] = dinv;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
          const double v = p <= c ? col[p] : 0.0;
          if (HB > 0) ring[(size_t)jslot * RS + p * 6 + c] = v;
          U[(size_t)j * ROWLEN + p * 6 + c] = v;
        }
      }
      for (int tk = tid; tk < HB * 6 + 1; tk += 64) {
        const bool isrhs = tk == HB * 6;
        const int d = isrhs ? 0 : 1 + tk / 6, cc = isrhs ? 0 : tk % 6;
        double v[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = isrhs ? Brow[ROWLEN + q] : Brow[d * 36 + q * 6 + cc];
#pragma unroll
        for (int q = 0; q < 6; ++q) {                  // forward substitution with U[j,j]^T (lower)
          double t = v[q];
#pragma unroll
          for (int p = 0; p < q; ++p) t -= Uf[q * (q - 1) / 2 + p] * v[p];
          v[q] = t * di[q];
        }
        if (isrhs) {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            if (HB > 0) yring[jslot * 6 + q] = v[q];
            y[6 * (size_t)j + q] = v[q];
          }
        } else {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            ring[(size_t)jslot * RS + d * 36 + q * 6 + cc] = v[q];
            U[(size_t)j * ROWLEN + d * 36 + q * 6 + cc] = v[q];
          }
        }
      }
    }
    if (HB > 0 && j + 1 < nco) {
      lds_barrier();                                  // U row j (ring) and the partial row j+1 are visible
      // ---- phase B, all wavefronts: row j+1 minus the contribution of row j (m = 1)
      for (int tk = tid; tk < NTASK; tk += kSolveThreads) Bnext[tk] -= term(jslot, 1, tk);
    }
    cur = 1 - cur;
    jslot = nslot;
  }
  __syncthreads();     // full barrier: U, y (global) and *bad of this workgroup are visible
  if (*bad) {
    if (tid == 0) *info = *bad;
    return;
  }
  if (tid == 0) {      // instrumentation: shader cycles / 100 MHz wall ticks of the forward sweep
    info[2] = (int)(clock64() - t_c0);
    info[3] = (int)(wall_clock64() - t_w0);
  }

  // ---- backward substitution, rows nco-1 .. 0, again in chunks staged through LDS by the
  // whole workgroup; the recurrence itself runs on wavefront 0 alone (no barriers inside a
  // chunk).  lane (a = lane % 6, g = lane / 6) owns row a of blocks d = g, g + 10, g + 20.
  const int a_ = tid % 6, g_ = tid / 6;               // g_ in 0..10 for wavefront 0 (lanes 60..63 idle)
  constexpr int NG = HB1 < 10 ? HB1 : 10;
  int xslot = HB > 0 ? (nco - 1) % HB : 0;            // slot of row j in xring
  for (int jend = nco; jend > 0; jend -= ch) {
    const int jbeg = max(0, jend - ch), rows = jend - jbeg;
    lds_barrier();                                    // wavefront 0 is done with the previous chunk
    copy_to_lds<kSolveThreads>(stage, U + (size_t)jbeg * ROWLEN, rows * ROWLEN, tid);
    for (int i = tid; i < rows * 6; i += kSolveThreads) {
      bstage[i] = y[(size_t)jbeg * 6 + i];
      dstage[i] = dinvg[(size_t)jbeg * 6 + i];
    }
    lds_barrier();
    if (tid >= 64) continue;
    for (int jj = rows - 1; jj >= 0; --jj) {
      const int j = jbeg + jj;
      const double* urow = stage + (size_t)jj * ROWLEN;
      if (g_ < NG) {
        double sacc = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int d = g_ + 10 * r;
          if (d >= 1 && d <= HB && j + d < nco) {
            int sl = xslot + d;
            if (sl >= HB) sl -= HB;
            const double* xr = xring + sl * 6;
            const double* ur = urow + d * 36 + a_ * 6;
#pragma unroll
            for (int c = 0; c < 6; ++c) sacc += ur[c] * xr[c];
          }
        }
        part[g_ * 6 + a_] = sacc;
      }
      // lanes 0..5 fetch what the second half needs while the partial sums land
      double t = 0.0, dv = 1.0, ud[6] = {0, 0, 0, 0, 0, 0};
      if (g_ == 0) {
        t = bstage[jj * 6 + a_];
        dv = dstage[jj * 6 + a_];
#pragma unroll
        for (int c = 0; c < 6; ++c) ud[c] = urow[a_ * 6 + c];
      }
      lds_wave_sync();
      if (g_ == 0) {
        double pg[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) pg[g] = part[g * 6 + a_];
#pragma unroll
        for (int g = 0; g < NG; ++g) t -= pg[g];
      }
      double xs = 0.0;
#pragma unroll
      for (int q = 5; q >= 0; --q) {
        const double xq = lane_bcast(t * dv, q);        // x_q, final once rows > q were eliminated
        if (a_ == q) xs = xq;
        if (a_ < q) t -= ud[q] * xq;
      }
      if (g_ == 0) {
        if (HB > 0) xring[xslot * 6 + a_] = xs;
        x[6 * (size_t)j + a_] = xs;
      }
      lds_wave_sync();
      xslot = xslot == 0 ? (HB > 0 ? HB - 1 : 0) : xslot - 1;
    }
  }
  if (tid == 0) {
    info[4] = (int)(clock64() - t_c0);
    info[5] = (int)(wall_clock64() - t_w0);
    *info = 0;
  }
}

}  // namespace ba
