// ba_bcr.h - parallel solve of the block-banded reduced camera system by block cyclic
// reduction (BCR), the multi-CU path of solve_motion_normal_eqns (bundle_adjuster.py:281-312).
//
// k_band_solve walks the band with ONE workgroup: nco dependent 6x6 pivots, a few
// microseconds each.  Here the cameras are grouped into super-blocks of hb cameras
// (B = 6*hb unknowns); because the band half-width is hb, the system is block
// TRIDIAGONAL in super-blocks:  T[I,I] = D_I,  T[I,I+1] = U_I.  Cyclic reduction then
// eliminates every other super-block in parallel, one workgroup per eliminated node,
// log2(N) levels deep:
//
//   level with stride s: node i (neighbours l = i-s, r = i+s):
//     D_i = G G^T (Cholesky),  P = G^-1 T[i,l],  Q = G^-1 T[i,r],  g = G^-1 f_i
//     D_l -= P^T P,  D_r -= Q^T Q,  T[l,r] = -P^T Q,  f_l -= P^T g,  f_r -= Q^T g
//   back-substitution, levels in reverse:  x_i = G^-T (g - P x_l - Q x_r)
//
// Every Schur complement of an SPD matrix is SPD, so this is the same arithmetic as a
// Cholesky factorisation in nested-dissection order: results agree with k_band_solve and
// with the reference's LU to round-off.  A non-positive pivot is reported through *info.
// All fp64, all blocks dense B x B (B <= 60) in LDS; no MFMA.
#pragma once

#include "ba_kernels.h"

// Build with -DBA_BCR_PROFILE (make PROFILE=1) to have node 2 of the first level write
// its per-phase shader-cycle counts to info[8..12] (printed under BA_SOLVE_TRACE=1).
#ifdef BA_BCR_PROFILE
#define BA_STAMP(var) const long long var = clock64()
#else
#define BA_STAMP(var)
#endif

namespace ba {

constexpr int kBcrThreads = 256;                 // assemble
constexpr int kBcrElimThreads = 1024;           // eliminate / backsolve: 16 wavefronts per node
constexpr int kBcrMaxHB = 10;                  // 4 matrices of B x (B+1) doubles must fit in LDS

__host__ __device__ inline size_t bcr_lds_bytes(int B) { return ((size_t)4 * B * (B + 1) + 4 * B + 8) * sizeof(double); }

// band (+ mask) -> D[N][B][B], U[N][B][B] = T[I,I+1], f[N][B]; cameras past nco and masked
// parameters become identity rows with zero right-hand side.
__global__ __launch_bounds__(kBcrThreads) void k_bcr_assemble(int nco, int hb, const double* __restrict__ S,
                                                              const double* __restrict__ b,
                                                              const unsigned char* __restrict__ mask,
                                                              double* __restrict__ Dm, double* __restrict__ Um,
                                                              double* __restrict__ fm, int* __restrict__ info) {
  const int B = 6 * hb, hb1 = hb + 1;
  const int I = blockIdx.x;
  if (I == 0 && threadIdx.x == 0) *info = 0;      // status word of this solve (the eliminate levels only ever set it)
  for (int e = threadIdx.x; e < B * B; e += kBcrThreads) {
    const int r = e / B, c = e - r * B;
    const int i = I * hb + r / 6, j = I * hb + c / 6, a = r % 6, bb = c % 6;
    double v;
    if (i >= nco || j >= nco) {
      v = r == c ? 1.0 : 0.0;
    } else {
      if (i == j) {
        const double* blk = S + band_block(i, i, hb1);                 // diagonal block: use its upper triangle
        v = a <= bb ? blk[a * 6 + bb] : blk[bb * 6 + a];
      } else if (i < j) {
        v = S[band_block(i, j, hb1) + a * 6 + bb];
      } else {
        v = S[band_block(j, i, hb1) + bb * 6 + a];
      }
      if (mask && (!mask[6 * i + a] || !mask[6 * j + bb])) v = (r == c) ? 1.0 : 0.0;
    }
    Dm[(size_t)I * B * B + e] = v;
    const int j2 = j + hb;
    double v2 = 0.0;
    if (i < nco && j2 < nco && j2 - i <= hb) {
      v2 = S[band_block(i, j2, hb1) + a * 6 + bb];
      if (mask && (!mask[6 * i + a] || !mask[6 * j2 + bb])) v2 = 0.0;
    }
    Um[(size_t)I * B * B + e] = v2;
  }
  for (int r = threadIdx.x; r < B; r += kBcrThreads) {
    const int i = I * hb + r / 6, a = r % 6;
    fm[(size_t)I * B + r] = (i < nco && (!mask || mask[6 * i + a])) ? b[6 * (size_t)i + a] : 0.0;
  }
}

// One elimination level.  blockIdx.x = k-th node of this level: i = s*(2k+1) - 1.
// Out: Gi[i] = G^-1 (lower triangular), Pm[i] = P, Qm[i] = Q, fm[i] = g; neighbours updated.
// HB is a template parameter so that the dense B x B (B = 6 HB) pieces unroll: a blocked
// (6-wide) right-looking Cholesky whose 6x6 diagonal factor runs on one wavefront with
// v_readlane broadcasts, and a forward substitution that keeps a whole solution column in
// registers (one thread per right-hand side, L read from LDS as broadcasts).
template <int HB>
__global__ __launch_bounds__(kBcrElimThreads) void k_bcr_eliminate(int N, int s, double* __restrict__ Dm,
                                                               double* __restrict__ Um, double* __restrict__ fm,
                                                               double* __restrict__ Pm, double* __restrict__ Qm,
                                                               double* __restrict__ Gi, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int B = 6 * HB, ld = B + 1;
  double* G = sm;                       // [B][ld]  D_i -> its Cholesky factor L (lower)
  double* Pl = G + (size_t)B * ld;      // [B][ld]  T[i,l] -> P
  double* Ql = Pl + (size_t)B * ld;     // [B][ld]  T[i,r] -> Q
  double* Xi = Ql + (size_t)B * ld;     // [B][ld]  identity -> G^-1
  double* g = Xi + (size_t)B * ld;      // [B]
  double* dinv = g + B;                 // [B]
  int* bad = reinterpret_cast<int*>(dinv + B + 2);
  const int tid = threadIdx.x;
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N) return;
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  constexpr size_t BB = (size_t)B * B;

  BA_STAMP(t0);
  if (tid == 0) *bad = 0;
  {
    // all global loads of a thread are issued before the first LDS store (one round trip)
    constexpr int NIT = (B * B + kBcrElimThreads - 1) / kBcrElimThreads;
    double vd[NIT], vp[NIT], vq[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * kBcrElimThreads;
      const bool ok = e < B * B;
      vd[it] = ok ? Dm[(size_t)i * BB + e] : 0.0;
      vp[it] = (ok && haveL) ? Um[(size_t)l * BB + e] : 0.0;
      vq[it] = (ok && haveR) ? Um[(size_t)i * BB + e] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * kBcrElimThreads;
      if (e < B * B) {
        const int rr = e / B, cc = e - rr * B;
        G[rr * ld + cc] = vd[it];
        Pl[cc * ld + rr] = vp[it];                                // T[i,l] = T[l,i]^T
        Ql[rr * ld + cc] = vq[it];                                // T[i,r]
        Xi[rr * ld + cc] = rr == cc ? 1.0 : 0.0;
      }
    }
  }
  for (int e = tid; e < B; e += kBcrElimThreads) g[e] = fm[(size_t)i * B + e];
  __syncthreads();

  BA_STAMP(t1);
  // ---- blocked Cholesky D_i = L L^T (lower, in place), block size 6
  for (int kb = 0; kb < HB; ++kb) {
    const int k0 = 6 * kb;
    if (tid < 64) {
      // 6x6 diagonal block on wavefront 0: lane c owns column c of the upper factor U (= L^T)
      const int c = tid < 6 ? tid : 5;
      double col[6];
#pragma unroll
      for (int p = 0; p < 6; ++p) col[p] = G[(k0 + c) * ld + k0 + p];      // A[p][c] from the lower triangle
      int fail = 0;
      double di = 0.0;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const double piv = lane_bcast(col[q], q);
        if (!(piv > 0.0) && !fail) fail = q + 1;
        const double inv = rsqrt_nr(piv);
        if (c == q) di = inv;
        const double uqc = c == q ? piv * inv : (c > q ? col[q] * inv : 0.0);
        col[q] = uqc;
#pragma unroll
        for (int p = q + 1; p < 6; ++p) {
          const double uqp = lane_bcast(uqc, p);
          if (p <= c) col[p] -= uqp * uqc;
        }
      }
      if (fail && tid == 0) *bad = k0 + fail;
      if (tid < 6) {
        dinv[k0 + c] = di;
#pragma unroll
        for (int p = 0; p < 6; ++p)
          if (p <= c) G[(k0 + c) * ld + k0 + p] = col[p];                    // L[c][p] = U[p][c]
      }
    }
    __syncthreads();
    // panel: rows below the diagonal block, X = A[i2][k0..k0+5] L_kk^-T (one row per thread)
    {
      double Lk[15], dk[6];
      int idx = 0;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        dk[q] = dinv[k0 + q];
#pragma unroll
        for (int p = 0; p < q; ++p) Lk[idx++] = G[(k0 + q) * ld + k0 + p];
      }
      for (int i2 = k0 + 6 + tid; i2 < B; i2 += kBcrElimThreads) {
        double x[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) x[q] = G[i2 * ld + k0 + q];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          double t = x[q];
#pragma unroll
          for (int p = 0; p < q; ++p) t -= x[p] * Lk[q * (q - 1) / 2 + p];
          x[q] = t * dk[q];
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) G[i2 * ld + k0 + q] = x[q];
      }
    }
    __syncthreads();
    // trailing update in 3x3 tiles of the lower triangle: A[i][j] -= sum_q X[i][q] X[j][q]
    {
      const int n3 = (B - k0 - 6) / 3;                       // tiles per side
      const int ntile = n3 * (n3 + 1) / 2;
      for (int t = tid; t < ntile; t += kBcrElimThreads) {
        int ti, tj;
        tri_decode(t, n3, tj, ti);                           // tj <= ti
        const int i0 = k0 + 6 + 3 * ti, j0 = k0 + 6 + 3 * tj;
        double xi[18], xj[18], a[9];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
          for (int q = 0; q < 6; ++q) { xi[u * 6 + q] = G[(i0 + u) * ld + k0 + q]; xj[u * 6 + q] = G[(j0 + u) * ld + k0 + q]; }
#pragma unroll
          for (int v = 0; v < 3; ++v) a[u * 3 + v] = G[(i0 + u) * ld + j0 + v];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
          for (int v = 0; v < 3; ++v) {
            double acc = a[u * 3 + v];
#pragma unroll
            for (int q = 0; q < 6; ++q) acc -= xi[u * 6 + q] * xj[v * 6 + q];
            a[u * 3 + v] = acc;
          }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
          for (int v = 0; v < 3; ++v) G[(i0 + u) * ld + j0 + v] = a[u * 3 + v];   // (the strict upper part of diagonal tiles is never read)
        }
      }
    }
    __syncthreads();
  }
  if (*bad) {
    if (tid == 0) atomicMax(info, i * B + *bad);
    return;
  }

  BA_STAMP(t2);
  // ---- forward substitution L Y = R for 3B+1 right-hand sides: P (B columns), Q (B), G^-1
  //      (B), g (1).  FOUR lanes per column: lane q of the quad owns the entries k = q mod 4
  //      of the solution (registers) and the matching quarter of every dot product; the
  //      four partial sums meet through DPP (no LDS traffic on the dependent chain).
  {
    constexpr int ncol = 3 * B + 1;
    constexpr int NQ = (B + 3) / 4;
    for (int task = tid; task < ncol * 4; task += kBcrElimThreads) {
      const int c = task >> 2, q4 = task & 3;
      double* X = c < B ? Pl + c : c < 2 * B ? Ql + (c - B) : c < 3 * B ? Xi + (c - 2 * B) : g;
      const int st = c < 3 * B ? ld : 1;
      double y[NQ];                                         // y[m] = solution entry 4m + q4
      // software pipeline: the quarter-row of L, the right-hand side and 1/diag of row i2+1
      // are fetched from LDS while row i2 runs its dependent chain (FMA -> DPP -> scale)
      double gn[NQ], xn = X[0], dn = dinv[0];
#pragma unroll
      for (int m = 0; m < NQ; ++m) gn[m] = 0.0;
#pragma unroll
      for (int i2 = 0; i2 < B; ++i2) {
        double gc[NQ];
#pragma unroll
        for (int m = 0; m < NQ; ++m) gc[m] = gn[m];
        const double xc = xn, dc = dn;
        if (i2 + 1 < B) {
#pragma unroll
          for (int m = 0; 4 * m < i2 + 1; ++m) {
            const int k2 = 4 * m + q4;
            gn[m] = k2 < i2 + 1 ? G[(i2 + 1) * ld + k2] : 0.0;
          }
          xn = X[(i2 + 1) * st];
          dn = dinv[i2 + 1];
        }
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;       // independent chains: fp64 FMA latency is 16 cycles
#pragma unroll
        for (int m = 0; 4 * m < i2; ++m) {                  // my quarter: k2 = 4m + q4 < i2 (gc is 0 beyond)
          const double t = gc[m] * y[m];
          if ((m & 3) == 0) a0 -= t;
          else if ((m & 3) == 1) a1 -= t;
          else if ((m & 3) == 2) a2 -= t;
          else a3 -= t;
        }
        double acc = (a0 + a1) + (a2 + a3);
        acc += dpp_pair<0xB1>(acc);
        acc += dpp_pair<0x4E>(acc);
        const double yi = (xc + acc) * dc;
        if ((i2 & 3) == q4) y[i2 >> 2] = yi;
        if (q4 == 0) X[i2 * st] = yi;                       // rows are final in order: no one reads X[i2] again
      }
    }
  }
  __syncthreads();

  BA_STAMP(t3);
  // ---- neighbour updates: 3x3 register tiles of P^T P, Q^T Q, P^T Q; P^T g, Q^T g
  {
    constexpr int T = B / 3, TT = T * T;
    for (int task = tid; task < 3 * TT; task += kBcrElimThreads) {
      const int which = task / TT, t2 = task - which * TT;
      if ((which == 0 && !haveL) || (which == 1 && !haveR) || (which == 2 && !(haveL && haveR))) continue;
      const int i0 = 3 * (t2 / T), j0 = 3 * (t2 % T);
      if (which < 2 && j0 > i0) continue;                   // D is only ever read in its lower triangle
      const double* A = which == 1 ? Ql : Pl;
      const double* Bm = which == 0 ? Pl : Ql;
      double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 6
      for (int k = 0; k < B; ++k) {
        const double a0 = A[k * ld + i0], a1 = A[k * ld + i0 + 1], a2 = A[k * ld + i0 + 2];
        const double b0 = Bm[k * ld + j0], b1 = Bm[k * ld + j0 + 1], b2 = Bm[k * ld + j0 + 2];
        acc[0] += a0 * b0; acc[1] += a0 * b1; acc[2] += a0 * b2;
        acc[3] += a1 * b0; acc[4] += a1 * b1; acc[5] += a1 * b2;
        acc[6] += a2 * b0; acc[7] += a2 * b1; acc[8] += a2 * b2;
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const size_t off = (size_t)(i0 + u) * B + j0 + v;
          if (which == 0) atomic_add_f64(Dm + (size_t)l * BB + off, -acc[u * 3 + v]);
          else if (which == 1) atomic_add_f64(Dm + (size_t)r * BB + off, -acc[u * 3 + v]);
          else Um[(size_t)l * BB + off] = -acc[u * 3 + v];          // new T[l,r]
        }
      }
    }
    for (int c = tid; c < 2 * B; c += kBcrElimThreads) {
      const bool left = c < B;
      if ((left && !haveL) || (!left && !haveR)) continue;
      const double* A = left ? Pl + c : Ql + (c - B);
      double acc = 0.0;
      for (int k = 0; k < B; ++k) acc += A[k * ld] * g[k];
      atomic_add_f64(fm + (size_t)(left ? l : r) * B + (left ? c : c - B), -acc);
    }
  }
  BA_STAMP(t4);
  // ---- keep what the back-substitution needs
  for (int e = tid; e < B * B; e += kBcrElimThreads) {
    const int rr = e / B, cc = e - rr * B;
    Pm[(size_t)i * BB + e] = Pl[rr * ld + cc];
    Qm[(size_t)i * BB + e] = Ql[rr * ld + cc];
    Gi[(size_t)i * BB + e] = cc <= rr ? Xi[rr * ld + cc] : 0.0;
  }
  __syncthreads();                       // the products above read g; only now overwrite fm[i]
  for (int e = tid; e < B; e += kBcrElimThreads) fm[(size_t)i * B + e] = g[e];
#ifdef BA_BCR_PROFILE
  if (tid == 0 && blockIdx.x == 1 && s == 1) {
    const long long t5 = clock64();
    int* o = info + 8;
    o[0] = (int)(t1 - t0); o[1] = (int)(t2 - t1); o[2] = (int)(t3 - t2); o[3] = (int)(t4 - t3); o[4] = (int)(t5 - t4);
  }
#endif
}

// One back-substitution level: x_i = G^-T (g - P x_l - Q x_r) for the nodes of that level.
// P, Q and G^-1 are staged into LDS in one round trip; the two matrix-vector products use
// four lanes per row, the last one four lanes per column.
__global__ __launch_bounds__(kBcrElimThreads) void k_bcr_backsolve(int N, int B, int s, const double* __restrict__ fm,
                                                                   const double* __restrict__ Pm,
                                                                   const double* __restrict__ Qm,
                                                                   const double* __restrict__ Gi,
                                                                   double* __restrict__ x) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int ld = B + 1;
  double* MP = sm;                       // [B][ld] P
  double* MQ = MP + (size_t)B * ld;      // [B][ld] Q
  double* MG = MQ + (size_t)B * ld;      // [B][ld] G^-1
  double* w = MG + (size_t)B * ld;       // [B]
  double* xl = w + B;                    // [B]
  double* xr = xl + B;                   // [B]
  const int tid = threadIdx.x;
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N) return;
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  const size_t BB = (size_t)B * B;
  for (int e = tid; e < B * B; e += kBcrElimThreads) {
    const int rr = e / B, cc = e - rr * B;
    const double vp = haveL ? Pm[(size_t)i * BB + e] : 0.0;
    const double vq = haveR ? Qm[(size_t)i * BB + e] : 0.0;
    const double vg = Gi[(size_t)i * BB + e];
    MP[rr * ld + cc] = vp; MQ[rr * ld + cc] = vq; MG[rr * ld + cc] = vg;
  }
  for (int e = tid; e < B; e += kBcrElimThreads) {
    w[e] = fm[(size_t)i * B + e];
    xl[e] = haveL ? x[(size_t)l * B + e] : 0.0;
    xr[e] = haveR ? x[(size_t)r * B + e] : 0.0;
  }
  __syncthreads();
  for (int task = tid; task < 4 * B; task += kBcrElimThreads) {      // w -= P xl + Q xr
    const int k = task >> 2, q4 = task & 3;
    double acc = 0.0;
    for (int c = q4; c < B; c += 4) acc += MP[k * ld + c] * xl[c] + MQ[k * ld + c] * xr[c];
    acc += dpp_pair<0xB1>(acc);
    acc += dpp_pair<0x4E>(acc);
    if (q4 == 0) w[k] -= acc;
  }
  __syncthreads();
  for (int task = tid; task < 4 * B; task += kBcrElimThreads) {      // x = (G^-1)^T w
    const int m = task >> 2, q4 = task & 3;
    double acc = 0.0;
    for (int k = m + q4; k < B; k += 4) acc += MG[k * ld + m] * w[k];
    acc += dpp_pair<0xB1>(acc);
    acc += dpp_pair<0x4E>(acc);
    if (q4 == 0) x[(size_t)i * B + m] = acc;
  }
}

}  // namespace ba
