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

namespace ba {

constexpr int kBcrThreads = 256;
constexpr int kBcrMaxHB = 10;                  // 4 matrices of B x (B+1) doubles must fit in LDS

__host__ __device__ inline size_t bcr_lds_bytes(int B) { return ((size_t)4 * B * (B + 1) + 4 * B + 8) * sizeof(double); }

// band (+ mask) -> D[N][B][B], U[N][B][B] = T[I,I+1], f[N][B]; cameras past nco and masked
// parameters become identity rows with zero right-hand side.
__global__ __launch_bounds__(kBcrThreads) void k_bcr_assemble(int nco, int hb, const double* __restrict__ S,
                                                              const double* __restrict__ b,
                                                              const unsigned char* __restrict__ mask,
                                                              double* __restrict__ Dm, double* __restrict__ Um,
                                                              double* __restrict__ fm) {
  const int B = 6 * hb, hb1 = hb + 1;
  const int I = blockIdx.x;
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
__global__ __launch_bounds__(kBcrThreads) void k_bcr_eliminate(int N, int s, double* __restrict__ Dm,
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

  if (tid == 0) *bad = 0;
  for (int e = tid; e < B * B; e += kBcrThreads) {
    const int rr = e / B, cc = e - rr * B;
    G[rr * ld + cc] = Dm[(size_t)i * BB + e];
    Pl[cc * ld + rr] = haveL ? Um[(size_t)l * BB + e] : 0.0;     // T[i,l] = T[l,i]^T
    Ql[rr * ld + cc] = haveR ? Um[(size_t)i * BB + e] : 0.0;     // T[i,r]
    Xi[rr * ld + cc] = rr == cc ? 1.0 : 0.0;
  }
  for (int e = tid; e < B; e += kBcrThreads) g[e] = fm[(size_t)i * B + e];
  __syncthreads();

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
      for (int i2 = k0 + 6 + tid; i2 < B; i2 += kBcrThreads) {
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
      for (int t = tid; t < ntile; t += kBcrThreads) {
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

  // ---- forward substitution L Y = R, one thread per right-hand-side column, the column kept
  //      in registers: P (B columns), Q (B columns), G^-1 (B columns), g (1 column)
  {
    constexpr int ncol = 3 * B + 1;
    for (int c = tid; c < ncol; c += kBcrThreads) {
      double* X = c < B ? Pl + c : c < 2 * B ? Ql + (c - B) : c < 3 * B ? Xi + (c - 2 * B) : g;
      const int st = c < 3 * B ? ld : 1;
      double y[B];
#pragma unroll
      for (int i2 = 0; i2 < B; ++i2) y[i2] = X[i2 * st];
#pragma unroll
      for (int i2 = 0; i2 < B; ++i2) {
        double a0 = y[i2], a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < i2; ++k2) {
          const double lv = G[i2 * ld + k2];                  // same address in every lane: one LDS broadcast
          if ((k2 & 3) == 0) a0 -= lv * y[k2];
          else if ((k2 & 3) == 1) a1 -= lv * y[k2];
          else if ((k2 & 3) == 2) a2 -= lv * y[k2];
          else a3 -= lv * y[k2];
        }
        y[i2] = ((a0 + a1) + (a2 + a3)) * dinv[i2];
      }
#pragma unroll
      for (int i2 = 0; i2 < B; ++i2) X[i2 * st] = y[i2];
    }
  }
  __syncthreads();

  // ---- neighbour updates: 3x3 register tiles of P^T P, Q^T Q, P^T Q; P^T g, Q^T g
  {
    constexpr int T = B / 3, TT = T * T;
    for (int task = tid; task < 3 * TT; task += kBcrThreads) {
      const int which = task / TT, t2 = task - which * TT;
      if ((which == 0 && !haveL) || (which == 1 && !haveR) || (which == 2 && !(haveL && haveR))) continue;
      const int i0 = 3 * (t2 / T), j0 = 3 * (t2 % T);
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
    for (int c = tid; c < 2 * B; c += kBcrThreads) {
      const bool left = c < B;
      if ((left && !haveL) || (!left && !haveR)) continue;
      const double* A = left ? Pl + c : Ql + (c - B);
      double acc = 0.0;
      for (int k = 0; k < B; ++k) acc += A[k * ld] * g[k];
      atomic_add_f64(fm + (size_t)(left ? l : r) * B + (left ? c : c - B), -acc);
    }
  }
  // ---- keep what the back-substitution needs
  for (int e = tid; e < B * B; e += kBcrThreads) {
    const int rr = e / B, cc = e - rr * B;
    Pm[(size_t)i * BB + e] = Pl[rr * ld + cc];
    Qm[(size_t)i * BB + e] = Ql[rr * ld + cc];
    Gi[(size_t)i * BB + e] = cc <= rr ? Xi[rr * ld + cc] : 0.0;
  }
  __syncthreads();                       // the products above read g; only now overwrite fm[i]
  for (int e = tid; e < B; e += kBcrThreads) fm[(size_t)i * B + e] = g[e];
}

// One back-substitution level: x_i = G^-T (g - P x_l - Q x_r) for the nodes of that level.
__global__ __launch_bounds__(kBcrThreads) void k_bcr_backsolve(int N, int B, int s, const double* __restrict__ fm,
                                                               const double* __restrict__ Pm,
                                                               const double* __restrict__ Qm,
                                                               const double* __restrict__ Gi,
                                                               double* __restrict__ x) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int ld = B + 1;
  double* M = sm;                        // [B][ld] staging of P, Q, then G^-1
  double* w = M + (size_t)B * ld;        // [B]
  double* xn = w + B;                    // [B] neighbour solution
  const int tid = threadIdx.x;
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N) return;
  const int l = i - s, r = i + s;
  const size_t BB = (size_t)B * B;
  for (int e = tid; e < B; e += kBcrThreads) w[e] = fm[(size_t)i * B + e];
  for (int side = 0; side < 2; ++side) {
    const int nb = side == 0 ? l : r;
    if (nb < 0 || nb >= N) continue;                 // uniform
    const double* src = (side == 0 ? Pm : Qm) + (size_t)i * BB;
    __syncthreads();
    for (int e = tid; e < B * B; e += kBcrThreads) M[(e / B) * ld + e % B] = src[e];
    for (int e = tid; e < B; e += kBcrThreads) xn[e] = x[(size_t)nb * B + e];
    __syncthreads();
    for (int k = tid; k < B; k += kBcrThreads) {
      double acc = 0.0;
      for (int c = 0; c < B; ++c) acc += M[k * ld + c] * xn[c];
      w[k] -= acc;
    }
  }
  __syncthreads();
  for (int e = tid; e < B * B; e += kBcrThreads) M[(e / B) * ld + e % B] = Gi[(size_t)i * BB + e];
  __syncthreads();
  for (int m = tid; m < B; m += kBcrThreads) {       // x = (G^-1)^T w : column m of the lower-triangular G^-1
    double acc = 0.0;
    for (int k = m; k < B; ++k) acc += M[k * ld + m] * w[k];
    x[(size_t)i * B + m] = acc;
  }
}

}  // namespace ba
