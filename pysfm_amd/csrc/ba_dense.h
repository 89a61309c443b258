// ba_dense.h - dense Cholesky solve of the reduced camera system on the device, for scenes whose band is
// as wide as the matrix (every camera shares tracks with every other one: the reference's own
// data/oleg_synthetic) or wider than the cyclic reduction's LDS blocks (hb > 21).
// Replaces solve_motion_normal_eqns' numpy.linalg.solve (bundle_adjuster.py:281-312) for those scenes;
// a non-positive pivot is reported through *info and the caller falls back to LU (the reference's
// factorisation, with its LinAlgError semantics).
//
// Layout: A is (n+1) x n row-major, n = 6 nco.  Rows 0..n-1 hold the LOWER triangle of S (masked
// parameters: identity rows), row n holds the right-hand side b.  Carrying b as one more ROW of the
// matrix makes the forward substitution part of the factorisation: after the last panel step row n is
// y = L^-1 b.  Right-looking, block columns of 48:
//
//   k_dense_gather     band-stored [S | b] -> A
//   k_dense_panel      one workgroup per 64 panel rows (128: 12 % slower per panel step, measured on the batched use of ba_bcr_big.h); EVERY workgroup factors the 48 x 48 diagonal block
//                      itself in LDS (12 x 12 steps: DPP pivots, one-row-per-lane panel, MFMA update - the
//                      pieces of ba_bcr.h) instead of waiting for another launch to do it once
//   k_dense_update     trailing matrix -= panel panel^T, 64 x 64 tiles, 16 x 16 x 4 fp64 MFMAs from LDS panels
//   k_dense_backsolve  x = L^-T y, one workgroup, w in LDS, block columns right to left
//
// 2 launches per block column: n = 594 (100 cameras) is 25 launches.
#pragma once

#include "ba_bcr_blocks.h"

namespace ba {

constexpr int kDcNB = 48;                      // block column width (a multiple of 12)
#ifndef BA_DC_ROWS
#define BA_DC_ROWS 64
#endif
constexpr int kDcRows = BA_DC_ROWS;            // panel rows per workgroup
constexpr int kDcLd = kDcNB + 1;
constexpr int kDcM = kDcNB + kDcRows + 16;     // LDS rows: diagonal block + panel rows + one tile of slack for the MFMA reads
constexpr int kDcTile = 64;

__host__ __device__ inline size_t dense_panel_lds_bytes() { return ((size_t)kDcM * kDcLd + kDcNB + 8 + 192 + kBcrIdtDoubles) * sizeof(double); }
__host__ __device__ inline size_t dense_backsolve_lds_bytes(int n) {
  return ((size_t)n + kDcNB * kDcLd + kDcNB + 1024 + 8) * sizeof(double);
}

// ---- band storage -> dense lower triangle (+ b as row n); one row per workgroup.  Also clears *info.
__global__ __launch_bounds__(256) void k_dense_gather(int nco, int hb, const double* __restrict__ S, const double* __restrict__ b,
                                                      const unsigned char* __restrict__ mask, double* __restrict__ A,
                                                      int* __restrict__ info) {
  const int n = 6 * nco, r = blockIdx.x;
  if (r == 0 && threadIdx.x == 0) *info = 0;
  double* row = A + (size_t)r * n;
  if (r == n) {
    for (int c = threadIdx.x; c < n; c += 256) row[c] = (!mask || mask[c]) ? b[c] : 0.0;
    return;
  }
  const int i = r / 6, a = r - 6 * i;
  const bool rok = !mask || mask[r];
  for (int c = threadIdx.x; c <= r; c += 256) {
    const int j = c / 6, d = c - 6 * j;                       // j <= i: S[j,i] is stored, S[i,j] = S[j,i]^T
    double v = (i - j <= hb) ? S[band_block(j, i, hb + 1) + d * 6 + a] : 0.0;
    if (!(rok && (!mask || mask[c]))) v = (c == r) ? 1.0 : 0.0;
    row[c] = v;
  }
}

// ---- panel step of block column [k0, k0 + nb): L_kk = chol(A_kk), panel = A[rows, k0..] L_kk^-T
// Band limit: S[r][c] = 0 for r - c > bw, and so is L; the rows of a block column that can be non-zero are
// kn .. rend-1 (rend = min(n, kn + bw)) and the right-hand side row n.  `total` = rend - kn + 1 counts them;
// list index idx -> matrix row:
__device__ __forceinline__ int dense_row(int idx, int total, int kn, int n) { return idx == total - 1 ? n : kn + idx; }

// One workgroup of a panel step (chunk = which kDcRows rows of the list are mine).  fnb > 0: the block column has NOT yet received
// what it owes to the panel before it, P = A[., fk0 .. fk0 + fnb) - the rows this workgroup holds take it here,
// C -= P_rows P_diag^T (k_dense_step: the step's own trailing update runs beside it on the rest of the matrix).
__device__ __forceinline__ void dense_panel_body(int n, int k0, int nb, int total, double* __restrict__ A, int* __restrict__ info,
                                                 int chunk, int fk0, int fnb, double* __restrict__ sm) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  constexpr int ld = kDcLd;
  double* G = sm;                                  // [kDcM][ld]: rows 0..nb-1 = A_kk, rows nb.. = my panel rows
  double* dinv = G + (size_t)kDcM * ld;            // [48]
  int* bad = reinterpret_cast<int*>(dinv + kDcNB);
  double* Li = dinv + kDcNB + 8;                   // [16][12]: inverse of the current 12 x 12 diagonal block, rows 12..15 zero
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int kn = k0 + nb;
  const int q0 = chunk * kDcRows;                  // my first panel row, as an index into the row list
  const int cnt = max(0, min(kDcRows, total - q0));
  const int M = nb + cnt;
  if (tid == 0) *bad = 0;
  if (tid < 192) Li[tid] = 0.0;
  double* Idt = Li + 192;
  bcr_identity_table(Idt, tid);
  if (fnb > 0) {
    // ---- the rows I hold have not yet received the update of the panel before mine: C -= P_rows P_diag^T, P = A[., fk0 .. fk0 + fnb).
    // P's rows (the diagonal block's nb, then mine) go through LDS IN G'S PLACE (lanes along a row: whole cache lines; operands read
    // straight from memory, a lane a row, cost 21 us a step); the tiles of C come from memory into the accumulators, and G is
    // written from them once every wavefront has read its operands.
    __builtin_amdgcn_s_setprio(2);                  // a panel workgroup is the step's critical path: ahead of the update's wavefronts on this unit
    double* X = G;                                  // [M][ld]
    {
      constexpr int U = ((kDcNB + kDcRows) * kDcNB + 1023) / 1024;
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = tid + 1024 * u, row = e / kDcNB, col = e - row * kDcNB;
        const int gr = row < nb ? k0 + row : dense_row(q0 + row - nb, total, kn, n);
        v[u] = (row < M && col < fnb) ? A[(size_t)gr * n + fk0 + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = tid + 1024 * u, row = e / kDcNB, col = e - row * kDcNB;
        if (row < kDcNB + kDcRows) X[row * ld + col] = v[u];
      }
    }
    const int ntr = (M + 15) >> 4, ntc = (nb + 15) >> 4;          // <= 7 x 3 tiles: at most two per wavefront
    mfma_acc hold[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    // (the tiles' own entries first: their loads are in flight while P arrives in LDS)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = wave + 16 * u;
      if (t < ntr * ntc) {
        const int tr = t / ntc, tc = t - tr * ntc, col = 16 * tc + lr;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * tr + lk + 4 * v;
          const bool ok = row < M && col < nb && (row >= nb || col <= row);
          const int gr = row < nb ? k0 + row : dense_row(q0 + (row < M ? row : nb) - nb, total, kn, n);
          hold[u][v] = ok ? A[(size_t)gr * n + k0 + col] : 0.0;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = wave + 16 * u;
      if (t < ntr * ntc) {
        const int tr = t / ntc, tc = t - tr * ntc;
        const double* pa = X + (16 * tr + lr) * ld + lk;
        const double* pb = X + (16 * tc + lr) * ld + lk;
        double av[kDcNB / 4], bv[kDcNB / 4];
#pragma unroll
        for (int q = 0; q < kDcNB / 4; ++q) { av[q] = pa[4 * q]; bv[q] = -pb[4 * q]; }
#pragma unroll
        for (int q = 0; q < kDcNB / 4; ++q) hold[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], hold[u], 0, 0, 0);
      }
    }
    __syncthreads();                                // every operand has been read: G takes X's place
    for (int e = tid; e < kDcM * ld; e += 1024) G[e] = 0.0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = wave + 16 * u;
      if (t < ntr * ntc) {
        const int tr = t / ntc, tc = t - tr * ntc, col = 16 * tc + lr;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * tr + lk + 4 * v;
          if (row < M && col < nb && (row >= nb || col <= row)) G[row * ld + col] = hold[u][v];
        }
      }
    }
  } else
  // fill: 9 entries per thread, loads first (their latencies overlap), then the LDS stores
  {
    constexpr int NE = kDcM * kDcNB, U = (NE + 1023) / 1024;
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + 1024 * u, row = e / kDcNB, col = e - row * kDcNB;
      const bool diag = row < nb;
      const bool in = col < nb && (diag ? col <= row : row < M);
      const int gr = diag ? k0 + row : dense_row(q0 + row - nb, total, kn, n);
      v[u] = in ? A[(size_t)gr * n + k0 + col] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + 1024 * u, row = e / kDcNB, col = e - row * kDcNB;
      if (e < NE) G[row * ld + col] = v[u];
    }
    if (tid < kDcM) G[tid * ld + kDcNB] = 0.0;      // the padding column
  }

  __syncthreads();
#pragma unroll 1
  for (int j0 = 0; j0 < nb; j0 += 12) {
    const int nbi = nb - j0 < 12 ? nb - j0 : 12;    // 12, or 6 at the very end of the matrix
    const int jn = j0 + nbi;
    if (wave == 0) {
      __builtin_amdgcn_s_setprio(3);                // (k_dense_step: wavefronts of the trailing update share this compute unit)
      if (nbi == 12) bcr_diag_block<12>(G, ld, dinv, bad, j0, lane, Li, Idt);
      else bcr_diag_block<6>(G, ld, dinv, bad, j0, lane, Li, Idt);
      if (fnb > 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    // rows below the diagonal block: X = A L_kk^-T, one 16-row tile per wavefront (below a 6-unknown block: the right-hand side row)
    for (int i0 = jn + 16 * wave; i0 < M; i0 += 256) {
      double pr[3];
      bcr_panel_tile(G, ld, M, j0, i0, Li, lr, lk, pr, nbi);
    }
    __syncthreads();
    if (jn < nb) {                                  // rank-12 update of the columns right of this block (nbi == 12 here)
      const int nct = (nb - jn + 15) >> 4;
      for (int task = wave;; task += 16) {
        int t = task, c0 = jn, found = 0;
        for (int ct = 0; ct < nct; ++ct) {
          c0 = jn + 16 * ct;
          const int nrt = (M - c0 + 15) >> 4;
          if (t < nrt) { found = 1; break; }
          t -= nrt;
        }
        if (!found) break;
        const int i0 = c0 + 16 * t;
        const int ao = (i0 + lr) * ld + j0 + lk, bo = (c0 + lr) * ld + j0 + lk, cb = (i0 + lk) * ld + c0 + lr;
        mfma_acc acc = {G[cb], G[cb + 4 * ld], G[cb + 8 * ld], G[cb + 12 * ld]};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(G[ao], -G[bo], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(G[ao + 4], -G[bo + 4], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(G[ao + 8], -G[bo + 8], acc, 0, 0, 0);
        const int rl = c0 + lr < nb ? M - i0 - lk : 0;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (4 * v < rl) G[cb + 4 * v * ld] = acc[v];
      }
    }
    __syncthreads();
  }
  if (*bad && tid == 0) atomicMax(info, k0 + *bad);
  // write back: the diagonal block once (1 / L_kk on its diagonal), every workgroup its own panel rows
  for (int e = tid; e < M * kDcNB; e += 1024) {
    const int row = e / kDcNB, col = e - row * kDcNB;
    if (col >= nb) continue;
    if (row < nb) {
      if (chunk == 0 && col <= row) A[(size_t)(k0 + row) * n + k0 + col] = col == row ? dinv[row] : G[row * ld + col];
    } else {
      A[(size_t)dense_row(q0 + row - nb, total, kn, n) * n + k0 + col] = G[row * ld + col];
    }
  }
}

__global__ __launch_bounds__(1024) void k_dense_panel(int n, int k0, int nb, int total, double* __restrict__ A,
                                                      int* __restrict__ info, size_t batch_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  dense_panel_body(n, k0, nb, total, A + (size_t)blockIdx.z * batch_stride, info, blockIdx.x, 0, 0, sm);
}

// ---- trailing update: A[i][j] -= sum_c P[i][c] P[j][c] for kn <= j <= i <= n, P = A[., k0 .. k0 + nb); tile (ti, tj) of the
// trailing rows' list.  kn: the first row of the region that is updated (the step's own k0 + nb - or, in k_dense_step, the row
// behind the NEXT block column, whose workgroups fold this update into their panel step), total: rows of that region + 1.
__host__ __device__ inline size_t dense_update_lds_bytes() { return (size_t)2 * kDcTile * kDcLd * sizeof(double); }
__device__ __forceinline__ void dense_update_body(int n, int k0, int nb, int kn, int total, double* __restrict__ A, int ti, int tj,
                                                  double* __restrict__ sm) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  double* Pi = sm;                                 // [kDcTile][kDcLd]
  double* Pj = sm + kDcTile * kDcLd;
  if (tj > ti) return;
  const int i0 = kDcTile * ti, j0 = kDcTile * tj;                        // list indices
  if (j0 >= total - 1) return;                                            // the last entry (row n) is not a column
  const int tid = threadIdx.x;
  {
    double vi[3], vj[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + 1024 * u, row = e / kDcNB, col = e - row * kDcNB;
      vi[u] = (i0 + row < total && col < nb) ? A[(size_t)dense_row(i0 + row, total, kn, n) * n + k0 + col] : 0.0;
      vj[u] = (j0 + row < total - 1 && col < nb) ? A[(size_t)(kn + j0 + row) * n + k0 + col] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + 1024 * u, row = e / kDcNB, col = e - row * kDcNB;
      Pi[row * kDcLd + col] = vi[u];
      Pj[row * kDcLd + col] = vj[u];
    }
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int wi = wave >> 2, wj = wave & 3;
  const int cidx = j0 + 16 * wj + lr, rowb = i0 + 16 * wi + lk;
  const bool cok = cidx < total - 1;
  double* Cp[4];
  mfma_acc acc;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int ridx = rowb + 4 * v;
    Cp[v] = A + (size_t)dense_row(ridx < total ? ridx : 0, total, kn, n) * n + kn + cidx;
    acc[v] = (cok && ridx < total) ? *Cp[v] : 0.0;
  }
  __syncthreads();
  if (ti == tj && wj > wi) return;                 // strictly upper tile of a diagonal block: never read
  const double* ap = Pi + (16 * wi + lr) * kDcLd + lk;
  const double* bp = Pj + (16 * wj + lr) * kDcLd + lk;
#pragma unroll
  for (int s = 0; s < kDcNB / 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * s], -bp[4 * s], acc, 0, 0, 0);
#pragma unroll
  for (int v = 0; v < 4; ++v)
    if (cok && rowb + 4 * v < total) *Cp[v] = acc[v];
}

__global__ __launch_bounds__(1024) void k_dense_update(int n, int k0, int nb, int total, double* __restrict__ A, size_t batch_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  dense_update_body(n, k0, nb, k0 + nb, total, A + (size_t)blockIdx.z * batch_stride, blockIdx.x, blockIdx.y, sm);
}

__host__ __device__ inline size_t dense_step_lds_bytes() { return dense_panel_lds_bytes() > dense_update_lds_bytes() ? dense_panel_lds_bytes() : dense_update_lds_bytes(); }
// ---- ONE launch per block column (round 5).  The panel step of block column k + 1 needs nothing of step k's trailing update
// but the part that lands on its own columns - and every panel workgroup can form that part for the rows it holds (the
// diagonal block's, redundantly, and its own 64): so the workgroups of panel k + 1 (blockIdx.x < npanel, the update folded in) run
// BESIDE those of update k on everything right of block column k + 1 (the lower-triangular tiles, in a line).  A step costs
// max(panel, update) instead of their sum, and half the launches; no workgroup waits for another.
//   n1 / total1: rows of block column k + 1 = [k1, k1 + n1) and of the list behind it (as k_dense_panel's nb / total)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_dense_step(int n, int k0, int nb, int k1, int n1, int total1, int npanel, int T,
                                                     double* __restrict__ A, int* __restrict__ info, size_t batch_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  A += (size_t)blockIdx.z * batch_stride;
  if ((int)blockIdx.x < npanel) {
    dense_panel_body(n, k1, n1, total1, A, info, blockIdx.x, k0, nb, sm);
    return;
  }
  // tile t of the lower triangle, row by row: t = ti (ti + 1) / 2 + tj
  const int t = blockIdx.x - npanel;
  int ti = (int)((__fsqrt_rn(8.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  while (ti * (ti + 1) / 2 > t) --ti;
  const int tj = t - ti * (ti + 1) / 2;
  if (ti >= T) return;
  dense_update_body(n, k0, nb, k1 + n1, total1, A, ti, tj, sm);
}

// ---- x = L^-T y (y = row n of A), one workgroup; x[n] out.  Block columns right to left; per block
//   (1) L_kk -> LDS (prefetched into registers during the previous block),
//   (2) wavefront 0 solves L_kk^T x_k = w_k column by column (lane c owns w[k0 + c]; x_r leaves lane r
//       through v_readlane, row r of L_kk sits in registers 12 rows at a time) WHILE every thread's loads
//       of its part of L[k0.., jlo..k0) and of the next L_kk are in flight,
//   (3) w[jlo..k0) -= L[k0.., jlo..k0)^T x_k: lanes along the columns (rows of L are contiguous), the 48
//       rows split over as many groups as the workgroup has threads for, reduced through LDS.
constexpr int kDcBsRows = 24;                      // update rows per thread that are prefetched

__device__ __forceinline__ double dense_lkk_entry(const double* __restrict__ A, int ld, int k0, int nb, int e) {
  const int row = e / kDcNB, col = e - row * kDcNB;
  return (e < kDcNB * kDcNB && row < nb && col <= row) ? A[(size_t)(k0 + row) * ld + k0 + col] : 0.0;
}

// w[0..n) in LDS (the caller has NOT synchronised after filling it) -> x = L^-T w in place; L = rows 0..n-1 of A, row stride ld
__device__ __forceinline__ void dense_backsolve_body(int n, int ld, int bw, const double* __restrict__ A, double* __restrict__ sm) {
  double* w = sm;                                  // [n]
  double* Lk = w + n;                              // [48][49]
  double* xk = Lk + kDcNB * kDcLd;                 // [48]
  double* red = xk + kDcNB;                        // [1024]
  const int tid = threadIdx.x, lane = tid & 63;
  const int nblk = (n + kDcNB - 1) / kDcNB;
  double lv[3];
  {
    const int k0 = kDcNB * (nblk - 1), nb = n - k0;
#pragma unroll
    for (int u = 0; u < 3; ++u) lv[u] = dense_lkk_entry(A, ld, k0, nb, tid + 1024 * u);
  }
#pragma unroll 1
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int k0 = kDcNB * kb, nb = min(kDcNB, n - k0);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + 1024 * u, row = e / kDcNB, col = e - row * kDcNB;
      if (e < kDcNB * kDcNB) Lk[row * kDcLd + col] = lv[u];
    }
    __syncthreads();
    // (2a) loads first: my rows of the update operand, the next diagonal block
    const int jlo = max(0, k0 - bw) & ~7, nj = k0 - jlo;      // columns left of jlo are zero in these rows of L
    const int cols = (nj + 63) & ~63;
    int ng = cols > 0 ? 1024 / cols : 1;
    ng = ng >= 16 ? 16 : ng >= 8 ? 8 : ng >= 4 ? 4 : ng >= 2 ? 2 : 1;
    const int rp = kDcNB / ng;                                 // 3, 6, 12, 24 or 48 rows per group
    const int g = cols > 0 ? tid / cols : 0, j = tid - g * cols;
    const bool mine = ng > 1 && g < ng && j < nj;
    const int rend = min(rp, nb - g * rp);
    double uv[kDcBsRows];
    {
      const double* Lp = A + (size_t)(k0 + g * rp) * ld + jlo + j;
#pragma unroll
      for (int r = 0; r < kDcBsRows; ++r) uv[r] = (mine && r < rend) ? Lp[(size_t)r * ld] : 0.0;
    }
    if (kb > 0) {
#pragma unroll
      for (int u = 0; u < 3; ++u) lv[u] = dense_lkk_entry(A, ld, k0 - kDcNB, kDcNB, tid + 1024 * u);
    }
    // (2b) the triangular solve
    if (tid < 64) {
      double wc = lane < nb ? w[k0 + lane] : 0.0;
      const int lc = lane < kDcNB ? lane : 0;
#pragma unroll
      for (int rb = kDcNB / 12 - 1; rb >= 0; --rb) {
        if (12 * rb < nb) {
          double lrow[12];
#pragma unroll
          for (int q = 0; q < 12; ++q) lrow[q] = Lk[(12 * rb + q) * kDcLd + lc];       // L[r][c], c = my lane; lane r: 1 / L[r][r]
#pragma unroll
          for (int q = 11; q >= 0; --q) {
            const int r = 12 * rb + q;
            if (r < nb) {
              const double wr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wc), r),
                                                 __builtin_amdgcn_readlane(__double2loint(wc), r));
              const double dr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(lrow[q]), r),
                                                 __builtin_amdgcn_readlane(__double2loint(lrow[q]), r));
              const double xr = wr * dr;
              wc = lane < r ? wc - lrow[q] * xr : (lane == r ? xr : wc);
            }
          }
        }
      }
      if (lane < nb) w[k0 + lane] = wc;
      if (lane < kDcNB) xk[lane] = lane < nb ? wc : 0.0;      // (a short last block: the update below multiplies entries nb.. by zero
                                                               //  rows of L - they must be zeros, not stale LDS: 0 x NaN = NaN)
    }
    __syncthreads();
    // (3) the update
    if (nj > 0) {
      if (ng > 1) {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < kDcBsRows; ++r) acc += uv[r] * xk[min(g * rp + r, kDcNB - 1)];
        if (g < ng) red[g * cols + j] = acc;
        __syncthreads();
        if (tid < nj) {
          double t = 0.0;
          for (int q = 0; q < ng; ++q) t += red[q * cols + tid];
          w[jlo + tid] -= t;
        }
      } else {
        for (int jj = jlo + tid; jj < k0; jj += 1024) {
          const double* Lp = A + (size_t)k0 * ld + jj;
          double acc = 0.0;
#pragma unroll 12
          for (int r = 0; r < nb; ++r) acc += Lp[(size_t)r * ld] * xk[r];
          w[jj] -= acc;
        }
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void k_dense_backsolve(int n, int bw, const double* __restrict__ A, double* __restrict__ x,
                                                          const int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  if (*info != 0) return;
  const int tid = threadIdx.x;
  for (int j = tid; j < n; j += 1024) sm[j] = A[(size_t)n * n + j];
  dense_backsolve_body(n, n, bw, A, sm);
  for (int j = tid; j < n; j += 1024) x[j] = sm[j];
}

}  // namespace ba
