// ba_bcr_wide.h - block cyclic reduction for half-bandwidths 12..23 (B = 6 hb = 72..138 unknowns per
// super-block).  Same algebra, node numbering and workspace layout as ba_bcr.h, but one B x B matrix is
// all that fits in LDS, so a level is three kernels instead of one:
//
//   k_bcrw_factor    one workgroup per node:  D_i = L L^T in LDS (the blocked, look-ahead Cholesky of
//                    ba_bcr.h without right-hand sides), L -> global (1 / L_kk on its diagonal)
//   k_bcrw_solve_mfma  one wavefront per (node, 16 right-hand sides):  [P | Q | G^-1 | g] = L^-1 [T_il | T_ir | I | f]
//   k_bcrw_products  one workgroup per (node, 64 x 64 outputs, MFMA):  D_l -= P^T P,  D_r -= Q^T Q,
//                    T[l,r] = -P^T Q,  f_l -= P^T g,  f_r -= Q^T g
//   k_bcrw_backsolve x_i = G^-T (g - P x_l - Q x_r)
//
// The single-workgroup band Cholesky (k_band_solve) these replace walks 1000 cameras in 7..20 ms at
// these widths; the levels here are ~80 us each at hb = 21 (factor 36, solve 24, products 10, backsolve 11).
// (The first form of the solve, one lane per right-hand side with L through scalar loads, took 442 us per level.)
#pragma once

#include <type_traits>

#include "ba_bcr_blocks.h"

namespace ba {

constexpr int kBcrwLvLdsMaxB = 126;            // up to here the inverses of the diagonal blocks sit in LDS next to L; beyond, in L2

__host__ __device__ inline size_t bcrw_factor_lds_bytes(int B) { return ((size_t)B * (B + 1) + B + 16 + 192 + kBcrIdtDoubles) * sizeof(double); }
#ifndef BA_BCRW_FS_THREADS
#define BA_BCRW_FS_THREADS 512
#endif
constexpr int kBcrwFsThreads = BA_BCRW_FS_THREADS;
__host__ __device__ inline size_t bcrw_factor_solve_lds_bytes(int B) { return bcrw_factor_lds_bytes(B) + 192 * sizeof(double); }
__host__ __device__ inline size_t bcrw_solve_lds_bytes(int B) {
  return ((size_t)B * (B + 1) + (B <= kBcrwLvLdsMaxB ? (size_t)((B + 11) / 12) * 144 : 0) + 64) * sizeof(double);
}

// B x B row-major global matrix -> LDS with row stride B + 1; loads issued U at a time so that their
// latencies overlap (one load - store pair per iteration costs a full memory round trip each)
template <int B, int THREADS, int U>
__device__ __forceinline__ void bcrw_fill(double* __restrict__ dst, const double* __restrict__ src, int tid) {
  constexpr int NE = B * B;
#pragma unroll 1
  for (int e0 = tid; e0 < NE; e0 += THREADS * U) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + THREADS * u;
      v[u] = e < NE ? src[e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + THREADS * u;
      if (e < NE) dst[e + e / B] = v[u];
    }
  }
}

// ---- factor: Lm[i] = Cholesky factor of Dm[i] (lower, row-major B x B, 1 / L_kk on the diagonal)
template <int HB>
__global__ __launch_bounds__(kBcrElimThreads) void k_bcrw_factor(int N, int s, const double* __restrict__ Dm,
                                                                double* __restrict__ Lm, double* __restrict__ Lvm,
                                                                int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int B = 6 * HB, ld = B + 1;
  double* G = sm;                       // [B][ld]
  double* dinv = G + (size_t)B * ld;    // [B]
  int* bad = reinterpret_cast<int*>(dinv + B + 2);
  double* Li = dinv + B + 16;           // [16][12]: inverse of the current diagonal block (rows 12..15 stay zero)
  const int tid = threadIdx.x;
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N) return;
  constexpr size_t BB = (size_t)B * B;
  if (tid == 0) *bad = 0;
  if (tid < 192) Li[tid] = 0.0;
  double* Idt = Li + 192;
  bcr_identity_table(Idt, tid);
  bcrw_fill<B, kBcrElimThreads, 16>(G, Dm + (size_t)i * BB, tid);
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  constexpr int NBLK = (B + 11) / 12;
  double pr[3] = {0.0, 0.0, 0.0};                           // wavefront 0: the first tile of the panel, handed from phase 2 to the next phase 1
  bcr_acc4 cpre = {0.0, 0.0, 0.0, 0.0};                     // and the tile it updates there
#pragma unroll 1
  for (int kb = 0; kb < NBLK; ++kb) {
    const int k0 = 12 * kb;
    const bool last = kb == NBLK - 1;
    const int nb = last ? B - k0 : 12;                      // this block: 12, or 6 at the end
    const int kn = k0 + nb;
    // ---------------- phase 1: diagonal factor (wavefront 0) | the late part of the previous step's update
    // what block column kb still owes to the panel of block kb - 1 (C -= panel panel^T, K = 12), one 16-row tile per call: the
    // tile holding the diagonal block by wavefront 0 right before it factors it, the tiles below as tasks of the other wavefronts
    auto urgent_tile = [&](int t) {
      typedef double mfma_acc __attribute__((ext_vector_type(4)));
      const int kp = k0 - 12, i0 = k0 + 16 * t;
      const int ao = (i0 + lr) * ld + kp + lk, bo = (k0 + lr) * ld + kp + lk, cb = (i0 + lk) * ld + k0 + lr;
      mfma_acc acc = {sm[cb], sm[cb + 4 * ld], sm[cb + 8 * ld], sm[cb + 12 * ld]};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao], -sm[bo], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao + 4], -sm[bo + 4], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao + 8], -sm[bo + 8], acc, 0, 0, 0);
      if (i0 + 16 <= B) {                                    // (wave-uniform) all 16 rows exist: one lane mask for the four stores
        if (lr < nb) {
#pragma unroll
          for (int v = 0; v < 4; ++v) sm[cb + 4 * v * ld] = acc[v];
        }
      } else {
        const int rl = lr < nb ? B - i0 - lk : 0;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (4 * v < rl) sm[cb + 4 * v * ld] = acc[v];
      }
    };
    if (wave == 0) {
      if (kb > 0) { bcr_urgent_tile0(sm, ld, B, k0, nb, lr, lk, pr, cpre); lds_wave_sync(); }
      double* Lv = Lvm + ((size_t)i * NBLK + kb) * 144;      // inverse of this diagonal block, [12][12], zero padded, for k_bcrw_solve_mfma
      if (nb == 12) {
        bcr_diag_block<12>(G, ld, dinv, bad, k0, lane, Li, Idt);
      } else {
        for (int e = lane; e < 144; e += 64) Li[e] = 0.0;
        lds_wave_sync();
        bcr_diag_block<6>(G, ld, dinv, bad, k0, lane, Li, Idt);
      }
      lds_wave_sync();
      for (int e = lane; e < 144; e += 64) Lv[e] = Li[e];
    } else if (kb > 0) {
      typedef double mfma_acc __attribute__((ext_vector_type(4)));
      const int kp = k0 - 12;
      const int ngt = (B - kn + 15) >> 4;                   // column tiles of the trailing matrix right of this block
      const int nsu = ((B - k0 + 15) >> 4) - 1;             // tiles of block column kb below the one wavefront 0 takes
      for (int task = wave - 1; task < ngt + nsu; task += kBcrElimThreads / 64 - 1) {
        if (task >= ngt) { urgent_tile(task - ngt + 1); continue; }
        const int c0 = kn + 16 * task;
        const bool cok = c0 + lr < B, full = c0 + 15 < B;
        const int bo = (c0 + lr) * ld + kp + lk;
        const double nb0 = -sm[bo], nb1 = -sm[bo + 4], nb2 = -sm[bo + 8];
        const int co = lk * ld + c0 + lr, c4 = 4 * ld;
        const int t1 = (B - kn + 15) >> 4;
        int ao = (kn + 16 * task + lr) * ld + kp + lk;
        int cb = co + (kn + 16 * task) * ld;
        int rows = B - (kn + 16 * task);
        for (int t = task; t < t1; ++t) {                    // lower tiles of this column tile, top to bottom
          const double a0 = sm[ao], a1 = sm[ao + 4], a2 = sm[ao + 8];
          mfma_acc acc = {sm[cb], sm[cb + c4], sm[cb + 2 * c4], sm[cb + 3 * c4]};
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, nb0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, nb1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, nb2, acc, 0, 0, 0);
          if (full && rows >= 16) {
#pragma unroll
            for (int v = 0; v < 4; ++v) sm[cb + v * c4] = acc[v];
          } else {
            const int rl = cok ? rows - lk : 0;
#pragma unroll
            for (int v = 0; v < 4; ++v)
              if (4 * v < rl) sm[cb + v * c4] = acc[v];
          }
          ao += 16 * ld; cb += 16 * ld; rows -= 16;
        }
      }
    }
    __syncthreads();
    // ---------------- phase 2: panel, rows below the diagonal block: X = A L_kk^-T, one 16-row tile per wavefront
    if (kn + 16 * wave < B) {                                // (B <= 138: at most 8 tiles; nb == 12 here)
      if (wave == 0) cpre = bcr_prefetch_tile0(sm, ld, kn, lr, lk);
      bcr_panel_tile(G, ld, B, k0, kn + 16 * wave, Li, lr, lk, pr);
    }
    __syncthreads();
  }
  if (*bad) {
    if (tid == 0) atomicMax(info, i * B + *bad);
    return;
  }
  for (int e = tid; e < B * B; e += kBcrElimThreads) {
    const int rr = e / B, cc = e - rr * B;
    Lm[(size_t)i * BB + e] = cc < rr ? G[rr * ld + cc] : (cc == rr ? dinv[rr] : 0.0);
  }
}

// ---- factor AND solve in one kernel (round 5): one workgroup per (node, 64 right-hand-side columns).  k_bcrw_factor keeps ONE compute
// unit per node busy with the chain of B pivots while the rest of the chip waits, then k_bcrw_solve_mfma reads L back on six
// workgroups per node.  Here each of those workgroups factors D_i for itself (redundant - the chain cannot be shared, and the
// compute units are idle anyway), and four of its eight wavefronts (4 .. 7; 512 threads: 256 registers each, the right-hand sides
// alone are 72 at B = 138) take 16 columns each of [T_il | T_ir | I | f] through
// the substitution WHILE the factorisation runs, left-looking: at block step kb they finish Y_kb-1 = L_pp^-1 acc (the inverse of
// diagonal block kb - 1 is in LDS since the step before) and form acc = R_kb - sum_{j < kb} L[kb, j] Y_j from the panels that are
// final - 3 + 3 kb MFMAs, under the 12 pivots of wavefront 0.  The right-hand sides live in registers as in k_bcrw_solve_mfma
// (3 NBLK values a lane).  A level: factor 17.9 + solve 12.9 us -> one launch (hb = 15).
template <int HB>
__global__ __launch_bounds__(kBcrwFsThreads) void k_bcrw_factor_solve(int N, int s, const double* __restrict__ Dm,
                                                                      const double* __restrict__ Um, double* __restrict__ fm,
                                                                      double* __restrict__ Pm, double* __restrict__ Qm,
                                                                      double* __restrict__ Gi, int* __restrict__ info) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int B = 6 * HB, ld = B + 1;
  double* G = sm;                       // [B][ld]
  double* dinv = G + (size_t)B * ld;    // [B]
  int* bad = reinterpret_cast<int*>(dinv + B + 2);
  double* Li = dinv + B + 16;           // [2][16][12]: inverse of diagonal block kb in half kb & 1 (rows 12..15 stay zero)
  const int tid = threadIdx.x;
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N) return;
  constexpr size_t BB = (size_t)B * B;
  if (tid == 0) *bad = 0;
  if (tid < 384) Li[tid] = 0.0;
  double* Idt = Li + 384;
  bcr_identity_table(Idt, tid);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  constexpr int NBLK = (B + 11) / 12;
  // ---- my 16 right-hand-side columns (wavefronts 12 .. 15; as in k_bcrw_solve_mfma)
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  constexpr int ncol = 3 * B + 1;
  constexpr int NW = kBcrwFsThreads / 64;                    // wavefront 0: the chain; 1 .. NW - 5: the trailing updates; the last four: right-hand sides
  const bool rhs_wave = wave >= NW - 4;
  const int ct = blockIdx.y * 4 + (wave - (NW - 4));
  const int c = 16 * ct + lr;
  const bool live = rhs_wave && c < ncol;
  const int grp = c < B ? 0 : c < 2 * B ? 1 : c < 3 * B ? 2 : 3;
  const int cc = c - grp * B;
  const bool loads = live && ((grp == 0 && haveL) || (grp == 1 && haveR) || grp == 3);
  const double* rbase = !loads ? fm + (size_t)i * B
                        : grp == 0 ? Um + (size_t)l * BB + (size_t)cc * B      // T[i,l] = T[l,i]^T
                        : grp == 1 ? Um + (size_t)i * BB + cc                  // T[i,r]
                                   : fm + (size_t)i * B;
  const int rstride = (loads && grp == 1) ? B : 1;
  const bool ident = live && grp == 2;
  auto rhs = [&](int row) -> double {
    const int rc = row < B ? row : B - 1;
    const double v = rbase[rc * rstride];                    // always finite: a valid entry of U or f
    return v * ((loads && row < B) ? 1.0 : 0.0) + ((ident && row == cc) ? 1.0 : 0.0);
  };
  double* out = grp == 0 ? Pm + (size_t)i * BB + cc : grp == 1 ? Qm + (size_t)i * BB + cc : grp == 2 ? Gi + (size_t)i * BB + cc
                                                                                              : fm + (size_t)i * B;
  const int ost = grp == 3 ? 1 : B;
  // identity columns are zero above their own row: the whole tile is zero in block rows above its first column
  const int c_lo = 16 * ct, zero_rows = (rhs_wave && c_lo >= 2 * B && c_lo + 15 < 3 * B) ? c_lo - 2 * B : 0;
  double ny[NBLK][3];                                        // R (until its block row is solved), then -Y
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb)
#pragma unroll
    for (int v = 0; v < 3; ++v) ny[kb][v] = rhs_wave ? rhs(12 * kb + lk + 4 * v) : 0.0;
  bcrw_fill<B, kBcrwFsThreads, 16>(G, Dm + (size_t)i * BB, tid);      // (the right-hand sides' loads are in flight under it)
  __syncthreads();
  mfma_acc racc = {ny[0][0], ny[0][1], ny[0][2], 0.0};      // R_kb - sum_{j < kb} L[kb, j] Y_j of the block row that is solved next
  // one block row of the substitution: Y_p = L_pp^-1 racc (stored), then racc for block row p + 1
  // (the block row a template parameter: with a run-time index the compiler keeps ny[][] in scratch memory)
  auto rhs_step_c = [&](auto pc) __attribute__((always_inline)) {      // p = 0 .. NBLK - 1: the block row whose diagonal inverse is in Li[p & 1]
    constexpr int p = decltype(pc)::value;
    const double* Lp = Li + 192 * (p & 1);
    constexpr int r0 = 12 * p;
    if (r0 + 12 <= zero_rows) {                              // (wave-uniform) identity columns right of this block row: Y_p = 0
      ny[p][0] = 0.0; ny[p][1] = 0.0; ny[p][2] = 0.0;
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const int row = r0 + lk + 4 * v;
        if (live && row < B) out[(size_t)row * ost] = 0.0;
      }
    } else {
      mfma_acc y = {0.0, 0.0, 0.0, 0.0};
      y = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[lr * 12 + lk], racc[0], y, 0, 0, 0);
      y = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[lr * 12 + 4 + lk], racc[1], y, 0, 0, 0);
      y = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[lr * 12 + 8 + lk], racc[2], y, 0, 0, 0);
      ny[p][0] = -y[0]; ny[p][1] = -y[1]; ny[p][2] = -y[2];
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const int row = r0 + lk + 4 * v;
        if (live && row < B) out[(size_t)row * ost] = y[v];
      }
    }
    if constexpr (p + 1 < NBLK) {
      // racc of block row p + 1: its right-hand side, then the panels 0 .. p (rows of block p + 1) times Y_0 .. Y_p
      constexpr int kb = p + 1, k0r = 12 * kb;
      constexpr bool tail = kb == NBLK - 1 && B % 12 != 0;   // 6 real rows: rows 6 .. 11 of the A operand lie beyond L - ZERO, not "whatever is there"
      const double rowok = (!tail || k0r + lr < B) ? 1.0 : 0.0;
      const int arow = (tail && k0r + lr >= B) ? B - 1 : k0r + lr;
      racc = mfma_acc{ny[kb][0], ny[kb][1], ny[kb][2], 0.0};
#pragma unroll
      for (int j = 0; j < kb; ++j) {
        if (12 * j + 12 > zero_rows) {                       // (wave-uniform; Y_j = 0 above the identity columns' first row)
          const double* ap = G + arow * ld + 12 * j + lk;
          const double a0 = tail ? ap[0] * rowok : ap[0], a1 = tail ? ap[4] * rowok : ap[4], a2 = tail ? ap[8] * rowok : ap[8];
          racc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, ny[j][0], racc, 0, 0, 0);
          racc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, ny[j][1], racc, 0, 0, 0);
          racc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, ny[j][2], racc, 0, 0, 0);
        }
      }
    }
  };
  auto rhs_step = [&](int p) __attribute__((always_inline)) {
    auto go = [&](auto self, auto c) __attribute__((always_inline)) {
      constexpr int V = decltype(c)::value;
      if constexpr (V < NBLK) {
        if (p == V) rhs_step_c(c);
        else self(self, std::integral_constant<int, V + 1>{});
      }
    };
    go(go, std::integral_constant<int, 0>{});
  };
  double pr[3] = {0.0, 0.0, 0.0};                           // wavefront 0: the first tile of the panel, handed from phase 2 to the next phase 1
  bcr_acc4 cpre = {0.0, 0.0, 0.0, 0.0};                     // and the tile it updates there
#pragma unroll 1
  for (int kb = 0; kb < NBLK; ++kb) {
    const int k0 = 12 * kb;
    const bool last = kb == NBLK - 1;
    const int nb = last ? B - k0 : 12;                      // this block: 12, or 6 at the end
    const int kn = k0 + nb;
    // ---------------- phase 1: diagonal factor (wavefront 0) | the late part of the previous step's update
    // what block column kb still owes to the panel of block kb - 1 (C -= panel panel^T, K = 12), one 16-row tile per call: the
    // tile holding the diagonal block by wavefront 0 right before it factors it, the tiles below as tasks of the other wavefronts
    auto urgent_tile = [&](int t) {
      const int kp = k0 - 12, i0 = k0 + 16 * t;
      const int ao = (i0 + lr) * ld + kp + lk, bo = (k0 + lr) * ld + kp + lk, cb = (i0 + lk) * ld + k0 + lr;
      mfma_acc acc = {sm[cb], sm[cb + 4 * ld], sm[cb + 8 * ld], sm[cb + 12 * ld]};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao], -sm[bo], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao + 4], -sm[bo + 4], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao + 8], -sm[bo + 8], acc, 0, 0, 0);
      if (i0 + 16 <= B) {                                    // (wave-uniform) all 16 rows exist: one lane mask for the four stores
        if (lr < nb) {
#pragma unroll
          for (int v = 0; v < 4; ++v) sm[cb + 4 * v * ld] = acc[v];
        }
      } else {
        const int rl = lr < nb ? B - i0 - lk : 0;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (4 * v < rl) sm[cb + 4 * v * ld] = acc[v];
      }
    };
    if (wave == 0) {
      __builtin_amdgcn_s_setprio(3);                        // the pivot chain is the critical path: ahead of the right-hand-side wavefront on its SIMD
      if (kb > 0) { bcr_urgent_tile0(sm, ld, B, k0, nb, lr, lk, pr, cpre); lds_wave_sync(); }
      double* Lik = Li + 192 * (kb & 1);
      if (nb == 12) {
        bcr_diag_block<12>(G, ld, dinv, bad, k0, lane, Lik, Idt);
      } else {
        for (int e = lane; e < 144; e += 64) Lik[e] = 0.0;
        lds_wave_sync();
        bcr_diag_block<6>(G, ld, dinv, bad, k0, lane, Lik, Idt);
      }
      __builtin_amdgcn_s_setprio(0);
    } else if (rhs_wave) {
      if (kb > 0 && 16 * ct < ncol) rhs_step(kb - 1);
    } else if (kb > 0) {
      const int kp = k0 - 12;
      const int ngt = (B - kn + 15) >> 4;                   // column tiles of the trailing matrix right of this block
      const int nsu = ((B - k0 + 15) >> 4) - 1;             // tiles of block column kb below the one wavefront 0 takes
      for (int task = wave - 1; task < ngt + nsu; task += NW - 5) {      // (wavefronts 1 .. NW - 5)
        if (task >= ngt) { urgent_tile(task - ngt + 1); continue; }
        const int c0 = kn + 16 * task;
        const bool cok = c0 + lr < B, full = c0 + 15 < B;
        const int bo = (c0 + lr) * ld + kp + lk;
        const double nb0 = -sm[bo], nb1 = -sm[bo + 4], nb2 = -sm[bo + 8];
        const int co = lk * ld + c0 + lr, c4 = 4 * ld;
        const int t1 = (B - kn + 15) >> 4;
        int ao = (kn + 16 * task + lr) * ld + kp + lk;
        int cb = co + (kn + 16 * task) * ld;
        int rows = B - (kn + 16 * task);
        for (int t = task; t < t1; ++t) {                    // lower tiles of this column tile, top to bottom
          const double a0 = sm[ao], a1 = sm[ao + 4], a2 = sm[ao + 8];
          mfma_acc acc = {sm[cb], sm[cb + c4], sm[cb + 2 * c4], sm[cb + 3 * c4]};
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, nb0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, nb1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, nb2, acc, 0, 0, 0);
          if (full && rows >= 16) {
#pragma unroll
            for (int v = 0; v < 4; ++v) sm[cb + v * c4] = acc[v];
          } else {
            const int rl = cok ? rows - lk : 0;
#pragma unroll
            for (int v = 0; v < 4; ++v)
              if (4 * v < rl) sm[cb + v * c4] = acc[v];
          }
          ao += 16 * ld; cb += 16 * ld; rows -= 16;
        }
      }
    }
    __syncthreads();
    // ---------------- phase 2: panel, rows below the diagonal block: X = A L_kk^-T, one 16-row tile per wavefront
    if (kn + 16 * wave < B) {                                // (B <= 138: at most 8 tiles; nb == 12 here)
      if (wave == 0) cpre = bcr_prefetch_tile0(sm, ld, kn, lr, lk);
      bcr_panel_tile(G, ld, B, k0, kn + 16 * wave, Li + 192 * (kb & 1), lr, lk, pr);
    }
    __syncthreads();
  }
  if (*bad) {
    if (tid == 0 && blockIdx.y == 0) atomicMax(info, i * B + *bad);
    return;                                                  // (what the right-hand-side wavefronts stored is never read: the status word says so)
  }
  if (rhs_wave && 16 * ct < ncol) rhs_step(NBLK - 1);        // the last block row: its diagonal inverse came with the last step
}


// ---- solve on the matrix cores: one workgroup (4 wavefronts) per (node, 4 column tiles of 16 right-hand
// sides); L and the inverses of its 12 x 12 diagonal blocks sit in LDS, a wavefront walks its tile block
// row by block row:  acc = R_kb - sum_{j<kb} L[kb,j] Y_j  (3 MFMAs per block, Y_j in registers in exactly
// the B-operand layout it came out of the MFMA in), then  Y_kb = L_kk^-1 acc  (3 more).
template <int HB>
__global__ __launch_bounds__(1024) void k_bcrw_solve_mfma(int N, int s, const double* __restrict__ Lm,
                                                         const double* __restrict__ Lvm, const double* __restrict__ Um,
                                                         double* __restrict__ fm, double* __restrict__ Pm,
                                                         double* __restrict__ Qm, double* __restrict__ Gi,
                                                         const int* __restrict__ info) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int B = 6 * HB, ld = B + 1, NBLK = (B + 11) / 12;
  constexpr bool LV_LDS = B <= kBcrwLvLdsMaxB;
  double* Ls = sm;                       // [B][ld]
  double* Lvs = Ls + (size_t)B * ld;     // [NBLK][144] (LV_LDS)
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N || *info != 0) return;
  const double* Lv = LV_LDS ? Lvs : Lvm + (size_t)i * NBLK * 144;
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  constexpr size_t BB = (size_t)B * B;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  constexpr int ncol = 3 * B + 1;
  const int ct = blockIdx.y * 4 + wave;
  const int c = 16 * ct + lr;
  const bool live = c < ncol && wave < 4;
  const int grp = c < B ? 0 : c < 2 * B ? 1 : c < 3 * B ? 2 : 3;
  const int cc = c - grp * B;
  // right-hand side entry (row, my column): one unconditional load per entry (address clamped, value
  // selected afterwards) - branches around the loads serialise their latencies
  const bool loads = live && ((grp == 0 && haveL) || (grp == 1 && haveR) || grp == 3);
  const double* rbase = !loads ? fm + (size_t)i * B
                        : grp == 0 ? Um + (size_t)l * BB + (size_t)cc * B      // T[i,l] = T[l,i]^T
                        : grp == 1 ? Um + (size_t)i * BB + cc                  // T[i,r]
                                   : fm + (size_t)i * B;
  const int rstride = (loads && grp == 1) ? B : 1;
  const bool ident = live && grp == 2;
  auto rhs = [&](int row) -> double {
    const int rc = row < B ? row : B - 1;
    const double v = rbase[rc * rstride];                    // always finite: a valid entry of U or f
    // arithmetic instead of a select: the compiler turns selects back into branches around the load
    return v * ((loads && row < B) ? 1.0 : 0.0) + ((ident && row == cc) ? 1.0 : 0.0);
  };
  double* out = grp == 0 ? Pm + (size_t)i * BB + cc : grp == 1 ? Qm + (size_t)i * BB + cc : grp == 2 ? Gi + (size_t)i * BB + cc
                                                                                              : fm + (size_t)i * B;
  const int ost = grp == 3 ? 1 : B;
  // identity columns are zero above their own row: the whole tile is zero in block rows above its first column
  const int c_lo = 16 * ct, zero_rows = (c_lo >= 2 * B && c_lo + 15 < 3 * B) ? c_lo - 2 * B : 0;
  double ny[NBLK][3];
  // the whole right-hand side tile goes to registers first (3 NBLK values per lane), its latency hides
  // under the fill of L below
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb)
#pragma unroll
    for (int v = 0; v < 3; ++v) ny[kb][v] = rhs(12 * kb + lk + 4 * v);
  {
    const double* Lg = Lm + (size_t)i * BB;
    bcrw_fill<B, 1024, 16>(Ls, Lg, tid);                      // all 16 wavefronts fetch, 4 of them compute
    if (LV_LDS)
      for (int e = tid; e < NBLK * 144; e += 1024) Lvs[e] = Lvm[(size_t)i * NBLK * 144 + e];
  }
  __syncthreads();
  if (wave >= 4 || 16 * ct >= ncol) return;
#pragma unroll
  for (int kb = 0; kb < NBLK; ++kb) {
    const int r0 = 12 * kb;
    if (r0 + 12 <= zero_rows) {                              // wave-uniform
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        ny[kb][v] = 0.0;
        const int row = r0 + lk + 4 * v;
        if (live && row < B) out[(size_t)row * ost] = 0.0;
      }
      continue;
    }
    mfma_acc acc = {ny[kb][0], ny[kb][1], ny[kb][2], 0.0};
    // The last block row of a B that is not a multiple of 12 has 6 real rows; rows 6..11 of its A operand lie beyond L.
    // They must be ZERO, not "whatever is there": the zero-padded inverse of the diagonal block multiplies them by 0,
    // and 0 x NaN = NaN (beyond B = 126 "whatever is there" is LDS no kernel of ours initialised - found by the
    // randomised sweep as a solve that failed only after other problems had run on the handle).
    const bool tail = kb == NBLK - 1 && B % 12 != 0;         // compile time (the loop is unrolled)
    const double rowok = (!tail || r0 + lr < B) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < kb; ++j) {
      if (12 * j + 12 <= zero_rows) continue;                // Y_j is zero
      const double* ap = Ls + (tail && r0 + lr >= B ? B - 1 : r0 + lr) * ld + 12 * j + lk;
      const double a0 = tail ? ap[0] * rowok : ap[0], a1 = tail ? ap[4] * rowok : ap[4], a2 = tail ? ap[8] * rowok : ap[8];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, ny[j][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, ny[j][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, ny[j][2], acc, 0, 0, 0);
    }
    const double* lp = Lv + kb * 144 + (lr < 12 ? lr : 11) * 12 + lk;      // rows lr >= 12 repeat row 11: unused rows of y, reads stay inside the block
    mfma_acc y = {0.0, 0.0, 0.0, 0.0};
    y = __builtin_amdgcn_mfma_f64_16x16x4f64(lp[0], acc[0], y, 0, 0, 0);
    y = __builtin_amdgcn_mfma_f64_16x16x4f64(lp[4], acc[1], y, 0, 0, 0);
    y = __builtin_amdgcn_mfma_f64_16x16x4f64(lp[8], acc[2], y, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      ny[kb][v] = -y[v];
      const int row = r0 + lk + 4 * v;
      if (live && row < B) out[(size_t)row * ost] = y[v];
    }
  }
}

// ---- products on the matrix cores: one workgroup (16 wavefronts) per (node, 64 x 64 output tile), the two
// operand panels [32 rows][64 columns] staged through LDS, next panel in flight during the MFMAs.
// blockIdx.y: tiles of P^T P (lower), then Q^T Q (lower), then P^T Q (all), last: the two vectors.
constexpr int kBcrwPTile = 64, kBcrwPKc = 32;

__global__ __launch_bounds__(1024) void k_bcrw_products(int N, int B, int s, double* __restrict__ Dm,
                                                        double* __restrict__ Um, double* __restrict__ fm,
                                                        const double* __restrict__ Pm, const double* __restrict__ Qm,
                                                        const int* __restrict__ info) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  constexpr int T = kBcrwPTile, KC = kBcrwPKc;
  __shared__ double pA[KC * T], pB[KC * T];
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N || *info != 0) return;
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  const size_t BB = (size_t)B * B;
  const int nt = (B + T - 1) / T, nsym = nt * (nt + 1) / 2;
  const int tid = threadIdx.x;
  const int task = blockIdx.y;
  if (task == 2 * nsym + nt * nt) {                          // f_l -= P^T g, f_r -= Q^T g
    for (int c = tid; c < 2 * B; c += 1024) {
      const bool left = c < B;
      if ((left && !haveL) || (!left && !haveR)) continue;
      const double* A = (left ? Pm : Qm) + (size_t)i * BB + (left ? c : c - B);
      double acc = 0.0;
#pragma unroll 8
      for (int k = 0; k < B; ++k) acc += A[(size_t)k * B] * fm[(size_t)i * B + k];
      atomic_add_f64(fm + (size_t)(left ? l : r) * B + (left ? c : c - B), -acc);
    }
    return;
  }
  int which, ti, tj;
  if (task < 2 * nsym) {
    which = task < nsym ? 0 : 1;
    tri_decode(task - which * nsym, nt, tj, ti);             // tj <= ti
  } else {
    which = 2;
    ti = (task - 2 * nsym) / nt; tj = (task - 2 * nsym) % nt;
  }
  if ((which == 0 && !haveL) || (which == 1 && !haveR) || (which == 2 && !(haveL && haveR))) return;
  const double* Asrc = (which == 1 ? Qm : Pm) + (size_t)i * BB;      // C = A^T B, A and B stored [k][column]
  const double* Bsrc = (which == 0 ? Pm : Qm) + (size_t)i * BB;
  const int i0 = T * ti, j0 = T * tj;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int wi = wave >> 2, wj = wave & 3;
  const int lrow = tid >> 6, lcol = tid & 63;                 // loader: rows lrow and lrow + 16 of the panel
  double va[2], vb[2];
  auto fetch = [&](int kk) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = kk + lrow + 16 * u;
      va[u] = (k < B && i0 + lcol < B) ? Asrc[(size_t)k * B + i0 + lcol] : 0.0;
      vb[u] = (k < B && j0 + lcol < B) ? Bsrc[(size_t)k * B + j0 + lcol] : 0.0;
    }
  };
  mfma_acc acc = {0.0, 0.0, 0.0, 0.0};
  const bool skip = which != 2 && ti == tj && wj > wi;       // strictly upper sub-tile of a symmetric diagonal tile
  fetch(0);
  for (int kk = 0; kk < B; kk += KC) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pA[(lrow + 16 * u) * T + lcol] = va[u];
      pB[(lrow + 16 * u) * T + lcol] = vb[u];
    }
    __syncthreads();
    if (kk + KC < B) fetch(kk + KC);
    if (!skip) {
#pragma unroll
      for (int q = 0; q < KC / 4; ++q)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pA[(4 * q + lk) * T + 16 * wi + lr], pB[(4 * q + lk) * T + 16 * wj + lr], acc, 0, 0,
                                                   0);
    }
  }
  if (skip) return;
  const int col = j0 + 16 * wj + lr;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = i0 + 16 * wi + lk + 4 * v;
    if (row >= B || col >= B) continue;
    if (which == 2) Um[(size_t)l * BB + (size_t)row * B + col] = -acc[v];                    // new T[l,r]
    else if (col <= row) atomic_add_f64(Dm + (size_t)(which == 0 ? l : r) * BB + (size_t)row * B + col, -acc[v]);
  }
}

// ---- back-substitution level: x_i = G^-T (g - P x_l - Q x_r), matrices read straight from global memory
// (8 lanes per row for the two products, then lanes along the columns of G^-1 so that both passes read
// whole cache lines)
__global__ __launch_bounds__(1024) void k_bcrw_backsolve(int N, int B, int s, const double* __restrict__ fm,
                                                         const double* __restrict__ Pm, const double* __restrict__ Qm,
                                                         const double* __restrict__ Gi, double* __restrict__ x) {
  __shared__ double w[144], xl[144], xr[144], red[8][144];     // B <= 138
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N) return;
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  const size_t BB = (size_t)B * B;
  const int tid = threadIdx.x;
  if (tid < B) {
    w[tid] = fm[(size_t)i * B + tid];
    xl[tid] = haveL ? x[(size_t)l * B + tid] : 0.0;
    xr[tid] = haveR ? x[(size_t)r * B + tid] : 0.0;
  }
  __syncthreads();
  const int q8 = tid & 7;
  for (int k0 = 0; k0 < B; k0 += 128) {                      // 128 rows per pass (B > 128: a second, short pass)
    const int k = k0 + (tid >> 3);
    double a0 = 0.0, a1 = 0.0;
    if (k < B) {
      const double* Pk = Pm + (size_t)i * BB + (size_t)k * B;
      const double* Qk = Qm + (size_t)i * BB + (size_t)k * B;
      if (haveL)
#pragma unroll 4
        for (int c = q8; c < B; c += 8) a0 += Pk[c] * xl[c];
      if (haveR)
#pragma unroll 4
        for (int c = q8; c < B; c += 8) a1 += Qk[c] * xr[c];
    }
    double acc = a0 + a1;
    acc += dpp_pair<0xB1>(acc);
    acc += dpp_pair<0x4E>(acc);
    acc += __shfl_xor(acc, 4);
    if (k < B && q8 == 0) w[k] -= acc;
  }
  __syncthreads();
  const int g = tid >> 7;                                    // x[m] = sum_{kk >= m} Ginv[kk][m] w[kk]
  for (int m = tid & 127; m < B; m += 128) {
    double acc = 0.0;
    const double* Gc = Gi + (size_t)i * BB + m;
    int kk = m + g;
#pragma unroll 4
    for (; kk < B; kk += 8) acc += Gc[kk * B] * w[kk];
    red[g][m] = acc;
  }
  __syncthreads();
  if (tid < B) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][tid];
    x[(size_t)i * B + tid] = t;
  }
}

// ---- ALL back-substitution levels in one launch (round 5; nodes of up to kBcrwFusedBackMaxHB cameras): the scheme of
// k_bcr_backsolve_fused (ba_bcr.h) for nodes whose three matrices do not fit in LDS - every node's workgroup takes a ticket, fetches
// ITS SHARE OF P, Q, G^-1 INTO REGISTERS (16 lanes a row: 2 x 7 entries of each matrix a thread at B = 102) while the levels above are
// still at work, polls the solution entries of its two neighbours one level up (k_bcr_assemble marked x "not yet"), forms x_i and
// publishes it.  `order`: the nodes from the root down, so that a workgroup only waits for workgroups that started before it.
constexpr int kBcrwFusedBackMaxHB = 21;      // (two rounds of 1024 threads at 16 lanes a row: B <= 128)
__device__ __forceinline__ double bcrw_wait_value(const double* p, int* status) {      // (bcr_wait_value of ba_bcr.h: bounded, looks at the status word)
  double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int spins = 0; __double_as_longlong(v) == kBcrNotYet; ++spins) {
    if (spins >= kBcrMaxSpins) { atomicMax(status, kBcrTimedOut); break; }
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    __builtin_amdgcn_s_sleep(2);
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return v;
}

template <int HB>
__global__ __launch_bounds__(1024) void k_bcrw_backsolve_fused(int N, const double* __restrict__ gm, const double* __restrict__ Pm,
                                                               const double* __restrict__ Qm, const double* __restrict__ Gi, double* x,
                                                               const int* __restrict__ order, int* ticket) {
  constexpr int B = 6 * HB, R = (16 * B + 1023) / 1024, E = (B + 15) / 16;
  __shared__ double w[B], xl[B], xr[B];
  __shared__ int my_ticket[2];
  int* status = ticket - kBcrTicketWord;
  const int tid = threadIdx.x;
  if (tid == 0) {                                            // (one thread reads the status word and takes the ticket: see k_bcr_backsolve_fused)
    const int st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    my_ticket[1] = st;
    my_ticket[0] = st != 0 ? 0 : atomicAdd(ticket, 1);
  }
  __syncthreads();
  if (my_ticket[1] != 0) return;                             // a failed elimination leaves solution entries unwritten: nobody may wait for them
  const int i = order[my_ticket[0]];
  const int s = (i + 1) & -(i + 1), l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  constexpr size_t BB = (size_t)B * B;
  const int q = tid & 15;
  // my share of the three matrices: row (or column, for G^-T) task >> 4 of round rd, entries q, q + 16, ...
  double vp[R][E], vq[R][E], vg[R][E];
#pragma unroll
  for (int rd = 0; rd < R; ++rd) {
    const int kraw = (rd * 1024 + tid) >> 4, k = kraw < B ? kraw : B - 1;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int c = q + 16 * e, cc = c < B ? c : 0;
      const double a = Pm[(size_t)i * BB + (size_t)k * B + cc], b = Qm[(size_t)i * BB + (size_t)k * B + cc];
      vp[rd][e] = (c < B && kraw < B && haveL) ? a : 0.0;
      vq[rd][e] = (c < B && kraw < B && haveR) ? b : 0.0;
      const int kk = k + c, kc = kk < B ? kk : k;            // x[m] = sum_{kk >= m} Ginv[kk][m] w[kk]: m = k here
      const double g = Gi[(size_t)i * BB + (size_t)kc * B + k];
      vg[rd][e] = (kk < B && kraw < B) ? g : 0.0;
    }
  }
  if (tid < B) {
    w[tid] = gm[(size_t)i * B + tid];
    xl[tid] = haveL ? bcrw_wait_value(x + (size_t)l * B + tid, status) : 0.0;
    xr[tid] = haveR ? bcrw_wait_value(x + (size_t)r * B + tid, status) : 0.0;
  }
  __syncthreads();
  double part[R];
#pragma unroll
  for (int rd = 0; rd < R; ++rd) {                           // w -= P xl + Q xr
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int c = q + 16 * e, cc = c < B ? c : 0;
      acc += vp[rd][e] * xl[cc] + vq[rd][e] * xr[cc];
    }
    acc += dpp_pair<0xB1>(acc);
    acc += dpp_pair<0x4E>(acc);
    acc += dpp_pair<0x141>(acc);
    acc += dpp_pair<0x140>(acc);
    part[rd] = acc;
  }
#pragma unroll
  for (int rd = 0; rd < R; ++rd) {
    const int kraw = (rd * 1024 + tid) >> 4;
    if (q == 0 && kraw < B) w[kraw] -= part[rd];
  }
  __syncthreads();
#pragma unroll
  for (int rd = 0; rd < R; ++rd) {                           // x = (G^-1)^T w
    const int mraw = (rd * 1024 + tid) >> 4, m = mraw < B ? mraw : B - 1;
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int kk = m + q + 16 * e;
      acc += vg[rd][e] * w[kk < B ? kk : m];
    }
    acc += dpp_pair<0xB1>(acc);
    acc += dpp_pair<0x4E>(acc);
    acc += dpp_pair<0x141>(acc);
    acc += dpp_pair<0x140>(acc);
    if (q == 0 && mraw < B) __hip_atomic_store(x + (size_t)i * B + m, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace ba
