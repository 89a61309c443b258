// ba_schur_kernels.h - damping + inversion of the point blocks and the Schur reduction into the reduced camera system
// (bundle_adjuster.py:238-278): the vector forms (k_schur_pairs, k_schur_groups), the producer / consumer matrix-core form for
// runs of points with identical camera lists (k_schur_groups_mfma2) and the dense-visibility reduction (k_dense_*).
// The window-group kernels are in ba_schur_window_kernels.h.  gfx950 (MI355X, CDNA4).
#pragma once

#include <type_traits>

#include "ba_device.h"
#include "ba_point_blocks.h"

namespace ba {

// (point_invert_values / point_invert_body / schur_init_body: ba_point_blocks.h - the group lineariser of a trial runs them too)
__global__ __launch_bounds__(kBlock) void k_point_invert(int nt, const double* __restrict__ HPP,
                                                         double damping, double rcond,
                                                         double* __restrict__ HPPinv,
                                                         int* __restrict__ singular_count,
                                                         int* __restrict__ next_count, const double* __restrict__ bP = nullptr,
                                                         double* __restrict__ fac = nullptr) {
  point_invert_body(blockIdx.x * kBlock + threadIdx.x, nt, HPP, damping, rcond, HPPinv, singular_count, next_count, bP, fac);
}

__global__ __launch_bounds__(kBlock) void k_schur_init(int nco, int hb1, const int* __restrict__ opt_cam,
                                                       const double* __restrict__ HCC,
                                                       const double* __restrict__ bC, double damping,
                                                       double* __restrict__ S, double* __restrict__ b, int use_hcc) {
  schur_init_body((long long)blockIdx.x * kBlock + threadIdx.x, nco, hb1, opt_cam, HCC, bC, damping, S, b, use_hcc);
}

// both of the above in ONE launch (they are independent; a dependent launch costs ~3 us on the stream):
// the first nbi blocks invert the point blocks, the rest initialise [S | b]
__global__ __launch_bounds__(kBlock) void k_point_invert_schur_init(int nbi, int nt, const double* __restrict__ HPP,
                                                                    double damping, double rcond,
                                                                    double* __restrict__ HPPinv,
                                                                    int* __restrict__ singular_count,
                                                                    int* __restrict__ next_count, int nco, int hb1,
                                                                    const int* __restrict__ opt_cam,
                                                                    const double* __restrict__ HCC,
                                                                    const double* __restrict__ bC,
                                                                    double* __restrict__ S, double* __restrict__ b,
                                                                    int use_hcc, const double* __restrict__ bP,
                                                                    double* __restrict__ fac) {
  if ((int)blockIdx.x < nbi)
    point_invert_body(blockIdx.x * kBlock + threadIdx.x, nt, HPP, damping, rcond, HPPinv, singular_count, next_count, bP, fac);
  else
    schur_init_body((long long)(blockIdx.x - nbi) * kBlock + threadIdx.x, nco, hb1, opt_cam, HCC, bC, damping, S, b, use_hcc);
}

// --------------------------------------------------------------------------
// compute_schur_complement, the reduction (bundle_adjuster.py:267-276):
//   b[i]   -= W_ik HPPinv_k bP_k
//   S[i,j] -= W_ik HPPinv_k W_jk^T     over the observation pairs of each point.
// Work unit = (point k, row tile r, col tile c >= r) of at most kTile x kTile
// observation pairs; one wavefront handles a unit.  Phase A: lanes recompute W for the
// tile's observations and stage T = W HPPinv (rows) and W (cols) in LDS.  Phase B: the 64
// lanes walk the (pair, entry) list so that 36 consecutive lanes hit the 36 contiguous
// doubles of one 6x6 block.  Only the upper block triangle (pos_i <= pos_j) is
// accumulated; S is symmetric.
//
// Where the products go: a workgroup owns a CHUNK of consecutive units.  Points are
// sorted, so a chunk only touches cameras in a narrow window [p0, p0 + wn) of
// optimised positions; the workgroup keeps that slice of the block band
// (wn rows x (hb+1) blocks, plus b) as an LDS tile, accumulates into it with LDS fp64
// atomics (ds_add_f64) and flushes the tile to HBM once, with one global atomic per
// touched entry.  That turns ~2000 global atomics per point into ~20.  Products that
// fall outside the window (possible for arbitrary scenes) go straight to global
// atomics, so the result never depends on the chunking.  wn == 0 disables the tile
// (bands too wide for LDS, e.g. dense co-visibility).
// --------------------------------------------------------------------------

__global__ __launch_bounds__(kSchurBlock) void k_schur_pairs(DevProblem P, const double* __restrict__ cams,
                                                        const double* __restrict__ X,
                                                        const SchurUnit* __restrict__ units,
                                                        const SchurChunk* __restrict__ chunks, int wn,
                                                        const double* __restrict__ HPPinv,
                                                        const double* __restrict__ bP,
                                                        double* __restrict__ S, double* __restrict__ b) {
  constexpr int NW = kSchurBlock / kWave;
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* sT = dyn;                                   // [NW][kTile][18]
  double* sW = sT + NW * kTile * 18;                  // [NW][kTile][18]
  int* sPosR = reinterpret_cast<int*>(sW + NW * kTile * 18);   // [NW][kTile]
  int* sPosC = sPosR + NW * kTile;                    // [NW][kTile]
  double* tile = reinterpret_cast<double*>(sPosC + NW * kTile);  // [wn][hb1*36] then tb [wn][6]
  const int hb1 = P.hb + 1;
  const int rowlen = hb1 * 36;
  double* tb = tile + (size_t)wn * rowlen;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const SchurChunk ck = chunks[blockIdx.x];
  const int p0 = ck.p0;
  for (int i = threadIdx.x; i < wn * (rowlen + 6); i += kSchurBlock) tile[i] = 0.0;
  double* mT = sT + wv * kTile * 18;
  double* mW = sW + wv * kTile * 18;
  int* mPosR = sPosR + wv * kTile;
  int* mPosC = sPosC + wv * kTile;
  __syncthreads();

  for (int u = ck.begin + wv; u < ck.end; u += NW) {     // wave-uniform loop
    const SchurUnit un = units[u];
    const int k = un.pt;
    const int s = P.pt_off[k];
    const int L = P.pt_off[k + 1] - s;
    const int nr = min(kTile, L - un.row0), ncol = min(kTile, L - un.col0);
    const bool diag = un.row0 == un.col0;
    // ---- phase A: lanes [0, nr) stage rows, lanes [32, 32+ncol) stage columns
    {
      const bool isRow = lane < 32;
      const int idx = isRow ? lane : lane - 32;
      const int cnt = isRow ? nr : ncol;
      if (idx < cnt && !(diag && !isRow)) {              // diagonal tile: cols = rows
        const double x[3] = {X[3 * (size_t)k], X[3 * (size_t)k + 1], X[3 * (size_t)k + 2]};
        double A[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) A[i] = HPPinv[6 * (size_t)k + i];
        const int n = s + (isRow ? un.row0 : un.col0) + idx;
        const int c = P.obs_cam[n];
        const double2 z = P.obs_z[n];
        double cm[12], e[2], r[2], Jc[12], Jp[6], W[18];
        load_cam(cams, c, cm);
        obs_linearize<true>(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
        block_W(Jc, Jp, W);
        const int pos = P.cam_opt_pos[c];
        if (isRow) {
          double T[18];
          block_T(W, A, T);
#pragma unroll
          for (int i = 0; i < 18; ++i) mT[idx * 18 + i] = T[i];
          mPosR[idx] = pos;
          if (diag) {
#pragma unroll
            for (int i = 0; i < 18; ++i) mW[idx * 18 + i] = W[i];
            mPosC[idx] = pos;
            if (pos >= 0) {   // b[i] -= T_i bP_k, once per observation
              const double g0 = bP[3 * (size_t)k], g1 = bP[3 * (size_t)k + 1], g2 = bP[3 * (size_t)k + 2];
              const int wr = pos - p0;
              const bool in = wr >= 0 && wr < wn;
#pragma unroll
              for (int a = 0; a < 6; ++a) {
                const double v = -(T[a * 3] * g0 + T[a * 3 + 1] * g1 + T[a * 3 + 2] * g2);
                if (in) atomic_add_f64(tb + wr * 6 + a, v);
                else atomic_add_f64(b + (size_t)pos * 6 + a, v);
              }
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < 18; ++i) mW[idx * 18 + i] = W[i];
          mPosC[idx] = pos;
        }
      }
    }
    lds_wave_sync();                                    // staging of this wavefront is visible to its lanes
    // ---- phase B: one lane per (pair (i,j), block row a): 3 + 18 staged values feed 18 FMAs and
    //      6 accumulations into consecutive (or stride-6, when transposed) entries of one block
    {
      const int npairs = diag ? nr * (nr + 1) / 2 : nr * ncol;
      const int items = npairs * 6;
      for (int q = lane; q < items; q += 64) {
        const int pr = q / 6, a = q - pr * 6;
        int i, j;
        if (diag) tri_decode(pr, nr, i, j);
        else { i = pr / ncol; j = pr - i * ncol; }
        const int pi = mPosR[i], pj = mPosC[j];
        if (pi < 0 || pj < 0) continue;
        const double t0 = mT[i * 18 + a * 3], t1 = mT[i * 18 + a * 3 + 1], t2 = mT[i * 18 + a * 3 + 2];
        double v[6];
#pragma unroll
        for (int c = 0; c < 6; ++c)
          v[c] = t0 * mW[j * 18 + c * 3] + t1 * mW[j * 18 + c * 3 + 1] + t2 * mW[j * 18 + c * 3 + 2];
        // block (pi,pj) row a; keep the upper block triangle (transpose when pi > pj)
        const bool up = pi <= pj;
        const int lo = up ? pi : pj, dd = up ? pj - pi : pi - pj;
        const int e0 = up ? a * 6 : a, es = up ? 1 : 6;
        const int wr = lo - p0;
        if (wr >= 0 && wr < wn) {
          double* dst = tile + (size_t)wr * rowlen + dd * 36 + e0;
#pragma unroll
          for (int c = 0; c < 6; ++c) atomic_add_f64(dst + c * es, -v[c]);
        } else {
          double* dst = S + ((size_t)lo * hb1 + dd) * 36 + e0;
#pragma unroll
          for (int c = 0; c < 6; ++c) atomic_add_f64(dst + c * es, -v[c]);
        }
      }
    }
    lds_wave_sync();                                    // all reads of the staging done before it is overwritten
  }
  if (wn == 0) return;
  __syncthreads();
  // ---- flush the tile: one global atomic per touched entry
  for (int i = threadIdx.x; i < wn * rowlen; i += kSchurBlock) {
    const double v = tile[i];
    const int wr = i / rowlen;
    if (v != 0.0 && p0 + wr < P.nco) atomic_add_f64(S + (size_t)(p0 + wr) * rowlen + (i - wr * rowlen), v);
  }
  for (int i = threadIdx.x; i < wn * 6; i += kSchurBlock) {
    const double v = tb[i];
    if (v != 0.0 && p0 + i / 6 < P.nco) atomic_add_f64(b + (size_t)p0 * 6 + i, v);
  }
}

// --------------------------------------------------------------------------
// The same reduction for scenes whose consecutive points share their camera list (image
// sequences: ~100 points per camera step at config 3).  A GROUP = up to kGroupMaxPts
// consecutive points with identical observation lists; one wavefront owns a group:
//   * lane p owns the pair (i, j) of the list (R rounds when there are more than 64 pairs)
//     and keeps the whole 6x6 block  sum_k W_ik HPPinv_k W_jk^T  in 36 registers while it
//     walks the group's points - ONE accumulation into S per block per group instead of
//     one per point, and 36 LDS values feed 162 FMAs;
//   * the points are linearised 64/L at a time so that all lanes work in phase A.
// Accumulation target and window logic as in k_schur_pairs.  Requires track length <= 15.
// --------------------------------------------------------------------------

template <int R>
__global__ __launch_bounds__(kGroupBlock) void k_schur_groups(DevProblem P, const double* __restrict__ cams,
                                                              const double* __restrict__ X,
                                                              const SchurGroup* __restrict__ groups,
                                                              const SchurChunk* __restrict__ chunks, int wn,
                                                              const double* __restrict__ HPPinv,
                                                              const double* __restrict__ bP,
                                                              double* __restrict__ S, double* __restrict__ b) {
  constexpr int NW = kGroupBlock / kWave;
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* sW = dyn;                                   // [NW][64][18]  W of the staged observations
  double* sA = sW + NW * 64 * 18;                     // [NW][64][6]   HPPinv of the staged points
  int* sPos = reinterpret_cast<int*>(sA + NW * 64 * 6);   // [NW][16]  optimised positions of the group's cameras
  double* tile = reinterpret_cast<double*>(sPos + NW * 16);
  const int hb1 = P.hb + 1;
  const int rowlen = hb1 * 36;
  double* tb = tile + (size_t)wn * rowlen;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const SchurChunk ck = chunks[blockIdx.x];
  const int p0 = ck.p0;
  for (int i = threadIdx.x; i < wn * (rowlen + 6); i += kGroupBlock) tile[i] = 0.0;
  double* mW = sW + wv * 64 * 18;
  double* mA = sA + wv * 64 * 6;
  int* mPos = sPos + wv * 16;
  __syncthreads();

  for (int g = ck.begin + wv; g < ck.end; g += NW) {       // wave-uniform
    const SchurGroup gr = groups[g];
    const int L = gr.L;
    const int NP = 64 / L;
    const int npairs = L * (L + 1) / 2;
    if (lane < L) mPos[lane] = P.cam_opt_pos[P.obs_cam[P.pt_off[gr.pt_begin] + lane]];
    lds_wave_sync();
    int pi_[R], pj_[R], oi_[R], oj_[R];
    bool act[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int pr = lane + 64 * r;
      int i = 0, j = 0;
      if (pr < npairs) tri_decode(pr, L, i, j);
      oi_[r] = i; oj_[r] = j;
      pi_[r] = mPos[i]; pj_[r] = mPos[j];
      act[r] = pr < npairs && pi_[r] >= 0 && pj_[r] >= 0;
    }
    double acc[R][36];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < 36; ++e) acc[r][e] = 0.0;
    double bacc[6] = {0, 0, 0, 0, 0, 0};
    const int slot = lane / L, oi = lane - slot * L;          // phase A role: (staged point, observation)
    const int mypos = lane < NP * L ? mPos[oi] : -1;

    for (int kb = gr.pt_begin; kb < gr.pt_end; kb += NP) {
      const int np = min(NP, gr.pt_end - kb);
      // ---- phase A: up to 64/L points at once, one observation per lane
      if (slot < np && lane < NP * L) {
        const int k = kb + slot;
        const int n = P.pt_off[k] + oi;
        const double x[3] = {X[3 * (size_t)k], X[3 * (size_t)k + 1], X[3 * (size_t)k + 2]};
        const int c = P.obs_cam[n];
        const double2 z = P.obs_z[n];
        double cm[12], e[2], r[2], Jc[12], Jp[6], W[18];
        load_cam(cams, c, cm);
        obs_linearize(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
        block_W(Jc, Jp, W);
#pragma unroll
        for (int q = 0; q < 18; ++q) mW[lane * 18 + q] = W[q];
        double A[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) A[q] = HPPinv[6 * (size_t)k + q];
        if (oi == 0) {
#pragma unroll
          for (int q = 0; q < 6; ++q) mA[slot * 6 + q] = A[q];
        }
        if (mypos >= 0) {                                     // b[i] -= T_i bP_k
          double T[18];
          block_T(W, A, T);
          const double g0 = bP[3 * (size_t)k], g1 = bP[3 * (size_t)k + 1], g2 = bP[3 * (size_t)k + 2];
#pragma unroll
          for (int a = 0; a < 6; ++a) bacc[a] -= T[a * 3] * g0 + T[a * 3 + 1] * g1 + T[a * 3 + 2] * g2;
        }
      }
      lds_wave_sync();
      // ---- phase B: every lane adds W_i A W_j^T of each staged point to its 6x6 block
      for (int sl = 0; sl < np; ++sl) {
        double A[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) A[q] = mA[sl * 6 + q];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (!act[r]) continue;
          double Wi[18], Wj[18], T[18];
          const double* wi = mW + (sl * L + oi_[r]) * 18;
          const double* wj = mW + (sl * L + oj_[r]) * 18;
#pragma unroll
          for (int q = 0; q < 18; ++q) { Wi[q] = wi[q]; Wj[q] = wj[q]; }
          block_T(Wi, A, T);
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = 0; c < 6; ++c)
              acc[r][a * 6 + c] += T[a * 3] * Wj[c * 3] + T[a * 3 + 1] * Wj[c * 3 + 1] + T[a * 3 + 2] * Wj[c * 3 + 2];
        }
      }
      lds_wave_sync();
    }
    // ---- one accumulation per block per group (upper block triangle; transpose when pi > pj)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!act[r]) continue;
      const int pi = pi_[r], pj = pj_[r];
      const bool up = pi <= pj;
      const int lo = up ? pi : pj, dd = up ? pj - pi : pi - pj;
      const int wr = lo - p0;
      double* dst = (wr >= 0 && wr < wn) ? tile + (size_t)wr * rowlen + dd * 36 : S + ((size_t)lo * hb1 + dd) * 36;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) atomic_add_f64(dst + (up ? a * 6 + c : c * 6 + a), -acc[r][a * 6 + c]);
    }
    if (mypos >= 0) {
      const int wr = mypos - p0;
      double* dst = (wr >= 0 && wr < wn) ? tb + wr * 6 : b + (size_t)mypos * 6;
#pragma unroll
      for (int a = 0; a < 6; ++a) atomic_add_f64(dst + a, bacc[a]);
    }
    lds_wave_sync();                                        // mPos is rewritten by the next group
  }
  if (wn == 0) return;
  __syncthreads();
  for (int i = threadIdx.x; i < wn * rowlen; i += kGroupBlock) {
    const double v = tile[i];
    const int wr = i / rowlen;
    if (v != 0.0 && p0 + wr < P.nco) atomic_add_f64(S + (size_t)(p0 + wr) * rowlen + (i - wr * rowlen), v);
  }
  for (int i = threadIdx.x; i < wn * 6; i += kGroupBlock) {
    const double v = tb[i];
    if (v != 0.0 && p0 + i / 6 < P.nco) atomic_add_f64(b + (size_t)p0 * 6 + i, v);
  }
}

// --------------------------------------------------------------------------
// The group reduction on the fp64 matrix cores.  For the points k of a group (consecutive points that share their cameras),
//     S_window -= sum_k Tstack_k Wstack_k^T ,   Tstack_k = [W_1k A_k; ...; W_Lk A_k]  (6L x 3),  A_k = HPPinv_k
// is ONE matrix product with inner dimension 3 * (#points): per batch of 6 points (K = 18, padded to 20) the observations
// are linearised one per lane, the operands staged K-major in LDS ([k][row], row = 6 * observation + a), and the product
// runs as k-steps x upper tiles of v_mfma_f64_16x16x4_f64 with the accumulators in registers across the whole group.
// (The vector kernels above spend 42 LDS reads and 162 FMA instructions per (pair, point); a batch here costs 40 LDS reads
// and 50 MFMAs.  The first form of this - one wavefront doing both phases, round 1 - is gone: the producer / consumer
// kernels below replaced it, 81 -> 63 us at config 3.)
// --------------------------------------------------------------------------
// The same reduction with the two phases on DIFFERENT wavefronts (producer / consumer), so that the
// vector unit (linearisation) and the matrix core (products) of a SIMD work at the same time:
// a workgroup is 4 producer + 4 consumer wavefronts, pair p = wavefronts p and p + 4, which the
// hardware places on the same SIMD; a pair owns one group at a time and two staging buffers.
//
// What makes two buffers per pair fit in LDS is the symmetric form of the product: with the
// per-point factorisation  HPPinv_k = L_k D_k L_k^T  (unit lower L, diagonal D: k_point_invert
// writes it next to the inverse, ba_math.h sym3_ldl)
//     S_window -= sum_k (Wstack_k L_k) D_k (Wstack_k L_k)^T
// needs ONE staged operand U = W L per observation (the other MFMA operand is the same rows scaled
// by D), and  b -= W (HPPinv_k bP_k)  needs no T either (the vector HPPinv bP comes with the factor).
//   producer  one observation per lane: linearise, U, stage it k-major, b and the camera-block sums
//   consumer  5 k-steps x 10 upper tiles of v_mfma_f64_16x16x4_f64 per batch, then the epilogue
// Hand-over through two counters per pair in LDS (batches staged / batches consumed); LDS executes
// one wavefront's instructions in order, so the data is there when the counter says so.
// fac[k] = {D0, D1, D2, L10, L20, L21, v0, v1, v2}.
// --------------------------------------------------------------------------

#ifdef BA_BCR_PROFILE
// PROFILE builds: when every workgroup of k_schur_groups_mfma2 started and ended (100 MHz wall clock): the rows of S are
// final when the LAST workgroup that touches them ends - how early that is decides what a consumer could overlap with
constexpr int kSchurTraceMax = 4096;
__device__ long long g_schur_trace[2 * kSchurTraceMax];
#endif
__global__ __launch_bounds__(kGm2Block) void k_schur_groups_mfma2(DevProblem P, const double* __restrict__ cams,
                                                                  const double* __restrict__ X,
                                                                  const SchurGroup* __restrict__ groups,
                                                                  const SchurChunk* __restrict__ chunks, int wn,
                                                                  const double* __restrict__ fac,
                                                                  double* __restrict__ S, double* __restrict__ b,
                                                                  double damping, int fuse_cam) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  constexpr int BUF = kGmK * kGmLd;
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* sU = dyn;                                            // [pair][2][kGmK][kGmLd]
  double* sD = sU + kGm2Pairs * 2 * BUF;                       // [pair][2][kGm2DRows]
  int* sPos = reinterpret_cast<int*>(sD + kGm2Pairs * 2 * kGm2DRows);   // [pair][16]
  int* sFlag = sPos + kGm2Pairs * 16;                          // [pair][4]: staged, consumed
  double* sDummy = reinterpret_cast<double*>(sFlag + kGm2Pairs * 4);   // [64]: where the epilogue's masked-out lanes add
  double* tile = sDummy + 64;
  const int hb1 = P.hb + 1;
  const int rowlen = hb1 * 36;
  double* tb = tile + (size_t)wn * rowlen;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int pair = wv & 3;
  const bool producer = wv < kGm2Pairs;
#ifdef BA_BCR_PROFILE
  const long long pkk = clock64();
  if (threadIdx.x == 0 && blockIdx.x < kSchurTraceMax) g_schur_trace[2 * blockIdx.x] = wall_clock64();      // (option solve_trace prints the spread)
#endif
  const SchurChunk ck = chunks[blockIdx.x];
  const int p0 = ck.p0;
  for (int i = threadIdx.x; i < wn * (rowlen + 6); i += kGm2Block) tile[i] = 0.0;
  for (int i = threadIdx.x; i < kGm2Pairs * 2 * (BUF + kGm2DRows); i += kGm2Block) sU[i] = 0.0;   // incl. sD, the zero k rows
  if (threadIdx.x < kGm2Pairs * 4) sFlag[threadIdx.x] = 0;
  __syncthreads();
  int* fStaged = sFlag + pair * 4;
  int* fConsumed = fStaged + 1;
  int nbatch = 0;                                              // batches this pair has handed over so far

#ifdef BA_BCR_PROFILE
  long long pw = 0, pc = 0, pe = 0, pn = 0;                   // cycles: waiting / working / epilogue, batches
  const long long pk0 = clock64();
#endif
  if (producer) {
    for (int g = ck.begin + pair; g < ck.end; g += kGm2Pairs) {
      const SchurGroup gr = groups[g];
      const int L = gr.L;
      const int NP = 64 / L < kGmPts ? 64 / L : kGmPts;
      const int slot = lane / L, oi = lane - slot * L;
      const bool stager = lane < NP * L;
      const int n0 = P.pt_off[gr.pt_begin] + oi;
      const int c = P.obs_cam[stager ? n0 : P.pt_off[gr.pt_begin]];
      const int mypos = stager ? P.cam_opt_pos[c] : -1;
      double cm[12];
      load_cam(cams, c, cm);
      double bacc[6] = {0, 0, 0, 0, 0, 0};
      double hc[21];
#pragma unroll
      for (int q = 0; q < 21; ++q) hc[q] = 0.0;
      struct PointIn { double x[3], f[9]; double2 z; };
      auto fetch = [&](int kb_, PointIn& in) {
        const int k = kb_ + slot;
        if (stager && k < gr.pt_end) {
          in.z = P.obs_z[n0 + (size_t)(k - gr.pt_begin) * L];
#pragma unroll
          for (int q = 0; q < 3; ++q) in.x[q] = X[3 * (size_t)k + q];
#pragma unroll
          for (int q = 0; q < 9; ++q) in.f[q] = fac[9 * (size_t)k + q];
        }
      };
      PointIn nxt;
      fetch(gr.pt_begin, nxt);
      for (int kb = gr.pt_begin; kb < gr.pt_end; kb += NP) {
        const int np = min(NP, gr.pt_end - kb);
        const PointIn cur = nxt;
        fetch(kb + NP, nxt);
        const bool live = stager && slot < np;
        double U[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) U[q] = 0.0;                // a short last batch stages zero k rows
        if (live) {
          double e[2], r[2], Jc[12], Jp[6], W[18];
          obs_linearize(P.K, cm, cur.x, cur.z.x, cur.z.y, P.sensor, e, r, Jc, Jp);
          block_W(Jc, Jp, W);
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            U[a * 3] = W[a * 3] + cur.f[3] * W[a * 3 + 1] + cur.f[4] * W[a * 3 + 2];
            U[a * 3 + 1] = W[a * 3 + 1] + cur.f[5] * W[a * 3 + 2];
            U[a * 3 + 2] = W[a * 3 + 2];
          }
          if (mypos >= 0) {
#pragma unroll
            for (int a = 0; a < 6; ++a) bacc[a] -= W[a * 3] * cur.f[6] + W[a * 3 + 1] * cur.f[7] + W[a * 3 + 2] * cur.f[8];
            if (fuse_cam) {                                     // HCC[i] += Jc^T Jc, b[i] += Jc^T r (k_camera_blocks' work)
              int idx = 0;
#pragma unroll
              for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int c2 = a; c2 < 6; ++c2) hc[idx++] += Jc[a] * Jc[c2] + Jc[6 + a] * Jc[6 + c2];
                bacc[a] += Jc[a] * r[0] + Jc[6 + a] * r[1];
              }
            }
          }
        }
#ifdef BA_BCR_PROFILE
        const long long w0 = clock64();
#endif
        gm2_wait(fConsumed, nbatch - 1);                         // the buffer's previous batch (nbatch - 2) has been read
#ifdef BA_BCR_PROFILE
        pw += clock64() - w0; ++pn;
#endif
        double* mU = sU + (pair * 2 + (nbatch & 1)) * BUF;
        double* mD = sD + (pair * 2 + (nbatch & 1)) * kGm2DRows;
        if (stager) {
          const int so = 3 * slot * kGmLd + 6 * oi;
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int d = 0; d < 3; ++d) mU[so + d * kGmLd + a] = U[a * 3 + d];
          if (oi == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) mD[3 * slot + d] = live ? cur.f[d] : 0.0;
          }
        }
        ++nbatch;
        gm2_post(fStaged, nbatch, lane);
      }
#ifdef BA_BCR_PROFILE
      const long long e0 = clock64();
#endif
      if (mypos >= 0) {
        const int wr = mypos - p0;
        const bool in = wr >= 0 && wr < wn;
        if (in) {
#pragma unroll
          for (int a = 0; a < 6; ++a) atomic_add_f64(tb + wr * 6 + a, bacc[a]);
        } else {
#pragma unroll
          for (int a = 0; a < 6; ++a) atomic_add_f64(b + (size_t)mypos * 6 + a, bacc[a]);
        }
        if (fuse_cam) {                                          // damped camera block onto the diagonal block (stored in full)
          int idx = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int c2 = a; c2 < 6; ++c2) {
              const double v = a == c2 ? hc[idx] * (1.0 + damping) : hc[idx];
              ++idx;
              if (in) {
                atomic_add_f64(tile + wr * rowlen + a * 6 + c2, v);
                if (a != c2) atomic_add_f64(tile + wr * rowlen + c2 * 6 + a, v);
              } else {
                atomic_add_f64(S + (size_t)mypos * rowlen + a * 6 + c2, v);
                if (a != c2) atomic_add_f64(S + (size_t)mypos * rowlen + c2 * 6 + a, v);
              }
            }
          }
        }
      }
#ifdef BA_BCR_PROFILE
      pe += clock64() - e0;
#endif
    }
  } else {
    const int lr = lane & 15, lk = lane >> 4;
    int* mPos = sPos + pair * 16;
    for (int g = ck.begin + pair; g < ck.end; g += kGm2Pairs) {
      const SchurGroup gr = groups[g];
      const int L = gr.L;
      const int NP = 64 / L < kGmPts ? 64 / L : kGmPts;
      const int nts = (6 * L + 15) >> 4;
      const int nb = (gr.pt_end - gr.pt_begin + NP - 1) / NP;
      if (lane < 16) mPos[lane] = lane < L ? P.cam_opt_pos[P.obs_cam[P.pt_off[gr.pt_begin] + lane]] : -1;
      mfma_acc acc[10];
#pragma unroll
      for (int t = 0; t < 10; ++t) acc[t] = mfma_acc{0.0, 0.0, 0.0, 0.0};
      // the batches of the group.  How many 16 x 16 tiles a side the window has is a property of the group: a compile-time
      // constant inside the loop (a uniform `if (tj < nts)` around every MFMA is a scalar branch around every MFMA - fifty a
      // batch, and nothing can be scheduled across them)
      auto consume = [&](auto nts_c) {
        constexpr int NTS = decltype(nts_c)::value;
        for (int ib = 0; ib < nb; ++ib) {
#ifdef BA_BCR_PROFILE
          const long long w0 = clock64();
#endif
          gm2_wait(fStaged, nbatch + 1);
#ifdef BA_BCR_PROFILE
          pw += clock64() - w0; ++pn;
#endif
          const double* mU = sU + (pair * 2 + (nbatch & 1)) * BUF;
          const double* mD = sD + (pair * 2 + (nbatch & 1)) * kGm2DRows;
#pragma unroll
          for (int s4 = 0; s4 < kGmK / 4; ++s4) {
            double ta[4], wb[4];
            const double dk = mD[4 * s4 + lk];
#pragma unroll
            for (int t = 0; t < 4; ++t) wb[t] = mU[(4 * s4 + lk) * kGmLd + 16 * t + lr];
            if (s4 == kGmK / 4 - 1) { ++nbatch; gm2_post(fConsumed, nbatch, lane); }     // everything of this buffer is in registers
#pragma unroll
            for (int t = 0; t < 4; ++t) ta[t] = wb[t] * dk;
            int q = 0;
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
              for (int tj = ti; tj < 4; ++tj, ++q)
                if (tj < NTS) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[ti], wb[tj], acc[q], 0, 0, 0);
          }
        }
      };
      switch (nts) {
        case 1: consume(std::integral_constant<int, 1>{}); break;
        case 2: consume(std::integral_constant<int, 2>{}); break;
        case 3: consume(std::integral_constant<int, 3>{}); break;
        default: consume(std::integral_constant<int, 4>{}); break;
      }
      lds_wave_sync();                                          // mPos
#ifdef BA_BCR_PROFILE
      const long long e0 = clock64();
#endif
      // ---- epilogue: C/D layout lane -> column n = 16 tj + lane%16, register v -> row m = 16 ti + lane/16 + 4 v.
      // Usual case (wave-uniform test): every optimised camera of the group lies inside the workgroup's LDS
      // window.  Then there is ONE unconditional ds_add_f64 per accumulator register (+ one for the mirrored
      // entry in the diagonal tiles): lanes that have nothing to add (frozen cameras, the lower triangle, padding)
      // add to a private dummy slot instead of branching around the instruction - the branchy form below costs
      // ~13 k cycles per group, mostly exec-mask bookkeeping.
      const int mp = lane < 16 ? mPos[lane] : -1;
      const bool allin = __all(mp < 0 || (mp - p0 >= 0 && mp - p0 < wn));
      if (allin) {
        double* dummy = sDummy + lane;
        int colpart[4], pjv[4], jn[4], cn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int n = 16 * t + lr;
          jn[t] = n / 6; cn[t] = n - 6 * jn[t];
          pjv[t] = jn[t] < L ? mPos[jn[t]] : -1;
          colpart[t] = pjv[t] * 36 + cn[t];
        }
        int q = 0;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
          int rowpart[4], pim[4], im[4], am[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int m = 16 * ti + lk + 4 * v;
            im[v] = m / 6; am[v] = m - 6 * im[v];
            pim[v] = im[v] < L ? mPos[im[v]] : -1;
            rowpart[v] = (pim[v] - p0) * rowlen - pim[v] * 36 + am[v] * 6;
          }
#pragma unroll
          for (int tj = ti; tj < 4; ++tj, ++q) {
            if (tj >= nts) continue;
            const int j = jn[tj], c = cn[tj];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int i = im[v], a = am[v];
              const bool both = pim[v] >= 0 && pjv[tj] >= 0;
              const double val = -acc[q][v];
              const int off = rowpart[v] + colpart[tj];
              const bool ok = both && (i < j || (i == j && a <= c));
              atomic_add_f64(ok ? tile + off : dummy, val);
              if (tj <= ti + 1) {                                // compile time: only these tiles can hold a piece of a diagonal
                const bool mirror = both && i == j && a < c;   // block (6 rows of a camera may straddle a tile edge); they are
                atomic_add_f64(mirror ? tile + off + 5 * (c - a) : dummy, val);      // stored in full
              }
            }
          }
        }
      } else {
        int colpart[4], pjv[4], jn[4], cn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int n = 16 * t + lr;
          jn[t] = n / 6; cn[t] = n - 6 * jn[t];
          pjv[t] = jn[t] < L ? mPos[jn[t]] : -1;
          colpart[t] = pjv[t] * 36 + cn[t];
        }
        int q = 0;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
          int rowpart[4], pim[4], im[4], am[4];
          bool inwin[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int m = 16 * ti + lk + 4 * v;
            im[v] = m / 6; am[v] = m - 6 * im[v];
            const int pi = im[v] < L ? mPos[im[v]] : -1;
            const int wr = pi - p0;
            pim[v] = pi;
            inwin[v] = wr >= 0 && wr < wn;
            rowpart[v] = (inwin[v] ? wr * rowlen : pi * rowlen) - pi * 36 + am[v] * 6;
          }
#pragma unroll
          for (int tj = ti; tj < 4; ++tj, ++q) {
            if (tj >= nts) continue;
            const int j = jn[tj], c = cn[tj];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int i = im[v], a = am[v];
              const bool ok = pim[v] >= 0 && pjv[tj] >= 0 && (i < j || (i == j && a <= c));
              if (ok) {
                const double val = -acc[q][v];
                const int off = rowpart[v] + colpart[tj];
                const int mir = off + 5 * (c - a);               // entry (c, a) of the same block
                if (inwin[v]) {
                  atomic_add_f64(tile + off, val);
                  if (i == j && a < c) atomic_add_f64(tile + mir, val);     // diagonal blocks are stored in full
                } else {
                  atomic_add_f64(S + off, val);
                  if (i == j && a < c) atomic_add_f64(S + mir, val);
                }
              }
            }
          }
        }
      }
      lds_wave_sync();                                          // mPos is rewritten by the next group
#ifdef BA_BCR_PROFILE
      pe += clock64() - e0;
#endif
    }
  }
#ifdef BA_BCR_PROFILE
  const long long pk1 = clock64();
  (void)pc;
#endif
  if (wn == 0) return;
  __syncthreads();
  for (int i = threadIdx.x; i < wn * rowlen; i += kGm2Block) {
    const double v = tile[i];
    const int wr = i / rowlen;
    if (v != 0.0 && p0 + wr < P.nco) atomic_add_f64(S + (size_t)(p0 + wr) * rowlen + (i - wr * rowlen), v);
  }
  for (int i = threadIdx.x; i < wn * 6; i += kGm2Block) {
    const double v = tb[i];
    if (v != 0.0 && p0 + i / 6 < P.nco) atomic_add_f64(b + (size_t)p0 * 6 + i, v);
  }
#ifdef BA_BCR_PROFILE
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x < kSchurTraceMax) g_schur_trace[2 * blockIdx.x + 1] = wall_clock64();
  if (blockIdx.x == 100 && lane == 0 && (wv == 0 || wv == 4))
    printf("[k_schur_groups_mfma2 wg 100 %s] batches %lld: total %lld cycles, waiting for the partner %lld, epilogue %lld; workgroup setup %lld, tail (barrier + flush) %lld\n",
           wv == 0 ? "producer" : "consumer", pn, pk1 - pk0, pw, pe, pk0 - pkk, clock64() - pk1);
#endif
}

// --------------------------------------------------------------------------
// Dense visibility (every track seen by most cameras: the reference's own data sets).  There the
// reduction  S -= sum_k Wstack_k HPPinv_k Wstack_k^T  is ONE dense matrix product with inner dimension
// 3 nt, and with the factorised point inverses (HPPinv = L D L^T, see k_schur_groups_mfma2) a symmetric
// one:  S -= Ud^T diag(Dd) Ud,  Ud [3 nt][6 nco]  (row 3 k + d, column 6 pos + a; zero where a camera
// does not see a point),  b -= Ud^T y  with  y_k = D_k L_k^T bP_k.
//   k_dense_stage  one observation per lane: linearise, U = W L, scatter into Ud; Dd, y per point
//   k_dense_syrk   upper 64 x 64 tiles of Ud^T D Ud on the matrix cores, split along the 3 nt rows so
//                  that a 594 x 594 result still fills the chip; partial sums to their own slabs
//   k_dense_apply  S_band -= sum of the slabs;  k_dense_rhs  b -= Ud^T y
// (The first version staged T and W and called the BLAS: a 594 x 594 x 3000 DGEMM ran at 12 TFLOP/s,
// 172 us, and the GEMV for b took another 118.)
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_dense_stage(DevProblem P, const double* __restrict__ cams,
                                                        const double* __restrict__ X,
                                                        const double* __restrict__ fac,
                                                        const double* __restrict__ bP, int M,
                                                        double* __restrict__ Ud, double* __restrict__ Dd,
                                                        double* __restrict__ yd) {
  const long long n = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (n < P.nt) {
    const double* f = fac + 9 * (size_t)n;
    const double g0 = bP[3 * n], g1 = bP[3 * n + 1], g2 = bP[3 * n + 2];
    Dd[3 * n] = f[0]; Dd[3 * n + 1] = f[1]; Dd[3 * n + 2] = f[2];
    yd[3 * n] = f[0] * (g0 + f[3] * g1 + f[4] * g2);          // D L^T bP
    yd[3 * n + 1] = f[1] * (g1 + f[5] * g2);
    yd[3 * n + 2] = f[2] * g2;
  }
  if (n >= P.nobs) return;
  const int c = P.obs_cam[n], k = P.obs_pt[n];
  const int pos = P.cam_opt_pos[c];
  if (pos < 0) return;
  const double2 z = P.obs_z[n];
  const double x[3] = {X[3 * (size_t)k], X[3 * (size_t)k + 1], X[3 * (size_t)k + 2]};
  double cm[12], e[2], r[2], Jc[12], Jp[6], W[18];
  load_cam(cams, c, cm);
  obs_linearize<true>(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
  block_W(Jc, Jp, W);
  const double l10 = fac[9 * (size_t)k + 3], l20 = fac[9 * (size_t)k + 4], l21 = fac[9 * (size_t)k + 5];
  const size_t row = (size_t)3 * k * M + 6 * (size_t)pos;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    Ud[row + a] = W[a * 3] + l10 * W[a * 3 + 1] + l20 * W[a * 3 + 2];
    Ud[row + M + a] = W[a * 3 + 1] + l21 * W[a * 3 + 2];
    Ud[row + 2 * (size_t)M + a] = W[a * 3 + 2];
  }
}


__global__ __launch_bounds__(1024) void k_dense_syrk(int M, int R, int chunk, const double* __restrict__ Ud,
                                                     const double* __restrict__ Dd, double* __restrict__ part) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  __shared__ double pA[kSyrkKc * kSyrkTile], pB[kSyrkKc * kSyrkTile];    // [k][column]: A = D Ud (rows of tile ti), B = Ud (tile tj)
  const int ti = blockIdx.x, tj = blockIdx.y, ks = blockIdx.z;
  if (tj < ti) return;                                                   // upper triangle of tiles
  const int i0 = kSyrkTile * ti, j0 = kSyrkTile * tj;
  const int r0 = ks * chunk, r1 = min(R, r0 + chunk);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int wi = wave >> 2, wj = wave & 3;
  // loader role: two entries of each panel per thread
  const int lrow = tid >> 6, lcol = tid & 63;                            // rows lrow and lrow + 16
  mfma_acc acc = {0.0, 0.0, 0.0, 0.0};
  double va[2], vb[2];
  auto fetch = [&](int kk) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = kk + lrow + 16 * u;
      const bool rok = r < r1;
      const double d = rok ? Dd[r] : 0.0;
      va[u] = (rok && i0 + lcol < M) ? Ud[(size_t)r * M + i0 + lcol] * d : 0.0;
      vb[u] = (rok && j0 + lcol < M) ? Ud[(size_t)r * M + j0 + lcol] : 0.0;
    }
  };
  fetch(r0);
  for (int kk = r0; kk < r1; kk += kSyrkKc) {
    __syncthreads();                                                     // the previous panel has been consumed
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pA[(lrow + 16 * u) * kSyrkTile + lcol] = va[u];
      pB[(lrow + 16 * u) * kSyrkTile + lcol] = vb[u];
    }
    __syncthreads();
    if (kk + kSyrkKc < r1) fetch(kk + kSyrkKc);                          // in flight during the MFMAs
    if (!(ti == tj && wj < wi)) {
#pragma unroll
      for (int s = 0; s < kSyrkKc / 4; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pA[(4 * s + lk) * kSyrkTile + 16 * wi + lr],
                                                   pB[(4 * s + lk) * kSyrkTile + 16 * wj + lr], acc, 0, 0, 0);
    }
  }
  if (ti == tj && wj < wi) return;
  double* out = part + (size_t)ks * M * M;
  const int col = j0 + 16 * wj + lr;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = i0 + 16 * wi + lk + 4 * v;
    if (row < M && col < M) out[(size_t)row * M + col] = acc[v];
  }
}

// S_band(i, j >= i) -= sum over the split slabs of part[.][6 i .. , 6 j ..];  one thread per entry of S
__global__ __launch_bounds__(kBlock) void k_dense_apply(int nco, int hb1, int M, int nsplit, const double* __restrict__ part,
                                                        double* __restrict__ S) {
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long nS = (long long)nco * hb1 * 36;
  if (tid >= nS) return;
  const int e = (int)(tid % 36);
  const long long blk = tid / 36;
  const int d = (int)(blk % hb1), i = (int)(blk / hb1), j = i + d;
  if (j >= nco) return;
  int ea = e / 6, ec = e % 6;
  if (d == 0 && ea > ec) { const int t = ea; ea = ec; ec = t; }            // only the upper triangle of the product is formed
  const size_t off = ((size_t)6 * i + ea) * M + 6 * (size_t)j + ec;
  double sum = 0.0;
  for (int q = 0; q < nsplit; ++q) sum += part[(size_t)q * M * M + off];
  S[tid] -= sum;
}

// b -= Ud^T y: lanes along the columns of Ud (whole cache lines), 32 rows per workgroup, one atomic per thread
__global__ __launch_bounds__(kBlock) void k_dense_rhs(int M, int R, const double* __restrict__ Ud, const double* __restrict__ y,
                                                      double* __restrict__ b) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  const int r0 = blockIdx.y * kDenseRhsRows;
  if (c >= M) return;
  double acc = 0.0;
#pragma unroll 8
  for (int r = r0; r < min(R, r0 + kDenseRhsRows); ++r) acc += Ud[(size_t)r * M + c] * y[r];
  atomic_add_f64(b + c, -acc);
}

}  // namespace ba
