// ba_schur_window_kernels.h - the Schur reduction (bundle_adjuster.py:259-278) for scenes whose points do NOT come in runs
// with identical camera lists: window groups on the fp64 matrix cores (k_schur_groups_mfma3, spans up to 40 cameras;
// k_schur_wide_mfma for the groups of 27 .. 40) and the rectangular products between the segments of longer tracks
// (k_schur_rect_mfma).  gfx950 (MI355X, CDNA4).
#pragma once

#include <type_traits>

#include "ba_device.h"

namespace ba {

// The producer / consumer reduction for ANY track length up to kGm3MaxL = 24 (ragged runs included): the
// 6L x 6L window of a group is NT = ceil(6L / 16) tiles on a side (up to 9), its upper triangle up to 45
// tiles - more accumulators than one wavefront has registers for beyond NT = 6.  So the kernel is a template
// on a range of tile COLUMNS [TJ0, TJ1): a launch forms the tiles (ti <= tj, TJ0 <= tj < TJ1) of every group
// (at most 15 tiles = 120 accumulator registers: the producer half of the kernel needs ~237 VGPRs, and 18 or 21
// tiles on the consumer side spill), and the host covers the window with one launch (NT <= 5: L <= 13) or two
// to four (NT = 6 ... 9), each of which linearises the observations again.  Only the first launch of a set adds
// the right-hand side and the camera blocks (do_rhs).
// Differences from k_schur_groups_mfma2, which this kernel contains as its <0, 4> instance:
//   * staged rows are Ld = 16 NT doubles long, a buffer holds Kbuf >= 4 ceil(3 NP / 4) k-rows, NP = points per
//     batch = min(64 / L, 6, np_cap) - sized by the host so that four pairs of buffers fit in LDS;
//   * a group runs ceil(3 NP / 4) k-steps, not always five; the k-rows of a short last step carry D = 0;
//   * the LDS accumulation window is optional (wn = 0 when hb is too wide for it: every group then adds its
//     window straight to S with global atomics, one per entry per ~100 points).
// --------------------------------------------------------------------------

template <int TJ0, int TJ1, int LDC = 0, int KSC = 0>      // LDC / KSC != 0: staged row length / k-steps per batch known at compile time
__global__ __launch_bounds__(kGm2Block) void k_schur_groups_mfma3(DevProblem P, const double* __restrict__ cams,
                                                                  const double* __restrict__ X,
                                                                  const WinGroup* __restrict__ groups,
                                                                  const int* __restrict__ wtab,
                                                                  const int* __restrict__ opt_cam,
                                                                  const SchurChunk* __restrict__ chunks, Gm3Params G,
                                                                  const double* __restrict__ fac,
                                                                  double* __restrict__ S, double* __restrict__ b,
                                                                  double damping, int fuse_cam) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  constexpr int NTILE = gm3_ntiles(TJ0, TJ1);
  static_assert(NTILE <= kGm3MaxTiles, "too many accumulator tiles for one wavefront");
  const int Ld = LDC ? LDC : G.Ld, BUF = G.Kbuf * Ld, wn = G.wn;
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* sU = dyn;                                            // [pair][2][Kbuf][Ld]
  double* sD = sU + kGm2Pairs * 2 * BUF;                       // [pair][2][kGm2DRows]
  int* sPos = reinterpret_cast<int*>(sD + kGm2Pairs * 2 * kGm2DRows);   // [pair][kGm3PosLen]
  int* sFlag = sPos + kGm2Pairs * kGm3PosLen;                  // [pair][4]: staged, consumed
  double* sDummy = reinterpret_cast<double*>(sFlag + kGm2Pairs * 4);   // [64]: where the epilogue's masked-out lanes add
  double* tile = sDummy + 64;
  const int hb1 = P.hb + 1;
  const int rowlen = hb1 * 36;
  const int wrow = G.wb1 * 36;                                 // the LDS window holds the first wb1 blocks of each band row (no group of this launch reaches further)
  double* tb = tile + (size_t)wn * wrow;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int pair = wv & 3;
  const bool producer = wv < kGm2Pairs;
  const SchurChunk ck = chunks[blockIdx.x];
  const int p0 = ck.p0;
  for (int i = threadIdx.x; i < wn * (wrow + 6); i += kGm2Block) tile[i] = 0.0;
  for (int i = threadIdx.x; i < kGm2Pairs * 2 * (BUF + kGm2DRows); i += kGm2Block) sU[i] = 0.0;   // incl. sD
  if (threadIdx.x < kGm2Pairs * 4) sFlag[threadIdx.x] = 0;
  __syncthreads();
  int* fStaged = sFlag + pair * 4;
  int* fConsumed = fStaged + 1;
  int nbatch = 0;                                              // batches this pair has handed over so far

  if (producer) {
    for (int g = ck.begin + pair; g < ck.end; g += kGm2Pairs) {
      const WinGroup gr = groups[g];
      const int L = gr.W;                                       // lanes per point = window columns (cameras lo .. lo + W - 1)
      if (TJ0 > 0 && ((6 * L + 15) >> 4) <= TJ0) continue;      // no tile column of this launch exists for the group (the consumer skips it too)
      const int NP = gm3_np(L, G.np_cap);
      const int ks = KSC ? KSC : (3 * NP + 3) >> 2;
      const int slot = lane / L, oi = lane - slot * L;
      const bool stager = lane < NP * L;
      // a lane keeps ONE camera for the whole group - the one at its window column - and handles whichever points observe it
      const int mypos = (stager && gr.lo + oi < P.nco) ? gr.lo + oi : -1;
      const int c = opt_cam[mypos >= 0 ? mypos : gr.lo];
      double cm[12];
      load_cam(cams, c, cm);
      double bacc[6] = {0, 0, 0, 0, 0, 0};
      double hc[21];
#pragma unroll
      for (int q = 0; q < 21; ++q) hc[q] = 0.0;
      struct PointIn { double x[3], f[9]; double2 z; int n; };
      auto fetch = [&](int kb_, PointIn& in) {
        const int k = kb_ + slot;
        in.n = -1;
        if (stager && k < gr.pt_end) {
          in.n = mypos >= 0 ? wtab[gr.tab + (k - gr.pt_begin) * L + oi] : -1;      // this point's observation in my camera, if any
          in.z = P.obs_z[in.n >= 0 ? in.n : 0];
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
        const bool live = stager && slot < np && cur.n >= 0;
        double U[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) U[q] = 0.0;                // a short last batch, a point that does not see my camera: zero rows
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
          if (mypos >= 0 && G.do_rhs) {
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
        gm2_wait(fConsumed, nbatch - 1);                         // the buffer's previous batch (nbatch - 2) has been read
        double* mU = sU + (pair * 2 + (nbatch & 1)) * BUF;
        double* mD = sD + (pair * 2 + (nbatch & 1)) * kGm2DRows;
        if (stager) {
          const int so = 3 * slot * Ld + 6 * oi;
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int d = 0; d < 3; ++d) mU[so + d * Ld + a] = U[a * 3 + d];
          if (oi == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) mD[3 * slot + d] = slot < np ? cur.f[d] : 0.0;
          }
        }
        if (lane >= 60 && 3 * NP + (lane - 60) < 4 * ks) mD[3 * NP + (lane - 60)] = 0.0;    // k rows that pad the last step
        ++nbatch;
        gm2_post(fStaged, nbatch, lane);
      }
      if (mypos >= 0 && G.do_rhs) {
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
                atomic_add_f64(tile + wr * wrow + a * 6 + c2, v);
                if (a != c2) atomic_add_f64(tile + wr * wrow + c2 * 6 + a, v);
              } else {
                atomic_add_f64(S + (size_t)mypos * rowlen + a * 6 + c2, v);
                if (a != c2) atomic_add_f64(S + (size_t)mypos * rowlen + c2 * 6 + a, v);
              }
            }
          }
        }
      }
    }
  } else {
    const int lr = lane & 15, lk = lane >> 4;
    int* mPos = sPos + pair * kGm3PosLen;
    for (int g = ck.begin + pair; g < ck.end; g += kGm2Pairs) {
      const WinGroup gr = groups[g];
      const int L = gr.W;
      const int NP = gm3_np(L, G.np_cap);
      const int ks = KSC ? KSC : (3 * NP + 3) >> 2;
      const int nts = (6 * L + 15) >> 4;                        // tiles per side that hold rows of THIS group
      if (TJ0 > 0 && nts <= TJ0) continue;
      const int nb = (gr.pt_end - gr.pt_begin + NP - 1) / NP;
      if (lane < kGm3PosLen) mPos[lane] = (lane < L && gr.lo + lane < P.nco) ? gr.lo + lane : -1;
      mfma_acc acc[NTILE];
#pragma unroll
      for (int t = 0; t < NTILE; ++t) acc[t] = mfma_acc{0.0, 0.0, 0.0, 0.0};
      // the batches of the group, the number of tile columns it has in this launch a compile-time constant inside the loop
      // (k_schur_groups_mfma2 has the story: a uniform condition around a load or an MFMA is a scalar branch around it)
      auto consume = [&](auto nts_c) {
        constexpr int NTS = decltype(nts_c)::value;             // min(nts, TJ1) > TJ0
        for (int ib = 0; ib < nb; ++ib) {
          gm2_wait(fStaged, nbatch + 1);
          const double* mU = sU + (pair * 2 + (nbatch & 1)) * BUF;
          const double* mD = sD + (pair * 2 + (nbatch & 1)) * kGm2DRows;
#pragma unroll KSC ? KSC : 1
          for (int s4 = 0; s4 < (KSC ? KSC : ks); ++s4) {
            double ta[NTS], wb[NTS];
            const double dk = mD[4 * s4 + lk];
            const double* row = mU + (4 * s4 + lk) * Ld + lr;
#pragma unroll
            for (int t = 0; t < NTS; ++t) wb[t] = row[16 * t];
            if (s4 == ks - 1) { ++nbatch; gm2_post(fConsumed, nbatch, lane); }     // everything of this buffer is in registers
#pragma unroll
            for (int t = 0; t < NTS; ++t) ta[t] = wb[t] * dk;
            int q = 0;
#pragma unroll
            for (int tj = TJ0; tj < TJ1; ++tj)
#pragma unroll
              for (int ti = 0; ti <= tj; ++ti, ++q)
                if (tj < NTS) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[ti], wb[tj], acc[q], 0, 0, 0);
          }
        }
      };
      const int ntc = nts < TJ1 ? nts : TJ1;
      auto dispatch = [&](auto self, auto c) {
        constexpr int V = decltype(c)::value;
        if constexpr (V >= TJ1) consume(std::integral_constant<int, TJ1>{});
        else if (ntc == V) consume(c);
        else self(self, std::integral_constant<int, V + 1>{});
      };
      dispatch(dispatch, std::integral_constant<int, TJ0 + 1>{});
      lds_wave_sync();                                          // mPos
      // ---- epilogue (see k_schur_groups_mfma2): C/D layout lane -> column n = 16 tj + lane%16, register v -> row
      // m = 16 ti + lane/16 + 4 v; block (i, j), i <= j, at row pos_i, offset (pos_j - pos_i) * 36 + a * 6 + c
      const int mp = lane < kGm3PosLen ? mPos[lane] : -1;
      const bool allin = wn > 0 && __all(mp < 0 || (mp - p0 >= 0 && mp - p0 < wn));
      double* dummy = sDummy + lane;
      int q = 0;
#pragma unroll
      for (int tj = TJ0; tj < TJ1; ++tj) {
        const int n = 16 * tj + lr;
        const int j = n / 6, c = n - 6 * j;
        const int pj = mPos[j];                                  // (j < 64 always: n <= 239)
        const int colpart = pj * 36 + c;
#pragma unroll
        for (int ti = 0; ti <= tj; ++ti, ++q) {
          if (tj >= nts) continue;                               // wave-uniform
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int m = 16 * ti + lk + 4 * v;
            const int i = m / 6, a = m - 6 * i;
            const int pi = mPos[i];
            const bool both = pi >= 0 && pj >= 0;
            const bool ok = both && (i < j || (i == j && a <= c));
            const bool mirror = both && i == j && a < c;       // diagonal blocks are stored in full
            const double val = -acc[q][v];
            if (allin) {
              // ONE unconditional ds_add_f64 per accumulator register (+ one for the mirrored entry in the tiles that can
              // hold a piece of a diagonal block): lanes with nothing to add hit a private dummy slot
              const int off = (pi - p0) * wrow - pi * 36 + a * 6 + colpart;
              atomic_add_f64(ok ? tile + off : dummy, val);
              if (tj <= ti + 1) atomic_add_f64(mirror ? tile + off + 5 * (c - a) : dummy, val);
            } else if (ok) {
              const int wr = pi - p0;
              if (wr >= 0 && wr < wn) {
                const int off = wr * wrow - pi * 36 + a * 6 + colpart;
                atomic_add_f64(tile + off, val);
                if (mirror) atomic_add_f64(tile + off + 5 * (c - a), val);
              } else {
                const size_t off = (size_t)pi * rowlen - pi * 36 + a * 6 + colpart;
                atomic_add_f64(S + off, val);
                if (mirror) atomic_add_f64(S + off + 5 * (c - a), val);
              }
            }
          }
        }
      }
      lds_wave_sync();                                          // mPos is rewritten by the next group
    }
  }
  if (wn == 0) return;
  __syncthreads();
  for (int i = threadIdx.x; i < wn * wrow; i += kGm2Block) {
    const double v = tile[i];
    const int wr = i / wrow;
    if (v != 0.0 && p0 + wr < P.nco) atomic_add_f64(S + (size_t)(p0 + wr) * rowlen + (i - wr * wrow), v);
  }
  for (int i = threadIdx.x; i < wn * 6; i += kGm2Block) {
    const double v = tb[i];
    if (v != 0.0 && p0 + i / 6 < P.nco) atomic_add_f64(b + (size_t)p0 * 6 + i, v);
  }
}

// --------------------------------------------------------------------------
// Window groups of 25 .. 40 cameras (NT = 10 .. 15 tiles per side, 55 .. 120 tiles): more accumulators than ONE consumer
// wavefront holds, so k_schur_groups_mfma3 covers them with one launch per tile column, each linearising every observation
// again (seven launches at W = 32).  Here ONE producer wavefront (the producer of k_schur_groups_mfma3: a lane keeps the
// camera at its window column, NP = 64 / W points per batch) stages for SEVEN consumers that split the group's tiles among
// themselves (column-major list of the tiles ti <= tj, ceil(T / 7) <= 18 consecutive ones each, operands read from LDS per
// tile): one workgroup per group, every observation linearised once.  A template on the tiles per side (the host launches it
// once per tile count that occurs): the consumer's loop over its tiles is straight-line code.  (Two producers and six
// consumers: slower - the consumers are what bounds it.)  Epilogue: global atomics (the window of
// such a group does not fit in LDS beside the staging).
// --------------------------------------------------------------------------

template <int NTS>
__global__ __launch_bounds__(kGwBlock) void k_schur_wide_mfma(DevProblem P, const double* __restrict__ cams, const double* __restrict__ X,
                                                              const WinGroup* __restrict__ groups, const int* __restrict__ glist,
                                                              const int* __restrict__ wtab, const int* __restrict__ opt_cam,
                                                              const double* __restrict__ fac, double* __restrict__ S,
                                                              double* __restrict__ b, double damping, int fuse_cam) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  constexpr int BUF = kGwK * kGwLd;
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* sU = dyn;                                            // [2][kGwK][kGwLd]
  double* sD = sU + 2 * BUF;                                   // [2][kGwK]
  int* sPos = reinterpret_cast<int*>(sD + 2 * kGwK);           // [64]
  int* sFlag = sPos + 64;                                      // [8]: staged, consumed by each of the seven
  int* sTile = sFlag + 8;                                      // [128]: tile q of the column-major list -> ti | tj << 8
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const WinGroup gr = groups[glist[blockIdx.x]];
  const int L = gr.W;                                          // window columns (cameras lo .. lo + W - 1): 25 .. 40
  const int NP = L <= 32 ? 2 : 1;                              // points per batch (a few narrower groups of a scene of wide ones come here too)
  const int ks = NP == 2 ? 2 : 1;                              // k-steps of four rows per batch
  constexpr int nts = NTS, T = nts * (nts + 1) / 2, OWN = gw_own(NTS);      // (6 L + 15) / 16 <= NTS for every group of this launch
  for (int i = threadIdx.x; i < 2 * (BUF + kGwK); i += kGwBlock) sU[i] = 0.0;      // incl. sD and the zero rows
  if (threadIdx.x < 8) sFlag[threadIdx.x] = 0;
  if (threadIdx.x < 64) sPos[lane] = (lane < L && gr.lo + lane < P.nco) ? gr.lo + lane : -1;
  if (threadIdx.x < 128) {
    int q = threadIdx.x, tj = 0;
    while (tj < nts && q > tj) { q -= tj + 1; ++tj; }          // column tj holds tiles ti = 0 .. tj
    sTile[threadIdx.x] = tj < nts ? (q | tj << 8) : -1;
  }
  __syncthreads();
  int* fStaged = sFlag;
  int* fConsumed = sFlag + 1;
  const int rowlen = (P.hb + 1) * 36;
  const int nb = (gr.pt_end - gr.pt_begin + NP - 1) / NP;
  if (wv == 0) {
    const int slot = lane / L, oi = lane - slot * L;
    const bool stager = lane < NP * L;
    // a lane keeps ONE camera for the whole group - the one at its window column - and handles whichever points observe it
    const int mypos = (stager && gr.lo + oi < P.nco) ? gr.lo + oi : -1;
    const int c = opt_cam[mypos >= 0 ? mypos : gr.lo];
    double cm[12];
    load_cam(cams, c, cm);
    double bacc[6] = {0, 0, 0, 0, 0, 0};
    double hc[21];
#pragma unroll
    for (int q = 0; q < 21; ++q) hc[q] = 0.0;
    struct PointIn { double x[3], f[9]; double2 z; int n; };
    auto fetch = [&](int kb_, PointIn& in) {
      const int k = kb_ + slot;
      in.n = -1;
      if (stager && k < gr.pt_end) {
        in.n = mypos >= 0 ? wtab[gr.tab + (k - gr.pt_begin) * L + oi] : -1;      // this point's observation in my camera, if any
        in.z = P.obs_z[in.n >= 0 ? in.n : 0];
#pragma unroll
        for (int q = 0; q < 3; ++q) in.x[q] = X[3 * (size_t)k + q];
#pragma unroll
        for (int q = 0; q < 9; ++q) in.f[q] = fac[9 * (size_t)k + q];
      }
    };
    PointIn nxt;
    fetch(gr.pt_begin, nxt);
    for (int ib = 0; ib < nb; ++ib) {
      const int kb = gr.pt_begin + ib * NP;
      const int np = min(NP, gr.pt_end - kb);
      const PointIn cur = nxt;
      fetch(kb + NP, nxt);
      const bool live = stager && slot < np && cur.n >= 0;
      double U[18];
#pragma unroll
      for (int q = 0; q < 18; ++q) U[q] = 0.0;                  // a short last batch, a point that does not see my camera: zero rows
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
          if (fuse_cam) {                                       // HCC[i] += Jc^T Jc, b[i] += Jc^T r (k_camera_blocks' work)
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
      // the buffer's previous batch (ib - 2) has been read by every consumer
      for (;;) {
        const int v = lane < kGw7 ? __hip_atomic_load(fConsumed + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 0x7fffffff;
        if (__all(v >= ib - 1)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      double* mU = sU + (ib & 1) * BUF;
      double* mD = sD + (ib & 1) * kGwK;
      if (stager) {
        const int so = 3 * slot * kGwLd + 6 * oi;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int d = 0; d < 3; ++d) mU[so + d * kGwLd + a] = U[a * 3 + d];
        if (oi == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) mD[3 * slot + d] = slot < np ? cur.f[d] : 0.0;
        }
      }
      gm2_post(fStaged, ib + 1, lane);                          // (k rows 3 NP .. 4 ks - 1 and their D stay zero: nobody writes them)
    }
    if (mypos >= 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a) atomic_add_f64(b + (size_t)mypos * 6 + a, bacc[a]);
      if (fuse_cam) {                                            // damped camera block onto the diagonal block (stored in full)
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c2 = a; c2 < 6; ++c2) {
            const double v = a == c2 ? hc[idx] * (1.0 + damping) : hc[idx];
            ++idx;
            atomic_add_f64(S + (size_t)mypos * rowlen + a * 6 + c2, v);
            if (a != c2) atomic_add_f64(S + (size_t)mypos * rowlen + c2 * 6 + a, v);
          }
        }
      }
    }
  } else {
    const int cw = wv - 1;
    const int q0 = cw * OWN, cnt = max(0, min(OWN, T - q0));    // my tiles: q0 .. q0 + cnt - 1 (the last consumer: a few less; it forms tile 0 in their place)
    int* mine = fConsumed + cw;
    const int lr = lane & 15, lk = lane >> 4;
    int tcode[OWN];                                             // (wave-uniform: scalar registers)
#pragma unroll
    for (int u = 0; u < OWN; ++u) tcode[u] = __builtin_amdgcn_readfirstlane(u < cnt ? sTile[q0 + u] : 0);
    mfma_acc acc[OWN];
#pragma unroll
    for (int u = 0; u < OWN; ++u) acc[u] = mfma_acc{0.0, 0.0, 0.0, 0.0};
    for (int ib = 0; ib < nb; ++ib) {
      gm2_wait(fStaged, ib + 1);
      const double* mU = sU + (ib & 1) * BUF;
      const double* mD = sD + (ib & 1) * kGwK;
      for (int s4 = 0; s4 < ks; ++s4) {
        const double dk = mD[4 * s4 + lk];
        const double* row = mU + (4 * s4 + lk) * kGwLd + lr;
        constexpr int CH = OWN > 15 ? OWN / 2 : OWN;            // operands of all my tiles at once, of half of them when 18 (registers)
#pragma unroll
        for (int u0 = 0; u0 < OWN; u0 += CH) {
          double ta[CH], wb[CH];
#pragma unroll
          for (int u = 0; u < CH; ++u) { ta[u] = row[16 * (tcode[u0 + u] & 255)]; wb[u] = row[16 * (tcode[u0 + u] >> 8)]; }
          if (s4 == ks - 1 && u0 + CH >= OWN) gm2_post(mine, ib + 1, lane);      // everything of this buffer is in registers
#pragma unroll
          for (int u = 0; u < CH; ++u) ta[u] *= dk;
#pragma unroll
          for (int u = 0; u < CH; ++u) acc[u0 + u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[u], wb[u], acc[u0 + u], 0, 0, 0);
        }
      }
    }
    if (nb == 0) gm2_post(mine, 1, lane);
    // ---- epilogue: C/D layout lane -> column n = 16 tj + lane % 16, register v -> row m = 16 ti + lane / 16 + 4 v;
    // block (i, j), i <= j, at band row pos_i, offset (pos_j - pos_i) * 36 + a * 6 + c; diagonal blocks are stored in full
#pragma unroll
    for (int u = 0; u < OWN; ++u) {
      if (u >= cnt) continue;
      const int ti = tcode[u] & 255, tj = tcode[u] >> 8;
      const int n = 16 * tj + lr, j = n / 6, c = n - 6 * j;
      const int pj = j < 64 ? sPos[j] : -1;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int m = 16 * ti + lk + 4 * v, i = m / 6, a = m - 6 * i;
        const int pi = i < 64 ? sPos[i] : -1;
        const bool both = pi >= 0 && pj >= 0;
        const bool ok = both && (i < j || (i == j && a <= c));
        const bool mirror = both && i == j && a < c;
        const double val = -acc[u][v];
        if (ok && val != 0.0) {
          const size_t off = (size_t)pi * rowlen + (size_t)(pj - pi) * 36 + a * 6 + c;
          atomic_add_f64(S + off, val);
          if (mirror) atomic_add_f64(S + off + 5 * (c - a), val);
        }
      }
    }
  }
}

// --------------------------------------------------------------------------
// Tracks that span MORE cameras than the widest window of k_schur_groups_mfma3 (kGm3MaxSpan): their optimised positions are
// cut along a grid of segments of kRectSeg = 32 positions (192 unknowns = 12 tiles), and what such a point adds to S is a sum
// over the PAIRS (A <= B) of segments it touches:  S[A, B] -= U_A^T D U_B  with the staged operand U = W L per observation.
// This kernel forms these products for groups of (listed) points that touch the same two segments, one group per workgroup:
// TWO producer wavefronts (lane = column of the table row [A | B]: 64 cameras, one point per batch; even / odd points of the
// group, a staging buffer each) stage for SIX consumer wavefronts, each of which keeps kRectTiles row tiles of A x TWO column
// tiles of B in registers for the whole group (24 accumulator tiles: 192 VGPRs) - the linearisation of an observation is paid
// once per pair of segments, not once per tile column, and the consumers' 24 MFMAs per batch hide behind it.
// A == B (the point inside one segment): the B operand is the A half of the staged row, tiles more than one below the
// diagonal are skipped, blocks with pj >= pi are added (diagonal blocks in full); the producer also adds the segment's share of
// the right-hand side and of the camera blocks - every observation of a long point lies in exactly one segment.
// No LDS window: the few long tracks of a video scene are what this is for (2 % of config 3's points seen by 80 cameras:
// 8 ms through the pair kernel's global atomics).
// --------------------------------------------------------------------------

__global__ __launch_bounds__(kRectBlock) void k_schur_rect_mfma(DevProblem P, const double* __restrict__ cams, const double* __restrict__ X,
                                                                const RectGroup* __restrict__ groups, int ngroups,
                                                                const int* __restrict__ rtab, const int* __restrict__ opt_cam,
                                                                const double* __restrict__ fac, double* __restrict__ S,
                                                                double* __restrict__ b, double damping, int fuse_cam) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  constexpr int BUF = 4 * kRectLd;
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* sU = dyn;                                            // [2][4][kRectLd]: three k rows of a point + a zero row
  double* sD = sU + 2 * BUF;                                   // [2][4]
  int* sPos = reinterpret_cast<int*>(sD + 2 * 4);              // [64]
  int* sFlag = sPos + 64;                                      // [8]: staged (one word per buffer), consumed by each of the six
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const RectGroup gr = groups[blockIdx.x];
  for (int i = threadIdx.x; i < 2 * (BUF + 4); i += kRectBlock) sU[i] = 0.0;      // incl. sD and the zero rows
  if (threadIdx.x < 8) sFlag[threadIdx.x] = 0;
  if (threadIdx.x < 64) {
    const int pos = lane < kRectSeg ? gr.loA + lane : gr.loB + (lane - kRectSeg);
    const bool there = lane < kRectSeg ? pos < P.nco : lane - kRectSeg < gr.WB;
    sPos[lane] = there ? pos : -1;
  }
  __syncthreads();
  const bool sym = gr.loA == gr.loB;
  int* fStaged = sFlag;                                        // [2]
  int* fConsumed = sFlag + 2;
  const int hb1 = P.hb + 1, rowlen = hb1 * 36;
  if (wv < 2) {
    // lane = column of the table row: A's cameras (positions loA ..), then B's (A == B: the second half stays empty)
    const int pos = lane < kRectSeg ? gr.loA + lane : gr.loB + (lane - kRectSeg);
    const bool col_ok = lane < kRectSeg ? pos < P.nco : (!sym && lane - kRectSeg < gr.WB);
    const bool rhs = sym && col_ok;                            // this lane's camera: right-hand side (and camera block) of the segment's observations
    const int c = opt_cam[col_ok ? pos : gr.loA];
    double cm[12];
    load_cam(cams, c, cm);
    double bacc[6] = {0, 0, 0, 0, 0, 0};
    double hc[21];
#pragma unroll
    for (int q = 0; q < 21; ++q) hc[q] = 0.0;
    struct PointIn { double x[3], f[9]; double2 z; int n; };
    auto fetch = [&](int q, PointIn& in) {
      in.n = -1;
      if (q < gr.n) {
        const size_t k = (size_t)rtab[gr.pts + q];
        in.n = col_ok ? rtab[gr.tab + q * 2 * kRectSeg + lane] : -1;
        in.z = P.obs_z[in.n >= 0 ? in.n : 0];
#pragma unroll
        for (int v = 0; v < 3; ++v) in.x[v] = X[3 * k + v];
#pragma unroll
        for (int v = 0; v < 9; ++v) in.f[v] = fac[9 * k + v];
      }
    };
    PointIn nxt;
    fetch(wv, nxt);
    for (int q = wv; q < gr.n; q += 2) {
      const PointIn cur = nxt;
      fetch(q + 2, nxt);
      double U[18];
#pragma unroll
      for (int v = 0; v < 18; ++v) U[v] = 0.0;
      if (cur.n >= 0) {
        double e[2], r[2], Jc[12], Jp[6], W[18];
        obs_linearize(P.K, cm, cur.x, cur.z.x, cur.z.y, P.sensor, e, r, Jc, Jp);
        block_W(Jc, Jp, W);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          U[a * 3] = W[a * 3] + cur.f[3] * W[a * 3 + 1] + cur.f[4] * W[a * 3 + 2];
          U[a * 3 + 1] = W[a * 3 + 1] + cur.f[5] * W[a * 3 + 2];
          U[a * 3 + 2] = W[a * 3 + 2];
        }
        if (rhs) {
#pragma unroll
          for (int a = 0; a < 6; ++a) bacc[a] -= W[a * 3] * cur.f[6] + W[a * 3 + 1] * cur.f[7] + W[a * 3 + 2] * cur.f[8];
          if (fuse_cam) {                                       // HCC[i] += Jc^T Jc, b[i] += Jc^T r (k_camera_blocks' work)
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
      // my buffer's previous batch (q - 2) has been read by every consumer
      for (;;) {
        const int v = lane < kRectConsumers ? __hip_atomic_load(fConsumed + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 0x7fffffff;
        if (__all(v >= q - 1)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      double* mU = sU + (q & 1) * BUF;
      double* mD = sD + (q & 1) * 4;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int d = 0; d < 3; ++d) mU[d * kRectLd + 6 * lane + a] = U[a * 3 + d];
      if (lane < 3) mD[lane] = lane == 0 ? cur.f[0] : lane == 1 ? cur.f[1] : cur.f[2];
      gm2_post(fStaged + (q & 1), q + 1, lane);
    }
    if (rhs) {
#pragma unroll
      for (int a = 0; a < 6; ++a) atomic_add_f64(b + (size_t)pos * 6 + a, bacc[a]);
      if (fuse_cam) {                                          // damped camera block onto the diagonal block (stored in full)
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c2 = a; c2 < 6; ++c2) {
            const double v = a == c2 ? hc[idx] * (1.0 + damping) : hc[idx];
            ++idx;
            atomic_add_f64(S + (size_t)pos * rowlen + a * 6 + c2, v);
            if (a != c2) atomic_add_f64(S + (size_t)pos * rowlen + c2 * 6 + a, v);
          }
        }
      }
    }
  } else {
    const int cw = wv - 2, tj0 = 2 * cw;                       // my tile columns of B: tj0, tj0 + 1
    int* mine = fConsumed + cw;
    if (16 * tj0 >= 6 * gr.WB) {                               // (a short last segment has fewer column tiles: nobody waits for me)
      gm2_post(mine, 0x7ffffff0, lane);
      return;
    }
    const int lr = lane & 15, lk = lane >> 4;
    const int* mPos = sPos;
    mfma_acc acc[2][kRectTiles];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < kRectTiles; ++t) acc[u][t] = mfma_acc{0.0, 0.0, 0.0, 0.0};
    const int boff = sym ? 16 * tj0 : 6 * kRectSeg + 16 * tj0;   // A == B: the B operand is the A half
    for (int q = 0; q < gr.n; ++q) {
      gm2_wait(fStaged + (q & 1), q + 1);
      const double* mU = sU + (q & 1) * BUF;
      const double dk = sD[(q & 1) * 4 + lk];                  // (k row 3: D = 0, the row itself is zero)
      const double* row = mU + lk * kRectLd + lr;
      double ta[kRectTiles];
#pragma unroll
      for (int t = 0; t < kRectTiles; ++t) ta[t] = row[16 * t] * dk;
      const double wb0 = row[boff], wb1 = row[boff + 16];
      gm2_post(mine, q + 1, lane);                             // everything of this buffer is in registers
#pragma unroll
      for (int t = 0; t < kRectTiles; ++t) {
        if (!sym || t <= tj0 + 1) acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[t], wb0, acc[0][t], 0, 0, 0);      // (wave-uniform)
        if (!sym || t <= tj0 + 2) acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[t], wb1, acc[1][t], 0, 0, 0);
      }
    }
    // C/D layout: lane -> column n = 16 tj + lane % 16 (of B), register v -> row m = 16 ti + lane / 16 + 4 v (of A);
    // block (i, j), position pi <= pj, sits in band row pi at (pj - pi) * 36 + a * 6 + c
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int tj = tj0 + u;
      const int n = 16 * tj + lr, j = n / 6, cc = n - 6 * j;
      const int pj = j < kRectSeg ? mPos[kRectSeg + j] : -1;
#pragma unroll
      for (int ti = 0; ti < kRectTiles; ++ti) {
        if (sym && ti > tj + 1) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int m = 16 * ti + lk + 4 * v, i = m / 6, a = m - 6 * i;
          const int pi = mPos[i];
          const double val = -acc[u][ti][v];
          if (pi >= 0 && pj >= pi && val != 0.0 && pj - pi <= P.hb)      // (a zero entry: no point of the group sees both cameras)
            atomic_add_f64(S + (size_t)pi * rowlen + (size_t)(pj - pi) * 36 + a * 6 + cc, val);
        }
      }
    }
  }
}
}  // namespace ba
