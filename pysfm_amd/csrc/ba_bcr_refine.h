// ba_bcr_refine.h - one step of iterative refinement of the reduced camera solve through the factors the block cyclic
// reduction keeps (ba_bcr.h: G^-1, P, Q of every node), solve_motion_normal_eqns of bundle_adjuster.py:281-312.
//
// The reference solves its reduced system with LAPACK's gesv (bundle_adjuster.py:302-305).  The cyclic reduction multiplies by
// explicit inverses of the nodes' Cholesky factors; on the damped system of a monocular scene at damping 1e-3 (condition number
// 1e13) that leaves a residual ||S x - b|| about 2.5 times LAPACK's.  The cure is the textbook one:
//     r = b - S x          (every product and sum carried in twice the working precision - TwoProduct / TwoSum,
//                           Ogita-Rump-Oishi's Dot2 - so that r is the residual of x and not the round-off of forming it)
//     S d = r              (k_bcr_refine: r goes up and down the SAME elimination tree, two matrix-vector products per node and
//                           direction; nothing is factored again)
//     x += d
// One launch behind the solve, ~38 us at 111 nodes (the solve: 105; twelve dependent hand-overs between workgroups at ~2 us each
// and ~11 us of its own): switched on where the walk is sensitive to the last digits of the solve (option refine = auto: damping
// below 1e-2), always (1) or never (0).
//
// k_bcr_refine is ONE launch for both sweeps, in the manner of k_bcr_eliminate_fused: a workgroup takes a ticket when it starts
// and works on work[ticket] - the residual items, then the N forward items level by level from the leaves up, then the backward
// items from the root down - so it only ever waits for workgroups that started before it.  What crosses workgroups (a node's contributions
// P^T g, Q^T g to its neighbours' right-hand sides, g itself, the corrections d) is written once into a slot of its own that the
// solve's k_bcr_assemble marked "not yet" (relaxed agent-scope accesses; the consumer polls the data itself, ba_bcr.h), and a
// right-hand side adds its contributions up in a fixed order: the same bits every run.  A forward item forms its node's rows of
// the residual itself, before it starts to wait.
#pragma once

#include "ba_bcr.h"

namespace ba {

constexpr int kRefineMaxLevels = 12;        // levels of the elimination tree a right-hand side may collect contributions from (4095 nodes; registers: two per level)

// ---- error-free transformations (no contraction: __dmul_rn / __dadd_rn are never fused)
__device__ __forceinline__ void two_sum(double a, double b, double& s, double& e) {
  s = __dadd_rn(a, b);
  const double bb = __dadd_rn(s, -a);
  e = __dadd_rn(__dadd_rn(a, -__dadd_rn(s, -bb)), __dadd_rn(b, -bb));
}
// (hi, lo) += a b, the product exact (fma), the sum's error kept
__device__ __forceinline__ void dot2_step(double& hi, double& lo, double a, double b) {
  const double p = __dmul_rn(a, b);
  const double ep = __fma_rn(a, b, -p);
  double s, es;
  two_sum(hi, p, s, es);
  hi = s;
  lo = __dadd_rn(lo, __dadd_rn(ep, es));
}
// (hi, lo) += (h2, l2)
__device__ __forceinline__ void dd_add(double& hi, double& lo, double h2, double l2) {
  double s, e;
  two_sum(hi, h2, s, e);
  hi = s;
  lo = __dadd_rn(__dadd_rn(lo, l2), e);
}

#ifndef BA_BCR_TEMPLATES_ONLY
constexpr int kRefineMaxTerms = ((2 * kBcrSplitMaxHB + 1) * 6 + 15) / 16;      // entries of a band row per lane, 16 lanes a row: 11

// r = b - S x for the rows of node I (cb cameras; the band's half-width is hb <= cb) into LDS, sixteen lanes per row: a lane's entries of
// S and x are all on their way before the first is used (a load per term of a dependent chain costs a memory round trip per
// term: 15 us instead of 3 for the stand-alone kernel this once was).  Rows of cameras past the end and masked parameters: 0
// (k_bcr_assemble made them identity rows).
__device__ __forceinline__ void refine_residual_rows(int I, int n1, int hb, int cb, const double* __restrict__ S, const double* __restrict__ b,
                                                     const unsigned char* __restrict__ mask, const double* __restrict__ x, double* __restrict__ out /*LDS [B]*/) {
  const int B = 6 * cb, hb1 = hb + 1, tid = threadIdx.x;
  for (int base = 0; base < 16 * B; base += kBcrElimThreads) {
    const int task = base + tid, rraw = task >> 4, q = task & 15;
    const int r = rraw < B ? rraw : B - 1;                      // (rows past the end repeat the last: DPP sources must be live lanes)
    const int i = I * cb + r / 6, a = r % 6;
    const bool live = i < n1 && (!mask || mask[6 * i + a]);
    const int jlo = i - hb > 0 ? i - hb : 0, jhi = i + hb < n1 - 1 ? i + hb : n1 - 1;
    const int terms = live ? (jhi - jlo + 1) * 6 : 0;
    double sv[kRefineMaxTerms], xv[kRefineMaxTerms];
#pragma unroll
    for (int u = 0; u < kRefineMaxTerms; ++u) {
      const int t = q + 16 * u;
      const bool in = t < terms;
      const int j = in ? jlo + t / 6 : 0, bb = in ? t % 6 : 0;
      const bool use = in && (!mask || mask[6 * j + bb]);       // (a deleted column; its x is zero anyway)
      const int ii = in ? i : 0;
      const int lc = ii < j ? ii : j, hc = ii < j ? j : ii;
      const bool flip = ii > j || (ii == j && a > bb);          // the band holds blocks (lc, hc), the diagonal ones by their upper triangle
      const double s1 = S[band_block(lc, hc, hb1) + (flip ? bb * 6 + a : a * 6 + bb)];
      const double x1 = x[6 * (size_t)j + bb];
      sv[u] = use ? -s1 : 0.0;
      xv[u] = use ? x1 : 0.0;
    }
    double hi = (live && q == 0) ? b[6 * (size_t)i + a] : 0.0, lo = 0.0;
#pragma unroll
    for (int u = 0; u < kRefineMaxTerms; ++u) dot2_step(hi, lo, sv[u], xv[u]);
    dd_add(hi, lo, dpp_pair<0xB1>(hi), dpp_pair<0xB1>(lo));
    dd_add(hi, lo, dpp_pair<0x4E>(hi), dpp_pair<0x4E>(lo));
    dd_add(hi, lo, dpp_pair<0x141>(hi), dpp_pair<0x141>(lo));
    dd_add(hi, lo, dpp_pair<0x140>(hi), dpp_pair<0x140>(lo));
    if (q == 0 && rraw < B) out[r] = live ? __dadd_rn(hi, lo) : 0.0;
  }
}

// bcr_wait_value (ba_bcr.h) with the poll of the value and the look at the status word on their way TOGETHER, and no sleep: a poll
// is one memory round trip instead of two and a nap - with fourteen hand-overs on the critical path (seven levels up, seven
// down) the poll interval is a tenth of the kernel.  One wavefront per workgroup polls.
__device__ __forceinline__ double refine_wait(const double* p, int* status, int nowait = 0) {
  double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (nowait) return 0.0;
  for (int spins = 0; __double_as_longlong(v) == kBcrNotYet; ++spins) {
    if (spins >= kBcrMaxSpins) { atomicMax(status, kBcrTimedOut); break; }
    const double v2 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = v2;
    if (st != 0) break;
  }
  return v;
}

constexpr int kRefineSeg = (6 * kBcrSplitMaxHB + 15) / 16;      // entries of a row or column of a node's matrix per lane, 16 lanes each: 5

// sum over the 16 lanes of a row group; every lane gets it
__device__ __forceinline__ double refine_sum16(double acc) {
  acc += dpp_pair<0xB1>(acc);
  acc += dpp_pair<0x4E>(acc);
  acc += dpp_pair<0x141>(acc);
  acc += dpp_pair<0x140>(acc);
  return acc;
}

// One node's item of a sweep.  1024 threads = 64 groups of 16 lanes; group o (< B, in up to two rounds for B > 64) owns entry o of
// every vector the item forms, its lane q the entries q, q + 16, ... of the matrix rows / columns that entry needs - fetched
// from the LDS copy into REGISTERS before the item starts to wait, so that what follows the arrival of the awaited vector is
// vector traffic only: LDS write, barrier, five LDS reads, the FMAs, four DPP steps.
template <int ROUNDS>      // 16 B / 1024 rounded up: 1 (B <= 64) or 2
__global__ __launch_bounds__(kBcrElimThreads) void k_bcr_refine(int N, int B, int LV, int n1, int hb, const double* __restrict__ S,
                                                                const double* __restrict__ bvec, const unsigned char* __restrict__ mask,
                                                                const double* __restrict__ Pm, const double* __restrict__ Qm,
                                                                const double* __restrict__ Gi, double* rq, double* gq, double* xq, double* slots,
                                                                double* __restrict__ x, const int* __restrict__ work, int* ticket, int nowait /* bit 0: experiment, nobody waits; bit 1: no residual items */) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, ld = B + 1;
  double* MP = sm;                       // [B][ld] P
  double* MQ = MP + (size_t)B * ld;      // [B][ld] Q
  double* MG = MQ + (size_t)B * ld;      // [B][ld] G^-1
  double* w = MG + (size_t)B * ld;       // [B]
  double* xl = w + B;                    // [B]
  double* xr = xl + B;                   // [B]
  int* my_ticket = reinterpret_cast<int*>(xr + B);
  int* status = ticket - kBcrRefineTicketWord;
  const int rows = 6 * n1;
  // (ONE thread reads the status word and takes the ticket: k_bcr_backsolve_fused.  A failed solve leaves nothing to refine.)
  if (tid == 0) {
    const int st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    my_ticket[1] = st;
    my_ticket[0] = st != 0 ? 0 : atomicAdd(ticket, 1);
  }
  __syncthreads();
  if (my_ticket[1] != 0) return;
  const int item = work[my_ticket[0]];
  const int i = item >> 2;
  const bool back = (item & 3) == 1, resid = (item & 3) == 2;
  const int s = (i + 1) & -(i + 1);      // the level that eliminated node i: i = s (2 k + 1) - 1
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  const bool root = !haveL && !haveR;    // (its forward item goes straight on to d = G^-T g: no backward item, one hand-over less)
  const size_t BB = (size_t)B * B;
  const size_t slot_node = (size_t)2 * LV * B;
  int lv = 0;
  while ((1 << lv) < s) ++lv;            // this node's level: contributions come from levels 0 .. lv - 1
  // r_i = this node's rows of b - S x, in twice the working precision: the RESIDUAL items, first in the work list (so that the
  // forward items, which wait for them, only ever wait for workgroups that started before them), one per node but the root
  // (its forward item forms its own: it has time) - off the forward sweep's critical path, where forming r_i sat for 6 us.
  // (x is the solve's: the first correction lands in it after the root's forward item has heard from every node, i.e. after every
  // node's rows have been formed.)
  const bool own_resid = (nowait & 2) != 0;      // no residual items (more nodes than the chip holds workgroups: every forward item forms its own rows)
  if (resid || ((root || own_resid) && !back)) {
    refine_residual_rows(i, n1, hb, B / 6, S, bvec, mask, x, xr);
    __syncthreads();
    if (resid) {
      if (tid < B) __hip_atomic_store(rq + (size_t)i * B + tid, xr[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  // stage P_i, Q_i, G_i^-1 (written by the solve's launches: plain loads), all loads of a thread before its first store
  {
    constexpr int U = (kBcrSplitMaxHB * 6 * kBcrSplitMaxHB * 6 + kBcrElimThreads - 1) / kBcrElimThreads;
    double vp[U], vq[U], vg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + kBcrElimThreads * u;
      const bool in = e < B * B;
      vp[u] = (in && haveL) ? Pm[(size_t)i * BB + e] : 0.0;
      vq[u] = (in && haveR) ? Qm[(size_t)i * BB + e] : 0.0;
      vg[u] = in ? Gi[(size_t)i * BB + e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + kBcrElimThreads * u;
      if (e < B * B) {
        const int rw = e / B, cc = e - rw * B;
        MP[rw * ld + cc] = vp[u]; MQ[rw * ld + cc] = vq[u]; MG[rw * ld + cc] = vg[u];
      }
    }
  }
  __syncthreads();
  const int q = tid & 15;
  // this thread's matrix entries, by round: forward  G^-1[o][c], P[c][o], Q[c][o];  backward  G^-1[c][o], P[o][c], Q[o][c]   (c = q + 16 u)
  double mg[ROUNDS][kRefineSeg], mp[ROUNDS][kRefineSeg], mq[ROUNDS][kRefineSeg];
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    const int oraw = (rd * kBcrElimThreads + tid) >> 4;
    const int o = oraw < B ? oraw : B - 1;
#pragma unroll
    for (int u = 0; u < kRefineSeg; ++u) {
      const int c = q + 16 * u;
      const bool in = c < B;
      const int cc = in ? c : 0;
      const double g_oc = MG[o * ld + cc], g_co = MG[cc * ld + o];
      const double p_oc = MP[o * ld + cc], p_co = MP[cc * ld + o];
      const double q_oc = MQ[o * ld + cc], q_co = MQ[cc * ld + o];
      mg[rd][u] = !in ? 0.0 : back ? (c >= o ? g_co : 0.0) : (c <= o ? g_oc : 0.0);      // (G^-1 is lower triangular)
      mp[rd][u] = !in ? 0.0 : back ? p_oc : p_co;
      mq[rd][u] = !in ? 0.0 : back ? q_oc : q_co;
    }
  }
  if (!back) {
    // ---- forward: f_i = r_i - (what the nodes eliminated before it added), g_i = G^-1 f_i, P^T g -> f_l, Q^T g -> f_r
    if (tid < B) {
      double c[2 * kRefineMaxLevels];
      const double* mine = slots + (size_t)i * slot_node + tid;
      // every slot fetched at once (one round trip when they are all there - all but the last two usually are) ...
#pragma unroll
      for (int u = 0; u < 2 * kRefineMaxLevels; ++u) {
        const int p = u >> 1, t = 1 << p;
        const bool want = p < lv && ((u & 1) ? i + t < N : i - t >= 0);
        c[u] = want ? __hip_atomic_load(mine + (size_t)u * B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
      // ... then, in a fixed order, waited for where needed and subtracted
      double acc = (root || own_resid) ? xr[tid] : refine_wait(rq + (size_t)i * B + tid, status, nowait & 1);
#pragma unroll
      for (int u = 0; u < 2 * kRefineMaxLevels; ++u) {
        if (__double_as_longlong(c[u]) == kBcrNotYet) c[u] = refine_wait(mine + (size_t)u * B, status, nowait & 1);
        acc -= c[u];
      }
      w[tid] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int oraw = (rd * kBcrElimThreads + tid) >> 4;
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < kRefineSeg; ++u) { const int c = q + 16 * u; acc += mg[rd][u] * w[c < B ? c : 0]; }
      acc = refine_sum16(acc);
      if (q == 0 && oraw < B) {
        xl[oraw] = acc;                                            // g_i (xl: free in this sweep)
        if (!root) __hip_atomic_store(gq + (size_t)i * B + oraw, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    if (!root) {
      double* dl = slots + (size_t)(haveL ? l : 0) * slot_node + (size_t)(2 * lv + 1) * B;      // node l hears from i = l + s: its right-hand contributor of level lv
      double* dr = slots + (size_t)(haveR ? r : 0) * slot_node + (size_t)(2 * lv) * B;          // node r hears from i = r - s: its left-hand contributor
#pragma unroll
      for (int rd = 0; rd < ROUNDS; ++rd) {
          const int oraw = (rd * kBcrElimThreads + tid) >> 4;
        double al = 0.0, ar = 0.0;
#pragma unroll
        for (int u = 0; u < kRefineSeg; ++u) {
          const int c = q + 16 * u;
          const double gv = xl[c < B ? c : 0];
          al += mp[rd][u] * gv; ar += mq[rd][u] * gv;
        }
        al = refine_sum16(al); ar = refine_sum16(ar);
        if (q == 0 && oraw < B) {
          if (haveL) __hip_atomic_store(dl + oraw, al, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (haveR) __hip_atomic_store(dr + oraw, ar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      return;
    }
    // the root: d = G^-T g at once (the column entries of G^-1 come from LDS here: one item in 2 N, off nobody's critical path but its own)
    for (int base = 0; base < 16 * B; base += kBcrElimThreads) {
      const int task = base + tid, mraw = task >> 4;
      const int m = mraw < B ? mraw : B - 1;
      double acc = 0.0;
      for (int k = m + q; k < B; k += 16) acc += MG[k * ld + m] * xl[k];
      acc = refine_sum16(acc);
      if (q == 0 && mraw < B) {
        __hip_atomic_store(xq + (size_t)i * B + m, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const size_t row = (size_t)i * B + m;
        if (row < (size_t)rows) x[row] += acc;
      }
    }
    return;
  }
  // ---- backward: d_i = G^-T (g_i - P d_l - Q d_r), x_i += d_i
  if (tid < B) {
    // (g_i has been there since the forward sweep passed this node; the two loads that may have to wait go out together)
    const double g0 = __hip_atomic_load(gq + (size_t)i * B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double a0 = haveL ? __hip_atomic_load(xq + (size_t)l * B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    double a1 = haveR ? __hip_atomic_load(xq + (size_t)r * B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    w[tid] = __double_as_longlong(g0) == kBcrNotYet ? refine_wait(gq + (size_t)i * B + tid, status, nowait & 1) : g0;
    if (__double_as_longlong(a0) == kBcrNotYet) a0 = refine_wait(xq + (size_t)l * B + tid, status, nowait & 1);
    if (__double_as_longlong(a1) == kBcrNotYet) a1 = refine_wait(xq + (size_t)r * B + tid, status, nowait & 1);
    xl[tid] = a0;
    xr[tid] = a1;
  }
  __syncthreads();
  double wv[ROUNDS];
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {                                 // w -= P d_l + Q d_r
    wv[rd] = 0.0;
    const int oraw = (rd * kBcrElimThreads + tid) >> 4;
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < kRefineSeg; ++u) { const int c = q + 16 * u, cc = c < B ? c : 0; acc += mp[rd][u] * xl[cc] + mq[rd][u] * xr[cc]; }
    acc = refine_sum16(acc);
    wv[rd] = w[oraw < B ? oraw : B - 1] - acc;
  }
  __syncthreads();
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    const int oraw = (rd * kBcrElimThreads + tid) >> 4;
    if (q == 0 && oraw < B) w[oraw] = wv[rd];
  }
  __syncthreads();
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {                                 // d = (G^-1)^T w
    const int oraw = (rd * kBcrElimThreads + tid) >> 4;
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < kRefineSeg; ++u) { const int c = q + 16 * u; acc += mg[rd][u] * w[c < B ? c : 0]; }
    acc = refine_sum16(acc);
    if (q == 0 && oraw < B) {
      __hip_atomic_store(xq + (size_t)i * B + oraw, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const size_t row = (size_t)i * B + oraw;
      if (row < (size_t)rows) x[row] += acc;
    }
  }
}
#endif

}  // namespace ba
