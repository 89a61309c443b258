// ba_solve.hip - ba_solve_reduced: the reduced camera system on the device - block cyclic reduction (nodes of up to 11 cameras), the one-workgroup band Cholesky, LU nodes, the solve spread over ranks.
#include "ba_internal.h"

#include "ba_bcr.h"
#include "ba_bcr_refine.h"
#include "ba_dist.h"

using namespace ba;

namespace ba {

// Cameras per node of the narrow cyclic reduction: the half-bandwidth - or, for systems of at most kBcrMaxHB cameras (the
// sliding-window caller's 10-camera windows, the reference's own small scenes), ALL of them: one node, one workgroup, one
// 6 nco x 6 nco Cholesky with the node kernel's pivot chain (~10 us where k_band_solve's nco dependent 6 x 6 pivots take 33).
// (With a border, ba_border.h, the band ends at the band cameras: the border cameras behind them are "cameras past the end".)
inline int bcr_node_size(const ba_handle* h) { return h->band_cams() <= kBcrMaxHB ? std::max(1, h->band_cams()) : std::max(1, h->hb); }

// Block cyclic reduction over super-blocks of hb cameras (ba_bcr.h): log2(N) levels, one
// workgroup per eliminated node.  Leaves the solution in h->dC and the status in flags[1].
static int solve_bcr_factor(ba_handle* h, const unsigned char* dmask, double* mark, long long mark_n);

// Does this solve take the refinement step (ba_bcr_refine.h)?  Not with a border (its solve goes on from the factors), not when
// the solve is spread over ranks (ba_dist.h); option refine = auto: where the reduced system is damped less than kRefineBelowDamping.
static bool refine_wanted(const ba_handle* h) {
  if (h->nbc > 0 || h->opt.refine == REFINE_OFF) return false;
  return h->opt.refine == REFINE_ON || h->schur_damping < kRefineBelowDamping;
}

struct RefinePlan { int N, B, LV; size_t slot_doubles, doubles; };
static RefinePlan refine_plan(const ba_handle* h) {
  const int n1 = h->band_cams();
  const int hb = bcr_node_size(h), B = 6 * hb, N = (n1 + hb - 1) / hb;
  int LV = 0;
  for (int s = 1; (N / s + 1) / 2 > 0; s *= 2) ++LV;
  const size_t slot_doubles = (size_t)2 * LV * B;
  return {N, B, LV, slot_doubles, (size_t)N * (3 * B + slot_doubles)};       // [r | g | d | contribution slots]
}

// x += S^-1 (b - S x) through the kept factors of the cyclic reduction (every level's G^-1, P, Q are in bcrG / bcrP / bcrQ whichever
// kernel eliminated it): ONE launch behind the solve (k_bcr_refine); what it waits on was marked by the solve's k_bcr_assemble.
static int refine_bcr(ba_handle* h, const unsigned char* dmask) {
  const RefinePlan rp = refine_plan(h);
  const int N = rp.N, B = rp.B, LV = rp.LV;
  // residual items of their own (off the forward sweep's critical path) while the launch fits the chip about once; beyond that the
  // sweep is bound by rounds of workgroups, not by hand-overs: a third fewer of them when every forward item forms its own rows
  const bool resid_items = 3 * N - 2 <= 2 * h->ncu;
  if (h->bcr_rwork_n != N) {
    std::vector<int> strides, work;
    for (int s = 1; (N / s + 1) / 2 > 0; s *= 2) strides.push_back(s);
    auto is_root = [&](int i) { const int s = (i + 1) & -(i + 1); return i - s < 0 && i + s >= N; };
    for (int i = 0; i < N; ++i)
      if (resid_items && !is_root(i)) work.push_back(4 * i + 2);       // residual items first: nobody they wait for, everybody waits for them
    for (int q = 0; q < LV; ++q)
      for (int k = 0, cnt = (N / strides[q] + 1) / 2; k < cnt; ++k) {
        const int i = strides[q] * (2 * k + 1) - 1;
        if (i < N) work.push_back(4 * i);
      }
    for (int q = LV - 1; q >= 0; --q)
      for (int k = 0, cnt = (N / strides[q] + 1) / 2; k < cnt; ++k) {
        const int i = strides[q] * (2 * k + 1) - 1;
        if (i < N && !is_root(i)) work.push_back(4 * i + 1);      // (the root's forward item goes straight on to its correction)
      }
    if ((int)work.size() != (resid_items ? 3 * N - 2 : 2 * N - 1)) return h->fail(BA_ERR_STATE, "refinement: %d items in the level lists of %d nodes", (int)work.size(), N);
    HIPCHECK(h, h->bcr_rwork.resize(work.size()));
    HIPCHECK(h, hipMemcpyAsync(h->bcr_rwork.p, work.data(), work.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));          // `work` goes out of scope
    h->bcr_rwork_n = N;
  }
  double* rq = h->bcrRm.p, *gq = rq + (size_t)N * B, *xq = gq + (size_t)N * B, *slots = xq + (size_t)N * B;
  const size_t lds = ((size_t)3 * B * (B + 1) + 3 * B + 8) * sizeof(double);
  const bool two_rounds = 16 * B > kBcrElimThreads;
  HIPCHECK(h, ensure_lds_attr(h, two_rounds ? (const void*)k_bcr_refine<2> : (const void*)k_bcr_refine<1>));
  ScopedTimer tm(h, BA_K_BCR_REFINE, 1);
  if (two_rounds)
    hipLaunchKernelGGL(k_bcr_refine<2>, dim3(resid_items ? 3 * N - 2 : 2 * N - 1), dim3(kBcrElimThreads), lds, h->stream, N, B, LV, h->band_cams(), h->hb, h->S, h->b, dmask,
                       h->bcrP.p, h->bcrQ.p, h->bcrG.p, rq, gq, xq, slots, h->dC.p, h->bcr_rwork.p, h->flags.p + 1 + kBcrRefineTicketWord, (h->opt.refine_debug & 1) | (resid_items ? 0 : 2));
  else
    hipLaunchKernelGGL(k_bcr_refine<1>, dim3(resid_items ? 3 * N - 2 : 2 * N - 1), dim3(kBcrElimThreads), lds, h->stream, N, B, LV, h->band_cams(), h->hb, h->S, h->b, dmask,
                       h->bcrP.p, h->bcrQ.p, h->bcrG.p, rq, gq, xq, slots, h->dC.p, h->bcr_rwork.p, h->flags.p + 1 + kBcrRefineTicketWord, (h->opt.refine_debug & 1) | (resid_items ? 0 : 2));
  HIPCHECK(h, hipGetLastError());
  ++h->refined;
  return BA_OK;
}

int solve_bcr(ba_handle* h, const unsigned char* dmask) {
  const RefinePlan rp = refine_plan(h);
  const bool refine = refine_wanted(h) && rp.LV <= kRefineMaxLevels;      // (more than 4095 nodes: the solve stands as it is)
  if (refine) HIPCHECK(h, h->bcrRm.resize(rp.doubles));
  int rc = solve_bcr_factor(h, dmask, refine ? h->bcrRm.p : nullptr, refine ? (long long)rp.doubles : 0);
  if (rc == BA_OK && refine) rc = refine_bcr(h, dmask);
  return rc;
}

static int solve_bcr_factor(ba_handle* h, const unsigned char* dmask, double* mark, long long mark_n) {
  const int n1 = h->band_cams();
  const int hb = bcr_node_size(h), B = 6 * hb, N = (n1 + hb - 1) / hb;      // (hb: cameras per node from here on)
  const size_t BB = (size_t)B * B;
  HIPCHECK(h, h->bcrD.resize(N * BB)); HIPCHECK(h, h->bcrU.resize(N * BB)); HIPCHECK(h, h->bcrP.resize(N * BB));
  HIPCHECK(h, h->bcrQ.resize(N * BB)); HIPCHECK(h, h->bcrG.resize(N * BB));
  HIPCHECK(h, h->bcrF.resize((size_t)N * B));
  HIPCHECK(h, h->bcrGv.resize((size_t)N * B));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_bcr_backsolve));
  // a node over three compute units (k_bcr_eliminate_split) on the levels whose nodes then still fit the chip in one
  // round of workgroups; one compute unit per node (k_bcr_eliminate) on the wide levels below them and with "bcr1".
  // (A split level takes its couplings from the factors of the level below, whichever kernel wrote them; a one-unit
  // level needs the couplings U the split kernel does not form: so never one-unit above split - node counts only fall.)
  // (nodes of 12 and 13 cameras: the one-unit kernel's four matrices do not fit in LDS - every level is a split one)
  const bool wide_node = hb > kBcrMaxHB;
  const bool split = h->opt.solver != SOLVER_BCR1 || wide_node;
  std::vector<char> level_split;
  const size_t lds = bcr_lds_bytes(B);
  std::vector<int> strides;
  for (int s = 1; (N / s + 1) / 2 > 0; s *= 2) strides.push_back(s);
  for (int s : strides) {
    const int cnt = (N / s + 1) / 2;
    level_split.push_back(split && (wide_node || 3 * cnt <= h->ncu || (!level_split.empty() && level_split.back())));
  }
  // the split levels in ONE launch (k_bcr_eliminate_fused): its work list = the (node, role) pairs of those levels, leaves first
  int s_fused = 0, nwork = 0;
  if (split && h->opt.fused_eliminate) {
    for (size_t q = 0; q < strides.size(); ++q)
      if (level_split[q]) { s_fused = strides[q]; break; }
    if (s_fused && !(h->bcr_work_n == N && h->bcr_work_s == s_fused)) {
      std::vector<int> work;
      for (size_t q = 0; q < strides.size(); ++q) {
        if (!level_split[q]) continue;
        const int s = strides[q];
        for (int k = 0, cnt = (N / s + 1) / 2; k < cnt; ++k) {
          const int i = s * (2 * k + 1) - 1;
          if (i >= N) continue;
          if (i - s >= 0) work.push_back(4 * i + 0);
          if (i + s < N) work.push_back(4 * i + 1);
          work.push_back(4 * i + 2);
        }
      }
      h->bcr_work_elim = (int)work.size();
      // ... followed by the back-substitution items, root down (k_bcr_eliminate_fused role 3): the whole solve behind
      // k_bcr_assemble is then ONE launch.
      for (int q = (int)strides.size() - 1; q >= 0; --q)
        for (int k = 0, cnt = (N / strides[q] + 1) / 2; k < cnt; ++k) {
          const int i = strides[q] * (2 * k + 1) - 1;
          if (i < N && (i - strides[q] >= 0 || i + strides[q] < N)) work.push_back(4 * i + 3);
        }
      HIPCHECK(h, h->bcr_work.resize(work.size()));
      HIPCHECK(h, hipMemcpyAsync(h->bcr_work.p, work.data(), work.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
      HIPCHECK(h, hipStreamSynchronize(h->stream));          // `work` goes out of scope
      h->bcr_work_n = N; h->bcr_work_s = s_fused; h->bcr_work_len = (int)work.size();
    }
    const bool back_in_launch = s_fused && h->opt.fused_backsolve && N <= 8 * h->ncu;
    nwork = s_fused ? (back_in_launch ? h->bcr_work_len : h->bcr_work_elim) : 0;
    if (s_fused) HIPCHECK(h, h->bcr_done.resize((size_t)4 * N));
  }
  {
    ScopedTimer tm(h, BA_K_BCR_ASSEMBLE);       // also clears the status word flags[1], marks the solution "not there yet", clears done[]
    hipLaunchKernelGGL(k_bcr_assemble<1>, dim3(N, (B * B + kBcrThreads - 1) / kBcrThreads), dim3(kBcrThreads), 0, h->stream, n1, h->hb, hb, h->S, h->b, dmask, h->bcrD.p,
                       h->bcrU.p, h->bcrF.p, h->flags.p + 1, h->dC.p, s_fused ? h->bcr_done.p : nullptr, (const int*)nullptr, mark, mark_n);
  }
  {
    int launches = 0;
    for (size_t q = 0; q < strides.size(); ++q) launches += (s_fused && level_split[q]) ? (strides[q] == s_fused ? 1 : 0) : 1;
    ScopedTimer tm(h, BA_K_BCR_ELIMINATE, launches);
    for (size_t q = 0; q < strides.size(); ++q) {
      const int s = strides[q], cnt = (N / s + 1) / 2;
      if (s_fused && level_split[q]) {
        if (s == s_fused)
          HIPCHECK(h, launch_bcr_fused(h, hb, nwork, h->stream, N, s_fused, h->bcrD.p, h->bcrU.p, h->bcrF.p, h->bcrP.p, h->bcrQ.p, h->bcrG.p,
                                       h->bcrGv.p, h->flags.p + 1, h->dC.p, h->bcr_work.p, h->bcr_done.p));
      } else if (level_split[q])
        HIPCHECK(h, launch_bcr_split(h, hb, cnt, h->stream, N, s, h->bcrD.p, h->bcrU.p, h->bcrF.p, h->bcrP.p, h->bcrQ.p, h->bcrG.p,
                                     h->bcrGv.p, h->flags.p + 1, h->dC.p));
      else
        HIPCHECK(h, launch_bcr_eliminate(h, hb, cnt, lds, h->stream, N, s, h->bcrD.p, h->bcrU.p, h->bcrF.p, h->bcrP.p, h->bcrQ.p,
                                         h->bcrG.p, h->flags.p + 1, h->dC.p));
    }
  }
  const size_t lds2 = ((size_t)3 * B * (B + 1) + 3 * B + 8) * sizeof(double);
  // a level whose only node has no neighbours (the root) was solved inside its eliminate kernel
  int top = (int)strides.size() - 1;
  if (top >= 0 && (N / strides[top] + 1) / 2 == 1 && 2 * strides[top] - 1 >= N) --top;
  int split_stride = INT32_MAX;                            // the first (smallest-stride) level eliminated by the split kernel
  for (size_t q = 0; q < strides.size(); ++q)
    if (level_split[q]) { split_stride = strides[q]; break; }
  if (s_fused && h->opt.fused_backsolve && N <= 8 * h->ncu) return BA_OK;      // (done inside k_bcr_eliminate_fused)
  if (h->opt.fused_backsolve && N <= 8 * h->ncu && top >= 0) {
    // every node's workgroup is resident at once: all levels in ONE launch, handing x down through flags
    if (h->bcr_order_n != N) {
      std::vector<int> order;
      for (int q = (int)strides.size() - 1; q >= 0; --q)
        for (int k = 0, cnt = (N / strides[q] + 1) / 2; k < cnt; ++k) {
          const int i = strides[q] * (2 * k + 1) - 1;
          if (i < N) order.push_back(i);
        }
      if ((int)order.size() != N) return h->fail(BA_ERR_STATE, "cyclic reduction: %d of %d nodes in the level lists", (int)order.size(), N);
      HIPCHECK(h, h->bcr_order.resize((size_t)N));
      HIPCHECK(h, hipMemcpyAsync(h->bcr_order.p, order.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, h->stream));
      HIPCHECK(h, hipStreamSynchronize(h->stream));          // `order` goes out of scope
      h->bcr_order_n = N;
    }
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_bcr_backsolve_fused));
    ScopedTimer tmb(h, BA_K_BCR_BACKSOLVE, 1);
    hipLaunchKernelGGL(k_bcr_backsolve_fused, dim3(N), dim3(kBcrElimThreads), lds2, h->stream, N, B, h->bcrGv.p, h->bcrF.p,
                       split_stride, h->bcrP.p, h->bcrQ.p, h->bcrG.p, h->dC.p,
                       h->bcr_order.p, h->flags.p + 1 + kBcrTicketWord);
    HIPCHECK(h, hipGetLastError());
    return BA_OK;
  }
  ScopedTimer tmb(h, BA_K_BCR_BACKSOLVE, top + 1);
  for (int q = top; q >= 0; --q) {
    const int s = strides[q], cnt = (N / s + 1) / 2;
    hipLaunchKernelGGL(k_bcr_backsolve, dim3(cnt), dim3(kBcrElimThreads), lds2, h->stream, N, B, s,
                       level_split[q] ? h->bcrGv.p : h->bcrF.p, h->bcrP.p, h->bcrQ.p, h->bcrG.p, h->dC.p);
  }
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

// The cyclic reduction with LU nodes (k_bcr_eliminate_lu): for reduced systems the Cholesky solvers reported as not positive
// definite.  Same layout and back-substitution as solve_bcr; leaves the solution in h->dC and the status in flags[1].
int solve_bcr_lu(ba_handle* h, const unsigned char* dmask, int ncams, const double* rhs) {
  // (ncams, rhs: the band part of a bordered system and any right-hand side of it - ba_border.hip; default: the whole system, b)
  const int n1 = ncams >= 0 ? ncams : h->nco;
  if (!rhs) rhs = h->b;
  const int hb = bcr_node_size(h), B = 6 * hb, N = (n1 + hb - 1) / hb;
  const size_t BB = (size_t)B * B;
  HIPCHECK(h, h->bcrD.resize(N * BB)); HIPCHECK(h, h->bcrU.resize(N * BB)); HIPCHECK(h, h->bcrP.resize(N * BB));
  HIPCHECK(h, h->bcrQ.resize(N * BB)); HIPCHECK(h, h->bcrG.resize(N * BB));
  HIPCHECK(h, h->bcrF.resize((size_t)N * B));
  {
    ScopedTimer tm(h, BA_K_BCR_ASSEMBLE);
    hipLaunchKernelGGL(k_bcr_assemble<3>, dim3(N), dim3(kBcrThreads), 0, h->stream, n1, h->hb, hb, h->S, rhs, dmask, h->bcrD.p,
                       h->bcrU.p, h->bcrF.p, h->flags.p + 1, h->dC.p);
  }
  std::vector<int> strides;
  for (int s = 1; (N / s + 1) / 2 > 0; s *= 2) strides.push_back(s);
  {
    ScopedTimer tm(h, BA_K_BCR_ELIMINATE, (int)strides.size());
    for (int s : strides) {
      const int cnt = (N / s + 1) / 2;
      HIPCHECK(h, launch_bcr_lu(h, hb, cnt, h->stream, N, s, h->bcrD.p, h->bcrU.p, h->bcrF.p, h->bcrP.p, h->bcrQ.p, h->bcrG.p, h->flags.p + 1, h->dC.p));
    }
  }
  const size_t lds2 = ((size_t)3 * B * (B + 1) + 3 * B + 8) * sizeof(double);
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_bcr_backsolve));
  int top = (int)strides.size() - 1;
  if (top >= 0 && (N / strides[top] + 1) / 2 == 1 && 2 * strides[top] - 1 >= N) --top;      // the root solved itself
  ScopedTimer tmb(h, BA_K_BCR_BACKSOLVE, top + 1);
  for (int q = top; q >= 0; --q) {
    const int s = strides[q], cnt = (N / s + 1) / 2;
    hipLaunchKernelGGL(k_bcr_backsolve, dim3(cnt), dim3(kBcrElimThreads), lds2, h->stream, N, B, s, h->bcrF.p, h->bcrP.p, h->bcrQ.p,
                       h->bcrG.p, h->dC.p);
  }
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

// ---- the reduced solve spread over the ranks (ba_dist.h) --------------------------------------------------------------
// Which super-block size cuts nco cameras (band half-width hb) into an elimination tree that splits evenly over nranks = 2^g
// ranks: the smallest number of levels L, then the smallest cb in [hb, kBcrMaxHB], such that every rank's interval has nodes
// (the last one at least a quarter of a full interval) and a rank has at least 8 nodes.  false = does not apply.
bool dist_plan_static(int nco, int hb, int nranks, int* cb_out, int* N_out, int* P_out) {
  if (nranks < 2 || (nranks & (nranks - 1)) || hb < 1 || hb > kBcrMaxHB) return false;
  int bestL = 1 << 30, best_cb = 0, bestN = 0, bestP = 0;
  for (int cb = hb; cb <= kBcrMaxHB; ++cb) {
    const int N = (nco + cb - 1) / cb;
    int L = 0;
    while ((1 << L) - 1 < N) ++L;
    const int P = (1 << L) / nranks;
    if (P < 8) continue;
    const int last = N - (nranks - 1) * P;                  // nodes of the last interval
    if (last < P / 4) continue;
    if (L < bestL) { bestL = L; best_cb = cb; bestN = N; bestP = P; }
  }
  if (!best_cb) return false;
  *cb_out = best_cb; *N_out = bestN; *P_out = bestP;
  return true;
}

int dist_build_plan(ba_handle* h, int rank, int nranks) {
  auto& d = h->dist;
  d.on = false;
  int cb, N, P;
  if (!h->have_problem || h->nco == 0 || !dist_plan_static(h->nco, h->hb, nranks, &cb, &N, &P)) return BA_OK;
  d.rank = rank; d.nranks = nranks; d.cb = cb; d.N = N; d.P = P;
  d.n_lo = rank * P; d.n_hi = std::min(N, rank * P + P - 1);
  d.own_lo = std::min(h->nco, d.n_lo * cb); d.own_hi = std::min(h->nco, (rank + 1) * P * cb);
  const int hb = h->hb, nco = h->nco;
  std::vector<int> rows, sep, sep_owner, root, root_owner, work, order;
  for (int t = P - 1; t < N; t += P) {                       // separators: nodes whose stride is >= P
    sep.push_back(t); sep_owner.push_back((t + 1) / P - 1);
    for (int c = t * cb; c < std::min(nco, (t + 1) * cb + hb); ++c) rows.push_back(c);      // its rows and the first hb behind it
  }
  for (int r = 0; r < nranks; ++r) {
    const int j = r * P + P / 2 - 1;                         // root of rank r's subtree (stride P / 2)
    if (j < N) { root.push_back(j); root_owner.push_back(r); }
  }
  auto push_items = [&](int i, int s) {
    if (i - s >= 0) work.push_back(4 * i + 0);
    if (i + s < N) work.push_back(4 * i + 1);
    work.push_back(4 * i + 2);
  };
  for (int s = 1; s < P; s *= 2)                             // local phase: own interval, leaves first
    for (int i = s - 1; i < N; i += 2 * s)
      if (i >= d.n_lo && i < d.n_hi) push_items(i, s);
  d.nwork_local = (int)work.size();
  int s_top = P;
  while (2 * s_top - 1 < N) s_top *= 2;                      // (the largest stride that has a node)
  for (int s = P; s <= s_top; s *= 2)                        // separator phase (every rank), leaves first
    for (int i = s - 1; i < N; i += 2 * s) push_items(i, s);
  d.nwork_top = (int)work.size() - d.nwork_local;
  for (int s = s_top; s >= 1; s /= 2)                        // back-substitution: separators, then the own interval, root down
    for (int i = s - 1; i < N; i += 2 * s)
      if (s >= P || (i >= d.n_lo && i < d.n_hi)) order.push_back(i);
  std::vector<int> asm_nodes(sep);
  for (int i = d.n_lo; i < d.n_hi; ++i) asm_nodes.push_back(i);
  d.nasm = (int)asm_nodes.size();
  d.nrows = (int)rows.size(); d.nsep = (int)sep.size(); d.nroot = (int)root.size(); d.norder = (int)order.size();
  const size_t B = 6 * (size_t)cb, BB = B * B;
  d.xcount[0] = (size_t)d.nrows * ((size_t)(hb + 1) * 36 + 6);
  d.xcount[1] = (size_t)d.nsep * (BB + B) + (size_t)d.nroot * 2 * BB;
  d.xcount[2] = (size_t)nco * 6;
  auto up = [&](DevBuf<int>& b, const std::vector<int>& v) -> hipError_t {
    if (hipError_t e = b.resize(std::max<size_t>(1, v.size())); e != hipSuccess) return e;
    return v.empty() ? hipSuccess : hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, h->stream);
  };
  HIPCHECK(h, up(d.rows, rows)); HIPCHECK(h, up(d.sep, sep)); HIPCHECK(h, up(d.sep_owner, sep_owner));
  HIPCHECK(h, up(d.root, root)); HIPCHECK(h, up(d.root_owner, root_owner)); HIPCHECK(h, up(d.work, work)); HIPCHECK(h, up(d.order, order));
  HIPCHECK(h, up(d.asm_nodes, asm_nodes));
  HIPCHECK(h, hipStreamSynchronize(h->stream));              // the vectors go out of scope
  const size_t need = std::max(d.xcount[0], std::max(d.xcount[1], d.xcount[2]));
  if (!d.xbuf || d.xcap < need) {
    HIPCHECK(h, d.xown.resize(need));
    d.xbuf = d.xown.p; d.xcap = need;
  }
  d.on = true;
  return BA_OK;
}

int dist_upload_mask(ba_handle* h, const uint8_t* cam_param_mask, const unsigned char** dmask) {
  *dmask = nullptr;
  if (!cam_param_mask) return BA_OK;
  bool all = true;
  for (int i = 0; i < h->nco * 6; ++i) all = all && cam_param_mask[i];
  if (all) return BA_OK;
  const unsigned char* m = cam_rows_in(h, cam_param_mask, h->mask_host, 6);
  HIPCHECK(h, hipMemcpyAsync(h->mask.p, m, (size_t)h->nco * 6, hipMemcpyHostToDevice, h->stream));
  *dmask = h->mask.p;
  return BA_OK;
}

// Stage 1: shared rows of this rank's partial [S | b] -> exchange buffer.  Stage 2 (after the sum): rows back, assemble,
// eliminate the own interval, separators + subtree roots -> buffer.  Stage 3 (after the sum): separators back, eliminate them,
// back-substitute separators + own interval, owned solution entries -> buffer.  Stage 4 (after the sum): the full dC.
int dist_stage(ba_handle* h, int stage, const uint8_t* cam_param_mask, size_t* count) {
  auto& d = h->dist;
  const int cb = d.cb, B = 6 * cb, N = d.N, hb1 = h->hb + 1;
  const size_t BB = (size_t)B * B;
  *count = 0;
  if (stage == 1) {
    hipLaunchKernelGGL(k_dist_rows, dim3(std::max(1u, std::min(1024u, blocks_for((long long)d.xcount[0])))), dim3(256), 0, h->stream, d.nrows,
                       d.rows.p, hb1, h->S, h->b, d.xbuf, 0);
    *count = d.xcount[0];
  } else if (stage == 2) {
    hipLaunchKernelGGL(k_dist_rows, dim3(std::max(1u, std::min(1024u, blocks_for((long long)d.xcount[0])))), dim3(256), 0, h->stream, d.nrows,
                       d.rows.p, hb1, h->S, h->b, d.xbuf, 1);
    HIPCHECK(h, h->bcrD.resize(N * BB)); HIPCHECK(h, h->bcrU.resize(N * BB)); HIPCHECK(h, h->bcrP.resize(N * BB));
    HIPCHECK(h, h->bcrQ.resize(N * BB)); HIPCHECK(h, h->bcrG.resize(N * BB));
    HIPCHECK(h, h->bcrF.resize((size_t)N * B)); HIPCHECK(h, h->bcrGv.resize((size_t)N * B));
    HIPCHECK(h, h->bcr_done.resize((size_t)4 * N));
    const unsigned char* dmask = nullptr;
    if (int rc = dist_upload_mask(h, cam_param_mask, &dmask); rc != BA_OK) return rc;
    {
      ScopedTimer tm(h, BA_K_BCR_ASSEMBLE);
      hipLaunchKernelGGL(k_bcr_assemble<3>, dim3(d.nasm), dim3(kBcrThreads), 0, h->stream, h->nco, h->hb, cb, h->S, h->b, dmask, h->bcrD.p,
                         h->bcrU.p, h->bcrF.p, h->flags.p + 1, h->dC.p, h->bcr_done.p, d.asm_nodes.p);
      hipLaunchKernelGGL(k_dist_zero_separators, dim3(8, std::max(1, d.nsep)), dim3(256), 0, h->stream, d.nsep, d.sep.p, d.sep_owner.p, d.rank, B,
                         h->bcrD.p, h->bcrF.p);
    }
    if (d.nwork_local > 0) {
      ScopedTimer tm(h, BA_K_BCR_ELIMINATE, 1);
      HIPCHECK(h, launch_bcr_fused(h, cb, d.nwork_local, h->stream, N, 1, h->bcrD.p, h->bcrU.p, h->bcrF.p, h->bcrP.p, h->bcrQ.p, h->bcrG.p,
                                   h->bcrGv.p, h->flags.p + 1, h->dC.p, d.work.p, h->bcr_done.p));
    }
    hipLaunchKernelGGL(k_dist_top, dim3(8, d.nsep + d.nroot), dim3(256), 0, h->stream, d.nsep, d.sep.p, d.nroot, d.root.p, d.root_owner.p,
                       d.rank, B, h->bcrD.p, h->bcrF.p, h->bcrP.p, h->bcrQ.p, d.xbuf, 0);
    *count = d.xcount[1];
  } else if (stage == 3) {
    hipLaunchKernelGGL(k_dist_top, dim3(8, d.nsep + d.nroot), dim3(256), 0, h->stream, d.nsep, d.sep.p, d.nroot, d.root.p, d.root_owner.p,
                       d.rank, B, h->bcrD.p, h->bcrF.p, h->bcrP.p, h->bcrQ.p, d.xbuf, 1);
    if (d.nwork_top > 0) {
      ScopedTimer tm(h, BA_K_BCR_ELIMINATE, 1);      // (the tickets go on from where the local phase stopped: one work list)
      HIPCHECK(h, launch_bcr_fused(h, cb, d.nwork_top, h->stream, N, d.P, h->bcrD.p, h->bcrU.p, h->bcrF.p, h->bcrP.p, h->bcrQ.p, h->bcrG.p,
                                   h->bcrGv.p, h->flags.p + 1, h->dC.p, d.work.p, h->bcr_done.p));
    }
    {
      const size_t lds2 = ((size_t)3 * B * (B + 1) + 3 * B + 8) * sizeof(double);
      HIPCHECK(h, ensure_lds_attr(h, (const void*)k_bcr_backsolve_fused));
      ScopedTimer tmb(h, BA_K_BCR_BACKSOLVE, 1);
      hipLaunchKernelGGL(k_bcr_backsolve_fused, dim3(d.norder), dim3(kBcrElimThreads), lds2, h->stream, N, B, h->bcrGv.p, h->bcrF.p, 1,
                         h->bcrP.p, h->bcrQ.p, h->bcrG.p, h->dC.p, d.order.p, h->flags.p + 1 + kBcrTicketWord);
    }
    hipLaunchKernelGGL(k_dist_solution, dim3(blocks_for((long long)h->nco * 6)), dim3(256), 0, h->stream, h->nco * 6, d.own_lo * 6, d.own_hi * 6,
                       h->dC.p, d.xbuf, 0);
    *count = d.xcount[2];
  } else if (stage == 4) {
    hipLaunchKernelGGL(k_dist_solution, dim3(blocks_for((long long)h->nco * 6)), dim3(256), 0, h->stream, h->nco * 6, 0, 0, h->dC.p, d.xbuf, 1);
    h->solve_kind = BA_SOLVE_BCR;
    h->have_solution = true;
  } else {
    return h->fail(BA_ERR_INVALID_ARG, "ba_dist_stage: stage %d", stage);
  }
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}


void launch_bcr_assemble(ba_handle* h, dim3 grid, int cb, const unsigned char* dmask, double* xsol, int* done, const int* nodes) {
  hipLaunchKernelGGL(k_bcr_assemble<3>, grid, dim3(kBcrThreads), 0, h->stream, h->nco, h->hb, cb, h->S, h->b, dmask, h->bcrD.p, h->bcrU.p, h->bcrF.p,
                     h->flags.p + 1, xsol, done, nodes);
}

}  // namespace ba

extern "C" {

int ba_solve_reduced(ba_handle* h, const uint8_t* cam_param_mask, int32_t* info) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_schur, BA_ERR_STATE, "ba_solve_reduced: call ba_schur first");
  REQUIRE(h, info, BA_ERR_INVALID_ARG, "ba_solve_reduced: info is NULL");
  if (h->nco == 0) { *info = 0; h->have_solution = true; return BA_OK; }
  const int force = h->opt.solver;                 // ba_set_option "solver"
  if (h->nbc > 0) {
    // band + border (ba_border.h): the cyclic reduction factors the band and solves for its right-hand side, its kept factors
    // take the border's columns through, one workgroup solves the border system.  A system that is not positive definite is
    // solved again with LU in every place of the block elimination (border_solve_lu: the reference's numpy.linalg.solve takes whatever
    // is not singular); option solver = lu goes there at once.
    REQUIRE(h, force == SOLVER_AUTO || force == SOLVER_BCR || force == SOLVER_BCR1 || force == SOLVER_LU, BA_ERR_STATE, "ba_solve_reduced: a problem with border cameras is solved by the cyclic reduction or by LU (option camera_order = off sets it up without a border)");
    REQUIRE(h, h->band_cams() <= kBcrMaxHB || (h->hb >= 1 && h->hb <= kBcrSplitMaxHB), BA_ERR_STATE, "ba_solve_reduced: border with a band the cyclic reduction does not take");
    HIPCHECK(h, hipSetDevice(h->device));
    const unsigned char* dmaskb = nullptr;
    if (int rcm = dist_upload_mask(h, cam_param_mask, &dmaskb); rcm != BA_OK) return rcm;
    const bool lu_nodes_b = bcr_node_size(h) <= kBcrMaxHB;
    int rcb = BA_OK;
    if (force == SOLVER_LU) {
      h->solve_kind = lu_nodes_b ? BA_SOLVE_BCR_LU : BA_SOLVE_BAND_LU;
      rcb = border_solve_lu(h, dmaskb);
    } else {
      h->solve_kind = BA_SOLVE_BCR;
      rcb = solve_bcr(h, dmaskb);
      if (rcb == BA_OK) rcb = border_solve(h, dmaskb);
    }
    if (rcb != BA_OK) return rcb;
    HIPCHECK(h, hipGetLastError());
    if (h->defer) { *info = 0; h->have_solution = true; return BA_OK; }
    int infb = 0;
    HIPCHECK(h, hipMemcpyAsync(&infb, h->flags.p + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    if (infb > 0 && infb != kBcrTimedOut && force != SOLVER_LU && h->opt.device_lu) {
      // not positive definite: the reference's LU would still solve it (bundle_adjuster.py:302-305)
      rcb = border_solve_lu(h, dmaskb);
      if (rcb != BA_OK) return rcb;
      HIPCHECK(h, hipMemcpyAsync(&infb, h->flags.p + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(h, hipStreamSynchronize(h->stream));
      h->solve_kind = lu_nodes_b ? BA_SOLVE_BCR_LU : BA_SOLVE_BAND_LU;
    }
    *info = infb;
    h->have_solution = infb == 0;
    return BA_OK;
  }
  const int nodes = h->hb > 0 ? (h->nco + h->hb - 1) / h->hb : 0;
  // (any number of nodes: even two levels beat k_band_solve's chain of nco pivots; nodes of 12 and 13 cameras exist spread over
  //  three workgroups only: option solver = bcr1 means the wide solver's one-unit-per-node kernels there)
  const bool bcr_ok = h->nco <= kBcrMaxHB || (h->hb >= 1 && h->hb <= kBcrMaxHB) || (h->hb <= kBcrSplitMaxHB && h->hb >= 1 && force != SOLVER_BCR1);
  const bool bcrw_ok = h->hb >= kBcrwMinHB && h->hb <= kBcrwMaxHB && nodes >= 4;
  const bool band_ok = h->hb <= kMaxBandSolve;       // (the single-workgroup band Cholesky is instantiated up to there)
  const bool dense_ok = 6 * h->nco <= kDcMaxN && force != SOLVER_LU;
  // wider than that: nodes that do not fit in LDS (ba_bcr_big.h), as long as there are a few of them to reduce over
  const int big_nodes = h->hb > kBcrwMaxHB ? (h->nco + big_node_cameras(h->hb) - 1) / big_node_cameras(h->hb) : 0;
  // (measured at 1000 cameras: 13 nodes of 80 cameras 1.6 ms against the dense factorisation's 3.4 ms, 5 nodes of 200 cameras 5.1 against 6.3)
  const bool big_ok = (force == SOLVER_BCR && big_nodes >= 4) || (force == SOLVER_AUTO && big_nodes >= (dense_ok ? 5 : 4));
  // ... and as long as their workspace (72 N B^2 bytes of K, 16 N B^2 of D and U) fits the device: otherwise the dense
  // factorisation, or - too large for that as well - *info = -1 (not a hard allocation error)
  bool use_big = big_ok;
  if (use_big) {
    const size_t cb = big_node_cameras(h->hb), B = 6 * cb, N = big_nodes;
    const size_t need = (N * big_matrix_doubles((int)B) + 2 * N * B * B + N * B) * sizeof(double);
    const size_t have = (h->bigK.n + h->bcrD.n + h->bcrU.n + h->bcrF.n) * sizeof(double);
    if (need > have) {
      size_t free_b = 0, total_b = 0;
      HIPCHECK(h, hipSetDevice(h->device));
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need - have > free_b - free_b / 16) use_big = false;
    }
  }
  // a scene without a narrow band under any order (an unordered photo collection): conjugate gradients over the blocks the tracks
  // define (ba_pcg.h) - by option, or when the band is wide, mostly structural zeros, and large enough for a dense factorisation to hurt
  const bool pcg_forced = force == SOLVER_PCG;
  // (the list of blocks comes from THIS handle's tracks: a shard of a sharded adjuster sees only its own, the dense-visibility mode has no list)
  REQUIRE(h, !pcg_forced || ((!h->comm || h->pcg.shared_lists) && !h->dense_mode), BA_ERR_STATE, "ba_solve_reduced: solver = pcg needs the whole scene on one handle (no communicator) and the sparse reductions (no dense-visibility mode)");
  REQUIRE(h, !h->pcg.packed || force == SOLVER_AUTO || pcg_forced, BA_ERR_STATE, "ba_solve_reduced: this problem's reduced system is stored as the list of its blocks (packed store): conjugate gradients are its solver (option packed_store = 0 before ba_set_problem keeps the band)");
  const bool use_pcg = h->nco > 0 && (pcg_forced || h->pcg.packed || (force == SOLVER_AUTO && !bcr_ok && !bcrw_ok && sparse_layout(h)));
  if (use_pcg) use_big = false;
  const bool use_dense = !use_pcg && !use_big && dense_ok && (force == SOLVER_DENSE || (!band_ok && !(bcrw_ok && (force == SOLVER_AUTO || force == SOLVER_BCR || force == SOLVER_BCR1))));
  const bool use_bcr = !use_dense && (force ? ((force == SOLVER_BCR || force == SOLVER_BCR1) && bcr_ok) : bcr_ok);
  const bool use_bcrw = !use_dense && !use_bcr && (force ? ((force == SOLVER_BCR || force == SOLVER_BCR1) && bcrw_ok) : bcrw_ok);
  // nothing of the above applies (or option solver = lu): LU with partial pivoting, the reference's own factorisation
  const bool use_lu = !use_pcg && (force == SOLVER_LU || (!use_big && !use_dense && !use_bcr && !use_bcrw && !band_ok));
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, h->Ufac.resize(std::max<size_t>(1, reduced_doubles(h))));
  const unsigned char* dmask = nullptr;
  if (cam_param_mask) {
    bool all = true;
    for (int i = 0; i < h->nco * 6; ++i) all = all && cam_param_mask[i];
    if (!all) {
      const unsigned char* m = cam_rows_in(h, cam_param_mask, h->mask_host, 6);      // (the caller's positions -> the internal camera order)
      HIPCHECK(h, hipMemcpyAsync(h->mask.p, m, (size_t)h->nco * 6, hipMemcpyHostToDevice, h->stream));
      dmask = h->mask.p;
    }
  }
  // multi-CU paths: block cyclic reduction when the band is narrow enough for dense (6 hb)^2 blocks in LDS and there
  // are enough super-blocks to parallelise over (decided above)
  const size_t lds_budget = 160 * 1024;
  const int ch = band_solve_chunk(h->hb, lds_budget);
  size_t lds = 0;
  h->solve_kind = use_pcg ? BA_SOLVE_PCG : use_lu ? BA_SOLVE_BAND_LU : use_big ? BA_SOLVE_BCR_BIG : use_dense ? BA_SOLVE_DENSE_CHOLESKY : use_bcr ? BA_SOLVE_BCR : use_bcrw ? BA_SOLVE_BCR_WIDE : BA_SOLVE_BAND;
  if (use_pcg) {
    int rc = solve_pcg(h, dmask);
    if (rc != BA_OK) return rc;
    if (h->pcg.last_status != 0 && !pcg_forced && dense_ok && !h->pcg.packed) {
      // chosen by the library and it did not converge (or found the matrix not positive definite): the dense factorisation has the last word
      h->solve_kind = BA_SOLVE_DENSE_CHOLESKY;
      rc = solve_dense_chol(h, dmask);
      if (rc != BA_OK) return rc;
    }
  } else if (use_lu) {
    int rc = solve_band_lu(h, dmask, -1, nullptr);
    if (rc != BA_OK) return rc;
  } else if (use_big) {
    int rc = solve_bcr_big(h, dmask);
    if (rc != BA_OK) return rc;
  } else if (use_dense) {
    int rc = solve_dense_chol(h, dmask);
    if (rc != BA_OK) return rc;
  } else if (use_bcr) {
    int rc = solve_bcr(h, dmask);
    if (rc != BA_OK) return rc;
  } else if (use_bcrw) {
    int rc = solve_bcr_wide(h, dmask);
    if (rc != BA_OK) return rc;
  } else {
    if (ch < 1) return h->fail(BA_ERR_STATE, "ba_solve_reduced: no LDS for the band solver at half-bandwidth %d", h->hb);
    lds = band_solve_lds_bytes(h->hb, ch);
    ScopedTimer tm(h, BA_K_BAND_SOLVE);
    hipError_t le = launch_band_solve(h, h->hb, lds, h->stream, h->nco, ch, h->S, h->b, dmask, h->Ufac.p, h->ysol.p,
                                      h->dinv.p, h->dC.p, h->flags.p + 1);
    if (le != hipSuccess) return h->fail(BA_ERR_HIP, "k_band_solve launch failed: %s", hipGetErrorString(le));
  }
  HIPCHECK(h, hipGetLastError());
  if (h->defer) { *info = 0; h->have_solution = true; return BA_OK; }   // status is read by ba_lm_trial
  int inf6[62] = {0};
  HIPCHECK(h, hipMemcpyAsync(inf6, h->flags.p + 1, sizeof(inf6), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  int inf = inf6[0];
  if (inf > 0 && inf != kBcrTimedOut && !use_lu && h->opt.device_lu && h->solve_kind != BA_SOLVE_PCG) {
    // not positive definite: the reference's LU would still solve it (bundle_adjuster.py:302-305) - the cyclic reduction with
    // LU nodes where the nodes are narrow, LU with partial pivoting down the band otherwise
    const bool lu_nodes = use_bcr && bcr_node_size(h) <= kBcrMaxHB;      // (k_bcr_eliminate_lu keeps a node's B x (3 B + 1) matrix in LDS)
    int rc = lu_nodes ? solve_bcr_lu(h, dmask, -1, nullptr) : solve_band_lu(h, dmask, -1, nullptr);
    if (rc != BA_OK) return rc;
    int inf2 = 0;
    HIPCHECK(h, hipMemcpyAsync(&inf2, h->flags.p + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    h->solve_kind = lu_nodes ? BA_SOLVE_BCR_LU : BA_SOLVE_BAND_LU;
    inf = inf2;
  }
#ifdef BA_BCR_PROFILE
  if (h->opt.solve_trace && use_bcr && h->bcr_trace_n > 0 && h->opt.fused_eliminate) {
    // time line of k_bcr_eliminate_fused: per level, when its workgroups passed each stage (us after the first workgroup started)
    std::vector<long long> tr((size_t)8 * h->bcr_trace_n);
    HIPCHECK(h, hipMemcpy(tr.data(), h->bcr_trace.p, tr.size() * sizeof(long long), hipMemcpyDeviceToHost));
    long long t0 = LLONG_MAX;
    for (int w = 0; w < h->bcr_trace_n; ++w) t0 = std::min(t0, tr[8 * w]);
    static const char* names[6] = {"start", "producers done", "loaded", "coupling formed", "factored", "handed on"};
    for (int s = 1; s < 2 * h->bcr_trace_n; s *= 2) {
      for (int role = 0; role < 3; ++role) {
        double lo[6], hi[6], sum[6]; int cnt = 0, xcds = 0;
        for (int k = 0; k < 6; ++k) { lo[k] = 1e30; hi[k] = -1e30; sum[k] = 0; }
        for (int w = 0; w < h->bcr_trace_n; ++w) {
          const int item = (int)tr[8 * w + 6], i = item >> 2;
          if (((i + 1) & -(i + 1)) != s || (item & 3) != role) continue;
          ++cnt; xcds |= 1 << (int)(tr[8 * w + 7] & 15);
          for (int k = 0; k < 6; ++k) { const double v = (tr[8 * w + k] - t0) * 0.01; lo[k] = std::min(lo[k], v); hi[k] = std::max(hi[k], v); sum[k] += v; }
        }
        if (!cnt) continue;
        fprintf(stderr, "[k_bcr_eliminate_fused stride %4d role %d: %3d workgroups]", s, role, cnt);
        for (int k = 0; k < 6; ++k) fprintf(stderr, " %s %.2f..%.2f (mean %.2f) |", names[k], lo[k], hi[k], sum[k] / cnt);
        fprintf(stderr, " us, on %d XCDs\n", __builtin_popcount(xcds));
      }
    }
  }
  if (h->opt.solve_trace && use_bcr && h->opt.solver != SOLVER_BCR1) {
    for (int role = 0; role < 3; ++role) {
      const int* o = inf6 + 8 + 10 * role;
      fprintf(stderr, "[k_bcr_eliminate_split level 2 node 1 role %d] load %d prologue %d | diag factor (wave 0, with block 0 and the urgent tiles) %d, phase 1 %d, phase 2 %d, urgent tile 0 %d | last rhs %d, products+store %d cycles\n",
              role, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
    }
    fprintf(stderr, "    prologue of role 0 per wavefront:");
    for (int w = 0; w < 16; ++w) fprintf(stderr, " %d", inf6[44 + w]);
    fprintf(stderr, "\n");
    fprintf(stderr, "    diagonal block 2 of role 0: loads %d, 12 pivots %d, stores of the inverse %d cycles\n", inf6[40], inf6[41], inf6[42]);
  } else
  if (h->opt.solve_trace && use_bcr)
    fprintf(stderr, "[k_bcr_eliminate level 0 node 2] load %d chol %d trsm %d products %d store %d cycles\n", inf6[8], inf6[9],
            inf6[10], inf6[11], inf6[12]);
  if (h->opt.solve_trace && use_bcr)
    fprintf(stderr, "    factor+solve, summed over the block steps: diagonal factor (wave 0) %d, phase 1 %d, phase 2 %d, phase 3 %d\n",
            inf6[14], inf6[15], inf6[16], inf6[17]);
  if (h->opt.solve_trace && use_bcr) {
    fprintf(stderr, "    phase 1 of block step 1, per wavefront:");
    for (int w = 0; w < 16; ++w) fprintf(stderr, " %d", inf6[44 + w]);
    fprintf(stderr, "\n");
  }
#endif
  if (h->opt.solve_trace && !use_bcr && !use_bcrw && !use_dense && !use_big)
    fprintf(stderr, "[k_band_solve] nco=%d hb=%d ch=%d lds=%zu B | forward: %d cycles, %d ticks(100MHz) | total: %d cycles, %d ticks\n",
            h->nco, h->hb, ch, lds, inf6[2], inf6[3], inf6[4], inf6[5]);
  *info = inf;
  h->have_solution = inf == 0;
  return BA_OK;
}

int ba_last_solve_kind(const ba_handle* h) { return h ? h->solve_kind : BA_SOLVE_NONE; }

int ba_get_solution(ba_handle* h, double* dC) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_solution && dC, BA_ERR_STATE, "ba_get_solution: no solution on the device (ba_solve_reduced)");
  HIPCHECK(h, hipSetDevice(h->device));
  if (h->nco) HIPCHECK(h, hipMemcpyAsync(dC, h->dC.p, (size_t)h->nco * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  cam_rows_out(h, dC, 6);                       // internal camera order -> the caller's positions
  return BA_OK;
}

// A solution of the reduced system computed by the caller (solve_motion_normal_eqns on host arrays, bundle_adjuster.py:281-312:
// the Python host's LU of a band + border system that is not positive definite) becomes the device's: what ba_backsubstitute,
// ba_apply_update and ba_get_solution use from here on.  dC in the caller's positions.
int ba_set_solution(ba_handle* h, const double* dC) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_schur && dC, BA_ERR_STATE, "ba_set_solution: call ba_schur first (and pass a solution)");
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, h->dC.resize((size_t)h->nco * 6 + 16));
  if (h->nco) {
    const double* src = cam_rows_in(h, dC, h->rows_host, 6);
    HIPCHECK(h, hipMemcpyAsync(h->dC.p, src, (size_t)h->nco * 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemsetAsync(h->flags.p + 1, 0, sizeof(int), h->stream));      // the status word of the solve this replaces
    HIPCHECK(h, hipStreamSynchronize(h->stream));                                 // (rows_host / dC are the caller's and ours to reuse)
  }
  h->have_solution = true;
  h->have_backsub = false;
  return BA_OK;
}

int ba_dist_plan(int32_t nco, int32_t half_bandwidth, int32_t nranks, int32_t* cams_per_node, int32_t* nodes, int32_t* nodes_per_rank) {
  int cb = 0, N = 0, P = 0;
  if (!dist_plan_static(nco, half_bandwidth, nranks, &cb, &N, &P)) return BA_ERR_STATE;
  if (cams_per_node) *cams_per_node = cb;
  if (nodes) *nodes = N;
  if (nodes_per_rank) *nodes_per_rank = P;
  return BA_OK;
}

int ba_dist_enable(ba_handle* h, int32_t rank, int32_t nranks) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_dist_enable: set the problem first");
  REQUIRE(h, nranks < 2 || (rank >= 0 && rank < nranks), BA_ERR_INVALID_ARG, "ba_dist_enable: bad rank");
  HIPCHECK(h, hipSetDevice(h->device));
  h->dist.on = false;
  if (nranks < 2 || h->dense_mode || h->pcg.packed) return BA_OK;   // (no band to cut: a dense or a packed store is solved by every rank)
  return dist_build_plan(h, rank, nranks);
}

int ba_dist_info(ba_handle* h, int64_t* out, int32_t n) {
  if (!h || !out) return BA_ERR_INVALID_ARG;
  const auto& d = h->dist;
  const int64_t v[12] = {d.on, d.cb, d.N, d.P, d.n_lo, d.n_hi, d.own_lo, d.own_hi, (int64_t)d.xcount[0], (int64_t)d.xcount[1],
                         (int64_t)d.xcount[2], d.nsep};
  for (int i = 0; i < n && i < 12; ++i) out[i] = d.on || i == 0 ? v[i] : 0;
  return BA_OK;
}

int ba_dist_bind_exchange(ba_handle* h, void* dev, int64_t doubles) {
  if (!h) return BA_ERR_INVALID_ARG;
  auto& d = h->dist;
  REQUIRE(h, d.on, BA_ERR_STATE, "ba_dist_bind_exchange: the distributed solve is off");
  const size_t need = std::max(d.xcount[0], std::max(d.xcount[1], d.xcount[2]));
  REQUIRE(h, dev && (size_t)doubles >= need, BA_ERR_INVALID_ARG, "ba_dist_bind_exchange: buffer too small");
  d.xbuf = (double*)dev; d.xcap = (size_t)doubles;
  return BA_OK;
}

int ba_dist_stage(ba_handle* h, int32_t stage, const uint8_t* cam_param_mask, int64_t* doubles_to_sum) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->dist.on, BA_ERR_STATE, "ba_dist_stage: the distributed solve is off (ba_dist_enable)");
  REQUIRE(h, h->have_schur, BA_ERR_STATE, "ba_dist_stage: call ba_lm_trial_begin first");
  HIPCHECK(h, hipSetDevice(h->device));
  size_t count = 0;
  h->have_solution = false;
  int rc = dist_stage(h, stage, cam_param_mask, &count);
  if (doubles_to_sum) *doubles_to_sum = (int64_t)count;
  return rc;
}

}  // extern "C"
