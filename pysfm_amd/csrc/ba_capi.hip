// ba_capi.hip - host side of libpysfm_ba.so: the opaque handle, device buffers,
// kernel launches and the C ABI declared in include/pysfm_ba.h.
#include "../../include/pysfm_ba.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ba_kernels.h"
#include "ba_bcr.h"
#include "ba_bcr_wide.h"
#include "ba_dense.h"
#include "ba_bcr_big.h"
#include "ba_dist.h"

#include <dlfcn.h>
#include <rccl/rccl.h>     // types only: the entry points are resolved at run time from the librccl torch has loaded

using namespace ba;

namespace {

thread_local std::string g_create_error;

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t resize(size_t count) {
    if (count <= n && p) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
    if (count == 0) return hipSuccess;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

struct TimedLaunch { int id; hipEvent_t a, b; int count; };

// Test / measurement switches (ba_set_option).  The defaults are the product path; nothing in the library
// reads the environment.
enum { SCHUR_AUTO = 0, SCHUR_PAIRS, SCHUR_GROUPS, SCHUR_MFMA1, SCHUR_MFMA2, SCHUR_MFMA };
enum { SOLVER_AUTO = 0, SOLVER_BCR, SOLVER_BAND, SOLVER_DENSE, SOLVER_LU, SOLVER_BCR1 };
struct Options {
  int schur = SCHUR_AUTO;
  int solver = SOLVER_AUTO;
  bool point_kernels_v1 = false;   // lanes-per-point k_linearize / k_backsub instead of the group-packed kernels
  bool fuse_cost = true;           // trial cost inside k_backsub_groups
  bool fuse_cam = true;            // camera blocks inside the MFMA reduction
  bool fuse_lin = false;           // point blocks + inverses inside the single-wavefront MFMA reduction
  bool sort_points = true;         // internal point order (ba_set_problem); off = the caller's order as given
  int gm_cap = 0;                  // points per MFMA group (0 = chosen by ba_set_problem)
  bool lds_window = true;          // k_schur_groups_mfma3 accumulates in an LDS window of the band when one fits
  bool fast_paths = true;          // K = I / unit-Gaussian short cuts of the per-observation arithmetic (ba_math.h)
  bool fused_backsolve = true;     // all back-substitution levels of the cyclic reduction in one launch when the nodes fit the chip
  bool fused_eliminate = true;     // all split elimination levels of the cyclic reduction in one launch (k_bcr_eliminate_fused)
  bool device_lu = true;           // a reduced system the Cholesky solve reports as not positive definite is solved again by the cyclic reduction with LU nodes
  bool solve_trace = false;        // per-phase cycle counts of the node kernels (PROFILE builds)
};

}  // namespace

struct ba_handle {
  int device = 0;
  int ncu = 256;             // compute units of the device
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  Options opt;
  std::vector<const void*> lds_attr_done;   // kernels whose dynamic-LDS limit has been raised on THIS handle's device

  // problem
  int nc = 0, nt = 0, nco = 0;
  int hb = 0;                // block half-bandwidth of the reduced system
  bool fac_valid = false;    // fac[] (L D L^T of the point inverses + HPPinv bP) matches HPPinv
  int solve_kind = 0;        // BA_SOLVE_*: what the last ba_solve_reduced launched
  int min_hb = 0;            // ba_set_min_half_bandwidth: lower bound for hb (ranks must agree on the band layout)
  long long nobs = 0;
  bool have_problem = false;
  bool have_params[2] = {false, false};
  bool have_linearization = false, have_schur = false, have_backsub = false;
  int lin_phys = 0;                  // physical parameter set of the linearisation
  bool point_blocks_valid = false;   // HPP / bP hold the point blocks of the linearisation (ba_lm_trial leaves them to the reduction too)
  bool cam_blocks_valid = false;     // HCC / bC hold the camera blocks of the linearisation (ba_lm_trial may leave them to the reduction)
  bool inv_valid = false;            // HPPinv holds pinv of the damped point blocks for (inv_damping, inv_rcond)
  double inv_damping = 0.0, inv_rcond = 0.0;
  bool dense_mode = false;           // ba_set_dense_visibility: the reduction is one SYRK over all points (k_dense_*)
  double* trial_result_dev = nullptr; // bound by ba_bind_trial_result: device copy of the cost partials + status words
  ncclComm_t comm = nullptr;          // ba_comm_init: the shards' communicator; collectives run on `stream`
  int comm_ranks = 0;
  double* comm_host = nullptr;        // pinned [kCostBlocks + 2]: the all-reduced trial record
  double trial_rcond = 0.0;          // ba_lm_trial_begin -> ba_lm_trial_end
  int glog = 0;              // lanes per point = 2^glog
  double K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  Sensor sensor{SENSOR_GAUSS, {1, 0, 0, 1}, 1.0, 1.0, FAST_UNIT_GAUSS};
  DevBuf<int> obs_cam, obs_pt, pt_off, cam_opt_pos, opt_cam, keep;
  DevBuf<double2> obs_z;
  DevBuf<unsigned char> pt_opt;
  DevBuf<SchurUnit> units;
  int nunits = 0;
  DevBuf<SchurChunk> chunks;
  int nchunks = 0, schur_wn = 0;
  DevBuf<SchurGroup> groups, mgroups;
  DevBuf<SchurChunk> gchunks, mchunks;
  int nmchunks = 0, nmgroups_total = 0;
  bool groups_ascending = false;
  int ngchunks = 0, group_rounds = 0;   // group_rounds == 0: k_schur_groups not applicable
  bool groups_worth = false;            // points really share camera lists (mean run >= 2 points)
  Gm3Params gm3{0, 0, 0, 0, 0, 1, 1};   // k_schur_groups_mfma3: tile count, staged row length, k-rows per buffer, points per batch, window
  DevBuf<SchurChunk> m3chunks;          // its chunks (window rows gm3.wn may differ from schur_wn)
  DevBuf<WinGroup> wgroups;             // its groups: points whose optimised cameras share a window of <= 24 positions
  DevBuf<int> wtab;                     // ... and their (point, window column) -> observation tables
  int nm3chunks = 0, nwgroups = 0;
  DevBuf<int> wide_list;                // window groups of 25 .. 40 cameras (k_schur_wide_mfma, one workgroup each): indices into wgroups,
  int nwide = 0;                        // by tiles per side: those of NT tiles are wide_list[wide_begin[NT - kGwMinTiles] .. wide_begin[NT - kGwMinTiles + 1])
  int wide_begin[kGwMaxTiles - kGwMinTiles + 2] = {0};
  // tracks that span more than kGm3MaxSpan cameras: groups of k_schur_rect_mfma, one per pair of segments (A <= B) they touch
  DevBuf<RectGroup> rgroups;
  DevBuf<int> rtab;
  int nrgroups = 0, nlong_points = 0;
  bool gm3_uniform_ks = false;          // every window group has 6 points per batch (width <= 10): five k-steps
  bool wgroups_worth = false;           // enough points per window group for the matrix-core reduction to pay
  int ngroups = 0;                      // groups[] (<= kGroupMaxPts points each)
  bool point_groups = false;            // every point sits in a group and groups are worth it: group-packed k_linearize / k_backsub
  int group_maxL = 0;                   // longest track (k_schur_groups_mfma takes <= kGmMaxL)
  DevBuf<int> cam_perm;
  DevBuf<CamUnit> cam_units;
  int ncam_units = 0;
  std::vector<int> h_cam_opt_pos;
  std::vector<unsigned char> h_pt_opt;
  // internal order (ba_set_problem): internal point i = the caller's track pperm[i], internal observation n = the
  // caller's operm[n]; empty = identity
  std::vector<int> pperm, operm;

  // parameters: cams[which] = nc x [R(9) | t(3)], X[which] = nt x 3
  DevBuf<double> cams[2], X[2];
  int cur = 0;               // physical index of BA_PARAMS_CUR

  // normal-equation blocks
  DevBuf<double> HCC, bC, HPP, bP, HPPinv, W, S_own, b_own, dC, dP, scratch, Ufac, ysol, dinv, bcrD, bcrU, bcrF, bcrP, bcrQ, bcrG, bcrGv, bcrL, bcrLv, denseA, bigK, fac, dUd, dDd, dyd, dpart, comm_dev;
  DevBuf<unsigned char> mask;
  DevBuf<int> bcr_order;     // k_bcr_backsolve_fused: the nodes level by level from the root down (for bcr_order_n nodes)
  int bcr_order_n = 0;
  DevBuf<int> bcr_work, bcr_done;   // k_bcr_eliminate_fused: 4 node + role of every workgroup, leaves first; "handed on" words [4 N]
  int bcr_work_n = 0, bcr_work_s = 0, bcr_work_len = 0, bcr_work_elim = 0;   // (elimination items first, then the back-substitution items)
  DevBuf<long long> bcr_trace;      // PROFILE builds, option solve_trace: the time line of k_bcr_eliminate_fused, 8 words per workgroup
  int bcr_trace_n = 0;
  // the reduced solve spread over the ranks of a sharded adjuster (ba_dist.h; ba_dist_enable)
  struct DistPlan {
    bool on = false;
    int rank = 0, nranks = 1;
    int cb = 0, N = 0, P = 0;           // cameras per node, nodes, nodes per interval (its separator included)
    int n_lo = 0, n_hi = 0;             // this rank's own nodes [n_lo, n_hi) (interior) ...
    int own_lo = 0, own_hi = 0;         // ... and the camera positions whose solution it contributes (interior + its separator)
    int nrows = 0, nsep = 0, nroot = 0, nwork_local = 0, nwork_top = 0, norder = 0;
    DevBuf<int> rows, sep, sep_owner, root, root_owner, work, order, asm_nodes;   // work = [local items | separator items]
    int nasm = 0;                       // nodes this rank assembles: its own interval and every separator
    double* xbuf = nullptr;             // the exchange buffer (bound by the caller, or our own)
    size_t xcap = 0;
    DevBuf<double> xown;
    size_t xcount[3] = {0, 0, 0};       // doubles of the three exchanges
  } dist;
  bool have_solution = false;
  bool defer = false;        // inside ba_lm_trial: leave status words / cost on the device, one read-back at the end
  DevBuf<int> flags;        // [0] unused, [1] solver status, [2..15] solver instrumentation, [40],[41] singular-point
                            // counters (alternate per ba_schur call)
  int sing_epoch = 0;       // which of the two counters the latest ba_schur used
  HostResult* host_result = nullptr;   // pinned, device-visible: cost + status words of a trial
  int cost_blocks = 0;      // partials the last k_cost launch wrote
  bool cost_fused = false;  // the last ba_backsubstitute evaluated the trial cost as well (k_backsub_groups)
  int* sing_counter() { return flags.p + 40 + (sing_epoch & 1); }
  double host_cost() const {           // second, deterministic stage of the cost reduction (after a stream sync)
    double s = 0.0;
    for (int i = 0; i < cost_blocks; ++i) s += host_result->partial[i];
    return s;
  }
  double* S = nullptr;       // nco*nco*36 (own or bound)
  double* b = nullptr;       // nco*6

  // timing
  bool timing = false;
  unsigned long long timing_mask = ~0ull;   // which kernel ids are bracketed with events
  int timing_stride = 1;                    // bracket every n-th eligible launch (an event pair costs stream time)
  unsigned timing_seen[BA_K_COUNT] = {0};
  std::vector<hipEvent_t> ev_pool;
  std::vector<TimedLaunch> pending;
  double ms[BA_K_COUNT] = {0};
  long long launches[BA_K_COUNT] = {0};

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
  int phys(int which) const { return which == BA_PARAMS_CUR ? cur : 1 - cur; }
};

#define HIPCHECK(h, call)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return (h)->fail(e_ == hipErrorOutOfMemory ? BA_ERR_NOMEM : BA_ERR_HIP, "%s failed: %s (%s:%d)", \
                       #call, hipGetErrorString(e_), __FILE__, __LINE__);                   \
  } while (0)

#define REQUIRE(h, cond, code, msg) \
  do { if (!(cond)) return (h)->fail(code, "%s", msg); } while (0)

namespace {

hipEvent_t get_event(ba_handle* h) {
  if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);      // (timing only: nobody reads memory on the strength of these events; the system-scope fence of a default event costs the stream ~20 us)
  return e;
}

void resolve_timings(ba_handle* h) {
  if (h->pending.empty()) return;
  (void)hipStreamSynchronize(h->stream);
  for (auto& t : h->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { h->ms[t.id] += ms; h->launches[t.id] += t.count; }
    h->ev_pool.push_back(t.a);
    h->ev_pool.push_back(t.b);
  }
  h->pending.clear();
}

struct ScopedTimer {
  // one event pair around `count` back-to-back launches of the same kernel (the cyclic-reduction levels):
  // an event pair costs a few microseconds of stream time, seven of them per 0.4 ms step would show
  ba_handle* h; int id; hipEvent_t a = nullptr, b = nullptr;
  bool on; int count;
  ScopedTimer(ba_handle* h_, int id_, int count_ = 1)
      : h(h_), id(id_), on(h_->timing && ((h_->timing_mask >> id_) & 1ull) && (h_->timing_seen[id_]++ % (unsigned)h_->timing_stride) == 0),
        count(count_) {
    if (on) { a = get_event(h); b = get_event(h); (void)hipEventRecord(a, h->stream); }
  }
  ~ScopedTimer() {
    if (on) {
      (void)hipEventRecord(b, h->stream);
      h->pending.push_back({id, a, b, count});
      if (h->pending.size() >= 8192) resolve_timings(h);
    }
  }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: remembered per handle (one handle = one device)
hipError_t ensure_lds_attr(ba_handle* h, const void* fn) {
  for (const void* f : h->lds_attr_done) if (f == fn) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) h->lds_attr_done.push_back(fn);
  return e;
}

// Host-facing per-point / per-observation arrays go through the internal order of ba_set_problem:
// rows of w doubles, perm[i] = the caller's index of internal row i.
void rows_to_internal(const std::vector<int>& perm, const double* src, double* dst, int w) {
  for (size_t i = 0; i < perm.size(); ++i) std::memcpy(dst + i * w, src + (size_t)perm[i] * w, w * sizeof(double));
}
void rows_to_caller(const std::vector<int>& perm, const double* src, double* dst, int w) {
  for (size_t i = 0; i < perm.size(); ++i) std::memcpy(dst + (size_t)perm[i] * w, src + i * w, w * sizeof(double));
}
// device rows -> caller's host array (synchronises the stream when a permutation is in the way)
int download_rows(ba_handle* h, const std::vector<int>& perm, const double* dev, double* host, size_t n, int w);

inline unsigned blocks_for(long long n) { return (unsigned)std::max<long long>(1, (n + kBlock - 1) / kBlock); }

DevProblem dev_problem(const ba_handle* h) {
  DevProblem P;
  P.nc = h->nc; P.nt = h->nt; P.nco = h->nco; P.hb = h->hb; P.nobs = h->nobs;
  P.obs_cam = h->obs_cam.p; P.obs_pt = h->obs_pt.p; P.obs_z = h->obs_z.p; P.pt_off = h->pt_off.p;
  P.cam_opt_pos = h->cam_opt_pos.p; P.pt_opt = h->pt_opt.p;
  std::memcpy(P.K, h->K, sizeof P.K);
  P.sensor = h->sensor;
  const double* K = h->K;
  if (K[0] == 1.0 && K[4] == 1.0 && K[8] == 1.0 && K[1] == 0.0 && K[2] == 0.0 && K[3] == 0.0 && K[5] == 0.0 && K[6] == 0.0 && K[7] == 0.0)
    P.sensor.fast |= FAST_K_IDENTITY;
  if (!h->opt.fast_paths) P.sensor.fast = 0;
  return P;
}

inline size_t reduced_doubles(const ba_handle* h) { return (size_t)h->nco * (h->hb + 1) * 36; }

// k_band_solve is instantiated per block half-bandwidth (compile-time unrolling)
template <int HB, bool MASKED>
hipError_t launch_band_solve_hbm(ba_handle* h, size_t lds, hipStream_t stream, int nco, int ch, const double* S, const double* b,
                                 const unsigned char* mask, double* U, double* y, double* dinv, double* x, int* info) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_band_solve<HB, MASKED>); e != hipSuccess) return e;
  hipLaunchKernelGGL((k_band_solve<HB, MASKED>), dim3(1), dim3(kSolveThreads), lds, stream, nco, ch, S, b, mask, U, y, dinv, x, info);
  return hipGetLastError();
}

template <int HB>
hipError_t launch_band_solve_hb(ba_handle* h, size_t lds, hipStream_t stream, int nco, int ch, const double* S, const double* b,
                                const unsigned char* mask, double* U, double* y, double* dinv, double* x, int* info) {
  return mask ? launch_band_solve_hbm<HB, true>(h, lds, stream, nco, ch, S, b, mask, U, y, dinv, x, info)
              : launch_band_solve_hbm<HB, false>(h, lds, stream, nco, ch, S, b, mask, U, y, dinv, x, info);
}

hipError_t launch_band_solve(ba_handle* h, int hb, size_t lds, hipStream_t stream, int nco, int ch, const double* S, const double* b,
                             const unsigned char* mask, double* U, double* y, double* dinv, double* x, int* info) {
#define BA_HB_CASE(N) case N: return launch_band_solve_hb<N>(h, lds, stream, nco, ch, S, b, mask, U, y, dinv, x, info);
  switch (hb) {
    BA_HB_CASE(0) BA_HB_CASE(1) BA_HB_CASE(2) BA_HB_CASE(3) BA_HB_CASE(4) BA_HB_CASE(5) BA_HB_CASE(6) BA_HB_CASE(7)
    BA_HB_CASE(8) BA_HB_CASE(9) BA_HB_CASE(10) BA_HB_CASE(11) BA_HB_CASE(12) BA_HB_CASE(13) BA_HB_CASE(14)
    BA_HB_CASE(15) BA_HB_CASE(16) BA_HB_CASE(17) BA_HB_CASE(18) BA_HB_CASE(19) BA_HB_CASE(20) BA_HB_CASE(21)
    default: return hipErrorInvalidValue;
  }
#undef BA_HB_CASE
}

int ensure_reduced(ba_handle* h) {
  if (!h->S) {
    HIPCHECK(h, h->S_own.resize(std::max<size_t>(1, reduced_doubles(h))));
    h->S = h->S_own.p;
  }
  if (!h->b) {
    HIPCHECK(h, h->b_own.resize(std::max<size_t>(1, (size_t)h->nco * 6)));
    h->b = h->b_own.p;
  }
  return BA_OK;
}

template <int HB>
hipError_t launch_bcr_eliminate_hb(ba_handle* h, int cnt, size_t lds, hipStream_t st, int N, int s, double* D, double* U, double* f,
                                   double* P, double* Q, double* G, int* info, double* x) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcr_eliminate<HB>); e != hipSuccess) return e;
  hipLaunchKernelGGL(k_bcr_eliminate<HB>, dim3(cnt), dim3(kBcrElimThreads), lds, st, N, s, D, U, f, P, Q, G, info, x);
  return hipSuccess;
}

template <int HB>
hipError_t launch_bcr_split_hb(ba_handle* h, int cnt, hipStream_t st, int N, int s, double* D, const double* U, double* f,
                               double* P, double* Q, double* G, double* gv, int* info, double* x) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcr_eliminate_split<HB>); e != hipSuccess) return e;
  hipLaunchKernelGGL(k_bcr_eliminate_split<HB>, dim3(cnt, 3), dim3(kBcrElimThreads), bcr_split_lds_bytes(6 * HB), st, N, s, D, U, f, P,
                     Q, G, gv, info, x);
  return hipSuccess;
}

template <int HB>
hipError_t launch_bcr_fused_hb(ba_handle* h, int nwork, hipStream_t st, int N, int s_first, double* D, const double* U, double* f,
                               double* P, double* Q, double* G, double* gv, int* info, double* x, const int* work, int* done) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcr_eliminate_fused<HB>); e != hipSuccess) return e;
  long long* trace = nullptr;
#ifdef BA_BCR_PROFILE
  if (h->opt.solve_trace) {
    if (hipError_t e = h->bcr_trace.resize((size_t)8 * nwork); e != hipSuccess) return e;
    trace = h->bcr_trace.p;
    h->bcr_trace_n = nwork;
  }
#endif
  hipLaunchKernelGGL(k_bcr_eliminate_fused<HB>, dim3(nwork), dim3(kBcrElimThreads), bcr_split_lds_bytes(6 * HB), st, N, s_first, D, U, f,
                     P, Q, G, gv, info, x, work, done, trace);
  return hipSuccess;
}

hipError_t launch_bcr_fused(ba_handle* h, int hb, int nwork, hipStream_t st, int N, int s_first, double* D, const double* U, double* f,
                            double* P, double* Q, double* G, double* gv, int* info, double* x, const int* work, int* done) {
#define BA_HB_CASE(K) case K: return launch_bcr_fused_hb<K>(h, nwork, st, N, s_first, D, U, f, P, Q, G, gv, info, x, work, done);
  switch (hb) {
    BA_HB_CASE(1) BA_HB_CASE(2) BA_HB_CASE(3) BA_HB_CASE(4) BA_HB_CASE(5) BA_HB_CASE(6) BA_HB_CASE(7) BA_HB_CASE(8)
    BA_HB_CASE(9) BA_HB_CASE(10) BA_HB_CASE(11)
    default: return hipErrorInvalidValue;
  }
#undef BA_HB_CASE
}

hipError_t launch_bcr_split(ba_handle* h, int hb, int cnt, hipStream_t st, int N, int s, double* D, const double* U, double* f,
                            double* P, double* Q, double* G, double* gv, int* info, double* x) {
#define BA_HB_CASE(K) case K: return launch_bcr_split_hb<K>(h, cnt, st, N, s, D, U, f, P, Q, G, gv, info, x);
  switch (hb) {
    BA_HB_CASE(1) BA_HB_CASE(2) BA_HB_CASE(3) BA_HB_CASE(4) BA_HB_CASE(5) BA_HB_CASE(6) BA_HB_CASE(7) BA_HB_CASE(8)
    BA_HB_CASE(9) BA_HB_CASE(10) BA_HB_CASE(11)
    default: return hipErrorInvalidValue;
  }
#undef BA_HB_CASE
}

hipError_t launch_bcr_eliminate(ba_handle* h, int hb, int cnt, size_t lds, hipStream_t st, int N, int s, double* D, double* U, double* f,
                                double* P, double* Q, double* G, int* info, double* x) {
#define BA_HB_CASE(K) case K: return launch_bcr_eliminate_hb<K>(h, cnt, lds, st, N, s, D, U, f, P, Q, G, info, x);
  switch (hb) {
    BA_HB_CASE(1) BA_HB_CASE(2) BA_HB_CASE(3) BA_HB_CASE(4) BA_HB_CASE(5) BA_HB_CASE(6) BA_HB_CASE(7) BA_HB_CASE(8)
    BA_HB_CASE(9) BA_HB_CASE(10) BA_HB_CASE(11)
    default: return hipErrorInvalidValue;
  }
#undef BA_HB_CASE
}

// Cameras per node of the narrow cyclic reduction: the half-bandwidth - or, for systems of at most kBcrMaxHB cameras (the
// sliding-window caller's 10-camera windows, the reference's own small scenes), ALL of them: one node, one workgroup, one
// 6 nco x 6 nco Cholesky with the node kernel's pivot chain (~10 us where k_band_solve's nco dependent 6 x 6 pivots take 33).
inline int bcr_node_size(const ba_handle* h) { return h->nco <= kBcrMaxHB ? std::max(1, h->nco) : std::max(1, h->hb); }

// Block cyclic reduction over super-blocks of hb cameras (ba_bcr.h): log2(N) levels, one
// workgroup per eliminated node.  Leaves the solution in h->dC and the status in flags[1].
int solve_bcr(ba_handle* h, const unsigned char* dmask) {
  const int hb = bcr_node_size(h), B = 6 * hb, N = (h->nco + hb - 1) / hb;      // (hb: cameras per node from here on)
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
  const bool split = h->opt.solver != SOLVER_BCR1;
  std::vector<char> level_split;
  const size_t lds = bcr_lds_bytes(B);
  std::vector<int> strides;
  for (int s = 1; (N / s + 1) / 2 > 0; s *= 2) strides.push_back(s);
  for (int s : strides) {
    const int cnt = (N / s + 1) / 2;
    level_split.push_back(split && (3 * cnt <= h->ncu || (!level_split.empty() && level_split.back())));
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
      // k_bcr_assemble is then ONE launch.  (Not with the two-stage words, which use the word the inverse role publishes.)
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
    const bool back_in_launch = s_fused && h->opt.fused_backsolve && !BA_BCR_TWO_STAGE && N <= 8 * h->ncu;
    nwork = s_fused ? (back_in_launch ? h->bcr_work_len : h->bcr_work_elim) : 0;
    if (s_fused) HIPCHECK(h, h->bcr_done.resize((size_t)4 * N));
  }
  {
    ScopedTimer tm(h, BA_K_BCR_ASSEMBLE);       // also clears the status word flags[1], marks the solution "not there yet", clears done[]
    hipLaunchKernelGGL(k_bcr_assemble, dim3(N), dim3(kBcrThreads), 0, h->stream, h->nco, h->hb, hb, h->S, h->b, dmask, h->bcrD.p,
                       h->bcrU.p, h->bcrF.p, h->flags.p + 1, h->dC.p, s_fused ? h->bcr_done.p : nullptr);
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
  if (s_fused && h->opt.fused_backsolve && !BA_BCR_TWO_STAGE && N <= 8 * h->ncu) return BA_OK;      // (done inside k_bcr_eliminate_fused)
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

template <int HB>
hipError_t launch_bcr_lu_hb(ba_handle* h, int cnt, hipStream_t st, int N, int s, double* D, double* U, double* f, double* P, double* Q,
                            double* G, int* info, double* x) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcr_eliminate_lu<HB>); e != hipSuccess) return e;
  hipLaunchKernelGGL(k_bcr_eliminate_lu<HB>, dim3(cnt), dim3(kBcrElimThreads), bcr_lu_lds_bytes(6 * HB), st, N, s, D, U, f, P, Q, G, info, x);
  return hipSuccess;
}

// The cyclic reduction with LU nodes (k_bcr_eliminate_lu): for reduced systems the Cholesky solvers reported as not positive
// definite.  Same layout and back-substitution as solve_bcr; leaves the solution in h->dC and the status in flags[1].
int solve_bcr_lu(ba_handle* h, const unsigned char* dmask) {
  const int hb = bcr_node_size(h), B = 6 * hb, N = (h->nco + hb - 1) / hb;
  const size_t BB = (size_t)B * B;
  HIPCHECK(h, h->bcrD.resize(N * BB)); HIPCHECK(h, h->bcrU.resize(N * BB)); HIPCHECK(h, h->bcrP.resize(N * BB));
  HIPCHECK(h, h->bcrQ.resize(N * BB)); HIPCHECK(h, h->bcrG.resize(N * BB));
  HIPCHECK(h, h->bcrF.resize((size_t)N * B));
  {
    ScopedTimer tm(h, BA_K_BCR_ASSEMBLE);
    hipLaunchKernelGGL(k_bcr_assemble, dim3(N), dim3(kBcrThreads), 0, h->stream, h->nco, h->hb, hb, h->S, h->b, dmask, h->bcrD.p,
                       h->bcrU.p, h->bcrF.p, h->flags.p + 1, h->dC.p);
  }
  std::vector<int> strides;
  for (int s = 1; (N / s + 1) / 2 > 0; s *= 2) strides.push_back(s);
  {
    ScopedTimer tm(h, BA_K_BCR_ELIMINATE, (int)strides.size());
    for (int s : strides) {
      const int cnt = (N / s + 1) / 2;
      hipError_t e = hipErrorInvalidValue;
#define BA_HB_CASE(K) case K: e = launch_bcr_lu_hb<K>(h, cnt, h->stream, N, s, h->bcrD.p, h->bcrU.p, h->bcrF.p, h->bcrP.p, h->bcrQ.p, h->bcrG.p, h->flags.p + 1, h->dC.p); break;
      switch (hb) {
        BA_HB_CASE(1) BA_HB_CASE(2) BA_HB_CASE(3) BA_HB_CASE(4) BA_HB_CASE(5) BA_HB_CASE(6) BA_HB_CASE(7) BA_HB_CASE(8)
        BA_HB_CASE(9) BA_HB_CASE(10) BA_HB_CASE(11)
        default: break;
      }
#undef BA_HB_CASE
      HIPCHECK(h, e);
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
  HIPCHECK(h, hipMemcpyAsync(h->mask.p, cam_param_mask, (size_t)h->nco * 6, hipMemcpyHostToDevice, h->stream));
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
      hipLaunchKernelGGL(k_bcr_assemble, dim3(d.nasm), dim3(kBcrThreads), 0, h->stream, h->nco, h->hb, cb, h->S, h->b, dmask, h->bcrD.p,
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

// factor + solve of one level (both are templates on the half-bandwidth)
template <int HB>
hipError_t launch_bcrw_factor_hb(ba_handle* h, int cnt, hipStream_t st, int N, int s, const double* D, double* L, double* Lv, const double* U,
                                 double* f, double* P, double* Q, double* G, int* info) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcrw_factor<HB>); e != hipSuccess) return e;
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcrw_solve_mfma<HB>); e != hipSuccess) return e;
  constexpr int B = 6 * HB;
  hipLaunchKernelGGL(k_bcrw_factor<HB>, dim3(cnt), dim3(kBcrElimThreads), bcrw_factor_lds_bytes(B), st, N, s, D, L, Lv, info);
  const int ntile = (3 * B + 1 + 15) / 16;
  hipLaunchKernelGGL(k_bcrw_solve_mfma<HB>, dim3(cnt, (ntile + 3) / 4), dim3(1024), bcrw_solve_lds_bytes(B), st, N, s, L, Lv, U, f,
                     P, Q, G, info);
  return hipSuccess;
}

hipError_t launch_bcrw_factor(ba_handle* h, int hb, int cnt, hipStream_t st, int N, int s, const double* D, double* L, double* Lv, const double* U,
                              double* f, double* P, double* Q, double* G, int* info) {
#define BA_HB_CASE(K) case K: return launch_bcrw_factor_hb<K>(h, cnt, st, N, s, D, L, Lv, U, f, P, Q, G, info);
  switch (hb) {
    BA_HB_CASE(12) BA_HB_CASE(13) BA_HB_CASE(14) BA_HB_CASE(15) BA_HB_CASE(16) BA_HB_CASE(17) BA_HB_CASE(18) BA_HB_CASE(19)
    BA_HB_CASE(20) BA_HB_CASE(21) BA_HB_CASE(22) BA_HB_CASE(23)
    default: return hipErrorInvalidValue;
  }
#undef BA_HB_CASE
}

// Block cyclic reduction for half-bandwidths 12..21 (ba_bcr_wide.h): three kernels per level.
int solve_bcr_wide(ba_handle* h, const unsigned char* dmask) {
  const int hb = h->hb, B = 6 * hb, N = (h->nco + hb - 1) / hb;
  const size_t BB = (size_t)B * B;
  HIPCHECK(h, h->bcrD.resize(N * BB)); HIPCHECK(h, h->bcrU.resize(N * BB)); HIPCHECK(h, h->bcrP.resize(N * BB));
  HIPCHECK(h, h->bcrQ.resize(N * BB)); HIPCHECK(h, h->bcrG.resize(N * BB)); HIPCHECK(h, h->bcrL.resize(N * BB));
  HIPCHECK(h, h->bcrF.resize((size_t)N * B));
  HIPCHECK(h, h->bcrLv.resize((size_t)N * ((B + 11) / 12) * 144));
  HIPCHECK(h, h->dC.resize((size_t)N * B + 16));
  {
    ScopedTimer tm(h, BA_K_BCR_ASSEMBLE);       // also clears the status word flags[1]
    hipLaunchKernelGGL(k_bcr_assemble, dim3(N), dim3(kBcrThreads), 0, h->stream, h->nco, hb, hb, h->S, h->b, dmask, h->bcrD.p,
                       h->bcrU.p, h->bcrF.p, h->flags.p + 1, (double*)nullptr);
  }
  std::vector<int> strides;
  for (int s = 1; (N / s + 1) / 2 > 0; s *= 2) strides.push_back(s);
  const int nt = (B + kBcrwPTile - 1) / kBcrwPTile, ntask = nt * (nt + 1) + nt * nt + 1;
  {
    ScopedTimer tm(h, BA_K_BCR_ELIMINATE, 3 * (int)strides.size());
    for (int s : strides) {
      const int cnt = (N / s + 1) / 2;
      HIPCHECK(h, launch_bcrw_factor(h, hb, cnt, h->stream, N, s, h->bcrD.p, h->bcrL.p, h->bcrLv.p, h->bcrU.p, h->bcrF.p, h->bcrP.p,
                                     h->bcrQ.p, h->bcrG.p, h->flags.p + 1));
      hipLaunchKernelGGL(k_bcrw_products, dim3(cnt, ntask), dim3(1024), 0, h->stream, N, B, s, h->bcrD.p, h->bcrU.p, h->bcrF.p, h->bcrP.p, h->bcrQ.p, h->flags.p + 1);
    }
  }
  ScopedTimer tmb(h, BA_K_BCR_BACKSOLVE, (int)strides.size());
  for (int q = (int)strides.size() - 1; q >= 0; --q) {
    const int s = strides[q], cnt = (N / s + 1) / 2;
    hipLaunchKernelGGL(k_bcrw_backsolve, dim3(cnt), dim3(1024), 0, h->stream, N, B, s, h->bcrF.p, h->bcrP.p, h->bcrQ.p, h->bcrG.p,
                       h->dC.p);
  }
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

// Block cyclic reduction with nodes too large for LDS (ba_bcr_big.h): half-bandwidths beyond kBcrwMaxHB.  A level is a batched
// partial dense Cholesky of one 3B x 3B matrix per eliminated node.
inline int big_node_cameras(int hb) { return (hb + 1) & ~1; }      // cb >= hb, even: B = 6 cb is a multiple of the panel kernel's 12-column steps
int solve_bcr_big(ba_handle* h, const unsigned char* dmask) {
  const int hb = h->hb, cb = big_node_cameras(hb), B = 6 * cb, N = (h->nco + cb - 1) / cb, n = 3 * B;
  const size_t BB = (size_t)B * B, KS = big_matrix_doubles(B);
  HIPCHECK(h, h->bcrD.resize(N * BB)); HIPCHECK(h, h->bcrU.resize(N * BB)); HIPCHECK(h, h->bcrF.resize((size_t)N * B));
  HIPCHECK(h, h->bigK.resize((size_t)N * KS));
  HIPCHECK(h, h->dC.resize((size_t)N * B + 16));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_dense_panel));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_big_backsolve));
  int* info = h->flags.p + 1;
  {
    ScopedTimer tm(h, BA_K_BCR_ASSEMBLE);       // also clears the status word flags[1]
    const int rounds = (int)((BB + 3 * kBcrThreads - 1) / (3 * kBcrThreads));        // (three entries per thread and round)
    hipLaunchKernelGGL(k_bcr_assemble, dim3(N, std::max(1, std::min(rounds, 2048 / N))), dim3(kBcrThreads), 0, h->stream, h->nco, hb, cb, h->S,
                       h->b, dmask, h->bcrD.p, h->bcrU.p, h->bcrF.p, info, (double*)nullptr);
  }
  struct Level { int s, cnt; size_t base; };
  std::vector<Level> levels;
  size_t slots = 0;
  for (int s = 1; (N / s + 1) / 2 > 0; s *= 2) { levels.push_back({s, (N / s + 1) / 2, slots}); slots += (N / s + 1) / 2; }
  const int npanels = (B + kDcNB - 1) / kDcNB;
  {
    ScopedTimer tm(h, BA_K_BCR_ELIMINATE, (int)levels.size() * (2 * npanels + 2));
    for (const Level& L : levels) {
      double* K = h->bigK.p + L.base * KS;
      hipLaunchKernelGGL(k_big_gather, dim3(std::min(n + 1, 128), L.cnt), dim3(256), 0, h->stream, N, B, L.s, h->bcrD.p, h->bcrU.p,
                         h->bcrF.p, K, KS);
      for (int k0 = 0; k0 < B; k0 += kDcNB) {
        const int nb = std::min(kDcNB, B - k0), kn = k0 + nb, total = n - kn + 1;      // every row below the block + the right-hand side row
        hipLaunchKernelGGL(k_dense_panel, dim3((total + kDcRows - 1) / kDcRows, 1, L.cnt), dim3(1024), dense_panel_lds_bytes(), h->stream,
                           n, k0, nb, total, K, info, KS);
        const int T = (total + kDcTile - 1) / kDcTile;
        hipLaunchKernelGGL(k_dense_update, dim3(T, T, L.cnt), dim3(1024), 0, h->stream, n, k0, nb, total, K, KS);
      }
      const int nsurv = (N + 1) / (2 * L.s);       // nodes m = 2 s (y + 1) - 1 < N
      if (nsurv > 0)
        hipLaunchKernelGGL(k_big_scatter, dim3(std::min(B + 1, 64), nsurv), dim3(256), 0, h->stream, N, B, L.s, L.cnt, h->bcrD.p, h->bcrU.p,
                           h->bcrF.p, K, KS);
    }
  }
  ScopedTimer tmb(h, BA_K_BCR_BACKSOLVE, (int)levels.size());
  for (int q = (int)levels.size() - 1; q >= 0; --q) {
    const Level& L = levels[q];
    hipLaunchKernelGGL(k_big_backsolve, dim3(L.cnt), dim3(1024), big_backsolve_lds_bytes(B), h->stream, N, B, L.s,
                       h->bigK.p + L.base * KS, KS, h->dC.p, info);
  }
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

// Dense Cholesky of the whole reduced system (ba_dense.h): bands wider than the cyclic reduction's blocks.
int solve_dense_chol(ba_handle* h, const unsigned char* dmask) {
  const int n = 6 * h->nco;
  HIPCHECK(h, h->denseA.resize((size_t)(n + 1) * n));
  HIPCHECK(h, h->dC.resize((size_t)n + 16));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_dense_panel));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_dense_backsolve));
  int* info = h->flags.p + 1;
  double* A = h->denseA.p;
  const int nsteps = (n + kDcNB - 1) / kDcNB;
  ScopedTimer tm(h, BA_K_DENSE_SOLVE, 2 * nsteps + 1);
  hipLaunchKernelGGL(k_dense_gather, dim3(n + 1), dim3(256), 0, h->stream, h->nco, h->hb, h->S, h->b, dmask, A, info);
  const int bw = std::min(n, 6 * (h->hb + 1) - 1);            // S[r][c] = 0 for |r - c| > bw
  for (int k0 = 0; k0 < n; k0 += kDcNB) {
    const int nb = std::min(kDcNB, n - k0), kn = k0 + nb;
    const int total = std::min(n, kn + bw) - kn + 1;           // rows below the block that can be non-zero + the rhs row
    hipLaunchKernelGGL(k_dense_panel, dim3((total + kDcRows - 1) / kDcRows), dim3(1024), dense_panel_lds_bytes(), h->stream, n,
                       k0, nb, total, A, info);
    if (total > 1) {
      const int T = (total + kDcTile - 1) / kDcTile;
      hipLaunchKernelGGL(k_dense_update, dim3(T, T), dim3(1024), 0, h->stream, n, k0, nb, total, A);
    }
  }
  hipLaunchKernelGGL(k_dense_backsolve, dim3(1), dim3(1024), dense_backsolve_lds_bytes(n), h->stream, n, bw, A, h->dC.p, info);
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

int download_rows(ba_handle* h, const std::vector<int>& perm, const double* dev, double* host, size_t n, int w) {
  if (n == 0) return BA_OK;
  if (perm.empty()) {
    HIPCHECK(h, hipMemcpyAsync(host, dev, n * w * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return BA_OK;
  }
  std::vector<double> tmp(n * w);
  HIPCHECK(h, hipMemcpyAsync(tmp.data(), dev, n * w * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  rows_to_caller(perm, tmp.data(), host, w);
  return BA_OK;
}

}  // namespace

namespace {
// ---- RCCL, resolved at run time (ba_comm_load): the library does not link against it, it uses the one the
// process already has (torch's), so that there is a single RCCL instance per process
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok() const { return GetUniqueId && CommInitRank && AllReduce && CommDestroy; }
};
RcclApi g_rccl;

#define RCCLCHECK(h, call)                                                                                   \
  do {                                                                                                       \
    ncclResult_t r_ = (call);                                                                                \
    if (r_ != ncclSuccess)                                                                                   \
      return (h)->fail(BA_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
  } while (0)

// in-place sum over the shards of the band-stored [S | b] (contiguous), on the handle's stream
int comm_allreduce_reduced(ba_handle* h) {
  const size_t nS = reduced_doubles(h), nb = (size_t)h->nco * 6;
  if (nS + nb == 0) return BA_OK;
  if (h->b == h->S + nS) {                            // the usual case: one contiguous [S | b]
    RCCLCHECK(h, g_rccl.AllReduce(h->S, h->S, nS + nb, ncclFloat64, ncclSum, h->comm, h->stream));
  } else {
    RCCLCHECK(h, g_rccl.AllReduce(h->S, h->S, nS, ncclFloat64, ncclSum, h->comm, h->stream));
    RCCLCHECK(h, g_rccl.AllReduce(h->b, h->b, nb, ncclFloat64, ncclSum, h->comm, h->stream));
  }
  return BA_OK;
}

}  // namespace

__global__ __launch_bounds__(256) void k_copy_doubles(const double* __restrict__ src, double* __restrict__ dst, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// The achievable HBM rate of this box (SURVEY 8d asks for the roofline fraction against it as well as
// against the 8 TB/s of the data sheet): a streaming copy, ONE 16-byte element per lane, no loop, non-temporal
// loads and stores.  tools/copy_probe.hip compares the forms on an MI355X (read + write, 1 GiB): this one
// 6.5 TB/s; the grid-stride loop it replaces 4.5 - 5.0; four elements per lane in flight 5.6 - 6.2;
// hipMemcpyAsync device-to-device 5.2.  (The guide quotes 6.29 TB/s for its float4 copy.)
typedef float copy_vec __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_copy(const copy_vec* __restrict__ src, copy_vec* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// test aid (ba_debug_poison): every LDS word of a compute unit / every double of a workspace buffer becomes a NaN
__global__ __launch_bounds__(1024) void k_poison_lds(int ndoubles) {
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  for (int i = threadIdx.x; i < ndoubles; i += 1024) dyn[i] = __longlong_as_double(0x7FF8DEADDEADDEADll);
  __syncthreads();
  if (dyn[(threadIdx.x * 7) % ndoubles] == 0.0) dyn[0] = 1.0;      // (keeps the stores)
}
__global__ __launch_bounds__(256) void k_poison_doubles(double* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = __longlong_as_double(0x7FF8DEADDEADDEADll);
}

extern "C" {

const char* ba_version(void) { return "pysfm_ba 0.1 (gfx950)"; }

const char* ba_kernel_name(int id) {
  static const char* names[BA_K_COUNT] = {"k_cost", "k_linearize", "k_point_invert", "k_schur_init",
                                          "k_schur_pairs", "k_backsub", "k_apply_update", "k_flatten",
                                          "k_band_solve", "k_eval", "k_camera_blocks", "k_triangulate",
                                          "k_bcr_assemble", "k_bcr_eliminate", "k_bcr_backsolve", "k_dense_solve"};
  return (id >= 0 && id < BA_K_COUNT) ? names[id] : "?";
}

const char* ba_last_error(const ba_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ba_create(int device_id, ba_handle** out) {
  if (!out) { g_create_error = "ba_create: out is NULL"; return BA_ERR_INVALID_ARG; }
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = std::string("ba_create: no HIP device available (") +
                     (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") + ")";
    return BA_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= ndev) {
    g_create_error = "ba_create: device_id out of range";
    return BA_ERR_INVALID_ARG;
  }
  if ((e = hipSetDevice(device_id)) != hipSuccess) {
    g_create_error = std::string("ba_create: hipSetDevice failed: ") + hipGetErrorString(e);
    return BA_ERR_HIP;
  }
  ba_handle* h = new ba_handle();
  h->device = device_id;
  (void)hipDeviceGetAttribute(&h->ncu, hipDeviceAttributeMultiprocessorCount, device_id);
  if (h->ncu <= 0) h->ncu = 256;
  if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) {
    g_create_error = std::string("ba_create: hipStreamCreate failed: ") + hipGetErrorString(e);
    delete h;
    return BA_ERR_HIP;
  }
  h->own_stream = true;
  if ((e = hipHostMalloc((void**)&h->host_result, sizeof(HostResult), hipHostMallocDefault)) != hipSuccess) {
    g_create_error = std::string("ba_create: hipHostMalloc failed: ") + hipGetErrorString(e);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return BA_ERR_HIP;
  }
  *out = h;
  return BA_OK;
}

int ba_destroy(ba_handle* h) {
  if (!h) return BA_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)ba_comm_destroy(h);
  for (auto& t : h->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (auto e : h->ev_pool) (void)hipEventDestroy(e);
  h->obs_cam.release(); h->obs_pt.release(); h->pt_off.release(); h->cam_opt_pos.release(); h->opt_cam.release();
  h->keep.release(); h->obs_z.release(); h->pt_opt.release(); h->units.release(); h->chunks.release(); h->groups.release(); h->mgroups.release(); h->gchunks.release(); h->mchunks.release(); h->m3chunks.release(); h->wide_list.release(); h->wgroups.release(); h->wtab.release(); h->cam_perm.release(); h->cam_units.release();
  for (int i = 0; i < 2; ++i) { h->cams[i].release(); h->X[i].release(); }
  h->HCC.release(); h->bC.release(); h->HPP.release(); h->bP.release(); h->HPPinv.release();
  h->W.release(); h->S_own.release(); h->b_own.release(); h->dC.release(); h->Ufac.release(); h->ysol.release(); h->dinv.release();
  h->bcrD.release(); h->bcrU.release(); h->bcrF.release(); h->bcrP.release(); h->bcrQ.release(); h->bcrG.release(); h->bcrGv.release(); h->bcr_order.release(); h->bcr_work.release(); h->bcr_done.release(); h->bcr_trace.release(); h->dist.rows.release(); h->dist.sep.release(); h->dist.sep_owner.release(); h->dist.root.release(); h->dist.root_owner.release(); h->dist.work.release(); h->rgroups.release(); h->rtab.release(); h->dist.order.release(); h->dist.asm_nodes.release(); h->dist.xown.release(); h->bcrL.release(); h->bcrLv.release(); h->denseA.release(); h->bigK.release(); h->fac.release(); h->dUd.release(); h->dDd.release(); h->dyd.release(); h->dpart.release(); h->mask.release(); h->dP.release();
  h->scratch.release(); h->flags.release();
  if (h->host_result) (void)hipHostFree(h->host_result);
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return BA_OK;
}

int ba_debug_poison(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_poison_lds));
  hipLaunchKernelGGL(k_poison_lds, dim3(8 * h->ncu), dim3(1024), 160 * 1024, h->stream, 160 * 1024 / 8);
  DevBuf<double>* bufs[] = {&h->HCC, &h->bC, &h->HPP, &h->bP, &h->HPPinv, &h->W, &h->dC, &h->dP, &h->scratch, &h->Ufac, &h->ysol, &h->dinv,
                            &h->bcrD, &h->bcrU, &h->bcrF, &h->bcrP, &h->bcrQ, &h->bcrG, &h->bcrGv, &h->bcrL, &h->bcrLv, &h->denseA, &h->bigK, &h->fac,
                            &h->dUd, &h->dDd, &h->dyd, &h->dpart, &h->cams[1 - h->cur], &h->X[1 - h->cur]};
  for (DevBuf<double>* b : bufs)
    if (b->p && b->n) hipLaunchKernelGGL(k_poison_doubles, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, h->stream, b->p, b->n);
  if (h->bcr_done.p && h->bcr_done.n >= 2)          // the "handed on" words of k_bcr_eliminate_fused: garbage that reads as "done" unless k_bcr_assemble clears it
    hipLaunchKernelGGL(k_poison_doubles, dim3((unsigned)((h->bcr_done.n / 2 + 255) / 256)), dim3(256), 0, h->stream, reinterpret_cast<double*>(h->bcr_done.p), h->bcr_done.n / 2);
  if (h->S) hipLaunchKernelGGL(k_poison_doubles, dim3((unsigned)((reduced_doubles(h) + 255) / 256)), dim3(256), 0, h->stream, h->S, reduced_doubles(h));
  if (h->b && h->nco) hipLaunchKernelGGL(k_poison_doubles, dim3((unsigned)(((size_t)h->nco * 6 + 255) / 256)), dim3(256), 0, h->stream, h->b, (size_t)h->nco * 6);
  HIPCHECK(h, hipGetLastError());
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  h->have_params[1 - h->cur] = false;
  h->have_linearization = h->have_schur = h->have_backsub = h->have_solution = false;
  h->inv_valid = h->fac_valid = h->point_blocks_valid = h->cam_blocks_valid = false;
  return BA_OK;
}

int ba_set_option(ba_handle* h, const char* name, const char* value) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, name && value, BA_ERR_INVALID_ARG, "ba_set_option: NULL argument");
  const std::string n(name), v(value);
  auto choice = [&](std::initializer_list<const char*> names, int& out) {
    int i = 0;
    for (const char* c : names) { if (v == c) { out = i; return true; } ++i; }
    return false;
  };
  auto flag = [&](bool& out) {
    if (v == "1" || v == "on" || v == "true") { out = true; return true; }
    if (v == "0" || v == "off" || v == "false") { out = false; return true; }
    return false;
  };
  bool ok = false;
  if (n == "schur") ok = choice({"auto", "pairs", "groups", "mfma1", "mfma2", "mfma"}, h->opt.schur);
  else if (n == "solver") ok = choice({"auto", "bcr", "band", "dense", "lu", "bcr1"}, h->opt.solver);
  else if (n == "point_kernels") { int c = 0; ok = choice({"auto", "v1"}, c); if (ok) h->opt.point_kernels_v1 = c == 1; }
  else if (n == "fuse_cost") ok = flag(h->opt.fuse_cost);
  else if (n == "fuse_cam") ok = flag(h->opt.fuse_cam);
  else if (n == "fuse_lin") ok = flag(h->opt.fuse_lin);
  else if (n == "sort_points") ok = flag(h->opt.sort_points);
  else if (n == "solve_trace") ok = flag(h->opt.solve_trace);
  else if (n == "lds_window") ok = flag(h->opt.lds_window);
  else if (n == "fused_backsolve") ok = flag(h->opt.fused_backsolve);
  else if (n == "fused_eliminate") ok = flag(h->opt.fused_eliminate);
  else if (n == "device_lu") ok = flag(h->opt.device_lu);
  else if (n == "fast_paths") ok = flag(h->opt.fast_paths);
  else if (n == "gm_cap") { char* end = nullptr; const long c = strtol(value, &end, 10); ok = end && *end == 0 && c >= 0; if (ok) h->opt.gm_cap = (int)c; }
  else return h->fail(BA_ERR_INVALID_ARG, "ba_set_option: unknown option '%s'", name);
  if (!ok) return h->fail(BA_ERR_INVALID_ARG, "ba_set_option: bad value '%s' for option '%s'", value, name);
  h->inv_valid = h->fac_valid = false;          // a different kernel family may need different by-products
  return BA_OK;
}

int ba_set_stream(ba_handle* h, void* hip_stream) {
  if (!h) return BA_ERR_INVALID_ARG;
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  resolve_timings(h);
  if (h->own_stream) { (void)hipStreamDestroy(h->stream); h->own_stream = false; }
  if (hip_stream) {
    h->stream = (hipStream_t)hip_stream;
  } else {
    HIPCHECK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  return BA_OK;
}

int ba_synchronize(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  return BA_OK;
}

int ba_set_problem(ba_handle* h, int32_t nc, int32_t nt, int64_t nobs, const int32_t* obs_cam,
                   const int32_t* obs_pt, const double* obs_z, const double* K,
                   const int32_t* cam_opt_pos, const uint8_t* pt_opt) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, nc >= 0 && nt >= 0 && nobs >= 0, BA_ERR_INVALID_ARG, "ba_set_problem: negative size");
  REQUIRE(h, nobs < (1ll << 31) - 64, BA_ERR_INVALID_ARG, "ba_set_problem: nobs must fit int32");
  REQUIRE(h, K && (nc == 0 || cam_opt_pos) && (nt == 0 || pt_opt), BA_ERR_INVALID_ARG,
          "ba_set_problem: NULL argument");
  REQUIRE(h, nobs == 0 || (obs_cam && obs_pt && obs_z), BA_ERR_INVALID_ARG, "ba_set_problem: NULL observation array");
  HIPCHECK(h, hipSetDevice(h->device));

  // validate; caller-order CSR by point (any observation order is accepted: stable counting sort)
  std::vector<int> coff((size_t)nt + 1, 0);
  for (int64_t n = 0; n < nobs; ++n) {
    const int c = obs_cam[n], k = obs_pt[n];
    if (c < 0 || c >= nc) return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: obs_cam[%lld]=%d out of range", (long long)n, c);
    if (k < 0 || k >= nt) return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: obs_pt[%lld]=%d out of range", (long long)n, k);
    coff[(size_t)k + 1] += 1;
  }
  for (int k = 0; k < nt; ++k) coff[(size_t)k + 1] += coff[k];
  // optimised-camera positions must be a permutation of 0..nco-1
  int nco = 0;
  for (int i = 0; i < nc; ++i) if (cam_opt_pos[i] >= 0) ++nco;
  {
    std::vector<char> seen((size_t)nco, 0);
    for (int i = 0; i < nc; ++i) {
      const int p = cam_opt_pos[i];
      if (p < 0) continue;
      if (p >= nco || seen[p]) return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: cam_opt_pos is not a permutation of 0..nco-1");
      seen[p] = 1;
    }
  }
  // ---- internal order.  The reference visits tracks and their measurements in any order
  // (bundle_adjuster.py:222-226); the kernels want (a) a track's observations by ascending optimised-camera
  // position, frozen cameras first, and (b) tracks with identical camera lists next to each other, lists
  // ordered by their first optimised camera (image sequences: the reduction's LDS window slides along the
  // band).  So: rank the cameras (frozen ones by index, then the optimised ones by position), sort each
  // track's observations by rank, sort the tracks by (first optimised position, rank list).  `pperm` /
  // `operm` map internal point / observation indices to the caller's; every host-facing array goes through
  // them (identity for a scene that already comes in this order: then they stay empty).
  std::vector<int> by_pt((size_t)nobs);                      // caller observation ids, grouped by caller point
  {
    std::vector<int> cursor(coff.begin(), coff.end() - 1);
    for (int64_t n = 0; n < nobs; ++n) by_pt[(size_t)cursor[obs_pt[n]]++] = (int)n;
  }
  std::vector<int> crank((size_t)nc);
  {
    int f = 0;
    const int nfrozen = nc - nco;
    for (int i = 0; i < nc; ++i) crank[i] = cam_opt_pos[i] < 0 ? f++ : nfrozen + cam_opt_pos[i];
  }
  const bool sort_points = h->opt.sort_points;
  if (sort_points) {
    for (int k = 0; k < nt; ++k) {
      int* b0 = by_pt.data() + coff[k];
      int* b1 = by_pt.data() + coff[(size_t)k + 1];
      auto less = [&](int a, int b) { return crank[obs_cam[a]] < crank[obs_cam[b]]; };
      if (!std::is_sorted(b0, b1, less)) std::stable_sort(b0, b1, less);
    }
  }
  for (int k = 0; k < nt; ++k) {                             // each (camera, track) pair at most once (bundle.py: a dict per track)
    std::vector<int> seen;
    const int L = coff[(size_t)k + 1] - coff[k];
    if (sort_points) {
      for (int q = coff[k] + 1; q < coff[(size_t)k + 1]; ++q)
        if (obs_cam[by_pt[q]] == obs_cam[by_pt[q - 1]])
          return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: track %d has two observations in camera %d", k, obs_cam[by_pt[q]]);
    } else if (L > 1) {
      seen.assign(by_pt.begin() + coff[k], by_pt.begin() + coff[(size_t)k + 1]);
      for (int& v : seen) v = obs_cam[v];
      std::sort(seen.begin(), seen.end());
      if (std::adjacent_find(seen.begin(), seen.end()) != seen.end())
        return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: track %d has two observations in one camera", k);
    }
  }
  std::vector<int> pperm((size_t)nt);
  for (int k = 0; k < nt; ++k) pperm[k] = k;
  if (sort_points && nt > 1) {
    std::vector<int> minpos((size_t)nt, INT32_MAX);
    for (int k = 0; k < nt; ++k)
      for (int q = coff[k]; q < coff[(size_t)k + 1]; ++q) {
        const int p = cam_opt_pos[obs_cam[by_pt[q]]];
        if (p >= 0) { minpos[k] = p; break; }               // (sorted by rank: the first optimised one is the smallest)
      }
    auto less = [&](int a, int b) {
      if (minpos[a] != minpos[b]) return minpos[a] < minpos[b];
      const int la = coff[(size_t)a + 1] - coff[a], lb = coff[(size_t)b + 1] - coff[b];
      const int* pa = by_pt.data() + coff[a];
      const int* pb = by_pt.data() + coff[b];
      for (int q = 0; q < std::min(la, lb); ++q) {
        const int ra = crank[obs_cam[pa[q]]], rb = crank[obs_cam[pb[q]]];
        if (ra != rb) return ra < rb;
      }
      return la < lb;
    };
    if (!std::is_sorted(pperm.begin(), pperm.end(), less)) std::stable_sort(pperm.begin(), pperm.end(), less);
  }
  // internal observation arrays + CSR
  std::vector<int> ic((size_t)nobs), ip((size_t)nobs), operm((size_t)nobs), off((size_t)nt + 1, 0);
  std::vector<double2> iz((size_t)nobs);
  std::vector<unsigned char> ipt_opt((size_t)nt);
  {
    size_t w = 0;
    for (int i = 0; i < nt; ++i) {
      const int k = pperm[i];
      ipt_opt[i] = pt_opt[k];
      for (int q = coff[k]; q < coff[(size_t)k + 1]; ++q, ++w) {
        const int n = by_pt[q];
        operm[w] = n; ic[w] = obs_cam[n]; ip[w] = i;
        iz[w] = double2{obs_z[2 * (size_t)n], obs_z[2 * (size_t)n + 1]};
      }
      off[(size_t)i + 1] = (int)w;
    }
  }
  bool pid = true, oid = true;
  for (int i = 0; i < nt && pid; ++i) pid = pperm[i] == i;
  for (int64_t n = 0; n < nobs && oid; ++n) oid = operm[(size_t)n] == (int)n;
  h->pperm = pid ? std::vector<int>() : pperm;
  h->operm = oid ? std::vector<int>() : operm;
  // from here on everything is in internal order
  obs_cam = ic.data(); obs_pt = ip.data(); pt_opt = ipt_opt.data();

  // Schur work units: (point, row tile, col tile >= row tile)
  std::vector<SchurUnit> units;
  units.reserve((size_t)nt);
  long long maxL = 0;
  for (int k = 0; k < nt; ++k) {
    const int L = off[(size_t)k + 1] - off[k];
    maxL = std::max<long long>(maxL, L);
    for (int r = 0; r < L; r += kTile)
      for (int c = r; c < L; c += kTile) units.push_back({k, r, c});
  }
  // camera-ordered view of the observations for k_camera_blocks: counting sort by camera
  std::vector<int> cam_off((size_t)nc + 1, 0), perm((size_t)nobs);
  for (int64_t n = 0; n < nobs; ++n) cam_off[(size_t)obs_cam[n] + 1] += 1;
  for (int i = 0; i < nc; ++i) cam_off[(size_t)i + 1] += cam_off[i];
  {
    std::vector<int> cursor(cam_off.begin(), cam_off.end() - 1);
    for (int64_t n = 0; n < nobs; ++n) perm[(size_t)cursor[obs_cam[n]]++] = (int)n;
  }
  std::vector<CamUnit> cam_units;
  // one wavefront per unit: few cameras with long observation lists (dense visibility) would leave the chip
  // empty at kCamChunk observations per unit, so shrink the chunk until there are about 500 units (measured: 100 000 observations of 100 cameras: 28 us at 2048 per unit, 14 at 256, 19 at 64); many
  // cameras with ~1000 observations each keep one unit per camera (one atomic result per camera)
  const int cam_chunk = (int)std::min<int64_t>(kCamChunk, std::max<int64_t>(64, (nobs / 512 + 63) / 64 * 64));
  for (int i = 0; i < nc; ++i)
    for (int s = cam_off[i]; s < cam_off[(size_t)i + 1]; s += cam_chunk)
      cam_units.push_back({i, s, std::min(s + cam_chunk, cam_off[(size_t)i + 1])});
  // block half-bandwidth of the reduced system: widest spread of optimised-camera
  // positions within one track
  int hb = 0;
  for (int k = 0; k < nt; ++k) {
    int lo = INT32_MAX, hi = -1;
    for (int n = off[k]; n < off[(size_t)k + 1]; ++n) {
      const int p = cam_opt_pos[obs_cam[n]];
      if (p < 0) continue;
      lo = std::min(lo, p); hi = std::max(hi, p);
    }
    if (hi >= 0) hb = std::max(hb, hi - lo);
  }
  hb = std::max(hb, std::min(h->min_hb, std::max(0, nco - 1)));   // sharded adjuster: every rank uses the widest band
  // Schur chunks: consecutive units whose optimised-camera positions fit a window of wn
  // band rows, so that a workgroup can accumulate them in an LDS tile
  int wn = (int)(kSchurTileBytes / (((size_t)(hb + 1) * 36 + 6) * sizeof(double)));
  wn = std::min(wn, 64);
  if (wn < hb + 2 || nco == 0) wn = 0;                 // band too wide for an LDS tile: global atomics only
  std::vector<SchurChunk> chunks;
  {
    std::vector<int> plo((size_t)nt, INT32_MAX), phi((size_t)nt, -1);
    for (int k = 0; k < nt; ++k)
      for (int n = off[k]; n < off[(size_t)k + 1]; ++n) {
        const int p = cam_opt_pos[obs_cam[n]];
        if (p < 0) continue;
        plo[k] = std::min(plo[k], p); phi[k] = std::max(phi[k], p);
      }
    int begin = 0, lo = INT32_MAX, hi = -1;
    for (int u = 0; u < (int)units.size(); ++u) {
      const int k = units[u].pt;
      const int nlo = std::min(lo, plo[k]), nhi = std::max(hi, phi[k]);
      const bool fits = wn == 0 || nhi < 0 || nhi - nlo + 1 <= wn;
      if (u > begin && (!fits || u - begin >= kSchurChunkUnits)) {
        chunks.push_back({begin, u, lo == INT32_MAX ? 0 : lo});
        begin = u; lo = plo[k]; hi = phi[k];
      } else {
        lo = nlo; hi = nhi;
      }
    }
    if (!units.empty()) chunks.push_back({begin, (int)units.size(), lo == INT32_MAX ? 0 : lo});
  }
  // Groups: runs of consecutive points (internal order) with identical observation lists.
  //   groups / gchunks   <= kGroupMaxPts points each: k_schur_groups (vector kernel, track length <= 15) and the
  //                      group-packed point kernels k_linearize_groups / k_backsub_groups (<= kGm3MaxL)
  //   mgroups / mchunks  longer runs for the matrix-core reductions; mchunks under the LDS window `wn` of the older
  //                      kernels (track length <= 10), m3chunks under k_schur_groups_mfma3's own window
  std::vector<SchurGroup> groups, mgroups;
  std::vector<SchurChunk> gchunks, mchunks, m3chunks;
  int group_rounds = 0;
  bool groups_worth = false;
  bool groups_ascending = true;                      // optimised positions ascend along every track
  Gm3Params gm3{0, 0, 0, 0, 0, 1, 1};
  if (maxL >= 1 && maxL <= kGm3MaxL) {
    auto build_groups = [&](int max_pts, std::vector<SchurGroup>& gs, std::vector<int>& glo, std::vector<int>& ghi) {
      for (int k = 0; k < nt;) {
        const int L = off[(size_t)k + 1] - off[k];
        if (L == 0) { ++k; continue; }
        int e = k + 1;
        while (e < nt && e - k < max_pts && off[(size_t)e + 1] - off[e] == L &&
               std::equal(obs_cam + off[k], obs_cam + off[k] + L, obs_cam + off[e]))
          ++e;
        int lo = INT32_MAX, hi = -1;
        for (int n = off[k]; n < off[k] + L; ++n) {
          const int p = cam_opt_pos[obs_cam[n]];
          if (p >= 0) {
            if (p <= hi) groups_ascending = false;
            lo = std::min(lo, p); hi = std::max(hi, p);
          }
        }
        gs.push_back({k, e, L, 0});
        glo.push_back(lo); ghi.push_back(hi);
        k = e;
      }
    };
    // consecutive groups whose optimised positions fit a window of `win` band rows (win == 0: no window, by count only)
    auto chunk_groups = [&](int limit, int win, const std::vector<SchurGroup>& gs, const std::vector<int>& glo, const std::vector<int>& ghi,
                            std::vector<SchurChunk>& out) {
      int begin = 0, lo = INT32_MAX, hi = -1;
      for (int g = 0; g < (int)gs.size(); ++g) {
        const int nlo = std::min(lo, glo[g]), nhi = std::max(hi, ghi[g]);
        const bool fits = win == 0 || nhi < 0 || nhi - nlo + 1 <= win;
        if (g > begin && (!fits || g - begin >= limit)) {
          out.push_back({begin, g, lo == INT32_MAX ? 0 : lo});
          begin = g; lo = glo[g]; hi = ghi[g];
        } else {
          lo = nlo; hi = nhi;
        }
      }
      if (!gs.empty()) out.push_back({begin, (int)gs.size(), lo == INT32_MAX ? 0 : lo});
    };
    std::vector<int> glo, ghi, mlo, mhi;
    build_groups(kGroupMaxPts, groups, glo, ghi);
    if (maxL <= kGroupMaxL) chunk_groups(kGroupChunk, wn, groups, glo, ghi, gchunks);
    // MFMA kernels: one group per wavefront pair, and the epilogue is expensive, so runs are cut
    // only where the chip would otherwise idle: about one group per wavefront-pair slot (4 per CU)
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
    const int slots = std::max(1, ncu * (kGmBlock / kWave));
    // The kernel lasts as long as its longest group (one round of workgroups), so: natural runs of points with
    // identical camera lists, runs longer than `cap` cut into EQUAL parts of whole batches, and the smallest cap
    // for which the groups still fit the wavefront-pair slots of one round (config 3: 991 runs of 101 +- 23
    // points, cap 120 -> 1021 groups in 256 workgroups; a fixed 1.25 x mean cap gave groups of 126).
    std::vector<SchurGroup> runs;
    std::vector<int> rlo, rhi;
    build_groups(INT32_MAX, runs, rlo, rhi);
    auto split_runs = [&](int cap, bool emit) -> size_t {
      size_t count = 0;
      for (size_t r = 0; r < runs.size(); ++r) {
        const int n = runs[r].pt_end - runs[r].pt_begin;
        const int k = (n + cap - 1) / cap;
        const int part = ((n + k - 1) / k + kGmPts - 1) / kGmPts * kGmPts;
        for (int b = runs[r].pt_begin; b < runs[r].pt_end; b += part) {
          ++count;
          if (emit) {
            mgroups.push_back({b, std::min(b + part, runs[r].pt_end), runs[r].L, 0});
            mlo.push_back(rlo[r]); mhi.push_back(rhi[r]);
          }
        }
      }
      return count;
    };
    int cap = std::max(kGroupMaxPts, (int)(((nt + slots - 1) / slots + kGmPts - 1) / kGmPts * kGmPts));
    int longest = 0;
    for (const SchurGroup& r : runs) longest = std::max(longest, r.pt_end - r.pt_begin);
    while (cap < longest && split_runs(cap, false) > (size_t)slots) cap += kGmPts;     // (beyond the longest run nothing changes)
    if (h->opt.gm_cap > 0) cap = std::max(kGmPts, h->opt.gm_cap);     // tuning aid (ba_set_option "gm_cap")
    split_runs(cap, true);
    if (maxL <= kGmMaxL && wn > 0) chunk_groups(kGmChunk, wn, mgroups, mlo, mhi, mchunks);
    // worth it only when points really share camera lists
    const double mean_group = groups.empty() ? 0.0 : (double)nt / groups.size();
    groups_worth = mean_group >= 2.0;
    if (groups_worth && maxL <= kGroupMaxL && wn > 0) group_rounds = (int)((maxL * (maxL + 1) / 2 + 63) / 64);
  }
  // Window groups for k_schur_groups_mfma3: consecutive points (internal order: by first optimised position) whose
  // optimised cameras all lie within `wmax` consecutive positions - identical camera lists are NOT required, so tracks
  // of different lengths, tracks with missing observations and tracks that start anywhere all join.  wmax = the
  // widest window that costs no more 16-row tiles than the widest track needs.
  std::vector<WinGroup> wgroups;
  std::vector<int> wtab;
  bool wgroups_worth = false;
  std::vector<RectGroup> rgroups;
  std::vector<int> rtab, wide_list;
  int wide_begin[kGwMaxTiles - kGwMinTiles + 2] = {0};
  int nlong_points = 0;
  {
    std::vector<int> plo((size_t)nt, INT32_MAX), phi((size_t)nt, -1);
    int maxspan = 0;
    for (int k = 0; k < nt; ++k) {
      for (int n = off[k]; n < off[(size_t)k + 1]; ++n) {
        const int p = cam_opt_pos[obs_cam[n]];
        if (p < 0) continue;
        plo[k] = std::min(plo[k], p); phi[k] = std::max(phi[k], p);
      }
      if (phi[k] >= 0) maxspan = std::max(maxspan, phi[k] - plo[k] + 1);
    }
    bool sorted_by_lo = true;                          // (the internal sort guarantees it; "sort_points" = 0 may not)
    for (int k = 1, last = -1; k < nt && sorted_by_lo; ++k) {
      if (phi[k - 1] >= 0) last = plo[k - 1];
      if (phi[k] >= 0 && plo[k] < last) sorted_by_lo = false;
    }
    // Points whose optimised cameras span MORE than the widest window (features that survive for a long stretch of a video):
    // their cameras are cut along a grid of SEGMENTS of kRectSeg positions, and what a point adds to S is a sum over the pairs
    // (A <= B) of segments it touches, S[A, B] -= U_A^T D U_B: k_schur_rect_mfma over the points that touch both (rgroups).
    // A == B also carries the right-hand side and the camera blocks - every observation lies in exactly one segment.
    auto is_long = [&](int k) { return phi[k] >= 0 && phi[k] - plo[k] + 1 > kGm3MaxSpan; };
    long long nlong = 0;
    int shortspan = 0;
    for (int k = 0; k < nt; ++k) {
      if (phi[k] < 0) continue;
      if (is_long(k)) ++nlong; else shortspan = std::max(shortspan, phi[k] - plo[k] + 1);
    }
    const bool hybrid = nlong > 0 && sorted_by_lo && nco > 0;
    struct SegTask { int qa, qb; std::vector<int> pts; };
    std::vector<SegTask> rect_tasks;
    if (hybrid) {
      maxspan = shortspan;
      nlong_points = (int)nlong;
      std::map<long long, int> rect_id;
      for (int k = 0; k < nt; ++k) {
        if (!is_long(k)) continue;
        std::vector<int> segs;
        for (int n = off[k]; n < off[(size_t)k + 1]; ++n) {
          const int p = cam_opt_pos[obs_cam[n]];
          if (p >= 0) segs.push_back(p / kRectSeg);
        }
        std::sort(segs.begin(), segs.end());
        segs.erase(std::unique(segs.begin(), segs.end()), segs.end());
        for (size_t x = 0; x < segs.size(); ++x) {
          for (size_t y = x; y < segs.size(); ++y) {
            const long long key = ((long long)segs[x] << 32) | (unsigned)segs[y];
            auto jt = rect_id.find(key);
            if (jt == rect_id.end()) { jt = rect_id.emplace(key, (int)rect_tasks.size()).first; rect_tasks.push_back({segs[x], segs[y], {}}); }
            rect_tasks[jt->second].pts.push_back(k);
          }
        }
      }
    }
    if ((maxspan >= 1 || hybrid) && maxspan <= kGm3MaxSpan && sorted_by_lo && nco > 0) {
      gm3.nts = (6 * maxspan + 15) / 16;
      gm3.Ld = 16 * gm3.nts;
      const int wmax = std::max(1, std::min(kGm3MaxSpan, gm3.Ld / 6));
      // natural groups: extend while the window still holds everybody
      struct Run { int b, e, lo, hi; };
      std::vector<Run> runs;
      for (int k = 0; k < nt;) {
        if (phi[k] < 0 || (hybrid && is_long(k))) { ++k; continue; }      // (points without an optimised camera add nothing to S or b; long ones are members of their segments' groups)
        int e = k + 1, lo = plo[k], hi = phi[k];
        while (e < nt && !(hybrid && is_long(e)) && (phi[e] < 0 || std::max(hi, phi[e]) - lo + 1 <= wmax)) {
          if (phi[e] >= 0) hi = std::max(hi, phi[e]);
          ++e;
        }
        while (e > k + 1 && phi[e - 1] < 0) --e;       // no trailing points without cameras
        runs.push_back({k, e, lo, hi});
        k = e;
      }
      // cut long groups into equal parts so that one round of workgroups holds them all and none lasts much longer
      // than the rest (as for the identical-list groups above)
      int ncu = 256;
      (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
      const int slots = std::max(1, ncu * kGm2Pairs);
      auto parts_of = [&](const Run& r, int cap) { return (r.e - r.b + cap - 1) / cap; };
      int cap = std::max(kGroupMaxPts, (int)(((nt + slots - 1) / slots + kGmPts - 1) / kGmPts * kGmPts));
      int longest = 0;
      for (const Run& r : runs) longest = std::max(longest, r.e - r.b);
      auto count = [&](int c) { size_t n = 0; for (const Run& r : runs) n += parts_of(r, c); return n; };
      while (cap < longest && count(cap) > (size_t)slots) cap += kGmPts;
      if (h->opt.gm_cap > 0) cap = std::max(kGmPts, h->opt.gm_cap);
      std::vector<int> wlo, whi;
      for (const Run& r : runs) {
        const int n = r.e - r.b, k = parts_of(r, cap);
        const int part = ((n + k - 1) / k + kGmPts - 1) / kGmPts * kGmPts;
        for (int b0 = r.b; b0 < r.e; b0 += part) {
          const int e0 = std::min(b0 + part, r.e);
          int lo = INT32_MAX, hi = -1;
          for (int q = b0; q < e0; ++q)
            if (phi[q] >= 0) { lo = std::min(lo, plo[q]); hi = std::max(hi, phi[q]); }
          if (hi < 0) continue;
          const int W = hi - lo + 1;
          WinGroup g{b0, e0, W, lo, (int)wtab.size(), 0, 0, 0};
          wtab.resize(wtab.size() + (size_t)(e0 - b0) * W, -1);
          for (int q = b0; q < e0; ++q)
            for (int n2 = off[q]; n2 < off[(size_t)q + 1]; ++n2) {
              const int p = cam_opt_pos[obs_cam[n2]];
              if (p >= 0) wtab[(size_t)g.tab + (size_t)(q - b0) * W + (p - lo)] = n2;
            }
          wgroups.push_back(g);
          wlo.push_back(lo); whi.push_back(hi);
        }
      }
      // groups of 27 .. 40 cameras go through k_schur_wide_mfma (every observation linearised once): narrow ones first.  When the
      // narrow ones are few beside them (tracks cut short by the end of the sequence), they go the same way - their own two to
      // five launches of k_schur_groups_mfma3 would each last as long as one group
      {
        auto tiles = [&](int g) { return (6 * wgroups[g].W + 15) >> 4; };
        long long pn = 0, pw = 0;
        for (size_t g = 0; g < wgroups.size(); ++g) (tiles((int)g) >= kGwMinTiles ? pw : pn) += wgroups[g].pt_end - wgroups[g].pt_begin;
        const bool all_wide = pw > 0 && 4 * pn <= pw;
        auto cls = [&](int g) { return (tiles(g) >= kGwMinTiles || all_wide) ? std::max(tiles(g), kGwMinTiles) : 0; };      // 0: narrow
        std::vector<int> order(wgroups.size());
        for (size_t g = 0; g < order.size(); ++g) order[g] = (int)g;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cls(x) < cls(y); });
        std::vector<WinGroup> wg2(wgroups.size());
        std::vector<int> lo2(wgroups.size()), hi2(wgroups.size()), cl2(wgroups.size());
        for (size_t g = 0; g < order.size(); ++g) { wg2[g] = wgroups[order[g]]; lo2[g] = wlo[order[g]]; hi2[g] = whi[order[g]]; cl2[g] = cls(order[g]); }
        wgroups.swap(wg2); wlo.swap(lo2); whi.swap(hi2);
        for (size_t g = 0; g < cl2.size(); ++g)
          if (cl2[g] > 0) wide_list.push_back((int)g);
        for (int t = kGwMinTiles; t <= kGwMaxTiles + 1; ++t) {      // wide_list is sorted by class: where class t begins
          int n = 0;
          for (size_t g = 0; g < cl2.size(); ++g) n += (cl2[g] > 0 && cl2[g] < t) ? 1 : 0;
          wide_begin[t - kGwMinTiles] = n;
        }
      }
      const int nshort_groups = (int)wgroups.size() - (int)wide_list.size();
      if (nshort_groups > 0) {                              // the narrow groups' launches: as many tiles as THEY need
        int wn = 1;
        for (int g = 0; g < nshort_groups; ++g) wn = std::max(wn, wgroups[g].W);
        gm3.nts = (6 * wn + 15) / 16;
        gm3.Ld = 16 * gm3.nts;
      }
      // points per group: a group is one serial chain of batches on one workgroup, and there are few long tracks - halve the
      // groups until they fill the chip twice (24 points at least: an epilogue of up to 144 tiles is paid per group)
      int rect_pts = kRectGroupPts;
      for (; rect_pts > 24; rect_pts /= 2) {
        size_t ngr = 0;
        for (const SegTask& t : rect_tasks) ngr += (t.pts.size() + rect_pts - 1) / rect_pts;
        if (ngr >= (size_t)2 * ncu) break;                 // one workgroup per group, one workgroup per compute unit (its consumers' registers)
      }
      // groups of the pairs of segments: table rows of 2 kRectSeg columns [A | B] (A == B: the B half stays empty)
      for (const SegTask& t : rect_tasks) {
        const int loA = t.qa * kRectSeg, loB = t.qb * kRectSeg, WB = std::min(kRectSeg, nco - loB);
        const size_t parts = (t.pts.size() + rect_pts - 1) / rect_pts, part = (t.pts.size() + parts - 1) / parts;
        for (size_t b0 = 0; b0 < t.pts.size(); b0 += part) {
          const int cnt = (int)std::min<size_t>(part, t.pts.size() - b0);
          RectGroup g{cnt, (int)rtab.size(), 0, loA, loB, WB, 0, 0};
          rtab.insert(rtab.end(), t.pts.begin() + b0, t.pts.begin() + b0 + cnt);
          g.tab = (int)rtab.size();
          rtab.resize(rtab.size() + (size_t)cnt * 2 * kRectSeg, -1);
          for (int q = 0; q < cnt; ++q) {
            const int k = t.pts[b0 + q];
            for (int n2 = off[k]; n2 < off[(size_t)k + 1]; ++n2) {
              const int p = cam_opt_pos[obs_cam[n2]];
              if (p >= loA && p < loA + kRectSeg) rtab[(size_t)g.tab + (size_t)q * 2 * kRectSeg + (p - loA)] = n2;
              else if (t.qa != t.qb && p >= loB && p < loB + WB) rtab[(size_t)g.tab + (size_t)q * 2 * kRectSeg + kRectSeg + (p - loB)] = n2;
            }
          }
          rgroups.push_back(g);
        }
      }
      // staging: four wavefront pairs with two buffers each must fit in LDS next to the (optional) accumulation window
      auto finish_set = [&](int g0, int g1, Gm3Params& G, std::vector<SchurChunk>& out) {
        if (g1 <= g0) return;
        const size_t lds_total = 160 * 1024, fixed = schur_mfma3_lds_bytes(0, 0, 0, hb + 1) + 1024;
        G.wb1 = 1;                                        // rows of the LDS window: as many blocks as the widest group spans
        for (int g = g0; g < g1; ++g) G.wb1 = std::max(G.wb1, std::min(hb + 1, wgroups[g].W));
        for (G.np_cap = kGmPts; G.np_cap >= 1; --G.np_cap) {
          int kmax = 4;
          for (int g = g0; g < g1; ++g) kmax = std::max(kmax, (3 * gm3_np(wgroups[g].W, G.np_cap) + 3) / 4 * 4);
          G.Kbuf = kmax;
          if ((size_t)kGm2Pairs * 2 * G.Kbuf * G.Ld * sizeof(double) <= 96 * 1024) break;
        }
        G.np_cap = std::max(1, G.np_cap);
        const size_t staging = (size_t)kGm2Pairs * 2 * G.Kbuf * G.Ld * sizeof(double);
        const size_t rowbytes = ((size_t)G.wb1 * 36 + 6) * sizeof(double);
        int w3 = (int)((lds_total - fixed - staging) / rowbytes);
        w3 = std::min(w3, std::max(16, G.wb1 + 6));
        if (w3 < G.wb1 + 1 || !h->opt.lds_window) w3 = 0;
        G.wn = w3;
        int begin = g0, lo = INT32_MAX, hi = -1;          // chunks of <= kGmChunk groups under the LDS window
        for (int g = g0; g < g1; ++g) {
          const int nlo = std::min(lo, wlo[g]), nhi = std::max(hi, whi[g]);
          const bool fits = w3 == 0 || nhi - nlo + 1 <= w3;
          if (g > begin && (!fits || g - begin >= kGmChunk)) {
            out.push_back({begin, g, lo});
            begin = g; lo = wlo[g]; hi = whi[g];
          } else {
            lo = nlo; hi = nhi;
          }
        }
        out.push_back({begin, g1, lo});
      };
      finish_set(0, nshort_groups, gm3, m3chunks);
      // an epilogue per >= 12 points (the short tracks' groups decide; a scene of nothing but long tracks: its segment groups)
      long long covered = 0;
      for (const WinGroup& g : wgroups) covered += g.pt_end - g.pt_begin;
      wgroups_worth = wgroups.empty() ? !rgroups.empty() : covered >= 12ll * (long long)wgroups.size();
    }
  }
  // lanes per point: smallest power of two >= mean track length, in [1, 64]
  int glog = 0;
  const double meanL = nt > 0 ? (double)nobs / nt : 1.0;
  while ((1 << glog) < meanL && glog < 6) ++glog;

  h->nc = nc; h->nt = nt; h->nco = nco; h->hb = hb; h->nobs = nobs; h->glog = glog;
  std::memcpy(h->K, K, sizeof h->K);
  h->h_cam_opt_pos.assign(cam_opt_pos, cam_opt_pos + nc);
  h->h_pt_opt.assign(pt_opt, pt_opt + nt);
  h->nunits = (int)units.size();
  h->nchunks = (int)chunks.size();
  h->schur_wn = wn;
  h->ngchunks = (int)gchunks.size();
  h->nmchunks = (int)mchunks.size();
  h->nm3chunks = (int)m3chunks.size();
  h->nwgroups = (int)wgroups.size();
  h->gm3_uniform_ks = !m3chunks.empty();
  for (const SchurChunk& c : m3chunks)
    for (int g = c.begin; g < c.end; ++g) h->gm3_uniform_ks = h->gm3_uniform_ks && gm3_np(wgroups[g].W, gm3.np_cap) == kGmPts;
  h->wgroups_worth = wgroups_worth;
  h->nrgroups = (int)rgroups.size();
  h->nwide = (int)wide_list.size();
  std::memcpy(h->wide_begin, wide_begin, sizeof wide_begin);
  h->nlong_points = rgroups.empty() ? 0 : nlong_points;
  h->gm3 = gm3;
  h->groups_worth = groups_worth;
  h->nmgroups_total = (int)mgroups.size();
  h->groups_ascending = groups_ascending;
  h->ngroups = (int)groups.size();
  {
    long long covered = 0;
    for (const SchurGroup& g : groups) covered += g.pt_end - g.pt_begin;
    h->point_groups = groups_worth && covered == nt;        // (points without observations are in no group)
  }
  h->group_rounds = group_rounds;
  h->group_maxL = maxL;
  h->ncam_units = (int)cam_units.size();

  HIPCHECK(h, h->obs_cam.resize(std::max<size_t>(1, nobs)));
  HIPCHECK(h, h->obs_pt.resize(std::max<size_t>(1, nobs)));
  HIPCHECK(h, h->obs_z.resize(std::max<size_t>(1, nobs)));
  HIPCHECK(h, h->pt_off.resize((size_t)nt + 1));
  HIPCHECK(h, h->cam_opt_pos.resize(std::max(1, nc)));
  HIPCHECK(h, h->pt_opt.resize(std::max(1, nt)));
  HIPCHECK(h, h->wide_list.resize(std::max<size_t>(1, wide_list.size())));
  if (!wide_list.empty())
    HIPCHECK(h, hipMemcpyAsync(h->wide_list.p, wide_list.data(), wide_list.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, h->rgroups.resize(std::max<size_t>(1, rgroups.size())));
  HIPCHECK(h, h->rtab.resize(std::max<size_t>(1, rtab.size())));
  if (!rgroups.empty()) {
    HIPCHECK(h, hipMemcpyAsync(h->rgroups.p, rgroups.data(), rgroups.size() * sizeof(RectGroup), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(h->rtab.p, rtab.data(), rtab.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHECK(h, h->units.resize(std::max<size_t>(1, units.size())));
  HIPCHECK(h, h->chunks.resize(std::max<size_t>(1, chunks.size())));
  if (!chunks.empty())
    HIPCHECK(h, hipMemcpyAsync(h->chunks.p, chunks.data(), chunks.size() * sizeof(SchurChunk), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, h->groups.resize(std::max<size_t>(1, groups.size())));
  HIPCHECK(h, h->gchunks.resize(std::max<size_t>(1, gchunks.size())));
  HIPCHECK(h, h->mchunks.resize(std::max<size_t>(1, mchunks.size())));
  HIPCHECK(h, h->m3chunks.resize(std::max<size_t>(1, m3chunks.size())));
  if (!m3chunks.empty())
    HIPCHECK(h, hipMemcpyAsync(h->m3chunks.p, m3chunks.data(), m3chunks.size() * sizeof(SchurChunk), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, h->wgroups.resize(std::max<size_t>(1, wgroups.size())));
  HIPCHECK(h, h->wtab.resize(std::max<size_t>(1, wtab.size())));
  if (!wgroups.empty()) {
    HIPCHECK(h, hipMemcpyAsync(h->wgroups.p, wgroups.data(), wgroups.size() * sizeof(WinGroup), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(h->wtab.p, wtab.data(), wtab.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHECK(h, h->mgroups.resize(std::max<size_t>(1, mgroups.size())));
  if (!mgroups.empty())
    HIPCHECK(h, hipMemcpyAsync(h->mgroups.p, mgroups.data(), mgroups.size() * sizeof(SchurGroup), hipMemcpyHostToDevice, h->stream));
  if (!mchunks.empty())
    HIPCHECK(h, hipMemcpyAsync(h->mchunks.p, mchunks.data(), mchunks.size() * sizeof(SchurChunk), hipMemcpyHostToDevice, h->stream));
  if (!groups.empty()) {
    HIPCHECK(h, hipMemcpyAsync(h->groups.p, groups.data(), groups.size() * sizeof(SchurGroup), hipMemcpyHostToDevice, h->stream));
    if (!gchunks.empty())
      HIPCHECK(h, hipMemcpyAsync(h->gchunks.p, gchunks.data(), gchunks.size() * sizeof(SchurChunk), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHECK(h, h->cam_perm.resize(std::max<size_t>(1, perm.size())));
  HIPCHECK(h, h->cam_units.resize(std::max<size_t>(1, cam_units.size())));
  if (!perm.empty())
    HIPCHECK(h, hipMemcpyAsync(h->cam_perm.p, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (!cam_units.empty())
    HIPCHECK(h, hipMemcpyAsync(h->cam_units.p, cam_units.data(), cam_units.size() * sizeof(CamUnit), hipMemcpyHostToDevice, h->stream));
  if (nobs) {
    HIPCHECK(h, hipMemcpyAsync(h->obs_cam.p, obs_cam, nobs * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(h->obs_pt.p, obs_pt, nobs * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(h->obs_z.p, iz.data(), nobs * sizeof(double2), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHECK(h, hipMemcpyAsync(h->pt_off.p, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (nc) HIPCHECK(h, hipMemcpyAsync(h->cam_opt_pos.p, cam_opt_pos, nc * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (nt) HIPCHECK(h, hipMemcpyAsync(h->pt_opt.p, pt_opt, nt, hipMemcpyHostToDevice, h->stream));
  if (!units.empty())
    HIPCHECK(h, hipMemcpyAsync(h->units.p, units.data(), units.size() * sizeof(SchurUnit), hipMemcpyHostToDevice, h->stream));

  for (int i = 0; i < 2; ++i) {
    HIPCHECK(h, h->cams[i].resize(std::max<size_t>(1, (size_t)nc * 12)));
    HIPCHECK(h, h->X[i].resize(std::max<size_t>(1, (size_t)nt * 3)));
  }
  HIPCHECK(h, h->HCC.resize(std::max<size_t>(1, (size_t)nc * 36)));
  HIPCHECK(h, h->bC.resize(std::max<size_t>(1, (size_t)nc * 6)));
  HIPCHECK(h, h->HPP.resize(std::max<size_t>(1, (size_t)nt * 6)));
  HIPCHECK(h, h->bP.resize(std::max<size_t>(1, (size_t)nt * 3)));
  HIPCHECK(h, h->HPPinv.resize(std::max<size_t>(1, (size_t)nt * 6)));
  HIPCHECK(h, h->dC.resize(((size_t)nco + 16) * 6));     // padded: the cyclic-reduction solve writes whole super-blocks
  HIPCHECK(h, h->ysol.resize(std::max<size_t>(1, (size_t)nco * 6)));
  HIPCHECK(h, h->dinv.resize(std::max<size_t>(1, (size_t)nco * 6)));
  HIPCHECK(h, h->mask.resize(std::max<size_t>(1, (size_t)nco * 6)));
  HIPCHECK(h, h->dP.resize(std::max<size_t>(1, (size_t)nt * 3)));
  HIPCHECK(h, h->flags.resize(64));
  HIPCHECK(h, hipMemsetAsync(h->flags.p, 0, 64 * sizeof(int), h->stream));
  {
    std::vector<int> opt_cam(std::max(1, nco), 0);
    for (int i = 0; i < nc; ++i) if (cam_opt_pos[i] >= 0) opt_cam[cam_opt_pos[i]] = i;
    HIPCHECK(h, h->opt_cam.resize(opt_cam.size()));
    HIPCHECK(h, hipMemcpyAsync(h->opt_cam.p, opt_cam.data(), opt_cam.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
  }
  // a reduced system bound for another problem size is no longer valid
  h->S = nullptr; h->b = nullptr;
  h->have_problem = true;
  h->dist.on = false;                 // (a cut of the solve over the ranks belongs to the problem it was made for: ba_dist_enable)
  h->have_params[0] = h->have_params[1] = false;
  h->have_linearization = h->have_schur = h->have_backsub = h->have_solution = false;
  h->cur = 0;
  HIPCHECK(h, hipStreamSynchronize(h->stream));   // host vectors go out of scope
  return BA_OK;
}

int ba_set_sensor(ba_handle* h, int kind, const double* params, int nparams) {
  if (!h) return BA_ERR_INVALID_ARG;
  Sensor s{kind, {1, 0, 0, 1}, 1.0, 1.0, 0};
  switch (kind) {
    case BA_SENSOR_GAUSS:
      REQUIRE(h, params && nparams == 4, BA_ERR_INVALID_ARG, "ba_set_sensor: Gaussian needs 4 params (L row-major)");
      for (int i = 0; i < 4; ++i) s.L[i] = params[i];
      if (s.L[0] == 1.0 && s.L[1] == 0.0 && s.L[2] == 0.0 && s.L[3] == 1.0) s.fast |= FAST_UNIT_GAUSS;
      break;
    case BA_SENSOR_CAUCHY:
      REQUIRE(h, params && nparams == 1 && params[0] > 0, BA_ERR_INVALID_ARG, "ba_set_sensor: Cauchy needs sigma > 0");
      s.sigma = params[0];
      break;
    case BA_SENSOR_HUBER:
      REQUIRE(h, params && nparams == 1 && params[0] > 0, BA_ERR_INVALID_ARG, "ba_set_sensor: Huber needs k > 0");
      s.k = params[0];
      break;
    default:
      return h->fail(BA_ERR_INVALID_ARG, "ba_set_sensor: unknown kind %d", kind);
  }
  h->sensor = s;
  h->have_linearization = h->have_schur = h->have_backsub = false;
  return BA_OK;
}

int ba_set_params(ba_handle* h, int which, const double* R, const double* t, const double* X) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_set_params: call ba_set_problem first");
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_set_params: bad parameter set");
  REQUIRE(h, (h->nc == 0 || (R && t)) && (h->nt == 0 || X), BA_ERR_INVALID_ARG, "ba_set_params: NULL argument");
  HIPCHECK(h, hipSetDevice(h->device));
  const int p = h->phys(which);
  std::vector<double> packed((size_t)h->nc * 12);
  for (int i = 0; i < h->nc; ++i) {
    std::memcpy(&packed[(size_t)i * 12], R + (size_t)i * 9, 9 * sizeof(double));
    std::memcpy(&packed[(size_t)i * 12 + 9], t + (size_t)i * 3, 3 * sizeof(double));
  }
  if (h->nc) HIPCHECK(h, hipMemcpyAsync(h->cams[p].p, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  std::vector<double> Xi;
  if (h->nt && !h->pperm.empty()) {
    Xi.resize((size_t)h->nt * 3);
    rows_to_internal(h->pperm, X, Xi.data(), 3);
    X = Xi.data();
  }
  if (h->nt) HIPCHECK(h, hipMemcpyAsync(h->X[p].p, X, (size_t)h->nt * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  h->have_params[p] = true;
  if (which == BA_PARAMS_CUR) h->have_linearization = h->have_schur = h->have_backsub = false;
  return BA_OK;
}

int ba_get_params(ba_handle* h, int which, double* R, double* t, double* X) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_get_params: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_problem && h->have_params[p], BA_ERR_STATE, "ba_get_params: parameter set is empty");
  HIPCHECK(h, hipSetDevice(h->device));
  std::vector<double> packed((size_t)h->nc * 12);
  if (h->nc) HIPCHECK(h, hipMemcpyAsync(packed.data(), h->cams[p].p, packed.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (X) { const int rc = download_rows(h, h->pperm, h->X[p].p, X, (size_t)h->nt, 3); if (rc != BA_OK) return rc; }
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < h->nc; ++i) {
    if (R) std::memcpy(R + (size_t)i * 9, &packed[(size_t)i * 12], 9 * sizeof(double));
    if (t) std::memcpy(t + (size_t)i * 3, &packed[(size_t)i * 12 + 9], 3 * sizeof(double));
  }
  return BA_OK;
}

int ba_swap_params(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem && h->have_params[1 - h->cur], BA_ERR_STATE, "ba_swap_params: trial set is empty");
  h->cur = 1 - h->cur;
  h->have_linearization = h->have_schur = h->have_backsub = false;
  return BA_OK;
}

int ba_cost(ba_handle* h, int which, double* cost_out) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, cost_out, BA_ERR_INVALID_ARG, "ba_cost: cost_out is NULL");
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_cost: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_problem && h->have_params[p], BA_ERR_STATE, "ba_cost: set problem and parameters first");
  HIPCHECK(h, hipSetDevice(h->device));
  const int nb = (int)std::max<long long>(1, std::min<long long>(kCostBlocks, blocks_for(h->nobs)));
  {
    ScopedTimer tm(h, BA_K_COST);
    hipLaunchKernelGGL(k_cost, dim3(nb), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p, h->X[p].p,
                       (const int*)h->sing_counter(), (const int*)(h->flags.p + 1), h->host_result, h->trial_result_dev);
  }
  h->cost_blocks = nb;
  HIPCHECK(h, hipGetLastError());
  if (h->defer) return BA_OK;
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  *cost_out = h->host_cost();
  return BA_OK;
}

int ba_eval_observations(ba_handle* h, int which, double* e, double* r, double* Jc, double* Jp) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_eval_observations: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_problem && h->have_params[p], BA_ERR_STATE, "ba_eval_observations: set problem and parameters first");
  if (h->nobs == 0) return BA_OK;
  HIPCHECK(h, hipSetDevice(h->device));
  const size_t N = (size_t)h->nobs;
  const size_t need = (e ? 2 * N : 0) + (r ? 2 * N : 0) + (Jc ? 12 * N : 0) + (Jp ? 6 * N : 0);
  if (!need) return BA_OK;
  HIPCHECK(h, h->scratch.resize(need));
  double* d = h->scratch.p;
  double* de = nullptr; double* dr = nullptr; double* dJc = nullptr; double* dJp = nullptr;
  if (e) { de = d; d += 2 * N; }
  if (r) { dr = d; d += 2 * N; }
  if (Jc) { dJc = d; d += 12 * N; }
  if (Jp) { dJp = d; d += 6 * N; }
  {
    ScopedTimer tm(h, BA_K_EVAL);
    hipLaunchKernelGGL(k_eval, dim3(blocks_for(h->nobs)), dim3(kBlock), 0, h->stream, dev_problem(h),
                       h->cams[p].p, h->X[p].p, de, dr, dJc, dJp);
  }
  HIPCHECK(h, hipGetLastError());
  int rc = BA_OK;
  if (e && rc == BA_OK) rc = download_rows(h, h->operm, de, e, N, 2);
  if (r && rc == BA_OK) rc = download_rows(h, h->operm, dr, r, N, 2);
  if (Jc && rc == BA_OK) rc = download_rows(h, h->operm, dJc, Jc, N, 12);
  if (Jp && rc == BA_OK) rc = download_rows(h, h->operm, dJp, Jp, N, 6);
  if (rc != BA_OK) return rc;
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  return BA_OK;
}

int ba_eval_sensor(ba_handle* h, int64_t n, const double* e, double* r, double* J) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, n >= 0 && (n == 0 || e), BA_ERR_INVALID_ARG, "ba_eval_sensor: bad arguments");
  if (n == 0 || (!r && !J)) return BA_OK;
  HIPCHECK(h, hipSetDevice(h->device));
  const size_t N = (size_t)n;
  HIPCHECK(h, h->scratch.resize(8 * N));
  double* de = h->scratch.p;
  double* dr = de + 2 * N;
  double* dJ = dr + 2 * N;
  HIPCHECK(h, hipMemcpyAsync(de, e, 2 * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  {
    ScopedTimer tm(h, BA_K_EVAL);
    hipLaunchKernelGGL(k_eval_sensor, dim3(blocks_for(n)), dim3(kBlock), 0, h->stream, h->sensor, (long long)n, de,
                       r ? dr : nullptr, J ? dJ : nullptr);
  }
  HIPCHECK(h, hipGetLastError());
  if (r) HIPCHECK(h, hipMemcpyAsync(r, dr, 2 * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (J) HIPCHECK(h, hipMemcpyAsync(J, dJ, 4 * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  return BA_OK;
}

namespace {

// Which kernel forms the Schur reduction (bundle_adjuster.py:259-278) for this problem and these options.
enum { KERN_PAIRS = 0, KERN_GROUPS, KERN_MFMA1, KERN_MFMA2, KERN_MFMA3, KERN_DENSE };
int pick_schur_kernel(const ba_handle* h) {
  if (h->dense_mode && h->nt > 0 && h->nco > 0) return KERN_DENSE;
  const bool asc = h->groups_ascending && h->group_maxL >= 1;
  const bool m3 = (h->nm3chunks > 0 && h->nwgroups > 0) || h->nrgroups > 0 || h->nwide > 0;           // window groups: no identical camera lists needed
  const bool m12 = asc && h->nmchunks > 0 && h->schur_wn > 0 && h->group_maxL <= kGmMaxL;      // the L <= 10 kernels
  const bool vec = h->ngchunks > 0 && h->schur_wn > 0 && h->group_maxL <= kGroupMaxL;
  switch (h->opt.schur) {
    case SCHUR_PAIRS: return KERN_PAIRS;
    case SCHUR_GROUPS: return vec ? KERN_GROUPS : KERN_PAIRS;
    case SCHUR_MFMA1: return m12 ? KERN_MFMA1 : KERN_PAIRS;
    case SCHUR_MFMA2: return m12 ? KERN_MFMA2 : KERN_PAIRS;
    case SCHUR_MFMA: return m3 ? KERN_MFMA3 : KERN_PAIRS;
    default: break;
  }
  if (h->groups_worth && h->opt.fuse_lin && m12) return KERN_MFMA1;   // (forms the point blocks itself: measured slower, kept tested)
  if (h->groups_worth && m12) return KERN_MFMA2;          // runs of identical camera lists, track length <= 10: the fixed-shape kernel (with the
                                                          // camera blocks folded in it is 6 % faster than the general one's <0, 4, 64, 5> instance)
  if (m3 && h->wgroups_worth) return KERN_MFMA3;
  if (h->groups_worth && vec && h->group_rounds >= 1 && h->group_rounds <= 2) return KERN_GROUPS;
  return KERN_PAIRS;
}
inline bool kern_is_mfma(int k) { return k == KERN_MFMA1 || k == KERN_MFMA2 || k == KERN_MFMA3; }

extern "C++" {
// k_schur_groups_mfma3 over the tile columns [TJ0, TJ1) of every group's window; one set of groups (chunks) with its parameters
struct M3Launch { Gm3Params G; const SchurChunk* chunks; int nchunks; bool uniform_ks; };
template <int TJ0, int TJ1, int LDC = 0, int KSC = 0>
int launch_mfma3(ba_handle* h, const M3Launch& L, int p, double damping, bool fuse_cam, bool first) {
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_groups_mfma3<TJ0, TJ1, LDC, KSC>));
  Gm3Params G = L.G;
  G.do_rhs = first ? 1 : 0;
  hipLaunchKernelGGL((k_schur_groups_mfma3<TJ0, TJ1, LDC, KSC>), dim3(L.nchunks), dim3(kGm2Block),
                     schur_mfma3_lds_bytes(G.Kbuf, G.Ld, G.wn, G.wb1), h->stream, dev_problem(h), h->cams[p].p, h->X[p].p,
                     h->wgroups.p, h->wtab.p, h->opt_cam.p, L.chunks, G, h->fac.p, h->S, h->b, damping, fuse_cam ? 1 : 0);
  return BA_OK;
}

int launch_mfma3_set(ba_handle* h, const M3Launch& L, int p, double damping, bool fuse_cam) {
  if (L.nchunks <= 0) return BA_OK;
  const int nts = L.G.nts;          // tiles per side of the widest window; at most 15 accumulator tiles per launch
  if (nts < 1 || nts > 15) return h->fail(BA_ERR_STATE, "k_schur_groups_mfma3: %d tiles per side", nts);
  // windows of at most 10 cameras with 6 points per batch everywhere (the north-star scenes): row length and k-steps fixed
  if (nts == 4 && L.G.np_cap == kGmPts && L.G.Kbuf == kGmK && L.uniform_ks) return launch_mfma3<0, 4, 64, 5>(h, L, p, damping, fuse_cam, true);
  int rc = nts == 5 ? launch_mfma3<0, 5>(h, L, p, damping, fuse_cam, true) : launch_mfma3<0, 4>(h, L, p, damping, fuse_cam, true);
  if (rc == BA_OK && nts >= 6) rc = launch_mfma3<4, 6>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts == 7) rc = launch_mfma3<6, 7>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 8) rc = launch_mfma3<6, 8>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 9) rc = launch_mfma3<8, 9>(h, L, p, damping, fuse_cam, false);
  // windows of 25 .. 40 cameras (tracks that long: video): one launch per further tile column (tj + 1 <= 15 tiles each)
  if (rc == BA_OK && nts >= 10) rc = launch_mfma3<9, 10>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 11) rc = launch_mfma3<10, 11>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 12) rc = launch_mfma3<11, 12>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 13) rc = launch_mfma3<12, 13>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 14) rc = launch_mfma3<13, 14>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 15) rc = launch_mfma3<14, 15>(h, L, p, damping, fuse_cam, false);
  return rc;
}
// k_schur_wide_mfma<NT> over the window groups of NT tiles per side (25 .. 40 cameras)
template <int NT>
int launch_wide(ba_handle* h, int p, double damping, bool fuse_cam) {
  const int g0 = h->wide_begin[NT - kGwMinTiles], n = h->wide_begin[NT - kGwMinTiles + 1] - g0;
  if (n <= 0) return BA_OK;
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_wide_mfma<NT>));
  hipLaunchKernelGGL(k_schur_wide_mfma<NT>, dim3(n), dim3(kGwBlock), schur_wide_lds_bytes(), h->stream, dev_problem(h), h->cams[p].p,
                     h->X[p].p, h->wgroups.p, h->wide_list.p + g0, h->wtab.p, h->opt_cam.p, h->fac.p, h->S, h->b, damping, fuse_cam ? 1 : 0);
  return BA_OK;
}
int launch_wide_all(ba_handle* h, int p, double damping, bool fuse_cam) {
  int rc = launch_wide<11>(h, p, damping, fuse_cam);
  if (rc == BA_OK) rc = launch_wide<12>(h, p, damping, fuse_cam);
  if (rc == BA_OK) rc = launch_wide<13>(h, p, damping, fuse_cam);
  if (rc == BA_OK) rc = launch_wide<14>(h, p, damping, fuse_cam);
  if (rc == BA_OK) rc = launch_wide<15>(h, p, damping, fuse_cam);
  return rc;
}
int wide_launches(const ba_handle* h) {
  int n = 0;
  for (int t = 0; t <= kGwMaxTiles - kGwMinTiles; ++t) n += h->wide_begin[t + 1] > h->wide_begin[t] ? 1 : 0;
  return n;
}

int launch_mfma3_all(ba_handle* h, int p, double damping, bool fuse_cam) {
  return launch_mfma3_set(h, M3Launch{h->gm3, h->m3chunks.p, h->nm3chunks, h->gm3_uniform_ks}, p, damping, fuse_cam);
}
inline int mfma3_launches(int nts) { return nts <= 5 ? 1 : nts == 6 ? 2 : nts <= 8 ? 3 : nts - 5; }
}  // extern "C++"

int launch_point_blocks(ba_handle* h, int p, double* Wd) {
  if (h->nt > 0) {
    ScopedTimer tm(h, BA_K_LINEARIZE);
    const long long threads = (long long)h->nt << h->glog;
    if (!Wd && h->point_groups && !h->opt.point_kernels_v1) {
      const int per_block = kBlock / kWave;
      hipLaunchKernelGGL(k_linearize_groups, dim3((h->ngroups + per_block - 1) / per_block), dim3(kBlock), 0, h->stream,
                         dev_problem(h), h->cams[p].p, h->X[p].p, h->groups.p, h->ngroups, h->HCC.p, h->bC.p, h->HPP.p, h->bP.p);
    } else {
      hipLaunchKernelGGL(k_linearize, dim3(blocks_for(threads)), dim3(kBlock), 0, h->stream, dev_problem(h),
                         h->cams[p].p, h->X[p].p, h->glog, h->HCC.p, h->bC.p, h->HPP.p, h->bP.p, Wd,
                         0.0, 0.0, (double*)nullptr, (int*)nullptr, (int*)nullptr);
    }
  }
  h->point_blocks_valid = true;
  h->cam_blocks_valid = false;          // k_linearize cleared HCC / bC
  return BA_OK;
}

int launch_camera_blocks(ba_handle* h, int p, bool clear) {
  if (clear) {
    HIPCHECK(h, hipMemsetAsync(h->HCC.p, 0, (size_t)h->nc * 36 * sizeof(double), h->stream));
    HIPCHECK(h, hipMemsetAsync(h->bC.p, 0, (size_t)h->nc * 6 * sizeof(double), h->stream));
  }
  if (h->ncam_units > 0) {
    ScopedTimer tm(h, BA_K_CAMERA_BLOCKS);
    const int per_block = kBlock / kWave;
    hipLaunchKernelGGL(k_camera_blocks, dim3((h->ncam_units + per_block - 1) / per_block), dim3(kBlock), 0, h->stream,
                       dev_problem(h), h->cams[p].p, h->X[p].p, h->cam_perm.p, h->cam_units.p, h->ncam_units, h->HCC.p,
                       h->bC.p);
  }
  h->cam_blocks_valid = true;
  return BA_OK;
}

// ba_linearize; with fuse (ba_lm_trial + MFMA reduction) the point inverses for (damping, rcond) are
// produced by the same kernel and the camera blocks are left to k_schur_groups_mfma
int linearize_impl(ba_handle* h, int which, int store_W, bool fuse, double damping, double rcond) {
  const int p = h->phys(which);
  HIPCHECK(h, hipSetDevice(h->device));
  double* Wd = nullptr;
  if (store_W) {
    HIPCHECK(h, h->W.resize(std::max<size_t>(1, (size_t)h->nobs * 18)));
    Wd = h->W.p;
  }
  fuse = fuse && h->nt > 0 && !store_W;
  h->inv_valid = false;
  h->fac_valid = false;
  h->cam_blocks_valid = false;
  h->point_blocks_valid = false;
  // fuse (ba_lm_trial with the MFMA reduction): the reduction kernel linearises every observation anyway and
  // adds the camera blocks on the way, so k_camera_blocks is skipped (ba_schur / ba_get_blocks run it lazily
  // if another path asks for HCC / bC).  The same kernel can form HPP, bP and HPPinv too (BA_FUSE_LIN=1:
  // then nothing is launched here at all), but that was measured 8 us per trial SLOWER: two more LDS round
  // trips and a 3x3 inversion per batch on a wavefront that has a SIMD to itself cost more than the two
  // kernels they replace.
  const bool fuse_lin = h->opt.fuse_lin;
  if (!(fuse && fuse_lin)) {
    int rc = launch_point_blocks(h, p, Wd);
    if (rc == BA_OK && !fuse) rc = launch_camera_blocks(h, p, h->nt == 0);      // k_linearize cleared HCC / bC otherwise
    if (rc != BA_OK) return rc;
  }
  HIPCHECK(h, hipGetLastError());
  h->have_linearization = true;
  h->lin_phys = p;
  h->have_schur = h->have_backsub = false;
  return BA_OK;
}

}  // namespace

int ba_problem_info(ba_handle* h, int64_t* out, int32_t n) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_problem_info: call ba_set_problem first");
  REQUIRE(h, out && n >= 1, BA_ERR_INVALID_ARG, "ba_problem_info: bad argument");
  const int kern = pick_schur_kernel(h);
  const int64_t v[BA_INFO_COUNT] = {
      h->pperm.empty() ? 0 : 1, h->operm.empty() ? 0 : 1, h->ngroups, (int64_t)(kern == KERN_MFMA3 ? h->nwgroups : h->nmgroups_total), h->point_groups ? 1 : 0,
      h->group_maxL, h->hb, kern_is_mfma(kern) ? 1 : 0, kern != KERN_PAIRS && kern != KERN_DENSE ? 1 : 0,
      kern == KERN_MFMA3 ? h->gm3.wn : h->schur_wn, h->nunits, kern, h->gm3.np_cap, h->gm3.Kbuf};
  for (int i = 0; i < n && i < BA_INFO_COUNT; ++i) out[i] = v[i];
  return BA_OK;
}

int ba_linearize(ba_handle* h, int which, int store_W) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_linearize: bad parameter set");
  REQUIRE(h, h->have_problem && h->have_params[h->phys(which)], BA_ERR_STATE, "ba_linearize: set problem and parameters first");
  return linearize_impl(h, which, store_W, false, 0.0, 0.0);
}

int ba_get_blocks(ba_handle* h, double* HCC, double* bC, double* HPP, double* bP, double* W) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_linearization, BA_ERR_STATE, "ba_get_blocks: call ba_linearize first");
  HIPCHECK(h, hipSetDevice(h->device));
  if ((HPP || bP) && !h->point_blocks_valid) {
    int rc = launch_point_blocks(h, h->lin_phys, nullptr);
    if (rc != BA_OK) return rc;
  }
  if ((HCC || bC) && !h->cam_blocks_valid) {       // ba_lm_trial left them to the reduction kernel
    int rc = launch_camera_blocks(h, h->lin_phys, true);
    if (rc != BA_OK) return rc;
  }
  std::vector<double> hpp6;
  if (HCC && h->nc) HIPCHECK(h, hipMemcpyAsync(HCC, h->HCC.p, (size_t)h->nc * 36 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (bC && h->nc) HIPCHECK(h, hipMemcpyAsync(bC, h->bC.p, (size_t)h->nc * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (HPP && h->nt) {
    hpp6.resize((size_t)h->nt * 6);
    HIPCHECK(h, hipMemcpyAsync(hpp6.data(), h->HPP.p, hpp6.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  if (bP) { const int rc = download_rows(h, h->pperm, h->bP.p, bP, (size_t)h->nt, 3); if (rc != BA_OK) return rc; }
  if (W && h->nobs) {
    REQUIRE(h, h->W.p && h->W.n >= (size_t)h->nobs * 18, BA_ERR_STATE, "ba_get_blocks: W was not stored (ba_linearize store_W=0)");
    const int rc = download_rows(h, h->operm, h->W.p, W, (size_t)h->nobs, 18);
    if (rc != BA_OK) return rc;
  }
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  if (HCC) {   // device keeps the upper triangle only
    for (int i = 0; i < h->nc; ++i)
      for (int a = 0; a < 6; ++a)
        for (int c = 0; c < a; ++c) HCC[(size_t)i * 36 + a * 6 + c] = HCC[(size_t)i * 36 + c * 6 + a];
  }
  if (HPP) {
    for (int k = 0; k < h->nt; ++k) {
      const double* s = &hpp6[(size_t)k * 6];
      double* d = HPP + (size_t)(h->pperm.empty() ? k : h->pperm[k]) * 9;
      d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[1]; d[4] = s[3]; d[5] = s[4]; d[6] = s[2]; d[7] = s[4]; d[8] = s[5];
    }
  }
  return BA_OK;
}

int ba_schur(ba_handle* h, int which, double damping, double pinv_rcond) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_schur: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_linearization && h->have_params[p], BA_ERR_STATE, "ba_schur: call ba_linearize first");
  HIPCHECK(h, hipSetDevice(h->device));
  int rc = ensure_reduced(h);
  if (rc != BA_OK) return rc;
  const int kern = pick_schur_kernel(h);         // (ba_set_option "schur" forces one: tests)
  const bool force_v1 = kern == KERN_MFMA1;      // the single-wavefront-per-group form
  const bool dense = kern == KERN_DENSE;
  const bool use_mfma = kern_is_mfma(kern);
  const bool use_groups = kern == KERN_GROUPS;
  // point blocks and camera blocks: normally in HPP / bP (k_linearize) and HCC / bC (k_camera_blocks);
  // ba_lm_trial leaves both to the MFMA reduction, which linearises every observation anyway
  const bool fuse_lin = (kern == KERN_MFMA1) && !h->point_blocks_valid;
  if (!h->point_blocks_valid && !fuse_lin) {
    rc = launch_point_blocks(h, h->lin_phys, nullptr);
    if (rc != BA_OK) return rc;
  }
  const bool hybrid = kern == KERN_MFMA3 && h->nrgroups > 0;       // long tracks: rectangular groups between their segments
  const bool fuse_cam = use_mfma && !h->cam_blocks_valid;
  if (!h->cam_blocks_valid && !fuse_cam) {
    rc = launch_camera_blocks(h, h->lin_phys, true);
    if (rc != BA_OK) return rc;
  }
  // the producer / consumer reductions and the dense one work from the factorised point inverses, which the merged
  // inversion + initialisation launch below writes (or has written, for the same damping)
  const bool want_fac = kern == KERN_MFMA2 || kern == KERN_MFMA3 || dense;
  const bool have_inv = fuse_lin || (h->inv_valid && h->inv_damping == damping && h->inv_rcond == pinv_rcond && (!want_fac || h->fac_valid));
  h->inv_valid = false;
  (void)force_v1;
  if (want_fac) HIPCHECK(h, h->fac.resize((size_t)9 * std::max(1, h->nt)));
  if (!have_inv) h->fac_valid = false;
  if (fuse_lin) h->sing_epoch ^= 1;   // the reduction kernel counts singular blocks like k_point_invert does
  const long long ninit = (long long)reduced_doubles(h) + (long long)h->nco * 6;
  if (!have_inv && h->nt > 0 && h->nco > 0) {
    // point inverses and the initialisation of [S | b] are independent: one launch for both
    h->sing_epoch ^= 1;     // this call counts singular blocks in sing_counter(); the kernel clears the other one
    ScopedTimer tm(h, BA_K_POINT_INVERT);
    const unsigned nbi = blocks_for(h->nt);
    hipLaunchKernelGGL(k_point_invert_schur_init, dim3(nbi + blocks_for(ninit)), dim3(kBlock), 0, h->stream, (int)nbi, h->nt,
                       h->HPP.p, damping, pinv_rcond, h->HPPinv.p, h->sing_counter(),
                       h->flags.p + 40 + ((h->sing_epoch ^ 1) & 1), h->nco, h->hb + 1, h->opt_cam.p, h->HCC.p, h->bC.p, h->S,
                       h->b, fuse_cam ? 0 : 1, h->bP.p, want_fac ? h->fac.p : (double*)nullptr);
    h->inv_valid = true; h->inv_damping = damping; h->inv_rcond = pinv_rcond;
    h->fac_valid = want_fac;
  } else {
    if (have_inv) {
      h->inv_valid = !fuse_lin;   // already inverted for this (damping, rcond) - or about to be, by the reduction kernel
    } else if (h->nt > 0) {
      h->sing_epoch ^= 1;
      ScopedTimer tm(h, BA_K_POINT_INVERT);
      hipLaunchKernelGGL(k_point_invert, dim3(blocks_for(h->nt)), dim3(kBlock), 0, h->stream, h->nt, h->HPP.p,
                         damping, pinv_rcond, h->HPPinv.p, h->sing_counter(), h->flags.p + 40 + ((h->sing_epoch ^ 1) & 1));
      h->inv_valid = true; h->inv_damping = damping; h->inv_rcond = pinv_rcond;
    } else {
      HIPCHECK(h, hipMemsetAsync(h->flags.p + 40, 0, 2 * sizeof(int), h->stream));
    }
    if (h->nco > 0) {
      ScopedTimer tm(h, BA_K_SCHUR_INIT);       // clears the band and writes the damped diagonal + b in one pass
      hipLaunchKernelGGL(k_schur_init, dim3(blocks_for(ninit)), dim3(kBlock), 0, h->stream, h->nco, h->hb + 1, h->opt_cam.p,
                         h->HCC.p, h->bC.p, damping, h->S, h->b, fuse_cam ? 0 : 1);
    }
  }
  if (dense) {
    // dense visibility: the reduction is one symmetric matrix product over all points (the kernels below
    // would do 36 global atomics per (pair, point): 258 M of them at 100 cameras x 1000 tracks)
    const int M = 6 * h->nco, R = 3 * h->nt;
    const int T = (M + kSyrkTile - 1) / kSyrkTile, pairs = T * (T + 1) / 2;
    int nsplit = std::max(1, std::min(16, (768 + pairs - 1) / pairs));
    const int chunk = ((R + nsplit - 1) / nsplit + kSyrkKc - 1) / kSyrkKc * kSyrkKc;
    nsplit = (R + chunk - 1) / chunk;
    HIPCHECK(h, h->dUd.resize((size_t)R * M)); HIPCHECK(h, h->dDd.resize(R)); HIPCHECK(h, h->dyd.resize(R));
    HIPCHECK(h, h->dpart.resize((size_t)nsplit * M * M));
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS, 5);
    HIPCHECK(h, hipMemsetAsync(h->dUd.p, 0, (size_t)R * M * sizeof(double), h->stream));
    const long long n = std::max<long long>(h->nobs, (long long)h->nt);
    hipLaunchKernelGGL(k_dense_stage, dim3(blocks_for(n)), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p, h->X[p].p,
                       h->fac.p, h->bP.p, M, h->dUd.p, h->dDd.p, h->dyd.p);
    hipLaunchKernelGGL(k_dense_syrk, dim3(T, T, nsplit), dim3(1024), 0, h->stream, M, R, chunk, h->dUd.p, h->dDd.p, h->dpart.p);
    hipLaunchKernelGGL(k_dense_apply, dim3(blocks_for(reduced_doubles(h))), dim3(kBlock), 0, h->stream, h->nco, h->hb + 1, M, nsplit,
                       h->dpart.p, h->S);
    hipLaunchKernelGGL(k_dense_rhs, dim3((M + kBlock - 1) / kBlock, (R + kDenseRhsRows - 1) / kDenseRhsRows), dim3(kBlock), 0,
                       h->stream, M, R, h->dUd.p, h->dyd.p, h->b);
  } else if (kern == KERN_MFMA3) {
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS, (h->nm3chunks ? mfma3_launches(h->gm3.nts) : 0) + (hybrid ? 1 : 0) + wide_launches(h));
    rc = launch_mfma3_all(h, p, damping, fuse_cam);
    if (rc != BA_OK) return rc;
    if (h->nwide > 0) {
      rc = launch_wide_all(h, p, damping, fuse_cam);
      if (rc != BA_OK) return rc;
    }
    if (hybrid) {
      HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_rect_mfma));
      hipLaunchKernelGGL(k_schur_rect_mfma, dim3(h->nrgroups), dim3(kRectBlock), schur_rect_lds_bytes(), h->stream, dev_problem(h), h->cams[p].p,
                         h->X[p].p, h->rgroups.p, h->nrgroups, h->rtab.p, h->opt_cam.p, h->fac.p, h->S, h->b, damping, fuse_cam ? 1 : 0);
    }
  } else if (kern == KERN_MFMA2) {
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS);
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_groups_mfma2));
    hipLaunchKernelGGL(k_schur_groups_mfma2, dim3(h->nmchunks), dim3(kGm2Block), schur_mfma2_lds_bytes(h->schur_wn, h->hb + 1), h->stream,
                       dev_problem(h), h->cams[p].p, h->X[p].p, h->mgroups.p, h->mchunks.p, h->schur_wn, h->fac.p, h->S, h->b, damping,
                       fuse_cam ? 1 : 0);
  } else if (use_mfma) {
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS);
    const int NW = kGmBlock / kWave;
    const size_t lds = (size_t)NW * 2 * kGmK * kGmLd * sizeof(double) + (size_t)NW * 16 * sizeof(int) + (size_t)NW * 64 * sizeof(double) +
                       (size_t)h->schur_wn * ((size_t)(h->hb + 1) * 36 + 6) * sizeof(double);
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_groups_mfma<false>));
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_groups_mfma<true>));
    if (fuse_lin)
      hipLaunchKernelGGL(k_schur_groups_mfma<true>, dim3(h->nmchunks), dim3(kGmBlock), lds, h->stream, dev_problem(h), h->cams[p].p,
                         h->X[p].p, h->mgroups.p, h->mchunks.p, h->schur_wn, h->HPPinv.p, h->bP.p, h->S, h->b, damping, fuse_cam ? 1 : 0,
                         h->HPP.p, pinv_rcond, h->sing_counter(), h->flags.p + 40 + ((h->sing_epoch ^ 1) & 1));
    else
      hipLaunchKernelGGL(k_schur_groups_mfma<false>, dim3(h->nmchunks), dim3(kGmBlock), lds, h->stream, dev_problem(h), h->cams[p].p,
                         h->X[p].p, h->mgroups.p, h->mchunks.p, h->schur_wn, h->HPPinv.p, h->bP.p, h->S, h->b, damping, fuse_cam ? 1 : 0,
                         h->HPP.p, pinv_rcond, h->sing_counter(), h->flags.p + 40 + ((h->sing_epoch ^ 1) & 1));
    if (fuse_lin) {
      h->point_blocks_valid = true;
      h->inv_valid = true; h->inv_damping = damping; h->inv_rcond = pinv_rcond;
      h->fac_valid = false;
    }
  } else if (use_groups) {
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS);
    const int NW = kGroupBlock / kWave;
    const size_t lds = (size_t)NW * 64 * 24 * sizeof(double) + (size_t)NW * 16 * sizeof(int) +
                       (size_t)h->schur_wn * ((size_t)(h->hb + 1) * 36 + 6) * sizeof(double);
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_groups<1>));
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_groups<2>));
    const int maxpairs_rounds = h->group_rounds >= 1 ? h->group_rounds : 2;
    if (maxpairs_rounds == 1)
      hipLaunchKernelGGL(k_schur_groups<1>, dim3(h->ngchunks), dim3(kGroupBlock), lds, h->stream, dev_problem(h), h->cams[p].p,
                         h->X[p].p, h->groups.p, h->gchunks.p, h->schur_wn, h->HPPinv.p, h->bP.p, h->S, h->b);
    else
      hipLaunchKernelGGL(k_schur_groups<2>, dim3(h->ngchunks), dim3(kGroupBlock), lds, h->stream, dev_problem(h), h->cams[p].p,
                         h->X[p].p, h->groups.p, h->gchunks.p, h->schur_wn, h->HPPinv.p, h->bP.p, h->S, h->b);
  } else if (h->nchunks > 0) {
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS);
    const int NW = kSchurBlock / kWave;
    const size_t lds = (size_t)NW * kTile * 18 * 2 * sizeof(double) + (size_t)NW * kTile * 2 * sizeof(int) +
                       (size_t)h->schur_wn * ((size_t)(h->hb + 1) * 36 + 6) * sizeof(double);
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_pairs));
    hipLaunchKernelGGL(k_schur_pairs, dim3(h->nchunks), dim3(kSchurBlock), lds, h->stream, dev_problem(h), h->cams[p].p,
                       h->X[p].p, h->units.p, h->chunks.p, h->schur_wn, h->HPPinv.p, h->bP.p, h->S, h->b);
  }
  HIPCHECK(h, hipGetLastError());
  h->have_schur = true;
  h->have_backsub = h->have_solution = false;
  if (pinv_rcond < 0.0 && !h->defer) {   // plain-inverse mode must report singular blocks (numpy.linalg.inv raises)
    int nsing = 0;
    HIPCHECK(h, hipMemcpyAsync(&nsing, h->sing_counter(), sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    if (nsing > 0) return h->fail(BA_ERR_SINGULAR, "ba_schur: %d singular 3x3 point block(s) in plain-inverse mode", nsing);
  }
  return BA_OK;
}

int ba_reduced_layout(ba_handle* h, int32_t* nco, int32_t* half_bandwidth, int64_t* S_doubles) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_reduced_layout: call ba_set_problem first");
  if (nco) *nco = h->nco;
  if (half_bandwidth) *half_bandwidth = h->hb;
  if (S_doubles) *S_doubles = (int64_t)reduced_doubles(h);
  return BA_OK;
}

int ba_get_reduced(ba_handle* h, double* S, double* b) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_schur, BA_ERR_STATE, "ba_get_reduced: call ba_schur first");
  HIPCHECK(h, hipSetDevice(h->device));
  const int nco = h->nco, hb1 = h->hb + 1;
  std::vector<double> band(S ? reduced_doubles(h) : 0);
  if (S && nco) HIPCHECK(h, hipMemcpyAsync(band.data(), h->S, band.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (b && nco) HIPCHECK(h, hipMemcpyAsync(b, h->b, (size_t)nco * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  if (S && nco) {   // expand the block band to the reference's dense (nco,nco,6,6), mirroring the upper triangle
    std::memset(S, 0, (size_t)nco * nco * 36 * sizeof(double));
    for (int i = 0; i < nco; ++i)
      for (int d = 0; d < hb1 && i + d < nco; ++d) {
        const double* src = &band[((size_t)i * hb1 + d) * 36];
        double* up = S + ((size_t)i * nco + (i + d)) * 36;
        std::memcpy(up, src, 36 * sizeof(double));
        if (d > 0) {
          double* lo = S + ((size_t)(i + d) * nco + i) * 36;
          for (int a = 0; a < 6; ++a)
            for (int c = 0; c < 6; ++c) lo[c * 6 + a] = src[a * 6 + c];
        }
      }
  }
  return BA_OK;
}

int ba_get_point_inverses(ba_handle* h, double* out) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_schur && out, BA_ERR_STATE, "ba_get_point_inverses: call ba_schur first");
  HIPCHECK(h, hipSetDevice(h->device));
  std::vector<double> s6((size_t)h->nt * 6);
  if (h->nt) HIPCHECK(h, hipMemcpyAsync(s6.data(), h->HPPinv.p, s6.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  for (int k = 0; k < h->nt; ++k) {
    const double* s = &s6[(size_t)k * 6];
    double* d = out + (size_t)(h->pperm.empty() ? k : h->pperm[k]) * 9;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[1]; d[4] = s[3]; d[5] = s[4]; d[6] = s[2]; d[7] = s[4]; d[8] = s[5];
  }
  return BA_OK;
}

int ba_reduced_device_ptrs(ba_handle* h, void** S_blocks, void** b) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_reduced_device_ptrs: call ba_set_problem first");
  HIPCHECK(h, hipSetDevice(h->device));
  int rc = ensure_reduced(h);
  if (rc != BA_OK) return rc;
  if (S_blocks) *S_blocks = h->S;
  if (b) *b = h->b;
  return BA_OK;
}

int ba_bind_reduced_buffers(ba_handle* h, void* S_blocks_dev, void* b_dev) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_bind_reduced_buffers: call ba_set_problem first");
  h->S = (double*)S_blocks_dev;
  h->b = (double*)b_dev;
  h->have_schur = false;
  return BA_OK;
}

int ba_flatten_reduced(ba_handle* h, const int32_t* keep, int32_t nkeep, void* A_dev, void* rhs_dev) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_schur, BA_ERR_STATE, "ba_flatten_reduced: call ba_schur first");
  REQUIRE(h, nkeep >= 0 && (nkeep == 0 || (keep && A_dev && rhs_dev)), BA_ERR_INVALID_ARG, "ba_flatten_reduced: NULL argument");
  for (int i = 0; i < nkeep; ++i)
    if (keep[i] < 0 || keep[i] >= h->nco * 6) return h->fail(BA_ERR_INVALID_ARG, "ba_flatten_reduced: keep[%d] out of range", i);
  if (nkeep == 0) return BA_OK;
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, h->keep.resize(nkeep));
  HIPCHECK(h, hipMemcpyAsync(h->keep.p, keep, nkeep * sizeof(int), hipMemcpyHostToDevice, h->stream));
  {
    ScopedTimer tm(h, BA_K_FLATTEN);
    hipLaunchKernelGGL(k_flatten, dim3(blocks_for((long long)nkeep * nkeep)), dim3(kBlock), 0, h->stream, h->nco, h->hb,
                       nkeep, h->keep.p, h->S, h->b, (double*)A_dev, (double*)rhs_dev);
  }
  HIPCHECK(h, hipGetLastError());
  HIPCHECK(h, hipStreamSynchronize(h->stream));   // `keep` is caller memory
  return BA_OK;
}

int ba_solve_reduced(ba_handle* h, const uint8_t* cam_param_mask, int32_t* info) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_schur, BA_ERR_STATE, "ba_solve_reduced: call ba_schur first");
  REQUIRE(h, info, BA_ERR_INVALID_ARG, "ba_solve_reduced: info is NULL");
  if (h->nco == 0) { *info = 0; h->have_solution = true; return BA_OK; }
  const int force = h->opt.solver;                 // ba_set_option "solver"
  const int nodes = h->hb > 0 ? (h->nco + h->hb - 1) / h->hb : 0;
  const bool bcr_ok = h->nco <= kBcrMaxHB || (h->hb >= 1 && h->hb <= kBcrMaxHB);      // (any number of nodes: even two levels beat k_band_solve's chain of nco pivots)
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
  const bool use_dense = !use_big && dense_ok && (force == SOLVER_DENSE || (!band_ok && !(bcrw_ok && (force == SOLVER_AUTO || force == SOLVER_BCR))));
  const bool use_bcr = !use_dense && (force ? ((force == SOLVER_BCR || force == SOLVER_BCR1) && bcr_ok) : bcr_ok);
  const bool use_bcrw = !use_dense && !use_bcr && (force ? (force == SOLVER_BCR && bcrw_ok) : bcrw_ok);
  if (force == SOLVER_LU || (!use_big && !use_dense && !use_bcr && !use_bcrw && !band_ok)) { *info = -1; return BA_OK; }     // caller's dense LU
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, h->Ufac.resize(std::max<size_t>(1, reduced_doubles(h))));
  const unsigned char* dmask = nullptr;
  if (cam_param_mask) {
    bool all = true;
    for (int i = 0; i < h->nco * 6; ++i) all = all && cam_param_mask[i];
    if (!all) {
      HIPCHECK(h, hipMemcpyAsync(h->mask.p, cam_param_mask, (size_t)h->nco * 6, hipMemcpyHostToDevice, h->stream));
      dmask = h->mask.p;
    }
  }
  // multi-CU paths: block cyclic reduction when the band is narrow enough for dense (6 hb)^2 blocks in LDS and there
  // are enough super-blocks to parallelise over (decided above)
  const size_t lds_budget = 160 * 1024;
  const int ch = band_solve_chunk(h->hb, lds_budget);
  size_t lds = 0;
  h->solve_kind = use_big ? BA_SOLVE_BCR_BIG : use_dense ? BA_SOLVE_DENSE_CHOLESKY : use_bcr ? BA_SOLVE_BCR : use_bcrw ? BA_SOLVE_BCR_WIDE : BA_SOLVE_BAND;
  if (use_big) {
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
    if (ch < 1) { *info = -1; return BA_OK; }
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
  if (inf > 0 && inf != kBcrTimedOut && use_bcr && h->opt.device_lu) {
    // not positive definite: the reference's LU would still solve it (bundle_adjuster.py:302-305) - cyclic reduction with LU nodes
    int rc = solve_bcr_lu(h, dmask);
    if (rc != BA_OK) return rc;
    int inf2 = 0;
    HIPCHECK(h, hipMemcpyAsync(&inf2, h->flags.p + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    h->solve_kind = BA_SOLVE_BCR_LU;
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
  return BA_OK;
}

int ba_backsubstitute(ba_handle* h, int which, const double* dC, double* dP) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_backsubstitute: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_schur && h->have_params[p], BA_ERR_STATE, "ba_backsubstitute: call ba_schur first");
  REQUIRE(h, dC || h->have_solution || h->nco == 0, BA_ERR_STATE,
          "ba_backsubstitute: dC is NULL and no device solution exists (ba_solve_reduced)");
  HIPCHECK(h, hipSetDevice(h->device));
  if (dC && h->nco) {
    HIPCHECK(h, hipMemcpyAsync(h->dC.p, dC, (size_t)h->nco * 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->have_solution = true;
  }
  // inside ba_lm_trial the update of the trial parameter set rides along (one launch less)
  const bool fuse_update = h->defer && which == BA_PARAMS_CUR && h->nt > 0;
  if (h->nt > 0) {
    ScopedTimer tm(h, BA_K_BACKSUB);
    const long long threads = (long long)h->nt << h->glog;
    h->cost_fused = false;
    if (h->point_groups && !h->opt.point_kernels_v1) {
      // inside ba_lm_trial the cost of the trial set rides along as well (k_cost's work)
      const int per_block = kBlock / kWave;
      const int nblk = std::min(kCostBlocks, (h->ngroups + per_block - 1) / per_block);
      const bool fuse_cost = fuse_update && h->opt.fuse_cost;
      hipLaunchKernelGGL(k_backsub_groups, dim3(nblk), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p, h->X[p].p,
                         h->groups.p, h->ngroups, h->dC.p, h->HPPinv.p, h->bP.p, h->dP.p, -1.0,
                         fuse_update ? h->cams[1 - p].p : (double*)nullptr, fuse_update ? h->X[1 - p].p : (double*)nullptr,
                         (const int*)h->sing_counter(), (const int*)(h->flags.p + 1),
                         fuse_cost ? h->host_result : (HostResult*)nullptr, fuse_cost ? h->trial_result_dev : (double*)nullptr);
      if (fuse_cost) { h->cost_fused = true; h->cost_blocks = nblk; }
    } else {
      hipLaunchKernelGGL(k_backsub, dim3(blocks_for(threads)), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p,
                         h->X[p].p, h->glog, h->dC.p, h->HPPinv.p, h->bP.p, h->dP.p, -1.0,
                         fuse_update ? h->cams[1 - p].p : (double*)nullptr, fuse_update ? h->X[1 - p].p : (double*)nullptr);
    }
  }
  HIPCHECK(h, hipGetLastError());
  if (dP) { const int rc = download_rows(h, h->pperm, h->dP.p, dP, (size_t)h->nt, 3); if (rc != BA_OK) return rc; }
  if (dC || dP) HIPCHECK(h, hipStreamSynchronize(h->stream));   // dC is caller memory
  h->have_backsub = true;
  return BA_OK;
}

int ba_apply_update(ba_handle* h, int src, int dst, const double* motion, const double* structure) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, (src == 0 || src == 1) && (dst == 0 || dst == 1), BA_ERR_INVALID_ARG, "ba_apply_update: bad parameter set");
  const int ps = h->phys(src), pd = h->phys(dst);
  REQUIRE(h, h->have_problem && h->have_params[ps], BA_ERR_STATE, "ba_apply_update: source parameter set is empty");
  REQUIRE(h, (motion == nullptr) == (structure == nullptr), BA_ERR_INVALID_ARG,
          "ba_apply_update: give both motion and structure, or neither");
  HIPCHECK(h, hipSetDevice(h->device));
  double sign = -1.0;
  if (motion) {
    sign = 1.0;
    if (h->nco) HIPCHECK(h, hipMemcpyAsync(h->dC.p, motion, (size_t)h->nco * 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->nt && !h->pperm.empty()) {
      std::vector<double> si((size_t)h->nt * 3);
      rows_to_internal(h->pperm, structure, si.data(), 3);
      HIPCHECK(h, hipMemcpyAsync(h->dP.p, si.data(), si.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
      HIPCHECK(h, hipStreamSynchronize(h->stream));         // `si` goes out of scope
    } else if (h->nt) {
      HIPCHECK(h, hipMemcpyAsync(h->dP.p, structure, (size_t)h->nt * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    h->have_backsub = h->have_solution = false;   // dC / dP now hold the caller's update
  } else {
    REQUIRE(h, h->have_backsub, BA_ERR_STATE, "ba_apply_update: no update on the device (call ba_backsubstitute)");
  }
  if (h->nc + h->nt > 0) {
    ScopedTimer tm(h, BA_K_UPDATE);
    hipLaunchKernelGGL(k_apply_update, dim3(blocks_for((long long)h->nc + h->nt)), dim3(kBlock), 0, h->stream, h->nc,
                       h->nt, h->cam_opt_pos.p, h->pt_opt.p, h->cams[ps].p, h->X[ps].p, h->dC.p, h->dP.p, sign,
                       h->cams[pd].p, h->X[pd].p);
  }
  HIPCHECK(h, hipGetLastError());
  if (motion) HIPCHECK(h, hipStreamSynchronize(h->stream));
  h->have_params[pd] = true;
  if (dst == BA_PARAMS_CUR) h->have_linearization = h->have_schur = false;
  return BA_OK;
}

int ba_set_dense_visibility(ba_handle* h, int32_t on) {
  if (!h) return BA_ERR_INVALID_ARG;
  h->dense_mode = on != 0;
  h->inv_valid = false;
  if (!on) { h->dUd.release(); h->dDd.release(); h->dyd.release(); h->dpart.release(); }
  return BA_OK;
}

// ---- the shards' collectives inside the library (RCCL over xGMI on the handle's own stream)
int ba_comm_load(const char* librccl_path) {
  if (g_rccl.ok()) return BA_OK;
  void* lib = dlopen(librccl_path && *librccl_path ? librccl_path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) { g_create_error = std::string("ba_comm_load: ") + dlerror(); return BA_ERR_HIP; }
  g_rccl.lib = lib;
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(lib, "ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!g_rccl.ok()) { g_create_error = "ba_comm_load: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce"; return BA_ERR_HIP; }
  return BA_OK;
}

int ba_comm_unique_id(void* id128) {
  if (!id128 || !g_rccl.ok()) return BA_ERR_STATE;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return BA_ERR_HIP;
  std::memcpy(id128, &id, sizeof id);
  return BA_OK;
}

int ba_comm_init(ba_handle* h, const void* id128, int32_t rank, int32_t nranks) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, g_rccl.ok(), BA_ERR_STATE, "ba_comm_init: call ba_comm_load first");
  REQUIRE(h, id128 && nranks >= 1 && rank >= 0 && rank < nranks, BA_ERR_INVALID_ARG, "ba_comm_init: bad argument");
  REQUIRE(h, !h->comm, BA_ERR_STATE, "ba_comm_init: communicator already attached");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  HIPCHECK(h, hipSetDevice(h->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  RCCLCHECK(h, g_rccl.CommInitRank(&h->comm, nranks, id, rank));
  h->comm_ranks = nranks;
  HIPCHECK(h, h->comm_dev.resize(kCostBlocks + 2));
  HIPCHECK(h, hipMemsetAsync(h->comm_dev.p, 0, (kCostBlocks + 2) * sizeof(double), h->stream));
  HIPCHECK(h, hipHostMalloc((void**)&h->comm_host, (kCostBlocks + 2) * sizeof(double), hipHostMallocDefault));
  h->trial_result_dev = h->comm_dev.p;               // k_cost / k_backsub_groups leave the trial record here
  return BA_OK;
}

int ba_comm_destroy(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  if (h->comm) {
    (void)hipStreamSynchronize(h->stream);
    (void)g_rccl.CommDestroy(h->comm);
    h->comm = nullptr; h->comm_ranks = 0;
    if (h->trial_result_dev == h->comm_dev.p) h->trial_result_dev = nullptr;
    if (h->comm_host) { (void)hipHostFree(h->comm_host); h->comm_host = nullptr; }
    h->comm_dev.release();
  }
  return BA_OK;
}

int ba_comm_allreduce_reduced(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->comm, BA_ERR_STATE, "ba_comm_allreduce_reduced: no communicator (ba_comm_init)");
  REQUIRE(h, h->have_schur, BA_ERR_STATE, "ba_comm_allreduce_reduced: call ba_schur first");
  HIPCHECK(h, hipSetDevice(h->device));
  return comm_allreduce_reduced(h);
}

int ba_comm_allreduce_sum(ba_handle* h, double* values, int32_t n) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->comm, BA_ERR_STATE, "ba_comm_allreduce_sum: no communicator (ba_comm_init)");
  REQUIRE(h, values && n >= 1 && n <= kCostBlocks, BA_ERR_INVALID_ARG, "ba_comm_allreduce_sum: bad argument");
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, h->scratch.resize((size_t)n));
  HIPCHECK(h, hipMemcpyAsync(h->scratch.p, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RCCLCHECK(h, g_rccl.AllReduce(h->scratch.p, h->scratch.p, (size_t)n, ncclFloat64, ncclSum, h->comm, h->stream));
  HIPCHECK(h, hipMemcpyAsync(values, h->scratch.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  return BA_OK;
}

int ba_set_min_half_bandwidth(ba_handle* h, int32_t min_hb) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, min_hb >= 0, BA_ERR_INVALID_ARG, "ba_set_min_half_bandwidth: negative");
  h->min_hb = min_hb;
  return BA_OK;
}

int ba_bind_trial_result(ba_handle* h, void* result_dev) {
  if (!h) return BA_ERR_INVALID_ARG;
  static_assert(BA_TRIAL_PARTIALS == kCostBlocks, "header and kernel disagree on the number of cost partials");
  h->trial_result_dev = static_cast<double*>(result_dev);
  return BA_OK;
}

int ba_lm_trial_begin(ba_handle* h, double damping, double pinv_rcond) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem && h->have_params[h->phys(BA_PARAMS_CUR)], BA_ERR_STATE, "ba_lm_trial: set problem and parameters first");
  h->defer = true;
  // with the MFMA reduction the camera blocks come out of the reduction itself: one launch and one pass
  // over the observations less
  const bool fuse = h->opt.schur == SCHUR_AUTO && h->opt.fuse_cam && kern_is_mfma(pick_schur_kernel(h));
  int rc = linearize_impl(h, BA_PARAMS_CUR, 0, fuse, damping, pinv_rcond);
  if (rc == BA_OK) rc = ba_schur(h, BA_PARAMS_CUR, damping, pinv_rcond);
  h->defer = false;
  h->trial_rcond = pinv_rcond;
  return rc;
}

// the tail of a trial once the solution is on the device: back-substitution, trial parameter set, trial cost - nothing read back
int ba_lm_trial_finish(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_schur && h->have_solution, BA_ERR_STATE, "ba_lm_trial_finish: no solution on the device");
  h->defer = true;
  double unused = 0.0;
  int rc = ba_backsubstitute(h, BA_PARAMS_CUR, nullptr, nullptr);
  if (rc == BA_OK) {
    if (h->nt > 0) h->have_params[h->phys(BA_PARAMS_TRIAL)] = true;      // k_backsub wrote the trial set
    else rc = ba_apply_update(h, BA_PARAMS_CUR, BA_PARAMS_TRIAL, nullptr, nullptr);
  }
  if (rc == BA_OK && !(h->cost_fused && h->nt > 0)) rc = ba_cost(h, BA_PARAMS_TRIAL, &unused);
  h->defer = false;
  return rc;
}

int ba_lm_trial_end(ba_handle* h, const uint8_t* cam_param_mask, int32_t* pre_info) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, pre_info, BA_ERR_INVALID_ARG, "ba_lm_trial_end: NULL output");
  REQUIRE(h, h->have_schur, BA_ERR_STATE, "ba_lm_trial_end: call ba_lm_trial_begin first");
  *pre_info = 0;
  h->defer = true;
  int rc = ba_solve_reduced(h, cam_param_mask, pre_info);
  h->defer = false;
  if (rc == BA_OK && *pre_info != 0) return BA_OK;   // band too wide: caller takes the dense path
  if (rc == BA_OK) rc = ba_lm_trial_finish(h);
  return rc;
}

// ---- the reduced solve spread over the ranks (ba_dist.h)
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
  if (nranks < 2 || h->dense_mode) return BA_OK;
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

int ba_lm_trial(ba_handle* h, double damping, double pinv_rcond, const uint8_t* cam_param_mask, double* next_cost,
                int32_t* info) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, next_cost && info, BA_ERR_INVALID_ARG, "ba_lm_trial: NULL output");
  *info = 0;
  // the dense-visibility reduction is driven by the caller (its matrix product is a library call), and systems
  // too large for the dense device solve go to the caller's LU: do not linearise and reduce just to find that out
  if (h->have_problem && h->hb > kBcrwMaxHB && 6 * h->nco > kDcMaxN) { *info = -1; return BA_OK; }
  int32_t pre = 0;
  int rc = ba_lm_trial_begin(h, damping, pinv_rcond);
  const bool dist = h->comm && h->dist.on && h->hb <= kBcrMaxHB;
  if (dist) {
    // the solve spread over the ranks (ba_dist.h): three small sums instead of one of the whole band
    for (int stage = 1; stage <= 4 && rc == BA_OK; ++stage) {
      size_t count = 0;
      rc = dist_stage(h, stage, cam_param_mask, &count);
      if (rc == BA_OK && count) RCCLCHECK(h, g_rccl.AllReduce(h->dist.xbuf, h->dist.xbuf, count, ncclFloat64, ncclSum, h->comm, h->stream));
    }
    if (rc == BA_OK) rc = ba_lm_trial_finish(h);
  } else {
    if (rc == BA_OK && h->comm) rc = comm_allreduce_reduced(h);      // sharded: the one data-path collective
    if (rc == BA_OK) rc = ba_lm_trial_end(h, cam_param_mask, &pre);
  }
  if (rc != BA_OK) return rc;
  if (pre != 0) { *info = pre; return BA_OK; }
  int st[2];
  if (h->comm) {
    // the shards' trial records (cost partials | singular blocks | solver status) are summed in place - 16 KB, the
    // latency of 8 bytes - and come back with one copy; the partials are added on the host in index order
    RCCLCHECK(h, g_rccl.AllReduce(h->comm_dev.p, h->comm_dev.p, (size_t)kCostBlocks + 2, ncclFloat64, ncclSum, h->comm, h->stream));
    // (a kernel storing into the pinned record: a 16 KB hipMemcpyAsync goes through the DMA engine and costs more)
    hipLaunchKernelGGL(k_copy_doubles, dim3((kCostBlocks + 2 + 255) / 256), dim3(256), 0, h->stream, h->comm_dev.p, h->comm_host,
                       kCostBlocks + 2);
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    double sum = 0.0;
    for (int i = 0; i < kCostBlocks; ++i) sum += h->comm_host[i];
    *next_cost = sum;
    st[0] = (int)std::llround(h->comm_host[kCostBlocks]);                          // over all shards
    st[1] = trial_status_of_sum(h->comm_host[kCostBlocks + 1], h->comm_ranks, dist);      // (a time-out on any rank stays a time-out)
  } else {
    HIPCHECK(h, hipStreamSynchronize(h->stream));    // k_cost left the cost partials + status words in pinned memory
    st[0] = h->host_result->singular_points; st[1] = h->host_result->solve_info;
    *next_cost = h->host_cost();
  }
  if (pinv_rcond < 0.0 && st[0] > 0)
    return h->fail(BA_ERR_SINGULAR, "ba_lm_trial: %d singular 3x3 point block(s) in plain-inverse mode", st[0]);
  *info = st[1];
  if (st[1] != 0) h->have_solution = h->have_backsub = false;
  return BA_OK;
}

int ba_triangulate(ba_handle* h, int which, double rcond, double* X) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_triangulate: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_problem && h->have_params[p], BA_ERR_STATE, "ba_triangulate: set problem and parameters first");
  HIPCHECK(h, hipSetDevice(h->device));
  if (rcond < 0) rcond = 2.220446049250313e-16 * std::max<double>(3.0, 2.0 * 64);   // numpy's default scale
  // QR of the 2L x 3 system (k_triangulate): full-rank systems are solved to cond(A) * eps like lstsq's; the rank decision
  // (|R_jj| <= rcond * max |R_ii|, never below 1e-13) sends what is rank deficient to working precision to the minimum-norm answer
  if (h->nt > 0) {
    ScopedTimer tm(h, BA_K_TRIANGULATE);
    const long long threads = (long long)h->nt << h->glog;
    hipLaunchKernelGGL(k_triangulate, dim3(blocks_for(threads)), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p,
                       h->glog, std::max(rcond, 1e-13), h->X[p].p);
  }
  HIPCHECK(h, hipGetLastError());
  if (which == BA_PARAMS_CUR) h->have_linearization = h->have_schur = h->have_backsub = h->have_solution = false;
  if (X && h->nt) {
    const int rc = download_rows(h, h->pperm, h->X[p].p, X, (size_t)h->nt, 3);
    if (rc != BA_OK) return rc;
    HIPCHECK(h, hipStreamSynchronize(h->stream));
  }
  return BA_OK;
}

int ba_enable_timing(ba_handle* h, int on) {
  if (!h) return BA_ERR_INVALID_ARG;
  if (!on) resolve_timings(h);
  h->timing = on != 0;
  return BA_OK;
}

int ba_set_timing_mask(ba_handle* h, uint64_t kernel_id_mask) {
  if (!h) return BA_ERR_INVALID_ARG;
  h->timing_mask = kernel_id_mask;
  return BA_OK;
}

int ba_set_timing_stride(ba_handle* h, int32_t stride) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, stride >= 1, BA_ERR_INVALID_ARG, "ba_set_timing_stride: stride must be >= 1");
  h->timing_stride = stride;
  for (auto& c : h->timing_seen) c = 0;
  return BA_OK;
}

int ba_measure_copy_bandwidth(ba_handle* h, int64_t bytes, int32_t repeats, double* gbytes_per_s) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, bytes >= 4096 && repeats >= 1 && gbytes_per_s, BA_ERR_INVALID_ARG, "ba_measure_copy_bandwidth: bad argument");
  HIPCHECK(h, hipSetDevice(h->device));
  const size_t n = (size_t)bytes / sizeof(copy_vec);
  REQUIRE(h, (n + 255) / 256 < (1ull << 31), BA_ERR_INVALID_ARG, "ba_measure_copy_bandwidth: too large");
  DevBuf<copy_vec> src, dst;
  HIPCHECK(h, src.resize(n)); HIPCHECK(h, dst.resize(n));
  HIPCHECK(h, hipMemsetAsync(src.p, 0, n * sizeof(copy_vec), h->stream));
  hipEvent_t a, b;
  HIPCHECK(h, hipEventCreate(&a)); HIPCHECK(h, hipEventCreate(&b));
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, h->stream, src.p, dst.p, n);     // warm-up
  HIPCHECK(h, hipEventRecord(a, h->stream));
  for (int r = 0; r < repeats; ++r) hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, h->stream, src.p, dst.p, n);
  HIPCHECK(h, hipEventRecord(b, h->stream));
  HIPCHECK(h, hipEventSynchronize(b));
  float ms = 0.f;
  HIPCHECK(h, hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  src.release(); dst.release();
  *gbytes_per_s = 2.0 * (double)(n * sizeof(copy_vec)) * repeats / (ms * 1e-3) / 1e9;           // read + write
  return BA_OK;
}

int ba_get_timings(ba_handle* h, double* ms, int64_t* launches, int reset) {
  if (!h) return BA_ERR_INVALID_ARG;
  HIPCHECK(h, hipSetDevice(h->device));
  resolve_timings(h);
  for (int i = 0; i < BA_K_COUNT; ++i) {
    if (ms) ms[i] = h->ms[i];
    if (launches) launches[i] = h->launches[i];
    if (reset) { h->ms[i] = 0; h->launches[i] = 0; }
  }
  return BA_OK;
}

}  // extern "C"
