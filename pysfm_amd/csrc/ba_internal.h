// ba_internal.h - what the translation units of libpysfm_ba.so share on the HOST side: the opaque handle behind the C ABI of
// include/pysfm_ba.h, its device buffers, the option table, error / timing helpers, and the few functions one unit offers
// the others.  Kernels live in the *_kernels.h / ba_bcr*.h / ba_dense.h / ba_band.h / ba_dist.h headers, each of which is
// included by exactly one unit (the Makefile builds the units in parallel).
#pragma once

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

#include "ba_types.h"

#include <rccl/rccl.h>     // types only: the entry points are resolved at run time from the librccl torch has loaded

namespace ba {

extern thread_local std::string g_create_error;

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

struct PcgStateRaw { int last_iter, done_iter, breakdown, pad; double rr, bb; };      // = PcgState of ba_pcg.h

struct TimedLaunch { int id; hipEvent_t a, b; int count; };

// Test / measurement switches (ba_set_option).  The defaults are the product path; nothing in the library
// reads the environment.
enum { SCHUR_AUTO = 0, SCHUR_PAIRS, SCHUR_GROUPS, SCHUR_MFMA2, SCHUR_MFMA };
enum { SOLVER_AUTO = 0, SOLVER_BCR, SOLVER_BAND, SOLVER_DENSE, SOLVER_LU, SOLVER_BCR1, SOLVER_PCG };
struct Options {
  int schur = SCHUR_AUTO;
  int solver = SOLVER_AUTO;
  bool point_kernels_v1 = false;   // lanes-per-point k_linearize / k_backsub instead of the group-packed kernels
  bool fuse_cost = true;           // trial cost inside k_backsub_groups
  bool fuse_invert = true;         // ba_lm_trial: point inverses and the clearing of [S | b] inside k_linearize_groups (no k_point_invert_schur_init launch)
  bool fuse_cam = true;            // camera blocks inside the MFMA reduction
  bool sort_points = true;         // internal point order (ba_set_problem); off = the caller's order as given
  int gm_cap = 0;                  // points per MFMA group (0 = chosen by ba_set_problem)
  int gm_chunk = 0;                // groups per workgroup of the MFMA reductions (0 = chosen by ba_set_problem)
  bool lds_window = true;          // k_schur_groups_mfma3 accumulates in an LDS window of the band when one fits
  bool fast_paths = true;          // K = I / unit-Gaussian short cuts of the per-observation arithmetic (ba_math.h)
  bool fused_backsolve = true;     // all back-substitution levels of the cyclic reduction in one launch when the nodes fit the chip
  bool bcrw_merged = true;         // wide cyclic reduction: factorisation and substitution of a level in one kernel (k_bcrw_factor_solve)
  bool six_tile_launch = true;     // k_schur_groups_mfma3: tile columns 0 .. 5 of the window in the first launch (21 accumulator tiles)
  bool dense_lookahead = true;     // dense / big-node factorisation: the next block column's panel step beside this one's trailing update (k_dense_step)
  bool fused_eliminate = true;     // all split elimination levels of the cyclic reduction in one launch (k_bcr_eliminate_fused)
  bool device_lu = true;           // a reduced system the Cholesky solve reports as not positive definite is solved again by the cyclic reduction with LU nodes
  bool packed_upload = true;       // ... and their arrays travel as one copy + one scatter launch (off: a copy per array)
  bool host_setup = true;          // ba_set_problem: small problems are ordered on the host (off: always the device pipeline)
  bool resident = true;            // ba_lm_resident applies to problems that fit one compute unit (off: ba_lm_resident_fits says no)
  int resident_fault = -1;         // test aid: the workgroup of the resident loop that reports a time-out whatever happened (-1: none)
  int resident_scatter_min = 5;    // ... whose launches of at least this many workgroups add their partial sums up in slices (two stages)
  bool solve_trace = false;        // per-phase cycle counts of the node kernels (PROFILE builds)
  bool reuse_linearization = true; // ba_lm_trial after a rejected trial: the point blocks of the unchanged current set are not formed again
  bool border_side_stream = true;  // the border's blocks and the preparation of its solve on a side stream beside the cyclic reduction (off: in line)
  bool border = true;              // ... and a border for the cameras at the far end of a few long-range tracks (ba_border.h)
  bool sparse_stage = false;       // sparse path's reduction in two steps: every observation linearised once, T and W left behind (288 bytes each) for the blocks to sum - measured SLOWER than every pair linearising its two observations itself (260 against 166 us at 5000 cameras): off, kept for the record
  bool packed_store = true;        // scenes the sparse path takes whole keep [S] as the list of the pattern's blocks, no band (0: the band of the camera order, as for every other scene)
  double pcg_tol = 1e-12;          // conjugate gradients (ba_pcg.h): converged at ||r|| <= pcg_tol ||b||
  int pcg_max_iter = 0;            // ... iteration budget (0: max(1000, min(20000, 4 nco)))
  int pcg_batch = 50;              // ... iterations enqueued between two looks at the state
  int refine_debug = 0;            // experiment: 1 = the refinement's items do not wait for each other (wrong numbers, the kernel's floor time)
  int refine = 0;                  // one step of iterative refinement behind the cyclic reduction (ba_bcr_refine.h): 0 auto (damping below kRefineBelowDamping), 1 always, 2 never
  int camera_order = 0;            // internal order of the optimised cameras (ba_order.hip): 0 auto (when the caller's is not provably as narrow as it can be), 1 off, 2 always try
};
enum { CAMORDER_AUTO = 0, CAMORDER_OFF, CAMORDER_ALWAYS };
enum { REFINE_AUTO = 0, REFINE_ON, REFINE_OFF };
constexpr int kPcgStalled = 0x7f000002;            // status word of conjugate gradients that ran out of iterations (BA_SOLVE_STALLED)
constexpr int kPcgMinCams = 1500;                  // auto: from here on (dense Cholesky: 25 ms and growing with the cube) ...
constexpr double kPcgMaxFill = 0.10;               // ... and when at most this fraction of the band's blocks can be non-zero
constexpr double kRefineBelowDamping = 1e-2;      // where the LM walk starts to feel the last digits of the reduced solve
constexpr int kBordMaxCamsHost = 21;       // (= kBordMaxCams of ba_border.h: 126 border unknowns fit the LDS of the border solve)

}  // namespace ba

using namespace ba;

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
  long long lin_reused = 0;          // trials that found the linearisation of the current set still valid (ba_lm_trial_begin)
  bool point_blocks_valid = false;   // HPP / bP hold the point blocks of the linearisation (ba_lm_trial leaves them to the reduction too)
  bool cam_blocks_valid = false;     // HCC / bC hold the camera blocks of the linearisation (ba_lm_trial may leave them to the reduction)
  bool inv_valid = false;            // HPPinv holds pinv of the damped point blocks for (inv_damping, inv_rcond)
  double inv_damping = 0.0, inv_rcond = 0.0;
  double schur_damping = 0.0;        // the damping of the reduced system [S | b] holds (the last ba_schur)
  long long refined = 0;             // solves that took the refinement step
  bool dense_mode = false;           // ba_set_dense_visibility: the reduction is one SYRK over all points (k_dense_*)
  double* trial_result_dev = nullptr; // bound by ba_bind_trial_result: device copy of the cost partials + status words
  ncclComm_t comm = nullptr;          // ba_comm_init: the shards' communicator; collectives run on `stream`
  int comm_ranks = 0;
  double* comm_host = nullptr;        // pinned [kCostBlocks + 2]: the all-reduced trial record
  double trial_rcond = 0.0;          // ba_lm_trial_begin -> ba_lm_trial_end
  int glog = 0;              // lanes per point = 2^glog
  double K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  Sensor sensor{SENSOR_GAUSS, {1, 0, 0, 1}, 1.0, 1.0, FAST_UNIT_GAUSS};
  DevBuf<double> sensor_table;       // BA_SENSOR_TABLE: the sampled robustifier
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
  bool groups_worth = false;            // points really share camera lists (mean run >= 9 points): the run-by-run lineariser / back-substitution pay
  bool mgroups_any = false;             // ... at least two points a run: the run-by-run reduction where there are no window groups
  bool mgroups_worth = false;           // ... nearly all of them in full runs (mean >= 20 of 24): the run-by-run matrix-core reduction pays
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
  bool cam_units_built = false, pair_units_built = false;      // work lists of kernels off the trial's path: built on first use
  // internal order (ba_set_problem): internal point i = the caller's track pperm[i], internal observation n = the caller's
  // operm[n].  On the device always; the host copy of pperm only when it is not the identity (empty = identity).
  std::vector<int> pperm;
  DevBuf<int> d_pperm, d_operm;
  bool operm_identity = true;
  // per internal point, kept on the host for the work lists that are built lazily: CSR offsets, lowest / highest optimised position
  std::vector<int> h_off, h_plo, h_phi;
  std::vector<unsigned char> h_same;   // point i has the camera list of point i - 1
  std::vector<int> h_cam_opt_pos;      // cam_opt_pos in the INTERNAL camera order (what the device holds)
  // internal order of the optimised cameras (ba_order.hip): the caller's position p sits at internal position cpos_in[p], internal
  // position q is the caller's cpos_out[q]; both empty = the caller's order.  Everything the C ABI takes or returns by optimised
  // position goes through cam_rows_in / cam_rows_out.
  std::vector<int> cpos_in, cpos_out;
  std::vector<int> forced_pos;         // ba_set_camera_layout: the layout the caller imposes (a sharded adjuster's ranks share one)
  int caller_hb = 0;                   // half-bandwidth the caller's camera order would have had
  // band + border (ba_border.h): the last nbc optimised positions are BORDER cameras - the band's kernels see them as cameras that
  // are not optimised (cam_band_pos), their blocks live in bord (C, D) and their solution comes out of the border solve
  int nbc = 0;
  DevBuf<int> cam_band_pos;            // [nc]: optimised position of a band camera, -1 for border and frozen cameras (= cam_opt_pos without a border)
  DevBuf<int> bord_obs;                // [BorderBlock x nbord_obs | the blocks' pairs of observations] (ba_border.h)
  int nbord_obs = 0, bord_ld = 0;      // ... the number of blocks; row length of C, D (6 nbc rounded up to 16)
  size_t bord_rows = 0;                // rows of C / F: the nodes of the cyclic reduction (>= 6 band cameras)
  int bord_nchunks = 0, bord_nrcams = 0;      // chunks of the blocks' pairs (one wavefront each); band cameras that have a block in C
  size_t bord_off_chunks = 0, bord_off_first = 0, bord_off_rcams = 0, bord_off_pairs = 0;      // where they sit in bord_obs (ints)
  DevBuf<double> bord_partial;         // the chunks' partial sums
  // The border's blocks need the point inverses and nothing of the band: they are formed on a SIDE stream beside the band's
  // reduction and the cyclic reduction (which leaves two thirds of the compute units idle) and join the main stream before the
  // first kernel that reads them (border_join).
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool bord_pending = false;           // work on the side stream that the main stream has not waited for yet
  DevBuf<double> bordV;                // a column of C on its way through the band's LU solver (border_solve_lu)
  DevBuf<double> bordC, bordF, bordD;  // C (as the reduction leaves it), F (work: C -> Y), [D | M | rv | x2]
  int band_cams() const { return nco - nbc; }
  std::vector<unsigned char> mask_host; // the mask of the last solve in the internal order (outlives its asynchronous upload)
  std::vector<double> rows_host;       // ... and the last per-camera rows that went up
  std::vector<int> plan_flags;         // the set-up's status record (SF_* of ba_setup_kernels.h)
  bool plan_pending = false;           // the work lists of the general kernels are not built yet (ensure_plan)
  // ba_set_problem's own device buffers (kept between calls: the sliding-window caller sets a problem per window)
  struct Setup {
    DevBuf<int> rc, rp, by_pt, cnt, coff, Lint, plo, phi, iota, crank, flags, vals;
    DevBuf<unsigned long long> key, key2, tkey, tkey2;
    DevBuf<double2> rz;
    DevBuf<unsigned char> rpo, same, tmp;
    DevBuf<unsigned> blob;             // device mirror of a packed upload (k_setup_scatter)
    void* host = nullptr;              // pinned staging for the read-backs
    size_t host_bytes = 0;
    void* up = nullptr;                // pinned arena for the small uploads
    size_t up_bytes = 0, up_used = 0;
    bool up_pending = false;           // uploads from the arena may still be in flight
    bool up_pageable = false;          // something did not fit the arena: the caller's memory is read until the stream is idle
  } su;

  // parameters: cams[which] = nc x [R(9) | t(3)], X[which] = nt x 3
  DevBuf<double> cams[2], X[2];
  int cur = 0;               // physical index of BA_PARAMS_CUR

  // normal-equation blocks
  DevBuf<double> HCC, bC, HPP, bP, HPPinv, W, S_own, b_own, dC, dP, scratch, scratch2, Ufac, ysol, dinv, bcrD, bcrU, bcrF, bcrP, bcrQ, bcrG, bcrGv, bcrL, bcrLv, denseA, bigK, fac, dUd, dDd, dyd, dpart, comm_dev;
  DevBuf<unsigned char> mask;
  DevBuf<double> bcrRm;      // refinement step (ba_bcr_refine.h): [g | d | contribution slots] of every node - what its items wait on, marked by k_bcr_assemble
  DevBuf<int> bcr_rwork;     // ... its work list: 2 node + sweep, forward items leaves first, then backward items root down (for bcr_rwork_n nodes)
  int bcr_rwork_n = 0;
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
  // conjugate gradients over the blocks of S the tracks define (ba_pcg.h / ba_pcg.hip)
  struct PcgPlan {
    bool built = false;
    long long nnz = 0, upper = 0;       // blocks of the full symmetric pattern; of its upper triangle (diagonal included)
    DevBuf<int> rowptr, col;
    DevBuf<long long> ublk;             // ... of the upper triangle's blocks alone (what the reductions write)
    DevBuf<int> bptr;                   // k_schur_blocks: the upper blocks' lists of observation pairs, [upper + 1] offsets into ...
    DevBuf<int2> pairs;                 // ... (observation of the camera at the lower position, of the camera at the higher position)
    DevBuf<int> cptr, cblk;             // ... cut into chunks of 16 pairs: a block's first chunk, a chunk's block
    DevBuf<double> partial;             // ... what the chunks leave behind (48 doubles each), added up per block by k_schur_blocks_sum
    long long nchunks = 0;
    // ba_set_pattern_lists: the camera lists of ALL tracks of a sharded scene (optimised positions in the caller's order) - a shard's
    // own tracks define only part of the pattern, and the ranks add their [S | b] element by element: one list of blocks for all
    std::vector<int> shared_loff, shared_lpos;
    bool shared_lists = false;
    bool pairs_built = false;
    // PACKED: [S] holds the pattern's upper blocks ONLY, one after the other in the list's order (block u at 36 u) - no band at all
    // (a band of 5000 cameras is 7 GB, of 30 000 it does not fit the device; the list is 43 MB and 260 MB).  Decided by ba_set_problem for
    // scenes the sparse path takes whole (sparse_layout, pair lists built, no group kernel worth its while); the reductions, the
    // initialisation, the conjugate gradients and ba_get_reduced follow the list, everything else that reads a band refuses.
    bool packed = false;
    DevBuf<int> udiag;                  // ... the list index of every camera's diagonal block
    std::vector<long long> h_ublk;      // host copy of ublk (ba_get_reduced of a packed system)
    bool band_clean = false;            // every block of the band outside the pattern is zero (one full initialisation, nothing scribbled since)
    DevBuf<int> cidx;                   // ... and in the packed array of a solve: the (upper) block of every entry of the full pattern
    DevBuf<double> TW;                  // k_sparse_stage: T = W HPPinv and W of every observation (36 doubles each)
    DevBuf<double> packed_blocks;       // the pattern's upper blocks, contiguous (k_pcg_gather, once per solve; not needed when [S] is stored packed)
    DevBuf<double> minv, r, z, q, p[2], part;
    DevBuf<PcgStateRaw> state;
    PcgStateRaw* host_state = nullptr;  // pinned
    int host_status = 0;
    int iterations = 0;                 // of the last solve
    int prev_iterations = 0;            // ... when it converged (the next solve's first batch)
    int last_status = 0;
    double rel_residual = 0.0;
  } pcg;
  bool have_solution = false;
  bool defer = false;        // inside ba_lm_trial: leave status words / cost on the device, one read-back at the end
  DevBuf<int> flags;        // [0] unused, [1] solver status, [2..15] solver instrumentation, [40],[41] singular-point
                            // counters (alternate per ba_schur call)
  int sing_epoch = 0;       // which of the two counters the latest ba_schur used
  bool trial_init_done = false;   // the linearisation just launched has inverted the point blocks and cleared [S | b] (launch_point_blocks, fused): the next ba_schur launches neither
  HostResult* host_result = nullptr;   // pinned, device-visible: cost + status words of a trial
  DevBuf<double> res_xb;               // ba_lm_resident: the exchange buffer of its workgroups
  DevBuf<long long> res_epoch;         // ... their epoch words (never reset: a launch starts above res_epoch0)
  long long res_epoch0 = 0;
  DevBuf<double> res_cost;             // ... their trial-cost words [2][32] (by the parity of the trial)
  int res_parity = 0;
  bool res_inflight = false;            // ... a launch ba_lm_resident_begin made and ba_lm_resident_end has not collected yet
  int res_inflight_phys = 0;
  DevBuf<double> res_stage;            // ... the set a launch ends on, before the host commits it (ba_lm_resident_end)
  int* res_exit = nullptr;             // ... pinned: every workgroup's (exit reason, trials walked)
  int res_inflight_groups = 0;
  void* res_log = nullptr;             // ... and its pinned log (ResidentLog of ba_resident.h)
  void* io = nullptr;                  // pinned staging of ba_set_params / ba_get_params (small parameter sets)
  size_t io_bytes = 0;
  bool io_pending = false;             // an upload from it may still be in flight
  double* res_out = nullptr;           // ... and, pinned, the parameter set it ended on: [nc x 12 | nt x 3] in the internal order
  size_t res_out_doubles = 0;
  int res_out_phys = -1;               // the physical set the copy mirrors (-1: none; whoever writes a set says so: params_written)
  void params_written(int phys) { if (res_out_phys == phys) res_out_phys = -1; }
  void* res_trace = nullptr;           // ... and, with option solve_trace, clock stamps of its first 64 trials
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

namespace ba {

hipEvent_t get_event(ba_handle* h);
void resolve_timings(ba_handle* h);

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
hipError_t ensure_lds_attr(ba_handle* h, const void* fn);

// Host-facing per-point / per-observation arrays go through the internal order of ba_set_problem:
// rows of w doubles, perm[i] = the caller's index of internal row i (on the device: k_rows_permute).
// device rows -> caller's host array (synchronises the stream when a permutation is in the way)
int download_rows(ba_handle* h, const int* dev_perm, const double* dev, double* host, size_t n, int w);

// rows of w values per optimised camera, caller's order -> internal order (`tmp` must outlive any asynchronous use of the result)
template <typename T>
inline const T* cam_rows_in(const ba_handle* h, const T* caller, std::vector<T>& tmp, int w) {
  if (!caller || h->cpos_in.empty()) return caller;
  tmp.resize((size_t)h->nco * w);
  for (int p = 0; p < h->nco; ++p) std::copy(caller + (size_t)p * w, caller + (size_t)(p + 1) * w, tmp.begin() + (size_t)h->cpos_in[p] * w);
  return tmp.data();
}
// ... and back, in place: data[q] (internal position q) -> data[cpos_out[q]]
template <typename T>
inline void cam_rows_out(const ba_handle* h, T* data, int w) {
  if (!data || h->cpos_out.empty()) return;
  std::vector<T> tmp(data, data + (size_t)h->nco * w);
  for (int q = 0; q < h->nco; ++q) std::copy(tmp.begin() + (size_t)q * w, tmp.begin() + (size_t)(q + 1) * w, data + (size_t)h->cpos_out[q] * w);
}

// ---- ba_border.hip: the border of the reduced system (ba_border.h)
int border_setup(ba_handle* h);                                  // after the problem is set: the border cameras' observations, buffers
int border_schur(ba_handle* h, int p, double damping);            // the blocks of the border cameras (C, D, the border part of b)
int border_solve_lu(ba_handle* h, const unsigned char* dmask);    // the same for a system that is not positive definite: LU everywhere (ba_border.hip)
int solve_bcr_lu(ba_handle* h, const unsigned char* dmask, int ncams = -1, const double* rhs = nullptr);      // cyclic reduction with LU nodes (ba_solve.hip)
int border_solve(ba_handle* h, const unsigned char* dmask);       // after the band solve: Y = B^-1 C, the border system, the correction of dC
int border_join(ba_handle* h);                                     // the main stream waits for the side stream's border kernels
int border_get_dense(ba_handle* h, std::vector<double>& C, std::vector<double>& D);      // host copies of C [6 n1][ld], D [ld][ld]
int border_flatten(ba_handle* h, int nkeep, double* A_dev, double* rhs_dev);              // ba_flatten_reduced with a border (h->keep holds the indices)
// Which cameras go to a border so that the others fit a band of half-width <= t: is_border[p] by position in the order given
// (lists: distinct camera lists as positions in that order).  Returns their number, or -1 when more than kmax would be needed.
bool plan_camera_layout(int nco, const std::vector<int>& loff, const std::vector<int>& lpos, const std::vector<int>& lmult, bool allow_border,
                        std::vector<int>& newpos, int* n1_out, int* hb_out);
int choose_border(int nco, const std::vector<int>& loff, const std::vector<int>& lpos, int t, int kmax, std::vector<char>& is_border);

// ---- ba_order.hip: Cuthill-McKee on the co-visibility hypergraph
void cuthill_mckee_order(int nco, const std::vector<int>& loff, const std::vector<int>& lpos, std::vector<int>& newpos);
int order_half_bandwidth(const std::vector<int>& loff, const std::vector<int>& lpos, const std::vector<int>& newpos);

inline unsigned blocks_for(long long n) { return (unsigned)std::max<long long>(1, (n + kBlock - 1) / kBlock); }
DevProblem dev_problem(const ba_handle* h);
DevProblem dev_problem_band(const ba_handle* h);      // ... as the reductions into the band see it: border cameras are not optimised

inline size_t reduced_doubles(const ba_handle* h) { return h->pcg.packed ? (size_t)h->pcg.upper * 36 : (size_t)h->nco * (h->hb + 1) * 36; }
int ensure_reduced(ba_handle* h);

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
extern RcclApi g_rccl;

#define RCCLCHECK(h, call)                                                                                   \
  do {                                                                                                       \
    ncclResult_t r_ = (call);                                                                                \
    if (r_ != ncclSuccess)                                                                                   \
      return (h)->fail(BA_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
  } while (0)

// in-place sum over the shards of the band-stored [S | b] (contiguous), on the handle's stream (ba_core.hip)
int comm_allreduce_reduced(ba_handle* h);
void launch_copy_doubles(ba_handle* h, const double* src, double* dst, int n);

// ---- ba_schur.hip
// Which kernel forms the Schur reduction (bundle_adjuster.py:259-278) for this problem and these options.
enum { KERN_PAIRS = 0, KERN_GROUPS, KERN_MFMA2 = 3, KERN_MFMA3, KERN_DENSE };      // (the values BA_INFO_SCHUR_KERNEL reports; 2 was the single-wavefront matrix-core kernel of round 1)
int pick_schur_kernel(const ba_handle* h);
inline bool kern_is_mfma(int k) { return k == KERN_MFMA2 || k == KERN_MFMA3; }

// ---- ba_schur_window.hip: the window-group reductions
int launch_mfma3_all(ba_handle* h, int p, double damping, bool fuse_cam);
int mfma3_launches(int nts);
int launch_wide_all(ba_handle* h, int p, double damping, bool fuse_cam);
int wide_launches(const ba_handle* h);
int launch_rect(ba_handle* h, int p, double damping, bool fuse_cam);

// ---- ba_sort.hip (rocPRIM) and ba_problem.hip
hipError_t sort_pairs_u64(ba_handle* h, const unsigned long long* keys_in, unsigned long long* keys_out, const int* values_in,
                          int* values_out, size_t n, int end_bit);
hipError_t exclusive_scan_i32(ba_handle* h, const int* in, int* out, size_t n);
int ensure_plan(ba_handle* h);             // groups / chunks / windows of the general kernels, if ba_set_problem left them for later
bool resident_shape(const ba_handle* h);   // the problem's shape fits the resident loop (ba_resident.h)
int ensure_cam_units(ba_handle* h);        // cam_perm / cam_units of k_camera_blocks
int ensure_pair_units(ba_handle* h);       // units / chunks of k_schur_pairs
// host rows (caller's order) -> device rows (internal order) and back; perm = h->d_pperm.p / h->d_operm.p or nullptr (identity)
int upload_rows(ba_handle* h, const int* dev_perm, const double* host, double* dev, size_t n, int w);

// ---- ba_points.hip: ba_linearize; with fuse (ba_lm_trial + MFMA reduction) the camera blocks are left to the reduction kernel
int launch_point_blocks(ba_handle* h, int p, double* Wd, bool fused = false, double damping = 0.0, double rcond = 0.0);
bool trial_init_fusable(ba_handle* h);                            // ba_lm_trial: the group lineariser may invert the point blocks and clear [S | b] itself
int launch_camera_blocks(ba_handle* h, int p, bool clear);
int linearize_impl(ba_handle* h, int which, int store_W, bool fuse, double damping, double rcond);

// ---- ba_solve.hip (cyclic reduction up to 11 cameras per node, the one-workgroup band solver, the solve spread over ranks)
// and ba_solve_wide.hip (wider bands); each leaves the solution in h->dC and the status in flags[1]
int solve_bcr_wide(ba_handle* h, const unsigned char* dmask);
int solve_bcr_big(ba_handle* h, const unsigned char* dmask);
int solve_dense_chol(ba_handle* h, const unsigned char* dmask);
int solve_pcg(ba_handle* h, const unsigned char* dmask);          // conjugate gradients over the blocks the tracks define (ba_pcg.hip)
bool sparse_layout(ba_handle* h);                                 // a wide band of mostly structural zeros: the pattern-driven initialisation and solver apply
int launch_schur_init_sparse(ba_handle* h, double damping, int use_hcc);
int launch_schur_blocks(ba_handle* h, int p);                     // the reduction block by block over the pattern's pair lists (k_schur_blocks)
double pcg_band_fill(ba_handle* h);                               // fraction of the band's blocks that can be non-zero (builds the pattern)
int solve_band_lu(ba_handle* h, const unsigned char* dmask, int ncams = -1, const double* rhs = nullptr);      // LU with partial pivoting, any band width (ba_band_lu.h)
// k_bcr_assemble (ba_bcr.h): band (+ mask) -> D, U, f of the nodes of cb cameras; clears the status word
void launch_bcr_assemble(ba_handle* h, dim3 grid, int cb, const unsigned char* dmask, double* xsol, int* done, const int* nodes);
// ba_bcr_split.hip / ba_bcr_levels.hip: the node kernels, instantiated per cameras-per-node
hipError_t launch_bcr_fused(ba_handle* h, int hb, int nwork, hipStream_t st, int N, int s_first, double* D, const double* U, double* f,
                            double* P, double* Q, double* G, double* gv, int* info, double* x, const int* work, int* done);
hipError_t launch_bcr_split(ba_handle* h, int hb, int cnt, hipStream_t st, int N, int s, double* D, const double* U, double* f,
                            double* P, double* Q, double* G, double* gv, int* info, double* x);
hipError_t launch_bcr_eliminate(ba_handle* h, int hb, int cnt, size_t lds, hipStream_t st, int N, int s, double* D, double* U, double* f,
                                double* P, double* Q, double* G, int* info, double* x);
hipError_t launch_bcr_lu(ba_handle* h, int hb, int cnt, hipStream_t st, int N, int s, double* D, double* U, double* f, double* P, double* Q,
                         double* G, int* info, double* x);
// ba_band_solve.hip, ba_band_solve_masked.hip
hipError_t launch_band_solve_masked(ba_handle* h, int hb, size_t lds, hipStream_t stream, int nco, int ch, const double* S, const double* b,
                                    const unsigned char* mask, double* U, double* y, double* dinv, double* x, int* info);
hipError_t launch_band_solve(ba_handle* h, int hb, size_t lds, hipStream_t stream, int nco, int ch, const double* S, const double* b,
                             const unsigned char* mask, double* U, double* y, double* dinv, double* x, int* info);
bool dist_plan_static(int nco, int hb, int nranks, int* cb_out, int* N_out, int* P_out);
int dist_build_plan(ba_handle* h, int rank, int nranks);
int dist_stage(ba_handle* h, int stage, const uint8_t* cam_param_mask, size_t* count);

inline int big_node_cameras(int hb) { return (hb + 1) & ~1; }      // cb >= hb, even: B = 6 cb is a multiple of the panel kernel's 12-column steps

}  // namespace ba
