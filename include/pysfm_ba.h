/* pysfm_ba.h - C ABI of the MI355X bundle-adjustment inner loop.
 *
 * Drop-in boundary for the hot path of alexflint/pysfm's BundleAdjuster
 * (reference: bundle_adjuster.py).  The reference is pure Python and has no FFI;
 * every entry point below therefore names the reference *method* (file:line) whose
 * arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a pysfm
 * maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, fp64 values, int32 indices, row-major.
 *   - every function returns a ba_status (0 = ok, negative = error) and never
 *     throws; ba_last_error() returns a message for the last failure.
 *   - "host" pointers are ordinary CPU memory; the library owns all device
 *     memory behind the opaque handle unless an external device buffer is bound
 *     with ba_bind_reduced_buffers().
 *   - one handle = one GPU + one HIP stream; calls on one handle are not
 *     re-entrant.  Calls return after the work they enqueue has been submitted;
 *     functions that hand results to host memory synchronise the stream first.
 *   - "camera position" / "track position" = index into the camera_ids /
 *     track_ids lists chosen in BundleAdjuster.set_bundle
 *     (bundle_adjuster.py:54-114).
 */
#ifndef PYSFM_BA_H
#define PYSFM_BA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ba_handle ba_handle;

typedef enum ba_status {
  BA_OK = 0,
  BA_ERR_INVALID_ARG = -1,
  BA_ERR_NO_DEVICE = -2,   /* no usable HIP device / runtime */
  BA_ERR_HIP = -3,         /* a HIP call failed; see ba_last_error */
  BA_ERR_STATE = -4,       /* call order violated (e.g. schur before set_params) */
  BA_ERR_SINGULAR = -5,    /* plain-inverse mode met a singular 3x3 point block */
  BA_ERR_NOMEM = -6
} ba_status;

/* sensor_model.py:7-32 (Gaussian), 37-72 (Cauchy); Huber is new (same protocol). */
typedef enum ba_sensor_kind {
  BA_SENSOR_GAUSS = 0,  /* params = L row-major 2x2, L = chol(cov^-1): r = L e      */
  BA_SENSOR_CAUCHY = 1, /* params[0] = sigma                                        */
  BA_SENSOR_HUBER = 2,  /* params[0] = k                                            */
  BA_SENSOR_TABLE = 3   /* any isotropic robustifier r = h(|e|) e (the reference's plug-in point, sensor_model.py:19-32),
                         * sampled: params = [log2 rho_0, nodes per octave, n, h_0, m_0, ..., h_(n-1), m_(n-1)] with node i
                         * at rho_i = 2^(log2 rho_0 + i / nodes per octave), h_i = h(rho_i), m_i = dh/d(log2 rho) there;
                         * cubic Hermite interpolation in log2 rho on the device; h = h_0 below rho_0              */
} ba_sensor_kind;

/* Two resident parameter sets: the accepted bundle and the LM trial
 * (bundle.py:301-310 clone_params is the reference's rollback snapshot). */
typedef enum ba_param_set { BA_PARAMS_CUR = 0, BA_PARAMS_TRIAL = 1 } ba_param_set;

/* kernel ids for ba_get_timings */
enum {
  BA_K_COST = 0, BA_K_LINEARIZE, BA_K_POINT_INVERT, BA_K_SCHUR_INIT, BA_K_SCHUR_PAIRS,
  BA_K_BACKSUB, BA_K_UPDATE, BA_K_FLATTEN, BA_K_BAND_SOLVE, BA_K_EVAL, BA_K_CAMERA_BLOCKS, BA_K_TRIANGULATE,
  BA_K_BCR_ASSEMBLE, BA_K_BCR_ELIMINATE, BA_K_BCR_BACKSOLVE, BA_K_DENSE_SOLVE, BA_K_BORDER_SCHUR, BA_K_BORDER_SOLVE, BA_K_BCR_REFINE, BA_K_PCG_SOLVE, BA_K_COUNT
};

/* ---- lifecycle ---------------------------------------------------------- */
int ba_create(int device_id, ba_handle** out);
int ba_destroy(ba_handle* h);
/* message of the last error on this handle (h may be NULL: last ba_create error) */
const char* ba_last_error(const ba_handle* h);
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL = own stream */
int ba_set_stream(ba_handle* h, void* hip_stream);
int ba_synchronize(ba_handle* h);
/* Test / measurement switches; the defaults are the product path and the library never reads the environment.
 *   "schur"         auto | pairs | groups | mfma2 | mfma   force a Schur-reduction kernel (falls back to pairs when not applicable)
 *   "lds_window"    1 | 0                                  LDS accumulation window of the matrix-core reduction (0: global atomics only)
 *   "fast_paths"    1 | 0                                  short cuts of the per-observation arithmetic for K = I and the unit Gaussian
 *                                                          sensor model (the reference's defaults); 0: the general formulas always
 *   "fused_backsolve" 1 | 0                                all back-substitution levels of the cyclic reduction in one launch (when its
 *                                                          nodes fit the chip at once) / one launch per level
 *   "six_tile_launch" 1 | 0                                window reduction: tile columns 0 .. 5 in the first launch (one launch less for windows of 14+ cameras) / 0 .. 3
 *   "bcrw_merged"   1 | 0                                  wide cyclic reduction (nodes of 14 .. 23 cameras): factorisation and substitution of a level in one kernel / two
 *   "dense_lookahead" 1 | 0                                dense Cholesky / big-node levels: one launch per block column (the next panel step beside the
 *                                                          trailing update) / two
 *   "solver"        auto | bcr | band | dense | lu | bcr1 | pcg   force the reduced solver (lu: always report -1 = caller's LU; bcr1: the
 *                                                          cyclic reduction with one compute unit per node instead of three -
 *                                                          for nodes of 12 and 13 cameras that is the wide solver's kernels; pcg:
 *                                                          conjugate gradients over the blocks the tracks define, see ba_pcg_info)
 *   "refine"        auto | 1 | 0                           one step of iterative refinement behind the cyclic reduction (residual in doubled
 *                                                          precision, correction through the kept factors): auto = where the reduced system is
 *                                                          damped below 1e-2; BA_INFO_SOLVES_REFINED counts
 *   "pcg_tol" x   "pcg_max_iter" n   "pcg_batch" n         conjugate gradients: converged at ||r|| <= x ||b|| (1e-12); iteration budget (0 =
 *                                                          max(1000, min(20000, 4 nco))); iterations enqueued between two looks at the state (50)
 *   "resident_fault" g   "refine_debug" 1 | 0              test aids: workgroup g of the resident loop REPORTS a time-out (-1: none); the
 *                                                          refinement's items do not wait for each other (wrong numbers: its floor time)
 *   "point_kernels" auto | v1                              lanes-per-point k_linearize / k_backsub instead of the group-packed ones
 *   "fuse_cost" "fuse_cam" "fuse_invert"   1 | 0           pieces of ba_lm_trial folded into neighbouring kernels (defaults 1, 1, 1)
 *   "sort_points"   1 | 0                                  internal point order chosen by ba_set_problem (default 1; see there)
 *   "gm_cap"        n                                      points per group of the MFMA reduction (0 = automatic)
 *   "gm_chunk"      n                                      groups per workgroup of the MFMA reductions (0 = automatic: 4, or 8 over several rounds)
 *   "camera_order"  auto | off | always                    internal order of the optimised cameras (see ba_set_problem): auto = when the
 *                                                          caller's order is not provably as narrow as an order can be
 *   "reuse_linearization" 1 | 0                            ba_lm_trial after a rejected trial does not form the point blocks of the unchanged
 *                                                          current set again (default 1; 0: every trial linearises, as the reference does)
 *   "border_side_stream" 1 | 0                             the border's blocks on a side stream beside the cyclic reduction (default 1)
 *   "border"        1 | 0                                  band + border layouts of the reduced system (see ba_set_problem; default 1)
 *   "solve_trace"   1 | 0                                  per-phase cycle counts of the node kernels on stderr (PROFILE builds)
 * Unknown names / values: BA_ERR_INVALID_ARG.  Options that shape the work lists ("sort_points", "gm_cap", "gm_chunk") take effect at
 * the next ba_set_problem. */
int ba_set_option(ba_handle* h, const char* name, const char* value);
/* Test aid: fills the LDS of every compute unit and every workspace buffer of the handle (normal-equation blocks, reduced
 * system, solver workspace, updates, the trial parameter set - not the problem, not the current parameter set) with NaNs
 * and forgets every cached intermediate.  A following computation that reads anything it did not write itself shows it. */
int ba_debug_poison(ba_handle* h);

/* ---- problem definition: BundleAdjuster.set_bundle (bundle_adjuster.py:54-114)
 * nc cameras, nt tracks, nobs observations in ANY order (the reference visits tracks and their measurements
 * in whatever order its containers yield, bundle_adjuster.py:222-226), each (camera, track) pair at most once.
 * The library keeps its own internal order - a track's observations by optimised-camera position, tracks with
 * identical camera lists next to each other, lists ordered by their first optimised camera - so that every
 * scene reaches the grouped kernels; all host-facing per-track / per-observation arrays (X, HPP, bP, dP, W,
 * e, r, Jc, Jp) are in the caller's order.  Option "sort_points" = 0 keeps the caller's order as it is.
 * The same holds for the OPTIMISED CAMERAS: the reference's dense S does not care in which order optim_camera_ids lists the
 * cameras (bundle_adjuster.py:259-312); the band stored here does, so when the widest spread of positions inside a track is
 * larger than a track of that length needs, the library orders the cameras itself (Cuthill-McKee on the co-visibility
 * hypergraph, pysfm_amd/csrc/ba_order.hip) and keeps that order if the band gets narrower.  Every array the ABI takes or returns
 * by optimised position (S, b, dC, cam_param_mask, motion updates) stays in the CALLER's positions.  Not done for a handle with a
 * communicator or a minimum band width set (the ranks of a sharded adjuster share one layout).
 * BAND + BORDER: a camera sequence with a few long-range tracks (a loop closure: camera 3 and camera 503 see the same point)
 * has a narrow band but for those tracks.  When the band would be wider than the narrow cyclic reduction's one-unit nodes (11 cameras) and
 * moving at most 21 cameras out of it makes it fit, those cameras become a border, S = [[B, C], [C^T, D]] with B the band of the
 * others (pysfm_amd/csrc/ba_border.h): ba_reduced_layout then describes B (rows of the border cameras unused), ba_get_reduced /
 * ba_get_solution return the full system / solution as always.  The layout with the lowest modelled solve cost among {caller's
 * order, Cuthill-McKee order} x {as it is, with a border} is taken (BA_INFO_BORDER_CAMERAS, BA_INFO_HALF_BANDWIDTH say which).
 * A bordered system that is not positive definite is solved again by LU like any other (the border's columns through the band's LU
 * solver one by one, the border system by Gaussian elimination with partial pivoting: BA_SOLVE_BCR_LU / BA_SOLVE_BAND_LU).
 *   obs_cam[nobs], obs_pt[nobs]  positions;  obs_z[nobs*2] measurements
 *   K[9]                         calibration (general 3x3)
 *   cam_opt_pos[nc]              position in optim_camera_ids, or -1 (frozen)
 *   pt_opt[nt]                   1 if the track is in optim_track_ids
 */
int ba_set_problem(ba_handle* h, int32_t nc, int32_t nt, int64_t nobs,
                   const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_z,
                   const double* K, const int32_t* cam_opt_pos, const uint8_t* pt_opt);

/* What ba_set_problem made of the scene (work lists are built once per problem): out[i] for i < n, see BA_INFO_*.
 * Tests and bench.py use it to assert that a scene takes the kernels it is meant to take. */
enum {
  BA_INFO_POINTS_PERMUTED = 0, /* 1: the internal point order differs from the caller's track order                */
  BA_INFO_OBS_PERMUTED,        /* 1: the internal observation order differs from the caller's                      */
  BA_INFO_GROUPS,              /* runs of points with identical camera lists (<= 24 points each)                   */
  BA_INFO_MFMA_GROUPS,         /* groups of the matrix-core reduction                                              */
  BA_INFO_POINT_GROUPS,        /* 1: the group-packed k_linearize_groups / k_backsub_groups are used               */
  BA_INFO_MAX_TRACK_LEN,
  BA_INFO_HALF_BANDWIDTH,
  BA_INFO_SCHUR_MFMA,          /* 1: the Schur reduction runs on the fp64 matrix cores                             */
  BA_INFO_SCHUR_GROUPS,        /* 1: a group reduction (vector or MFMA) applies; 0: k_schur_pairs                  */
  BA_INFO_LDS_WINDOW_ROWS,     /* band rows of the reduction's LDS accumulation window (0: global atomics only)   */
  BA_INFO_PAIR_UNITS,
  BA_INFO_SCHUR_KERNEL,        /* 0 pairs, 1 vector groups, 2 / 3 / 4 matrix-core reductions (single wavefront, producer-consumer L <= 10,
                                  producer-consumer any L <= 24), 5 dense visibility                                                   */
  BA_INFO_MFMA_POINTS_PER_BATCH_CAP,
  BA_INFO_MFMA_K_ROWS,
  BA_INFO_CAMERAS_PERMUTED,    /* 1: the library ordered the optimised cameras itself (see ba_set_problem)                */
  BA_INFO_CALLER_HALF_BANDWIDTH, /* the half-bandwidth the caller's camera order would have had                         */
  BA_INFO_BORDER_CAMERAS,      /* cameras in the border of the reduced system (band + border, see ba_set_problem)          */
  BA_INFO_LINEARIZATIONS_REUSED, /* trials of ba_lm_trial since ba_set_problem that reused the linearisation of an unchanged current set */
  BA_INFO_SOLVES_REFINED,      /* reduced solves since ba_set_problem that took the step of iterative refinement (option refine) */
  BA_INFO_PACKED_STORE,        /* 1: the reduced system is stored as the list of the blocks the tracks define, no band (scenes the sparse path takes whole; see ba_reduced_layout) */
  BA_INFO_COUNT
};
int ba_problem_info(ba_handle* h, int64_t* out, int32_t n);

/* The camera ordering ba_set_problem applies, as a pure function (no handle, no GPU): nco optimised cameras, nlists camera
 * lists (list l = the optimised positions list_pos[list_off[l] .. list_off[l + 1]) one track sees; duplicates allowed).
 * new_pos[p] = the position Cuthill-McKee on the co-visibility hypergraph gives the caller's position p (a permutation of
 * 0 .. nco - 1); *half_bandwidth = the widest spread of new positions inside a list (either output may be NULL). */
int ba_order_cameras(int32_t nco, int32_t nlists, const int32_t* list_off, const int32_t* list_pos, int32_t* new_pos, int32_t* half_bandwidth);
/* ... and the whole layout decision of ba_set_problem as a pure function: the candidates {caller's order, Cuthill-McKee, Cuthill-McKee
 * without the few single-point lists that tie far-apart cameras} x {as it is, with a border of at most 21 cameras that brings the
 * band down to <= 11} ranked by a model of their solve cost.  list_points[l] = points that have list l (NULL: not known).
 * new_pos[p]: final position of the caller's position p - positions >= *band_cameras are border cameras; *half_bandwidth: the
 * band's.  The caller's order comes back unchanged (band_cameras = nco) when nothing beats it. */
int ba_plan_camera_layout(int32_t nco, int32_t nlists, const int32_t* list_off, const int32_t* list_pos, const int32_t* list_points,
                          int32_t allow_border, int32_t* new_pos, int32_t* band_cameras, int32_t* half_bandwidth);

/* ---- multi-GPU: the shards' collectives inside the library (one all-reduce of the reduced camera
 * system per linearisation, plus the 16 KB trial record), issued with RCCL on the handle's OWN stream - no
 * stream hand-over to a framework's communicator, no host round trip between the halves of a trial.
 * ba_comm_load resolves RCCL at run time from the given librccl.so (pass the one the process has already
 * loaded, e.g. torch's: one RCCL instance per process; NULL = "librccl.so" on the loader path).
 * Rank 0 calls ba_comm_unique_id and ships the 128 bytes to the other ranks by any means; every rank then
 * calls ba_comm_init (collective, like ncclCommInitRank).  With a communicator attached ba_lm_trial is the
 * sharded trial: linearise + reduce the shard, all-reduce [S | b], solve (replicated), back-substitute and
 * update the shard, all-reduce the trial record, return the global cost. */
int ba_comm_load(const char* librccl_path);
int ba_comm_unique_id(void* id128 /*[128] out*/);
int ba_comm_init(ba_handle* h, const void* id128, int32_t rank, int32_t nranks);
int ba_comm_destroy(ba_handle* h);
int ba_comm_allreduce_reduced(ba_handle* h);                          /* after ba_schur: [S | b] summed over the shards, in place */
int ba_comm_allreduce_sum(ba_handle* h, double* values /*host, in place*/, int32_t n);

/* Lower bound for the block half-bandwidth chosen by the NEXT ba_set_problem.  The sharded adjuster
 * sets it to the maximum over the ranks so that every rank stores [S | b] in the same band layout
 * (the all-reduce adds the buffers element by element). */
int ba_set_min_half_bandwidth(ba_handle* h, int32_t min_hb);

/* The layout of the optimised cameras for the following ba_set_problem calls, imposed by the caller instead of chosen by the
 * library: new_pos[p] = internal position of the caller's position p (a permutation of 0 .. nco - 1; no border).  For the ranks of
 * a sharded adjuster, which add their [S | b] buffers element by element and so must all use ONE layout: every rank plans it from
 * the WHOLE scene (ba_plan_camera_layout with allow_border = 0: the same input gives the same layout) and imposes it; the band
 * width they agree on (ba_set_min_half_bandwidth) is then the one of the new positions.  new_pos = NULL: the library chooses again. */
int ba_set_camera_layout(ba_handle* h, const int32_t* new_pos, int32_t nco);
/* The layout ba_set_problem ended up with: new_pos[p] (nco entries) = the INTERNAL position of the caller's optimised position p
 * (the identity when BA_INFO_CAMERAS_PERMUTED is 0), *band_cameras = how many internal positions form the band (nco minus the
 * border cameras, which sit last).  Host arrays cross this API in the caller's positions and need none of this; the DEVICE views
 * (ba_reduced_device_ptrs, ba_bind_reduced_buffers, ba_reduced_layout) are in the internal order - see there. */
int ba_get_camera_layout(ba_handle* h, int32_t* new_pos, int32_t* band_cameras);

/* bundle.sensor_model (sensor_model.py:19-32 protocol) */
int ba_set_sensor(ba_handle* h, int kind, const double* params, int nparams);

/* camera / point parameters: R[nc*9], t[nc*3], X[nt*3] (host) */
int ba_set_params(ba_handle* h, int which, const double* R, const double* t, const double* X);
int ba_get_params(ba_handle* h, int which, double* R, double* t, double* X);
/* make the trial set the current one (optimize() accept branch, bundle_adjuster.py:151) */
int ba_swap_params(ba_handle* h);

/* ---- BundleAdjuster.compute_cost (bundle_adjuster.py:165-171) ----------- */
int ba_cost(ba_handle* h, int which, double* cost_out);

/* ---- per-observation evaluation: Bundle.reproj_error / residual / Jresidual
 * (bundle.py:243-277).  Any output may be NULL.  e[nobs*2], r[nobs*2],
 * Jc[nobs*12] (2x6 row-major), Jp[nobs*6] (2x3). */
int ba_eval_observations(ba_handle* h, int which, double* e, double* r, double* Jc, double* Jp);

/* sensor_model protocol on a batch of n errors (sensor_model.py:19-32): r[n*2] =
 * residual_from_error(e), J[n*4] = Jresidual_from_error(e) (2x2 row-major), with the
 * sensor set by ba_set_sensor.  Needs no problem.  r or J may be NULL. */
int ba_eval_sensor(ba_handle* h, int64_t n, const double* e, double* r, double* J);

/* ---- BundleAdjuster.prepare_schur_complement (bundle_adjuster.py:211-234)
 * Undamped blocks stay on the device.  store_W != 0 also keeps the per-observation
 * HCP blocks W = Jc^T Jp (6x3) for ba_get_blocks. */
int ba_linearize(ba_handle* h, int which, int store_W);
/* HCC[nc*36], bC[nc*6], HPP[nt*9], bP[nt*3], W[nobs*18]; any may be NULL */
int ba_get_blocks(ba_handle* h, double* HCC, double* bC, double* HPP, double* bP, double* W);

/* ---- apply_damping + compute_schur_complement (bundle_adjuster.py:238-278)
 * damping: diag *= (1+lambda) on every HCC / HPP block (optimize.py:7-9).
 * pinv_rcond >= 0: numpy.linalg.pinv(HPP, rcond) semantics; < 0: plain inverse
 * (SCHUR_COMPLIMENT_PINV_THRESHOLD = None).  Needs ba_linearize first.
 * The reduced system stays on the device in block-band form (see ba_reduced_layout). */
int ba_schur(ba_handle* h, int which, double damping, double pinv_rcond);
/* Layout of the device-resident reduced system.  S is symmetric; only 6x6 blocks (i, j)
 * with i <= j <= i + hb can be non-zero, hb = widest spread of optimised-camera
 * positions inside one track (hb = nco-1 is a dense system).  Block (i, i+d) starts at
 * ((i*(hb+1) + d)*36 doubles; S_doubles = nco*(hb+1)*36.  b has nco*6 doubles.
 * PACKED STORE (BA_INFO_PACKED_STORE; option packed_store, default 1, read by ba_set_problem): a scene without a band under any
 * camera order that the sparse path takes whole (BA_SOLVE_PCG: at least 1500 optimised cameras or solver = pcg set before
 * ba_set_problem, at most a tenth of the band's blocks non-zero) keeps NO band: S is the list of the blocks its tracks define, block
 * u at 36 u, S_doubles = 36 x (blocks of ba_pcg_info) - 43 MB instead of 7 GB at 5000 cameras.  half_bandwidth is still the spread
 * of the camera order.  ba_get_reduced / ba_get_solution / the whole trial work as always; what needs a band refuses with
 * BA_ERR_STATE: ba_flatten_reduced, the direct solvers (solver = dense | lu | ...), the dense-visibility mode, a communicator. */
int ba_reduced_layout(ba_handle* h, int32_t* nco, int32_t* half_bandwidth, int64_t* S_doubles);
/* S[nco*nco*36] laid out (nco,nco,6,6) and b[nco*6], expanded to the reference's dense
 * symmetric form (host) */
int ba_get_reduced(ba_handle* h, double* S, double* b);
/* HPP_invs[nt*9] (host) */
int ba_get_point_inverses(ba_handle* h, double* HPP_inv);

/* Device views of the reduced system for the collective and the dense solver
 * (sizes: ba_reduced_layout).  With ba_bind_reduced_buffers the caller supplies the
 * device memory (e.g. one torch tensor holding [S | b]) instead.  (A caller that WRITES the band itself keeps every block no
 * track touches zero: on sparse layouts - BA_SOLVE_PCG - only those blocks are initialised again by the next ba_schur.)
 * These views are in the INTERNAL order of the optimised cameras (camera_order = auto is the default: an unordered scene is
 * reordered, BA_INFO_CAMERAS_PERMUTED): row / block i belongs to the caller's position p with new_pos[p] == i
 * (ba_get_camera_layout).  With border cameras (BA_INFO_BORDER_CAMERAS > 0) the band covers the first band_cameras positions
 * only; the border's blocks live in library-owned buffers and band rows past band_cameras are not meaningful.  A caller that
 * solves from the device views itself sets option camera_order = off and border = 0 before ba_set_problem (the caller's order,
 * one band, as wide as that order makes it) - or reads the system through ba_get_reduced / ba_flatten_reduced, which are in the
 * caller's positions whatever the layout. */
int ba_reduced_device_ptrs(ba_handle* h, void** S_band, void** b);
int ba_bind_reduced_buffers(ba_handle* h, void* S_band_dev, void* b_dev);

/* Dense visibility (every track seen by most cameras: the reference's data/oleg_synthetic).  With `on`,
 * ba_schur forms the reduction as ONE symmetric matrix product over all points,
 * S -= Ud^T diag(D) Ud with Ud [3 nt][6 nco] (library-owned, 8 * 3 nt * 6 nco bytes), on the fp64 matrix
 * cores, instead of the per-pair kernels whose global atomics dominate when every camera pair shares
 * every track.  The caller decides (the Python host: band wider than 21 blocks and >= 25 % of the
 * (camera, track) pairs observed); results do not depend on it beyond rounding. */
int ba_set_dense_visibility(ba_handle* h, int32_t on);

/* ---- BundleAdjuster.solve_motion_normal_eqns (bundle_adjuster.py:281-312)
 * Device-resident solve of the reduced camera system.  Cholesky first, by block half-bandwidth hb: block cyclic
 * reduction with LDS-resident nodes (hb <= 23: all levels and the back-substitution in ONE launch up to 13 cameras per node,
 * four kernels per level beyond), a single-workgroup band Cholesky (fewer than 4 super-blocks), block
 * cyclic reduction with nodes in device memory - every level a batched partial dense Cholesky (hb > 23 and at least
 * four nodes of hb cameras: BA_SOLVE_BCR_BIG, any number of cameras), or a dense blocked Cholesky of the whole matrix
 * (hb > 23 and fewer nodes, up to 16000 unknowns).  A system the Cholesky solver reports as not positive definite is
 * solved again by LU with partial pivoting, which is what the reference's numpy.linalg.solve does with it (option
 * device_lu, default on): the cyclic reduction with LU nodes for hb <= 11 (BA_SOLVE_BCR_LU: pivoting inside a node),
 * LU down the band for every other shape (BA_SOLVE_BAND_LU: gesv's pivot choices on the same matrix; also what
 * option solver = lu forces, and what systems no Cholesky solver takes go through).
 * cam_param_mask[nco*6] (host, may be NULL = all kept): 0 deletes that camera parameter
 * from the system (its solution entry is 0).  *info: 0 = solved, solution stays on the
 * device for ba_backsubstitute / ba_get_solution; > 0 = the LU met an exactly zero pivot (1-based column, gesv's info:
 * where the reference raises LinAlgError -> NormalEquationsIllconditioned) - or, with device_lu off, the index of the
 * Cholesky pivot that was not positive; 0x7f000001 = a workgroup of the one-launch cyclic reduction gave up waiting for
 * another - a fault of the solver, reported instead of hanging the GPU (solve again with option solver = lu).
 * ba_last_solve_kind: which solver produced the state of the last ba_solve_reduced. */
enum { BA_SOLVE_NONE = 0, BA_SOLVE_BCR, BA_SOLVE_BCR_WIDE, BA_SOLVE_BAND, BA_SOLVE_DENSE_CHOLESKY, BA_SOLVE_BCR_LU, BA_SOLVE_BCR_BIG, BA_SOLVE_BAND_LU, BA_SOLVE_PCG };
#define BA_SOLVE_TIMED_OUT 0x7f000001   /* *info of a cyclic reduction whose workgroups gave up waiting for each other */
#define BA_SOLVE_STALLED 0x7f000002     /* *info of conjugate gradients (BA_SOLVE_PCG) that did not converge within their iteration budget */
/* Conjugate gradients over the blocks of S the tracks define (csrc/ba_pcg.h; solver = pcg, or chosen when the scene has no narrow
 * band under any camera order: at least 1500 optimised cameras, at most a tenth of the band's blocks non-zero): the blocks of the
 * upper triangle that can be non-zero, the iterations and ||r|| / ||b|| of the last such solve, the fraction of the band they fill. */
int ba_pcg_info(ba_handle* h, int64_t* blocks, int32_t* iterations, double* rel_residual, double* band_fill);
/* A SHARDED scene on the sparse path: every rank hands over the camera lists of ALL the scene's tracks (list l = optimised positions
 * list_pos[list_off[l] .. list_off[l + 1]) in the caller's order; duplicates welcome), for the following ba_set_problem calls: the
 * list of blocks - the layout of the packed [S | b] the ranks add up element by element - is then the same on every rank, whatever
 * tracks it holds; its own tracks supply the observation pairs.  The ranks must also take the same decisions (solver = pcg and
 * schur = pairs set by all or by none before ba_set_problem: pysfm_amd.BundleAdjuster does).  nlists = 0: forget the lists. */
int ba_set_pattern_lists(ba_handle* h, int32_t nlists, const int32_t* list_off, const int32_t* list_pos);
int ba_solve_reduced(ba_handle* h, const uint8_t* cam_param_mask, int32_t* info);
int ba_last_solve_kind(const ba_handle* h);
int ba_get_solution(ba_handle* h, double* dC /*[nco*6] host*/);
/* the caller's own solution of the reduced system (e.g. numpy.linalg.solve of ba_get_reduced's arrays) in place of the device's:
 * ba_backsubstitute / ba_apply_update / ba_get_solution use it from here on; clears the solver's status word */
int ba_set_solution(ba_handle* h, const double* dC /*[nco*6] host*/);
/* For callers that want the reduced system as the reference forms it before its solve: the flat (6nco x 6nco) matrix
 * with rows / columns of masked camera parameters deleted (bundle_adjuster.py:290-299).  keep[nkeep] lists the kept
 * flat parameter indices (host).  A_dev[nkeep*nkeep], rhs_dev[nkeep] are caller-owned DEVICE buffers.  (The library's
 * own solve does not go through it.) */
int ba_flatten_reduced(ba_handle* h, const int32_t* keep, int32_t nkeep, void* A_dev, void* rhs_dev);

/* ---- BundleAdjuster.backsubstitute (bundle_adjuster.py:316-331)
 * dC[nco*6] host (solution of the reduced system, zeros at masked parameters), or NULL
 * to use the device-resident solution of ba_solve_reduced;
 * dP[nt*3] for every track position (host, may be NULL: result stays on device). */
int ba_backsubstitute(ba_handle* h, int which, const double* dC, double* dP);

/* ---- update_motion / update_structure (bundle_adjuster.py:334-343;
 * Camera.perturb bundle.py:76-80; SO3.exp lie.py:21-34)
 * dst = src (+) update.  motion[nco*6] / structure[nt*3] are host arrays in the
 * sign the reference passes to perturb(); if both are NULL the update is
 * -(dC, dP) of the last ba_backsubstitute, taken from device memory.
 * Only optimised cameras / tracks move. */
int ba_apply_update(ba_handle* h, int src, int dst, const double* motion, const double* structure);

/* ---- one Levenberg-Marquardt trial, the inner-loop body of BundleAdjuster.optimize
 * (bundle_adjuster.py:132-146): compute_update(damping) on the current set, apply it to
 * the trial set, evaluate the trial cost.  Identical to calling ba_linearize, ba_schur,
 * ba_solve_reduced, ba_backsubstitute, ba_apply_update, ba_cost in turn, but enqueued as
 * one batch with a single host synchronisation.  *info as in ba_solve_reduced: when it is
 * non-zero the trial set is garbage and the caller repeats the step through the dense
 * solve path.  The accept / reject decision stays with the caller (ba_swap_params). */
int ba_lm_trial(ba_handle* h, double damping, double pinv_rcond, const uint8_t* cam_param_mask, double* next_cost,
                int32_t* info);

/* The same trial in two halves for the sharded (multi-GPU) adjuster, where the partial reduced
 * systems have to be summed over the ranks in between (one all-reduce of [S | b],
 * enqueued by the caller on the handle's stream):
 *   ba_lm_trial_begin : ba_linearize + ba_schur, nothing read back;
 *   ba_lm_trial_end   : ba_solve_reduced + ba_backsubstitute + update of the trial set + ba_cost,
 *                       nothing read back either.  The rank's partial trial cost is left in the
 *                       device buffer bound with ba_bind_trial_result: result[0 .. BA_TRIAL_PARTIALS)
 *                       = partial sums (add them), result[BA_TRIAL_PARTIALS] = singular point blocks,
 *                       result[BA_TRIAL_PARTIALS + 1] = solver status (*info of ba_solve_reduced).
 * The caller sums / all-reduces that buffer and synchronises once per trial. */
#define BA_TRIAL_PARTIALS 2048
int ba_bind_trial_result(ba_handle* h, void* result_dev /* BA_TRIAL_PARTIALS + 2 doubles, or NULL */);
int ba_lm_trial_begin(ba_handle* h, double damping, double pinv_rcond);
int ba_lm_trial_end(ba_handle* h, const uint8_t* cam_param_mask, int32_t* pre_info);
/* the tail of ba_lm_trial_end once a solution is on the device: back-substitution, trial set, trial cost */
int ba_lm_trial_finish(ba_handle* h);

/* ---- the whole loop of BundleAdjuster.optimize() / step() (bundle_adjuster.py:117-162) for a problem that fits a few compute
 * units, as one resident launch of a few workgroups (pysfm_amd/csrc/ba_resident.h): <= 16 optimised cameras, <= 32 cameras, <= 1024 tracks of <= 16
 * observations, no communicator.  The sliding-window caller
 * (window_slam.py:17-48) solves one such problem per frame; at that size a trial costs 68 us through ba_lm_trial (six
 * launches and one synchronisation for a thousand observations) and ~26 us here, with no round trip to the host between trials.
 *   ba_lm_resident_fits  1 when the handle's problem, sensor model and options allow it, else 0
 *   ba_lm_resident       runs the reference's schedule on the device from the state given (damping, steps taken, inside a
 *                        step or not, converged, cost of the current set or < 0 when unknown) until it converges, max_steps
 *                        are taken, the log is full, or a trial needs the general path; the current parameter set is updated
 *                        in place, *log says what happened trial by trial.  exit_reason: 0 done, 1 log full (call again
 *                        with the state of the log), 2 the reduced system of the next trial is not positive definite
 *                        (exit_info = 1 + the first unknown of the block of 12 columns in which the factorisation broke down), 3 a
 *                        singular point block in plain-inverse mode - for 2 and 3 the caller runs that one trial through
 *                        ba_lm_trial / the stepwise calls and comes back; 4 the workgroups of the launch lost each other (compute
 *                        units taken away underneath it): NOTHING happened - no trial is logged, the current set is the one the
 *                        launch was given - and the caller goes on through ba_lm_trial (exit_info = trials the launch had walked).
 *   ba_lm_resident_begin / _end   the same in two halves: _begin launches and returns, _end waits and fills *log.  For a caller
 *                        with host work that does not depend on the run (window_slam.py prepares the NEXT window on a second
 *                        handle meanwhile); nothing else may be asked of the handle in between. */
#define BA_RESIDENT_MAX_TRIALS 1000
typedef struct ba_resident_log {
  int32_t ntrials, nsteps, converged, in_step;
  int32_t exit_reason, exit_info, accepted, have_cost0;
  double damping, cost0, cur_cost, reserved;
  double trial_damping[BA_RESIDENT_MAX_TRIALS];
  double trial_cost[BA_RESIDENT_MAX_TRIALS];
  int32_t trial_accepted[BA_RESIDENT_MAX_TRIALS];
} ba_resident_log;
int ba_lm_resident_fits(ba_handle* h);
int ba_lm_resident(ba_handle* h, int32_t max_steps, int32_t steps_taken, int32_t in_step, int32_t converged, double damping,
                   double improvement_threshold, double pinv_rcond, double cur_cost,
                   const uint8_t* cam_param_mask /* [6 nco] as in ba_solve_reduced, or NULL */, ba_resident_log* log);
int ba_lm_resident_begin(ba_handle* h, int32_t max_steps, int32_t steps_taken, int32_t in_step, int32_t converged, double damping,
                         double improvement_threshold, double pinv_rcond, double cur_cost, const uint8_t* cam_param_mask);
int ba_lm_resident_end(ba_handle* h, ba_resident_log* log);
/* with option solve_trace: 16 words per trial for the first 64 trials of the last ba_lm_resident - wall_clock64 (100 MHz) at the
 * phase boundaries [0..7], [8] = 1 when the trial linearised */
int ba_lm_resident_trace(ba_handle* h, int64_t* out /* [64 * 16] */);
/* ... and the reduced system of its FIRST trial as the workgroups assembled it (dense, 6 nco x 6 nco | 6 nco) with the solution
 * they found: what the parity tests compare with ba_get_reduced / ba_get_solution of the stepwise path */
int ba_lm_resident_debug(ba_handle* h, double* S_out, double* b_out, double* dC_out);

/* ---- the reduced solve SPREAD OVER THE RANKS of a sharded adjuster (pysfm_amd/csrc/ba_dist.h; no counterpart in the
 * single-process reference: same arithmetic as ba_solve_reduced, bundle_adjuster.py:281-312).  For block-banded systems whose
 * band is too large to be summed over the ranks and solved by each of them (BASELINE config 5).  The elimination tree of the
 * cyclic reduction is cut along the ranks: super-blocks of cams_per_node >= half_bandwidth cameras, rank r owns the
 * nodes_per_rank - 1 nodes from r * nodes_per_rank, the separator node behind them, and must be given the tracks whose first
 * optimised camera lies in those (positions [r, r + 1) * nodes_per_rank * cams_per_node).  Needs nranks = 2^g >= 2.
 *   ba_dist_plan    pure function: the cut for (nco, half_bandwidth, nranks); BA_ERR_STATE when it does not apply
 *   ba_dist_enable  after ba_set_problem (with the common band width): switches ba_lm_trial (communicator attached) to the
 *                   distributed solve; nranks <= 1 switches it off.  ba_dist_info: [on, cams_per_node, nodes, nodes_per_rank,
 *                   first / end own node, first / end own camera position, doubles of the three exchanges, separators]
 *   ba_dist_stage   for callers that run the collectives themselves (torch.distributed): after ba_lm_trial_begin, stages
 *                   1, 2, 3 each leave *doubles_to_sum doubles in the exchange buffer (ba_dist_bind_exchange: caller-owned
 *                   device memory of at least the largest exchange), to be summed over the ranks in place before the next
 *                   stage; stage 4 takes the solution out of it; then ba_lm_trial_finish.  cam_param_mask as in ba_solve_reduced
 *                   (stage 2 reads it).  The status word of the solve is this rank's: non-zero on ANY rank = failed. */
int ba_dist_plan(int32_t nco, int32_t half_bandwidth, int32_t nranks, int32_t* cams_per_node, int32_t* nodes, int32_t* nodes_per_rank);
int ba_dist_enable(ba_handle* h, int32_t rank, int32_t nranks);
int ba_dist_info(ba_handle* h, int64_t* out, int32_t n);
int ba_dist_bind_exchange(ba_handle* h, void* exchange_dev, int64_t doubles);
int ba_dist_stage(ba_handle* h, int32_t stage, const uint8_t* cam_param_mask, int64_t* doubles_to_sum);

/* ---- Bundle.triangulate_all (bundle.py:313-321; triangulate.algebraic_lsq triangulate.py:6-18)
 * Re-initialise every point of parameter set `which` by linear least squares from its
 * observations and the set's cameras (the step before the path: test_bundle.py:175,
 * window_slam.py:82).  rcond: numpy.linalg.lstsq cut-off on the singular values of A
 * (< 0 = numpy's default; the kernel solves the 2L x 3 system by QR - Givens row updates - so a track
 * seen under little parallax keeps cond(A) * eps accuracy like lstsq's; a system that is rank deficient
 * at max(rcond, 1e-13) relative to the largest diagonal entry of R gets the minimum-norm point).
 * X[nt*3] (host) may be NULL: the result stays on the device. */
int ba_triangulate(ba_handle* h, int which, double rcond, double* X);

/* ---- instrumentation ---------------------------------------------------- */
int ba_enable_timing(ba_handle* h, int on);
/* bracket only the kernel ids whose bit is set (default: all); an event pair costs a few
 * microseconds of stream time, which matters when kernels are ~10 us long */
int ba_set_timing_mask(ba_handle* h, uint64_t kernel_id_mask);
/* bracket only every stride-th launch (run of launches) of each selected kernel: an event pair costs
 * a few microseconds of stream time, which shows in a 0.36 ms step */
int ba_set_timing_stride(ba_handle* h, int32_t stride);
/* accumulated HIP-event time (ms) and launch count per kernel id since the last reset */
int ba_get_timings(ba_handle* h, double* ms /*[BA_K_COUNT]*/, int64_t* launches /*[BA_K_COUNT]*/, int reset);
/* achievable HBM rate of this device: a streaming copy of `bytes` bytes, `repeats` times (read + write counted) */
int ba_measure_copy_bandwidth(ba_handle* h, int64_t bytes, int32_t repeats, double* gbytes_per_s);
const char* ba_kernel_name(int kernel_id);
const char* ba_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PYSFM_BA_H */
