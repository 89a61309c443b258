// ba_resident.h - the whole Levenberg-Marquardt loop of BundleAdjuster.optimize() (bundle_adjuster.py:117-162) for a SMALL
// problem as one resident launch: a sliding window of <= 16 optimised cameras and up to 1024 tracks
// (window_slam.py:17-48 runs one such problem per frame; the reference's own tests and its config-1 scenes are this size).
//
// At this size the six launches of ba_lm_trial cost 68 us per trial of which the kernels' own work is a fraction: every launch
// starts a grid for a thousand observations, the host synchronises to take the accept / reject decision, the next trial
// starts cold.  Here a small cluster of workgroups stays resident for the whole loop: workgroup g owns 16 points (their
// observations, W = Jc^T Jp in registers, their point blocks in LDS), every workgroup keeps both camera sets and takes the
// decisions of the reference's schedule itself; workgroup 0 leaves a log of (damping, outcome, cost) per trial that the Python
// side replays into costs / trial_log / num_steps.  No launch, no host synchronisation, no PCIe traffic between trials.
//
// One trial (p = point, j = its j-th observation, pos = optimised position of a camera):
//   1 linearise  lane group of 16 = one point, lane = observation: r, Jc, Jp (ba_math.h), W = Jc^T Jp kept in registers,
//                Jc | r to LDS for the camera blocks, HPP / bP by a DPP butterfly over the group.  SKIPPED after a rejected
//                trial: the parameters have not moved (prepare_schur_complement would recompute the same numbers,
//                bundle_adjuster.py:211-234).
//   2 damp+pinv  every lane of the group: (1 + damping) on the diagonal, pinv / inv of the 3 x 3 block, its L D L^T, U = W L into
//                the k-major staging array At[3 p + k][6 pos + a] (zeros where a camera does not see the point), D and
//                y = D L^T bP beside it                                                       (bundle_adjuster.py:238-256)
//   3 reduce     camera blocks HCC / bC of my points: thread = (pos, entry), points in index order; P_g = At^T diag(D) At on the
//                fp64 matrix cores, one wavefront per 16 x 16 tile of the upper triangle; At^T y with vector FMAs
//                                                                                                (bundle_adjuster.py:259-278)
//   exchange 1   every workgroup publishes its partial sums (relaxed agent-scope stores, one epoch word), waits for the others
//                and adds ALL partials in workgroup order: every workgroup holds the same [S | b], bit for bit.  From five
//                workgroups on in two stages: workgroup g adds up slice g of all records, everybody fetches the record of sums
//                (G^2 records through the fabric otherwise: 64 workgroups 110 -> 59 us per trial)
//   4 solve      S = damped HCC - P, b = bC - At^T y; Cholesky in steps of 12 columns with the diagonal-block and panel routines
//                of the cyclic reduction (ba_bcr_blocks.h), the updates of the next block of columns on the matrix cores.
//                Wavefront 0 is the critical path and nothing else: diagonal block (its INVERSE comes out of the same chain),
//                first tile of the panel, that tile's update of the next diagonal block straight from registers, next block;
//                the other three do the rest of the panel, its updates and the look-ahead beside the pivot chain.  The
//                right-hand side rides along as one more row; back-substitution in one wavefront, a block at a time through
//                the inverses of the diagonal blocks (no chain of dependent unknowns).  Every workgroup solves (redundantly: a
//                hand-over costs more than 10 us of one CU's time are worth).  A pivot <= 0 ends the run (exit reason 2):
//                the host repeats that trial through the general path, which solves it as the reference's gesv would
//                                                                                                (bundle_adjuster.py:281-312)
//   5 back-sub   dP = HPPinv (bP - sum W^T dC), the trial set = perturb(-dC), x - dP, and the cost of my points at the trial set
//                                                                                       (bundle_adjuster.py:316-343, 165-171)
//   exchange 2   the cost partials, one word per workgroup that IS its own flag (a NaN no computation produces says "not yet");
//                every workgroup adds them in workgroup order and takes the same decision.
// fp64 throughout; every sum has a fixed order: results are reproducible run to run and identical in every workgroup.
#pragma once

#include "ba_bcr_blocks.h"     // the diagonal-block and panel routines of the cyclic reduction's nodes

namespace ba {

constexpr int kResThreads = 256;                    // 4 wavefronts, one per SIMD: 512 VGPRs each (the loop's state stays in registers)
constexpr int kResG = 16;                           // lanes per point
constexpr int kResP = kResThreads / kResG;          // points per workgroup
constexpr int kResWaves = kResThreads / 64;
constexpr int kResK = 3 * kResP;                    // staged k rows
constexpr int kResMaxNco = 16;                      // optimised cameras: 96 unknowns
constexpr int kResMaxN = 6 * kResMaxNco;
constexpr int kResLd = kResMaxN;                    // staged row length (6 tiles)
constexpr int kResMaxTileRows = kResMaxN / 16;      // 16 x 16 tiles per side
constexpr int kResOwn = (kResMaxTileRows * (kResMaxTileRows + 1) / 2 + 3) / 4;      // upper tiles per wavefront at most: 6 of 21
constexpr int kResMaxNc = 32;
constexpr int kResMaxGroups = 64;                   // workgroups of a launch (a lane of the first wavefront polls each)
constexpr int kResMaxNt = kResP * kResMaxGroups;
constexpr int kResMaxL = kResG;
constexpr int kResSLd = kResMaxN + 1;               // row length of S in LDS (odd: rows fall on different banks)
constexpr int kResJcLd = 15;                        // Jc (12) | r (2), padded to an odd length
constexpr int kResSteps = kResMaxN / 12;               // block steps of the factorisation (12 columns each)
constexpr int kResMaxTrials = 1000;
// A workgroup's record in the exchange buffer, laid out by the THREAD that adds it up (element-major, so that a wavefront's
// load reads 512 contiguous bytes): thread t owns 28 doubles = its (at most six) accumulator tiles, 4 doubles each, and the pairs
// t and t + 256 of the "misc" block: right-hand side [96] | camera blocks [16 x 27] | scalars [8] ([0] cost of its points at
// the current set, [1] singular point blocks, [2] trial cost).  Tiles a problem does not have are neither stored nor loaded.
constexpr int kResRecPerThread = 4 * kResOwn + 4;
constexpr int kResRec = kResRecPerThread * 256;
constexpr int kResMiscRhs = 0, kResMiscCam = kResMaxN, kResMiscScal = kResMiscCam + 27 * kResMaxNco, kResMisc = kResMiscScal + 8;
static_assert(kResMisc <= 4 * 256, "two pairs of the misc block per thread");
__host__ __device__ constexpr int res_misc_at(int idx) {      // element e of thread t lives at e * 256 + t
  return (4 * kResOwn + 2 * ((idx >> 1) >= 256 ? 1 : 0) + (idx & 1)) * 256 + ((idx >> 1) & 255);
}
constexpr int kResMaxSpins = 1 << 22;
constexpr long long kResNotYet = 0x7FFA5A5A5A5A5A5All;      // a NaN no computation produces: the trial cost of a workgroup that is not there yet

enum { RES_DONE = 0, RES_LOG_FULL = 1, RES_NOT_POSITIVE_DEFINITE = 2, RES_SINGULAR_POINT = 3, RES_TIMED_OUT = 4 };

struct ResidentLog {
  int ntrials, nsteps, converged, in_step;
  int exit_reason, exit_info, accepted, have_cost0;
  double damping, cost0, cur_cost, pad;
  double trial_damping[kResMaxTrials];
  double trial_cost[kResMaxTrials];
  int trial_accepted[kResMaxTrials];
};

struct ResidentArgs {
  int nc, nt, nco, maxL, nobs, ngroups;
  const int* obs_cam;
  const double2* obs_z;
  const int* pt_off;
  const int* cam_opt_pos;
  const unsigned char* pt_opt;
  double K[9];
  Sensor sensor;
  const double* cams;           // the current set the launch is given (read only: a launch whose workgroups lose each other must leave it as it was)
  const double* X;
  double* stage;                // device memory, [nc x 12 | nt x 3]: the set the run ends on - the HOST copies it over the current set once every workgroup has reported the same ending
  int* group_exit;              // pinned host memory, [2 * ngroups]: every workgroup's (exit reason, trials walked)
  int fault_group;              // test aid (option resident_fault): this workgroup REPORTS a time-out whatever happened (-1: none)
  double* out;                  // pinned host memory: the set the run ends on, [nc x 12 | nt x 3] (what ba_get_params will be asked for)
  double* xb;                   // exchange buffer: ngroups records of kResRec doubles
  long long* epoch;             // [2 * ngroups]: what each workgroup has published (partial sums | trial cost)
  long long epoch0;             // epochs of this launch start above it (the words are never reset)
  double* cost_slots;           // [2][kResMaxGroups]: the workgroups' trial costs, by the parity of the trial; kResNotYet between uses
  int parity0;                  // parity of this launch's first trial
  int scatter_min;              // launches of at least this many workgroups add the records up in slices (exchange 1 in two stages)
  // the schedule
  int max_steps, max_trials, nsteps, in_step, converged;
  double damping, improvement_threshold, rcond, cur_cost;      // cur_cost < 0: not known yet
  int have_mask;                // camera parameters deleted from the solve (solve_motion_normal_eqns' param_mask, bundle_adjuster.py:290-299)
  unsigned char mask[kResMaxN]; // ... 1 = kept
  ResidentLog* log;             // pinned host memory
  long long* trace;             // optional: clock stamps at the phase boundaries of the first trials, 16 per trial (workgroup 0)
  double* dbg;                  // optional: [S | b] (lower triangle, 98 x 97) and dC (128) of the FIRST trial, for the parity tests
};

// LDS carve-up of a workgroup
struct ResidentLds {
  int cam, X, HPP, Hinv, Dk, red, misc, dC, fact, z, stage;     // offsets in doubles
  int flag_i, off_i, pos_i, stab_i, tab_b, opt_b, oc_b, mask_b;        // offsets in bytes
  size_t bytes;
};
__host__ __device__ inline ResidentLds resident_lds(int nc, int nco, int maxL) {
  ResidentLds l;
  int o = 0;
  l.cam = o; o += 2 * nc * 12;
  l.X = o; o += 2 * kResP * 3;
  l.HPP = o; o += kResP * 9 + 1;                  // HPP (6) | bP (3) per point, undamped
  l.Hinv = o; o += kResP * 9 + 1;                 // HPPinv (6) | HPPinv bP (3)
  l.Dk = o; o += 2 * kResK;                       // D | y
  l.red = o; o += kResWaves * 64 + 16;            // partial right-hand sides [waves][64], wavefront partials
  l.misc = o; o += kResMisc;                   // the summed right-hand side | camera blocks | scalars
  l.dC = o; o += 128;
  l.fact = o; o += kResSteps * 192 + kBcrIdtDoubles;      // the factorisation: the inverses of the diagonal blocks [step][16][12], identity table
  l.z = o; o += 2 * kResP * maxL;                 // the measurements of my points
  l.stage = o;
  const int at = kResK * kResLd + kResP * maxL * kResJcLd;
  const int sm = (kResMaxN + 2) * kResSLd;
  o += at > sm ? at : sm;
  o += o & 1;
  size_t b = (size_t)o * 8;
  l.flag_i = (int)b; b += 16;                     // status words of a trial
  l.off_i = (int)b; b += (size_t)(kResP + 1) * 4;
  l.pos_i = (int)b; b += (size_t)nc * 4;
  l.stab_i = (int)b; b += (size_t)4 * kResOwn * kResThreads * 4;      // where my accumulator entries go in S (res_s_entry)
  l.tab_b = (int)b; b += (size_t)kResP * nco;
  l.opt_b = (int)b; b += (size_t)kResP;
  l.oc_b = (int)b; b += (size_t)kResP * maxL;     // camera of an observation
  l.mask_b = (int)b; b += kResMaxN;               // kept camera parameters
  l.bytes = (b + 15) & ~(size_t)15;
  return l;
}

typedef double res_acc __attribute__((ext_vector_type(4)));

// what crosses workgroups inside the launch: relaxed agent-scope accesses (they bypass / write through the caches that are
// not coherent across the chip's eight L2s; no fences - ba_bcr.h has the long story)
__device__ __forceinline__ void res_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double res_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// a value every lane holds alike, said so to the compiler (scalar registers, uniform branches)
__device__ __forceinline__ int res_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double res_uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// my stores have been acknowledged -> one word says so
__device__ __forceinline__ void res_publish(long long* word, long long epoch, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  if (tid == 0) __hip_atomic_store(word, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every workgroup has published `epoch`: lane g of the first wavefront polls word g.  False: timed out.
__device__ __forceinline__ bool res_wait_all(const long long* words, int stride, int ngroups, long long epoch, int tid, int* timed_out) {
  if (tid < 64) {
    bool ok = tid >= ngroups;
    for (int spins = 0; !__all(ok); ++spins) {
      if (!ok) ok = __hip_atomic_load(words + (size_t)tid * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
      if (spins >= kResMaxSpins) { if (tid == 0) *timed_out = 1; break; }
      if (!__all(ok)) __builtin_amdgcn_s_sleep(1);
    }
  }
  lds_barrier();
  return res_uniform(*timed_out) == 0;
}

// Entry (row, col) of the upper triangle of S, as the thread that holds its sum in an accumulator tile needs it once a trial:
// where it goes in LDS (the lower triangle, [col][row]) | which entry of the camera blocks it starts from (diagonal blocks only)
// | is it damped | is its parameter kept (solve_motion_normal_eqns' mask) | is it on the diagonal | is it there at all.
// Packed once per launch: no division, no branch in the trial loop.
__device__ __forceinline__ unsigned res_s_entry(int row, int col, int n, bool mine, const unsigned char* __restrict__ maskL) {
  if (!mine || row > col || col >= n) return 0u;
  const int pr = row / 6, pc = col / 6, a = row - 6 * pr, b = col - 6 * pc;
  unsigned w = (unsigned)(col * kResSLd + row) | 1u << 27;
  if (pr == pc) w |= (unsigned)(pr * 27 + (a * (11 - a)) / 2 + b) << 14 | 1u << 23;      // entry (a, b >= a) of the camera's block, rows first
  if (row == col) w |= 1u << 24;
  if (maskL[row] && maskL[col]) w |= 1u << 25;
  return w;
}
static_assert((kResMaxN + 1) * kResSLd < (1 << 14) && kResMaxNco * 27 < (1 << 9), "res_s_entry's fields");

// Exchange 1 in two stages, stage one: KN elements per thread of my slice of the records, every record's in flight at once
// (eight records a batch), added in workgroup order; the sums go to the sum record.
template <int KN>
__device__ __forceinline__ void res_reduce_slice(const double* __restrict__ xb, double* __restrict__ sumrec, int G, const int (&off)[4],
                                                 const bool (&val)[4]) {
  double s[KN];
#pragma unroll
  for (int k = 0; k < KN; ++k) s[k] = 0.0;
  for (int g0 = 0; g0 < G; g0 += 8) {
    double v[8][KN];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double* q = xb + (size_t)min(g0 + u, G - 1) * kResRec;
#pragma unroll
      for (int k = 0; k < KN; ++k) v[u][k] = res_ld(q + off[k]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool in = g0 + u < G;
#pragma unroll
      for (int k = 0; k < KN; ++k) s[k] += in ? v[u][k] : 0.0;
    }
  }
#pragma unroll
  for (int k = 0; k < KN; ++k)
    if (val[k]) res_st(sumrec + off[k], s[k]);
}

// Columns [C0, C0 + NW) of the factorisation (NW = 12, or 6 at the end), row tile `tile` (rows C0 + 16 tile ..):
// S[row][C0 + c] -= sum over K0 <= m < K1 of L[row][m] L[C0 + c][m] on the matrix cores (a 16 x 16 tile of which NW columns are
// wanted).  Template parameters: the k steps are known to the compiler - no load under a condition, no branch.  The accumulator
// starts as the tile of S itself (its loads travel with the operands'), the products come off it: one LDS round trip.
template <int K0, int K1, int C0, int NW>
__device__ __forceinline__ void res_update_block(double* __restrict__ Sm, int tile, int n, int ln, int lk) {
  static_assert(K0 % 4 == 0 && K1 % 4 == 0, "whole k steps");
  constexpr int KS = (K1 - K0) / 4;
  const int r0 = C0 + 16 * tile;
  if (r0 > n) return;                             // (uniform: a wavefront takes one tile)
  const double* ar = Sm + min(r0 + ln, n) * kResSLd + K0 + lk;
  const double* br = Sm + min(C0 + ln, n) * kResSLd + K0 + lk;
  double av[KS], bv[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { av[ks] = ar[4 * ks]; bv[ks] = br[4 * ks]; }
  res_acc acc;
  bool wr[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = r0 + lk + 4 * v;
    wr[v] = ln < NW && row <= n && C0 + ln <= row;
    acc[v] = Sm[min(row, n) * kResSLd + C0 + min(ln, NW - 1)];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], -bv[ks], acc, 0, 0, 0);
#pragma unroll
  for (int v = 0; v < 4; ++v)
    if (wr[v]) Sm[(r0 + lk + 4 * v) * kResSLd + C0 + ln] = acc[v];
}
// step S of the factorisation (columns 12 S .. 12 S + 11): what the NEXT block of columns gets from the columns left of this
// step's (while this step's diagonal block is being factorised) ...
template <int S>
__device__ __forceinline__ void res_update_ahead(double* __restrict__ Sm, int tile, int n, int nw, int ln, int lk) {
  if constexpr (S >= 1) {
    if (nw == 12) res_update_block<0, 12 * S, 12 * S + 12, 12>(Sm, tile, n, ln, lk);
    else res_update_block<0, 12 * S, 12 * S + 12, 6>(Sm, tile, n, ln, lk);
  }
}
// ... and from this step's own panel
template <int S>
__device__ __forceinline__ void res_update_panel(double* __restrict__ Sm, int tile, int n, int nw, int ln, int lk) {
  if (nw == 12) res_update_block<12 * S, 12 * S + 12, 12 * S + 12, 12>(Sm, tile, n, ln, lk);
  else res_update_block<12 * S, 12 * S + 12, 12 * S + 12, 6>(Sm, tile, n, ln, lk);
}
// Wavefront 0 between two diagonal blocks (the cyclic reduction's bcr_prefetch_tile0 / bcr_urgent_tile0, rows clamped to the
// matrix): the tile that holds the next diagonal block, rows kn .. kn + 15, owes this step's panel C -= X X^T over the first 16
// rows X of the panel - the tile wavefront 0 has just computed and still holds in registers, in the layout of both operands.
__device__ __forceinline__ res_acc res_prefetch_tile0(const double* __restrict__ Sm, int n, int kn, int ln, int lk) {
  res_acc acc;
#pragma unroll
  for (int v = 0; v < 4; ++v) acc[v] = Sm[min(kn + lk + 4 * v, n) * kResSLd + kn + ln];
  return acc;
}

// L^T x = y in one wavefront, lane i holding the unknown i (TWO: and 64 + i), block by block from the last: what the factorisation
// left of a diagonal block is the INVERSE of its factor, so a block is x_k = Li^T (y_k - sum over later blocks) - twelve
// independent broadcasts, no chain of dependent unknowns - and then comes off the unknowns before it through the panel rows.
template <bool TWO, int NB>
__device__ __forceinline__ void res_backsolve_block(const double* __restrict__ Sm, const double* __restrict__ Li, int k0, int lane, int c1,
                                                    double& v0, double& v1, double& x0, double& x1) {
  const int i0 = lane - k0, i1 = 64 + lane - k0;                         // my position in the block, by slot
  const bool in0 = i0 >= 0 && i0 < NB, in1 = TWO && i1 >= 0 && i1 < NB;
  const int ic = in0 ? i0 : (in1 ? i1 : 0);
  double li[NB], r0[NB], r1[NB];
#pragma unroll
  for (int p = 0; p < NB; ++p) {
    li[p] = Li[p * 12 + ic];
    r0[p] = Sm[(k0 + p) * kResSLd + lane];
    r1[p] = TWO ? Sm[(k0 + p) * kResSLd + c1] : 0.0;
  }
  __builtin_amdgcn_sched_barrier(0);
  double xa = 0.0, xb = 0.0;
#pragma unroll
  for (int p = 0; p < NB; ++p) {
    const int g = k0 + p;
    const double tp = lane_bcast(!TWO || g < 64 ? v0 : v1, g & 63);
    if (p & 1) xb = fma(li[p], tp, xb); else xa = fma(li[p], tp, xa);
  }
  const double xs = in0 || in1 ? xa + xb : 0.0;
  x0 = in0 ? xs : x0;
  x1 = in1 ? xs : x1;
  const bool b0 = lane < k0, b1 = 64 + lane < k0;                        // unknowns before the block
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const double xq = lane_bcast(xs, (k0 + q) & 63);
    v0 = fma(b0 ? -r0[q] : 0.0, xq, v0);
    if (TWO) v1 = fma(b1 ? -r1[q] : 0.0, xq, v1);
  }
}
template <bool TWO>
__device__ __forceinline__ void res_backsolve(const double* __restrict__ Sm, const double* __restrict__ LiL, double* __restrict__ dCl, int n, int lane) {
  double v0 = lane < n ? Sm[n * kResSLd + lane] : 0.0;
  double v1 = TWO && 64 + lane < n ? Sm[n * kResSLd + 64 + lane] : 0.0;
  double x0 = 0.0, x1 = 0.0;
  const int c1 = min(64 + lane, kResMaxN - 1);
  int s = (n + 11) / 12 - 1;
  if (n - 12 * s < 12) {                              // (an odd number of cameras: the last block has one)
    res_backsolve_block<TWO, 6>(Sm, LiL + s * 192, 12 * s, lane, c1, v0, v1, x0, x1);
    --s;
  }
  for (; s >= 0; --s) res_backsolve_block<TWO, 12>(Sm, LiL + s * 192, 12 * s, lane, c1, v0, v1, x0, x1);
  if (lane < n) dCl[lane] = x0;
  if (TWO && 64 + lane < n) dCl[64 + lane] = x1;
}

#define RES_SWITCH_STEP(fn, ...)                                                              \
  switch (step) {                                                                             \
    case 0: fn<0>(__VA_ARGS__); break; case 1: fn<1>(__VA_ARGS__); break; case 2: fn<2>(__VA_ARGS__); break; \
    case 3: fn<3>(__VA_ARGS__); break; case 4: fn<4>(__VA_ARGS__); break; case 5: fn<5>(__VA_ARGS__); break; \
    default: fn<6>(__VA_ARGS__); break;                                                       \
  }

// TABLE: the sensor model may be a caller-defined robustifier sampled into a table (ba_math.h SENSOR_TABLE); its own instance,
// so that the closed-form models keep their registers
template <bool TABLE>
__global__ __launch_bounds__(kResThreads) void k_resident_lm(ResidentArgs A) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = blockIdx.x, G = A.ngroups;
  const int nc = A.nc, nt = A.nt, nco = A.nco, maxL = A.maxL, n = 6 * nco;
  const ResidentLds lo = resident_lds(nc, nco, maxL);
  double* camL = sm + lo.cam;
  double* XL = sm + lo.X;
  double* HPPl = sm + lo.HPP;
  double* Hinvl = sm + lo.Hinv;
  double* Dk = sm + lo.Dk;
  double* yk = Dk + kResK;
  double* red = sm + lo.red;
  double* dCl = sm + lo.dC;
  double* miscL = sm + lo.misc;
  double* LiL = sm + lo.fact;
  double* IdtL = LiL + kResSteps * 192;
  double* At = sm + lo.stage;
  double* JC = At + kResK * kResLd;
  double* Sm = sm + lo.stage;                      // aliases At / JC: used after the reduction only
  unsigned char* base = reinterpret_cast<unsigned char*>(sm);
  int* sflag = reinterpret_cast<int*>(base + lo.flag_i);      // [0]: pivot index of a failed factorisation, [1]: timed out
  int* offL = reinterpret_cast<int*>(base + lo.off_i);
  int* posL = reinterpret_cast<int*>(base + lo.pos_i);
  unsigned* stabL = reinterpret_cast<unsigned*>(base + lo.stab_i);
  signed char* tabj = reinterpret_cast<signed char*>(base + lo.tab_b);
  unsigned char* optL = base + lo.opt_b;
  unsigned char* ocL = base + lo.oc_b;
  unsigned char* maskL = base + lo.mask_b;
  double2* zL = reinterpret_cast<double2*>(sm + lo.z);

  // ---- once per launch: the cameras, my points and their observations, the (point, position) -> observation table
  const int p0 = grp * kResP, np = min(kResP, nt - p0);        // my points [p0, p0 + np)
  const int ob0 = A.pt_off[p0], nob = A.pt_off[p0 + np] - ob0;
  int cur = 0;                                     // which of the two LDS parameter sets is the current one
  for (int i = tid; i < nc * 12; i += kResThreads) camL[i] = A.cams[i];
  for (int i = tid; i < np * 3; i += kResThreads) XL[i] = A.X[(size_t)p0 * 3 + i];
  for (int i = tid; i <= np; i += kResThreads) offL[i] = A.pt_off[p0 + i] - ob0;
  for (int i = tid; i < nc; i += kResThreads) posL[i] = A.cam_opt_pos[i];
  for (int i = tid; i < np; i += kResThreads) optL[i] = A.pt_opt[p0 + i];
  for (int i = tid; i < kResP * nco; i += kResThreads) tabj[i] = -1;
  for (int i = tid; i < nob; i += kResThreads) { ocL[i] = (unsigned char)A.obs_cam[ob0 + i]; zL[i] = A.obs_z[ob0 + i]; }
  if (tid == 0) { sflag[0] = 0; sflag[1] = 0; }
  if (tid < kResMaxN) maskL[tid] = A.have_mask ? A.mask[tid] : 1;
  for (int i = tid; i < kResSteps * 192; i += kResThreads) LiL[i] = 0.0;      // (rows 12 .. 15 stay zero; a 6-column block leaves the rest of its own alone)
  bcr_identity_table(IdtL, tid);
  lds_barrier();
  for (int p = tid; p < np; p += kResThreads)
    for (int q = offL[p]; q < offL[p + 1]; ++q) {
      const int pos = posL[ocL[q]];
      if (pos >= 0) tabj[p * nco + pos] = (signed char)(q - offL[p]);
    }
  lds_barrier();

  // the schedule's state: identical in every thread of every workgroup (all decisions are taken on sums every workgroup
  // forms from the same numbers in the same order)
  double damping = A.damping, cur_cost = A.cur_cost;
  int nsteps = A.nsteps, in_step = A.in_step, converged = A.converged, ntrials = 0, accepted_any = 0;
  int exit_reason = RES_DONE, exit_info = 0;
  bool need_lin = true, have_cost0 = false;
  double cost0 = 0.0;
  long long epoch = A.epoch0;

  // my observation: lane group of 16 = point slot pl, lane j = the point's j-th observation
  const int pl = tid >> 4, j = tid & 15;
  const bool pv = pl < np;
  const int ob = pv ? offL[pl] : 0, L = pv ? offL[pl + 1] - ob : 0;
  const bool ov = j < L;
  const int oi = ob + j;
  const int ocam = ov ? ocL[oi] : 0;
  const int pos = ov ? posL[ocam] : -1;
  const bool popt = pv && optL[pl] != 0;
  double2 z = {0.0, 0.0};
  if (ov) z = zL[oi];
  double W[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) W[i] = 0.0;

  const int NT = (n + 15) >> 4, ntiles = NT * (NT + 1) / 2;
  const int ln = lane & 15, lk = lane >> 4;
  int tti[kResOwn], ttj[kResOwn];                   // my tiles: wave, wave + 4, wave + 8, ...
  bool own[kResOwn];
#pragma unroll
  for (int t = 0; t < kResOwn; ++t) {
    tti[t] = ttj[t] = 0;
    own[t] = wave + kResWaves * t < ntiles;
    if (own[t]) tri_decode(wave + kResWaves * t, NT, tti[t], ttj[t]);
  }
  const bool more_tiles = ntiles > 3 * kResWaves;      // (uniform over the launch: beyond 10 cameras)
#pragma unroll
  for (int t = 0; t < kResOwn; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) stabL[(4 * t + v) * kResThreads + tid] = res_s_entry(16 * tti[t] + lk + 4 * v, 16 * ttj[t] + ln, n, own[t], maskL);
  double* myrec = A.xb + (size_t)grp * kResRec;

  int tr_i = 0;
#define RES_STAMP(k) do { if (A.trace && grp == 0 && tid == 0 && tr_i < 64) A.trace[tr_i * 16 + (k)] = (long long)wall_clock64(); } while (0)
  for (;;) {
    // ---- BundleAdjuster.optimize / step (bundle_adjuster.py:117-162)
    if (!in_step) {
      if (converged || nsteps >= A.max_steps) break;
      ++nsteps;
      in_step = 1;
    }
    if (converged || !(damping < 1e+8)) { in_step = 0; continue; }      // the inner loop of step() ends
    if (ntrials >= A.max_trials) { exit_reason = RES_LOG_FULL; break; }

    RES_STAMP(0);
    if (A.trace && grp == 0 && tid == 0 && tr_i < 64) A.trace[tr_i * 16 + 10] = (long long)clock64();
    const double* cm_cur = camL + cur * nc * 12;
    const double* X_cur = XL + cur * kResP * 3;
    double* cm_tr = camL + (1 - cur) * nc * 12;
    double* X_tr = XL + (1 - cur) * kResP * 3;
    const double dampf = 1.0 + damping;
    const double ldl_tol = sym3_ldl_tolerance(A.rcond);
    const double x[3] = {pv ? X_cur[3 * pl] : 0.0, pv ? X_cur[3 * pl + 1] : 0.0, pv ? X_cur[3 * pl + 2] : 0.0};

    double hpp[6], bp[3];
    double lin_cost = 0.0;
    if (need_lin) {
      double r[2] = {0.0, 0.0}, Jc[12], Jp[6];
#pragma unroll
      for (int i = 0; i < 12; ++i) Jc[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) Jp[i] = 0.0;
      if (ov) {
        double cm[12], e[2];
#pragma unroll
        for (int i = 0; i < 12; ++i) cm[i] = cm_cur[ocam * 12 + i];
        obs_linearize<TABLE>(A.K, cm, x, z.x, z.y, A.sensor, e, r, Jc, Jp);
        if (pos >= 0 && popt) lin_cost = r[0] * r[0] + r[1] * r[1];
        if (pos >= 0) {
          double* jq = JC + (pl * maxL + j) * kResJcLd;
#pragma unroll
          for (int i = 0; i < 12; ++i) jq[i] = Jc[i];
          jq[12] = r[0]; jq[13] = r[1];
        }
      }
      block_W(Jc, Jp, W);
      hpp[0] = Jp[0] * Jp[0] + Jp[3] * Jp[3];
      hpp[1] = Jp[0] * Jp[1] + Jp[3] * Jp[4];
      hpp[2] = Jp[0] * Jp[2] + Jp[3] * Jp[5];
      hpp[3] = Jp[1] * Jp[1] + Jp[4] * Jp[4];
      hpp[4] = Jp[1] * Jp[2] + Jp[4] * Jp[5];
      hpp[5] = Jp[2] * Jp[2] + Jp[5] * Jp[5];
      bp[0] = Jp[0] * r[0] + Jp[3] * r[1];
      bp[1] = Jp[1] * r[0] + Jp[4] * r[1];
      bp[2] = Jp[2] * r[0] + Jp[5] * r[1];
#pragma unroll
      for (int i = 0; i < 6; ++i) hpp[i] = group_sum<16>(hpp[i]);
#pragma unroll
      for (int i = 0; i < 3; ++i) bp[i] = group_sum<16>(bp[i]);
      if (pv && j == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) HPPl[pl * 9 + i] = hpp[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) HPPl[pl * 9 + 6 + i] = bp[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) hpp[i] = pv ? HPPl[pl * 9 + i] : 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) bp[i] = pv ? HPPl[pl * 9 + 6 + i] : 0.0;
    }
    // ---- damping, inverse, its factors (apply_damping, pinv: bundle_adjuster.py:238-256)
    double Hi[6], Dd[3], Lo[3];
    hpp[0] *= dampf; hpp[3] *= dampf; hpp[5] *= dampf;
    bool singular = false;
    if (A.rcond >= 0.0) sym3_pinv_fast(hpp, A.rcond, Hi);
    else singular = !sym3_inv(hpp, Hi) && pv;
    sym3_ldl(Hi, ldl_tol, Dd, Lo);
    if (j == 0) {
      if (pv) {
        double tp[3];
        sym3_apply(Hi, bp, tp);
#pragma unroll
        for (int i = 0; i < 6; ++i) Hinvl[pl * 9 + i] = Hi[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) Hinvl[pl * 9 + 6 + i] = tp[i];
      }
      const double y0 = bp[0] + Lo[0] * bp[1] + Lo[1] * bp[2], y1 = bp[1] + Lo[2] * bp[2], y2 = bp[2];
      Dk[3 * pl] = pv ? Dd[0] : 0.0; Dk[3 * pl + 1] = pv ? Dd[1] : 0.0; Dk[3 * pl + 2] = pv ? Dd[2] : 0.0;
      yk[3 * pl] = pv ? Dd[0] * y0 : 0.0; yk[3 * pl + 1] = pv ? Dd[1] * y1 : 0.0; yk[3 * pl + 2] = pv ? Dd[2] * y2 : 0.0;
    }
    // U = W L of my observation; zeros for the position j of this point if no camera there sees it
    if (pos >= 0) {
      double* q = At + (3 * pl) * kResLd + 6 * pos;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        q[a] = W[a * 3] + Lo[0] * W[a * 3 + 1] + Lo[1] * W[a * 3 + 2];
        q[kResLd + a] = W[a * 3 + 1] + Lo[2] * W[a * 3 + 2];
        q[2 * kResLd + a] = W[a * 3 + 2];
      }
    }
    if (j < nco && (!pv || tabj[pl * nco + j] < 0)) {
      double* q = At + (3 * pl) * kResLd + 6 * j;
#pragma unroll
      for (int a = 0; a < 6; ++a) { q[a] = 0.0; q[kResLd + a] = 0.0; q[2 * kResLd + a] = 0.0; }
    }
    const int nsing = __syncthreads_count(singular && j == 0);      // (a full barrier: the LDS writes above are visible below)
    RES_STAMP(1);

    // ---- camera blocks of my points (bundle_adjuster.py:230, 233), in index order: thread = (pos, entry of HCC | bC)
    if (need_lin) {
      for (int ct = tid; ct < nco * 27; ct += kResThreads) {
        const int cpos = ct / 27, cent = ct - cpos * 27;
        int ca = 0, cb = 0;                              // entry -> (a, b) of the upper triangle, or (a, -) of bC
        if (cent < 21) { int e = cent; while (e >= 6 - ca) { e -= 6 - ca; ++ca; } cb = ca + e; } else { ca = cent - 21; cb = -1; }
        double acc = 0.0;
        {
          constexpr int q0 = 0;
          double v[kResP];
#pragma unroll
          for (int u = 0; u < kResP; ++u) {
            const int q = q0 + u;
            const int jr = tabj[q * nco + cpos];                 // (rows of point slots beyond my last point hold -1)
            const int jj = q < np ? jr : -1;
            const double* jq = JC + (q * maxL + max(jj, 0)) * kResJcLd;
            const double t = cb >= 0 ? jq[ca] * jq[cb] + jq[6 + ca] * jq[6 + cb] : jq[ca] * jq[12] + jq[6 + ca] * jq[13];
            v[u] = jj < 0 ? 0.0 : t;
          }
#pragma unroll
          for (int u = 0; u < kResP; ++u) acc += v[u];
        }
        res_st(myrec + res_misc_at(kResMiscCam + ct), acc);
      }
    }
    // ---- P_g = At^T diag(D) At on the matrix cores, At^T y beside it
    {
      // (all kResK rows every time: rows of point slots beyond my last point are zeros)
      constexpr int KS = kResK / 4;
      double dkv[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) dkv[ks] = Dk[4 * ks + lk];
#pragma unroll
      for (int t = 0; t < kResOwn; ++t) {
        if (!own[t]) continue;
        const double* a0 = At + 16 * tti[t] + ln + lk * kResLd;
        const double* b0 = At + 16 * ttj[t] + ln + lk * kResLd;
        double av[KS], bv[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { av[ks] = a0[4 * ks * kResLd]; bv[ks] = b0[4 * ks * kResLd]; }
        res_acc acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks] * dkv[ks], acc, 0, 0, 0);
        // my record: tiles in the accumulator layout [tile][lane][4]
        double* q = myrec + (4 * t) * 256 + tid;
        // (plain agent-scope stores: the compiler knows the wait states between a matrix-core result and a store that reads it;
        //  inside an asm statement it does not - rows 4..7 x columns 12..15 of every tile came out stale)
#pragma unroll
        for (int v = 0; v < 4; ++v) res_st(q + v * 256, acc[v]);
      }
      double rhs_part = 0.0;
      if (n <= 64) {                                  // (uniform) four slices of the k rows ...
        const int r = tid & 63, sl = tid >> 6;
        if (r < n) {
#pragma unroll
          for (int q = 0; q < kResK / 4; ++q) {
            const int k = sl * (kResK / 4) + q;
            rhs_part = fma(At[k * kResLd + r], yk[k], rhs_part);
          }
        }
        red[sl * 64 + r] = rhs_part;
      } else {                                        // ... or two
        const int r = tid & 127, sl = tid >> 7;
        if (r < n) {
#pragma unroll
          for (int q = 0; q < kResK / 2; ++q) {
            const int k = sl * (kResK / 2) + q;
            rhs_part = fma(At[k * kResLd + r], yk[k], rhs_part);
          }
        }
        red[sl * 128 + r] = rhs_part;
      }
    }
    if (need_lin) {
      const double c = wave_sum(lin_cost);
      if (lane == 0) red[kResWaves * 64 + wave] = c;
    }
    lds_barrier();
    if (tid < n) res_st(myrec + res_misc_at(kResMiscRhs + tid), n <= 64 ? (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]) : red[tid] + red[128 + tid]);
    if (tid == 128) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < kResWaves; ++w) s += red[kResWaves * 64 + w];
      res_st(myrec + res_misc_at(kResMiscScal), need_lin ? s : 0.0);
      res_st(myrec + res_misc_at(kResMiscScal + 1), (double)nsing);
    }
    RES_STAMP(2);
    // ---- exchange 1: everybody's partial sums, added in workgroup order
    ++epoch;
    res_publish(A.epoch + 2 * grp, epoch, tid);
    if (!res_wait_all(A.epoch, 2, G, epoch, tid, sflag + 1)) { exit_reason = RES_TIMED_OUT; break; }
    RES_STAMP(3);
    // (everybody is in this trial: nobody reads the last trial's cost words any more - mine goes back to "not yet")
    const int par = (A.parity0 + ntrials) & 1;
    if (tid == 0) __hip_atomic_store(reinterpret_cast<long long*>(A.cost_slots) + (1 - par) * kResMaxGroups + grp, kResNotYet, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
      // tiles: C[row = 16 ti + lk + 4 v][col = 16 tj + ln], ti <= tj: entry (col, row) of the lower triangle of S
      res_acc ssum[kResOwn];
#pragma unroll
      for (int t = 0; t < kResOwn; ++t) ssum[t] = res_acc{0.0, 0.0, 0.0, 0.0};
      double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;  // pairs `tid` and `tid + 256` of the right-hand side | camera blocks | scalars
      const bool mine2 = 2 * (tid + 256) < kResMisc;
      const double* mybase = A.xb + tid;
      if (G >= A.scatter_min) {                       // (uniform over the launch)
        // Many workgroups: a record per workgroup read by every workgroup is G^2 records through the fabric.  Two stages
        // instead - workgroup g adds up slice g of all records (in workgroup order, as below: the same bits), publishes, and
        // everybody fetches the one record of sums.
        const int NR = (more_tiles ? 4 * kResOwn : 12) + 4;      // rows of 256 elements in use: tiles | the misc block
        const int E = NR * 256;
        const int chunk = ((E + G - 1) / G + 63) & ~63;
        const int e0 = grp * chunk, e1 = min(e0 + chunk, E);
        double* sumrec = A.xb + (size_t)G * kResRec;
        for (int eb = e0; eb < e1; eb += 4 * 256) {
          int off[4];
          bool val[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int idx = eb + 256 * k + tid;
            val[k] = idx < e1;
            const int ic = min(idx, E - 1), r = ic >> 8;
            off[k] = (r < NR - 4 ? r : 4 * kResOwn + r - (NR - 4)) * 256 + (ic & 255);
          }
          const int kn = (min(e1 - eb, 4 * 256) + 255) >> 8;
          if (kn == 1) res_reduce_slice<1>(A.xb, sumrec, G, off, val);
          else if (kn == 2) res_reduce_slice<2>(A.xb, sumrec, G, off, val);
          else if (kn == 3) res_reduce_slice<3>(A.xb, sumrec, G, off, val);
          else res_reduce_slice<4>(A.xb, sumrec, G, off, val);
        }
        RES_STAMP(9);
        res_publish(A.epoch + 2 * grp + 1, epoch, tid);
        if (!res_wait_all(A.epoch + 1, 2, G, epoch, tid, sflag + 1)) { exit_reason = RES_TIMED_OUT; break; }
        RES_STAMP(12);
        const double* q = sumrec + tid;
#pragma unroll
        for (int t = 0; t < kResOwn; ++t) {
          if (t >= 3 && !more_tiles) break;            // (uniform)
#pragma unroll
          for (int e = 0; e < 4; ++e) ssum[t][e] = res_ld(q + (4 * t + e) * 256);      // (tiles that are not mine: never used)
        }
        m0 = res_ld(q + (4 * kResOwn) * 256); m1 = res_ld(q + (4 * kResOwn + 1) * 256);
        if (mine2) { m2 = res_ld(q + (4 * kResOwn + 2) * 256); m3 = res_ld(q + (4 * kResOwn + 3) * 256); }
      } else {
        for (int g0 = 0; g0 < G; g0 += 2) {
          // (at most two workgroups here) both records in flight before the first use (compiler-visible agent-scope loads: it
          // counts the waits itself); more would not fit the registers
          double v[2][16];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const double* q = mybase + (size_t)min(g0 + u, G - 1) * kResRec;
#pragma unroll
            for (int e = 0; e < 12; ++e) v[u][e] = res_ld(q + e * 256);
            v[u][12] = res_ld(q + (4 * kResOwn) * 256); v[u][13] = res_ld(q + (4 * kResOwn + 1) * 256);
            v[u][14] = v[u][15] = 0.0;
            if (mine2) { v[u][14] = res_ld(q + (4 * kResOwn + 2) * 256); v[u][15] = res_ld(q + (4 * kResOwn + 3) * 256); }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const bool in = g0 + u < G;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
              for (int e = 0; e < 4; ++e) ssum[t][e] += in && own[t] ? v[u][4 * t + e] : 0.0;
            m0 += in ? v[u][12] : 0.0;
            m1 += in ? v[u][13] : 0.0;
            m2 += in && mine2 ? v[u][14] : 0.0;
            m3 += in && mine2 ? v[u][15] : 0.0;
          }
        }
        if (more_tiles) {                               // more than 10 cameras: my tiles 3 .. 5
          for (int g0 = 0; g0 < G; g0 += 2) {
            double v[2][12];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const double* q = mybase + (size_t)min(g0 + u, G - 1) * kResRec + 12 * 256;
#pragma unroll
              for (int e = 0; e < 12; ++e) v[u][e] = res_ld(q + e * 256);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const bool in = g0 + u < G;
#pragma unroll
              for (int t = 3; t < kResOwn; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) ssum[t][e] += in && own[t] ? v[u][4 * (t - 3) + e] : 0.0;
            }
          }
        }
      }
      miscL[2 * tid] = m0; miscL[2 * tid + 1] = m1;
      RES_STAMP(13);
      if (mine2) { miscL[2 * (tid + 256)] = m2; miscL[2 * (tid + 256) + 1] = m3; }
      lds_barrier();
      const double c0sum = miscL[kResMiscScal];
      if (res_uniform(miscL[kResMiscScal + 1]) > 0.0) { exit_reason = RES_SINGULAR_POINT; break; }      // plain-inverse mode: the general path raises (bundle_adjuster.py:254)
      if (need_lin && !have_cost0) {
        cost0 = res_uniform(c0sum);
        have_cost0 = true;
        if (cur_cost < 0.0) cur_cost = cost0;
      }
      // S = damped camera blocks - sums, b = bC - sums (the camera blocks of a trial that did not linearise are the records' own
      // of the trial before: the same numbers).  Sm aliases At: everybody has been through a barrier since the last read of it.
      const double rsum = tid < kResMaxN ? miscL[tid] : 0.0;
      const double bc = tid < n ? miscL[kResMiscCam + (tid / 6) * 27 + 21 + tid % 6] : 0.0;
      {
        unsigned w[12];
        double hv[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) w[e] = stabL[e * kResThreads + tid];
#pragma unroll
        for (int e = 0; e < 12; ++e) hv[e] = miscL[kResMiscCam + ((w[e] >> 14) & 511)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 12; ++e) {
          double h = w[e] & 1u << 23 ? hv[e] : 0.0;
          h = w[e] & 1u << 24 ? h * dampf : h;
          // a deleted parameter keeps an identity row / column and a zero right-hand side: its update is zero
          const double val = w[e] & 1u << 25 ? h - ssum[e >> 2][e & 3] : (w[e] & 1u << 24 ? 1.0 : 0.0);
          if (w[e] & 1u << 27) Sm[w[e] & 16383] = val;
        }
      }
      if (more_tiles) {
        unsigned w[4 * kResOwn - 12];
        double hv[4 * kResOwn - 12];
#pragma unroll
        for (int e = 12; e < 4 * kResOwn; ++e) w[e - 12] = stabL[e * kResThreads + tid];
#pragma unroll
        for (int e = 12; e < 4 * kResOwn; ++e) hv[e - 12] = miscL[kResMiscCam + ((w[e - 12] >> 14) & 511)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 12; e < 4 * kResOwn; ++e) {
          const unsigned x = w[e - 12];
          double h = x & 1u << 23 ? hv[e - 12] : 0.0;
          h = x & 1u << 24 ? h * dampf : h;
          const double val = x & 1u << 25 ? h - ssum[e >> 2][e & 3] : (x & 1u << 24 ? 1.0 : 0.0);
          if (x & 1u << 27) Sm[x & 16383] = val;
        }
      }
      if (tid < n) Sm[n * kResSLd + tid] = maskL[tid] ? bc - rsum : 0.0;
    }
    lds_barrier();
    if (A.dbg && grp == 0 && ntrials == 0)
      for (int i = tid; i < (n + 1) * kResSLd; i += kResThreads) A.dbg[i] = Sm[i];
    RES_STAMP(4);

    // ---- Cholesky in steps of 12 columns (two cameras; 6 at the end when the cameras are odd); row n = the right-hand side
    //      (becomes L^-1 b).  The diagonal block and the panel below it are the cyclic reduction's (ba_bcr_blocks.h): the block
    //      is factorised in one wavefront, a lane per column in every 16-lane row, every broadcast of the pivot chain fused
    //      into its v_fmac_f64 (DPP row_newbcast) - 85 cycles per pivot, and the lanes left over carry the identity along, so
    //      that the block's inverse comes out too; the panel is X = A L^-T on the matrix cores.  While wavefront 0 runs a
    //      block's chain the others give the next block of columns what the columns left of this step owe it.
#define RES_CSTAMP(k) do { if (A.trace && grp == 0 && tid == 0 && ntrials == 3) A.trace[(32 + step) * 16 + (k)] = (long long)clock64(); } while (0)
    //      Wavefront 0 owns the critical path - diagonal block, the first tile of the panel below it (which holds the rows of the
    //      NEXT diagonal block), that tile's update - and goes straight on to the next diagonal block; the other three do the rest
    //      of the panel, its update of the next block of columns and the look-ahead while that block's chain runs: two barriers
    //      a step.
    for (int step = 0; 12 * step < n; ++step) {
      const int k0 = 12 * step, nbw = min(12, n - k0), kn = k0 + nbw;      // this step's columns [k0, kn), the next block from kn
      const int nwn = min(12, n - kn);                                      // ... of nwn columns (0: this is the last step)
      double* LiS = LiL + step * 192;
      RES_CSTAMP(0);
      if (wave == 0) {
        if (nbw == 12) bcr_diag_block<12, false>(Sm, kResSLd, nullptr, sflag, k0, lane, LiS, IdtL);
        else bcr_diag_block<6, false>(Sm, kResSLd, nullptr, sflag, k0, lane, LiS, IdtL);
      } else if (nwn > 0) {
        for (int tile = wave - 1; kn + 16 * tile <= n; tile += kResWaves - 1) RES_SWITCH_STEP(res_update_ahead, Sm, tile, n, nwn, ln, lk);
      }
      RES_CSTAMP(1);
      lds_barrier();
      if (res_uniform(sflag[0])) break;
      // the panel: rows kn .. n (the right-hand side among them), a tile of 16 per wavefront.  Wavefront 0 takes the first and
      // gives the next diagonal block what it owes this panel straight from its registers
      if (wave == 0) {
        res_acc acc = {0.0, 0.0, 0.0, 0.0};
        if (nwn > 0) acc = res_prefetch_tile0(Sm, n, kn, ln, lk);
        double pr[3];
        bcr_panel_tile(Sm, kResSLd, n + 1, k0, kn, LiS, ln, lk, pr, nbw);
        if (nwn > 0) bcr_urgent_tile0(Sm, kResSLd, n + 1, kn, nwn, ln, lk, pr, acc);
      } else {
        for (int i0 = kn + 16 * wave; i0 <= n; i0 += 16 * (kResWaves - 1)) {
          double pr[3];
          bcr_panel_tile(Sm, kResSLd, n + 1, k0, i0, LiS, ln, lk, pr, nbw);
        }
      }
      RES_CSTAMP(2);
      lds_barrier();
      // the tiles below it in the other wavefronts, before their look-ahead of the next step - wavefront 0 is on the next
      // diagonal block by then
      if (nwn > 0 && wave != 0)
        for (int tile = wave; kn + 16 * tile <= n; tile += kResWaves - 1) RES_SWITCH_STEP(res_update_panel, Sm, tile, n, nwn, ln, lk);
      RES_CSTAMP(3);
    }
    lds_barrier();
    if (res_uniform(sflag[0])) { exit_reason = RES_NOT_POSITIVE_DEFINITE; exit_info = res_uniform(sflag[0]); break; }
    RES_STAMP(5);
    // ---- L^T x = y in one wavefront (res_backsolve)
    if (wave == 0) {
      if (n <= 64) res_backsolve<false>(Sm, LiL, dCl, n, lane);
      else res_backsolve<true>(Sm, LiL, dCl, n, lane);
    }
    lds_barrier();
    if (A.dbg && grp == 0 && ntrials == 0 && tid < 128) A.dbg[(kResMaxN + 2) * kResSLd + tid] = tid < n ? dCl[tid] : 0.0;
    RES_STAMP(6);

    // ---- the trial set: cameras (update_motion with the sign of compute_update, bundle_adjuster.py:203-208, 334-337)
    for (int c = tid; c < nc; c += kResThreads) {
      const int cp = posL[c];
      double cm[12], out[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) cm[q] = cm_cur[c * 12 + q];
      if (cp >= 0) {
        double d[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) d[q] = -dCl[cp * 6 + q];
        camera_perturb(cm, d, out);
      } else {
#pragma unroll
        for (int q = 0; q < 12; ++q) out[q] = cm[q];
      }
#pragma unroll
      for (int q = 0; q < 12; ++q) cm_tr[c * 12 + q] = out[q];
    }
    lds_barrier();
    // ---- back-substitution, the trial points and the trial cost (bundle_adjuster.py:316-331, 340-343, 165-171)
    double tc = 0.0;
    {
      double v0 = 0.0, v1 = 0.0, v2 = 0.0;
      if (pos >= 0) {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double d = dCl[pos * 6 + a];
          v0 = fma(W[a * 3], d, v0);
          v1 = fma(W[a * 3 + 1], d, v1);
          v2 = fma(W[a * 3 + 2], d, v2);
        }
      }
      v0 = group_sum<16>(v0); v1 = group_sum<16>(v1); v2 = group_sum<16>(v2);
      if (pv) {
        const double* hq = Hinvl + pl * 9;
        const double* bq = HPPl + pl * 9 + 6;
        const double w[3] = {bq[0] - v0, bq[1] - v1, bq[2] - v2};
        const double Hq[6] = {hq[0], hq[1], hq[2], hq[3], hq[4], hq[5]};
        double dp[3];
        sym3_apply(Hq, w, dp);
        const double xt[3] = {popt ? x[0] - dp[0] : x[0], popt ? x[1] - dp[1] : x[1], popt ? x[2] - dp[2] : x[2]};
        if (j == 0) { X_tr[3 * pl] = xt[0]; X_tr[3 * pl + 1] = xt[1]; X_tr[3 * pl + 2] = xt[2]; }
        if (pos >= 0 && popt) {
          double cm[12], e[2], r[2];
#pragma unroll
          for (int q = 0; q < 12; ++q) cm[q] = cm_tr[ocam * 12 + q];
          obs_residual<TABLE>(A.K, cm, xt, z.x, z.y, A.sensor, e, r);
          tc = r[0] * r[0] + r[1] * r[1];
        }
      }
    }
    tc = wave_sum(tc);
    if (lane == 0) red[kResWaves * 64 + wave] = tc;
    lds_barrier();
    RES_STAMP(7);
    // ---- exchange 2: the cost of the trial set.  One word per workgroup, which is its own flag.
    if (tid < 64) {
      long long* slots = reinterpret_cast<long long*>(A.cost_slots) + par * kResMaxGroups;
      if (tid == 0) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kResWaves; ++w) s += red[kResWaves * 64 + w];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the word went back to "not yet" long ago: now for certain)
        __hip_atomic_store(slots + grp, __double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      long long bits = 0;
      bool ok = tid >= G;
      for (int spins = 0; !__all(ok); ++spins) {
        if (!ok) { bits = __hip_atomic_load(slots + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = bits != kResNotYet; }
        if (spins >= kResMaxSpins) { if (tid == 0) sflag[1] = 1; break; }
        if (!__all(ok)) __builtin_amdgcn_s_sleep(1);
      }
      if (tid < G) miscL[tid] = __longlong_as_double(bits);
    }
    lds_barrier();
    if (res_uniform(sflag[1])) { exit_reason = RES_TIMED_OUT; break; }
    double next_cost = 0.0;
    for (int g = 0; g < G; ++g) next_cost += miscL[g];
    next_cost = res_uniform(next_cost);
    // ---- accept / reject and the damping schedule (bundle_adjuster.py:136-157)
    const bool accept = next_cost < cur_cost;
    if (grp == 0 && tid == 0) {
      A.log->trial_damping[ntrials] = damping;
      A.log->trial_cost[ntrials] = next_cost;
      A.log->trial_accepted[ntrials] = accept ? 1 : 0;
    }
    RES_STAMP(8);
    if (A.trace && grp == 0 && tid == 0 && tr_i < 64) A.trace[tr_i * 16 + 11] = (long long)clock64();
    if (A.trace && grp == 0 && tid == 0 && tr_i < 64) A.trace[tr_i * 16 + 15] = need_lin ? 1 : 0;
    ++tr_i;
    ++ntrials;
    if (accept) {
      damping *= 0.1;
      converged = fabs(cur_cost - next_cost) < A.improvement_threshold;
      cur_cost = next_cost;
      cur = 1 - cur;
      need_lin = true;
      accepted_any = 1;
      in_step = 0;
    } else {
      damping *= 10.0;
      converged = damping > 1e+8;
      need_lin = false;
    }
  }

  // ---- the current set back to device memory (my points; workgroup 0: the cameras), the state of the schedule to the log
  {
    const double* cm_cur = camL + cur * nc * 12;
    const double* X_cur = XL + cur * kResP * 3;
    // Never over the set the launch was given: workgroups that lost each other need not agree on how far they came (one passes its
    // last wait and ends DONE while another's spin budget runs out on the same epoch), and a set updated by some of them only would
    // corrupt the optimisation silently.  Every workgroup writes its slice to a staging copy and says how it ended; the host
    // commits the copy when all of them ended alike (ba_lm_resident_end).
    if (grp == 0)
      for (int i = tid; i < nc * 12; i += kResThreads) { A.stage[i] = cm_cur[i]; A.out[i] = cm_cur[i]; }
    for (int i = tid; i < np * 3; i += kResThreads) {
      A.stage[(size_t)nc * 12 + (size_t)p0 * 3 + i] = X_cur[i];
      A.out[(size_t)nc * 12 + (size_t)p0 * 3 + i] = X_cur[i];
    }
    if (tid == 0) { A.group_exit[2 * grp] = grp == A.fault_group ? (int)RES_TIMED_OUT : exit_reason; A.group_exit[2 * grp + 1] = ntrials; }
  }
  if (grp == 0 && tid == 0) {
    ResidentLog* g = A.log;
    g->ntrials = ntrials; g->nsteps = nsteps; g->converged = converged; g->in_step = in_step;
    g->exit_reason = exit_reason; g->exit_info = exit_info; g->accepted = accepted_any; g->have_cost0 = have_cost0 ? 1 : 0;
    g->damping = damping; g->cost0 = cost0; g->cur_cost = cur_cost;
  }
}

#undef RES_STAMP
#undef RES_CSTAMP
#undef RES_SWITCH_STEP

}  // namespace ba
