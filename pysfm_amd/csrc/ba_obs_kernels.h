// ba_obs_kernels.h - the per-observation / per-point kernels of the bundle-adjustment inner loop (cost, evaluation,
// linearisation, point-block inversion, back-substitution, parameter update, triangulation).  gfx950 (MI355X, CDNA4).
//
//
// Everything here is fp64.  The individual blocks are no larger than 6x6; the per-observation
// kernels are vector code bound by fp64 issue and gather latency, and the one step that becomes a
// real matrix product once points are grouped - the Schur reduction over points that share their
// cameras - runs on the fp64 matrix cores (k_schur_groups_mfma; the dense nodes of the reduced solve
// do the same in ba_bcr.h).  Layout rules used throughout:
//   * observations are a structure of arrays sorted by point (CSR `pt_off`), so a
//     wavefront reads `obs_cam` / `obs_z` as contiguous, coalesced runs;
//   * a camera is one 96-byte record [R | t] and is gathered (L1/L2 resident:
//     1000 cameras = 96 KB); a point is read once per track;
//   * per-observation 2x6 / 2x3 Jacobian blocks and W = Jc^T Jp live in registers
//     and are RECOMPUTED in the Schur and back-substitution kernels instead of being
//     written to and re-read from HBM (24 B/obs of input instead of 144 B/obs);
//   * 64-lane wavefronts everywhere: a power-of-two group of lanes owns one point
//     and reduces with cross-lane shuffles; the Schur kernel gives one wavefront a
//     tile of a point's track staged in LDS.
#pragma once

#include "ba_device.h"
#include "ba_point_blocks.h"

namespace ba {

// --------------------------------------------------------------------------
// compute_cost (bundle_adjuster.py:165-171): one observation per lane,
// wavefront + block reduction, one partial per block.  The partials go straight into a
// pinned host record (with the two status words of the trial) and the CPU adds them in
// index order after synchronising: deterministic, no second launch, no copy kernels.
// --------------------------------------------------------------------------

__global__ __launch_bounds__(kBlock) void k_cost(DevProblem P, const double* __restrict__ cams,
                                                 const double* __restrict__ X,
                                                 const int* __restrict__ singular_points,
                                                 const int* __restrict__ solve_info, HostResult* __restrict__ host,
                                                 double* __restrict__ dev_result) {
  __shared__ double wsum[kBlock / kWave];
  double acc = 0.0;
  const long long stride = (long long)gridDim.x * kBlock;
  for (long long n = (long long)blockIdx.x * kBlock + threadIdx.x; n < P.nobs; n += stride) {
    const int c = P.obs_cam[n];
    const int k = P.obs_pt[n];
    if (P.cam_opt_pos[c] < 0 || !P.pt_opt[k]) continue;
    const double2 z = P.obs_z[n];
    double cm[12], e[2], r[2];
    load_cam(cams, c, cm);
    const double x[3] = {X[3 * (size_t)k], X[3 * (size_t)k + 1], X[3 * (size_t)k + 2]};
    obs_residual<true>(P.K, cm, x, z.x, z.y, P.sensor, e, r);
    acc += r[0] * r[0] + r[1] * r[1];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) s += wsum[w];
    host->partial[blockIdx.x] = s;
    if (dev_result) dev_result[blockIdx.x] = s;             // sharded adjuster: the ranks' costs meet on the device
    if (blockIdx.x == 0) {
      host->singular_points = *singular_points;
      host->solve_info = *solve_info;
      if (dev_result) { dev_result[kCostBlocks] = (double)*singular_points; dev_result[kCostBlocks + 1] = trial_status_word(*solve_info); }
    }
  }
  if (dev_result && blockIdx.x == 0)                     // the sharded adjuster sums ALL kCostBlocks entries
    for (int i = gridDim.x + threadIdx.x; i < kCostBlocks; i += kBlock) dev_result[i] = 0.0;
}

// --------------------------------------------------------------------------
// Bundle.reproj_error / residual / Jresidual for every observation
// (bundle.py:243-277) - the per-observation API and parity probe.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_eval(DevProblem P, const double* __restrict__ cams,
                                                 const double* __restrict__ X, double* __restrict__ oe,
                                                 double* __restrict__ orr, double* __restrict__ oJc,
                                                 double* __restrict__ oJp) {
  const long long n = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (n >= P.nobs) return;
  const int c = P.obs_cam[n];
  const int k = P.obs_pt[n];
  const double2 z = P.obs_z[n];
  double cm[12], e[2], r[2], Jc[12], Jp[6];
  load_cam(cams, c, cm);
  const double x[3] = {X[3 * (size_t)k], X[3 * (size_t)k + 1], X[3 * (size_t)k + 2]};
  obs_linearize<true>(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
  if (oe) { oe[2 * n] = e[0]; oe[2 * n + 1] = e[1]; }
  if (orr) { orr[2 * n] = r[0]; orr[2 * n + 1] = r[1]; }
  if (oJc) {
#pragma unroll
    for (int i = 0; i < 12; ++i) oJc[12 * n + i] = Jc[i];
  }
  if (oJp) {
#pragma unroll
    for (int i = 0; i < 6; ++i) oJp[6 * n + i] = Jp[i];
  }
}

// sensor_model.residual_from_error / Jresidual_from_error on a batch (sensor_model.py:19-32)
__global__ __launch_bounds__(kBlock) void k_eval_sensor(Sensor s, long long n, const double* __restrict__ e,
                                                        double* __restrict__ r, double* __restrict__ J) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double rr[2], JJ[4];
  sensor_eval<true>(s, e[2 * i], e[2 * i + 1], rr, JJ);
  if (r) { r[2 * i] = rr[0]; r[2 * i + 1] = rr[1]; }
  if (J) {
#pragma unroll
    for (int q = 0; q < 4; ++q) J[4 * i + q] = JJ[q];
  }
}

// --------------------------------------------------------------------------
// prepare_schur_complement (bundle_adjuster.py:211-234).
// A group of G = 2^glog lanes owns one point: lane l takes observations
// s+l, s+l+G, ...  HPP / bP are reduced across the group with shuffles and
// written once (deterministic).  W is written only on request.  The camera blocks
// HCC / bC are produced by k_camera_blocks below (a second, camera-ordered pass over
// the observations) because 1000 observations per camera hammering 27 addresses with
// atomics is ~50x slower than re-reading 24 bytes per observation.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_linearize(DevProblem P, const double* __restrict__ cams,
                                                      const double* __restrict__ X, int glog,
                                                      double* __restrict__ HCC, double* __restrict__ bC,
                                                      double* __restrict__ HPP, double* __restrict__ bP,
                                                      double* __restrict__ Wout, double damping, double rcond,
                                                      double* __restrict__ HPPinv, int* __restrict__ singular_count,
                                                      int* __restrict__ next_count) {
  const int G = 1 << glog;
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  // HCC / bC are accumulated with atomics by k_camera_blocks, which runs next: clear them here
  // (saves two memset launches of ~5 us each on the trial's critical path)
  if (HCC) {
    const long long nthreads = (long long)gridDim.x * kBlock;
    for (long long i = tid; i < (long long)P.nc * 36; i += nthreads) HCC[i] = 0.0;
    for (long long i = tid; i < (long long)P.nc * 6; i += nthreads) bC[i] = 0.0;
  }
  const long long k = tid >> glog;
  const int l = (int)(tid & (G - 1));
  const bool valid = k < P.nt;
  int s = 0, e_ = 0;
  double x[3] = {0, 0, 0};
  if (valid) {
    s = P.pt_off[k]; e_ = P.pt_off[k + 1];
    x[0] = X[3 * k]; x[1] = X[3 * k + 1]; x[2] = X[3 * k + 2];
  }
  double hpp[6] = {0, 0, 0, 0, 0, 0}, bp[3] = {0, 0, 0};
  for (int n = s + l; n < e_; n += G) {
    const int c = P.obs_cam[n];
    const double2 z = P.obs_z[n];
    double cm[12], e[2], r[2], Jc[12], Jp[6];
    load_cam(cams, c, cm);
    obs_linearize<true>(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
    hpp[0] += Jp[0] * Jp[0] + Jp[3] * Jp[3];
    hpp[1] += Jp[0] * Jp[1] + Jp[3] * Jp[4];
    hpp[2] += Jp[0] * Jp[2] + Jp[3] * Jp[5];
    hpp[3] += Jp[1] * Jp[1] + Jp[4] * Jp[4];
    hpp[4] += Jp[1] * Jp[2] + Jp[4] * Jp[5];
    hpp[5] += Jp[2] * Jp[2] + Jp[5] * Jp[5];
    bp[0] += Jp[0] * r[0] + Jp[3] * r[1];
    bp[1] += Jp[1] * r[0] + Jp[4] * r[1];
    bp[2] += Jp[2] * r[0] + Jp[5] * r[1];
    if (Wout) {
      double W[18];
      block_W(Jc, Jp, W);
#pragma unroll
      for (int i = 0; i < 18; ++i) Wout[(size_t)n * 18 + i] = W[i];
    }
  }
  for (int m = G >> 1; m >= 1; m >>= 1) {
#pragma unroll
    for (int i = 0; i < 6; ++i) hpp[i] += __shfl_xor(hpp[i], m, 64);
#pragma unroll
    for (int i = 0; i < 3; ++i) bp[i] += __shfl_xor(bp[i], m, 64);
  }
  if (HPPinv && tid == 0) *next_count = 0;       // as k_point_invert: the counter the NEXT inversion will use
  if (valid && l == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) HPP[6 * k + i] = hpp[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) bP[3 * k + i] = bp[i];
    if (HPPinv) {                                  // ba_lm_trial: k_point_invert's work rides along (one launch less)
      double out[6];
      const double f = 1.0 + damping;
      hpp[0] *= f; hpp[3] *= f; hpp[5] *= f;
      if (rcond >= 0.0) {
        sym3_pinv_fast(hpp, rcond, out);
      } else if (!sym3_inv(hpp, out)) {
        atomicAdd(singular_count, 1);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) HPPinv[6 * k + i] = out[i];
    }
  }
}

// --------------------------------------------------------------------------
// HCC[i] += Jc^T Jc, bC[i] += Jc^T r (bundle_adjuster.py:230,233), camera-ordered:
// one wavefront per (camera, chunk of <= kCamChunk of its observations) walks the
// camera's observation list `perm` (observation ids sorted by camera), accumulates the
// 21 + 6 sums in registers, reduces across the 64 lanes and adds ONE result per unit.
// --------------------------------------------------------------------------

__global__ __launch_bounds__(kBlock) void k_camera_blocks(DevProblem P, const double* __restrict__ cams,
                                                          const double* __restrict__ X,
                                                          const int* __restrict__ perm,
                                                          const CamUnit* __restrict__ units, int nunits,
                                                          double* __restrict__ HCC, double* __restrict__ bC) {
  const int u = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (u >= nunits) return;                       // whole wavefront
  const CamUnit un = units[u];
  double cm[12];
  load_cam(cams, un.cam, cm);
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.0;
  // Four observations of a lane at a time, every level of the chain perm -> (obs_pt, obs_z) -> X fetched for all four before the
  // next: three trips to memory per FOUR observations (one observation at a time it was three per observation - 32 dependent
  // iterations of a 2048-observation unit: 59 us for two million observations; round 5).  Same order of additions per lane.
  constexpr int U = 4;
  for (int q0 = un.begin + lane; q0 < un.end; q0 += 64 * U) {
    int n[U], k[U];
    double2 z[U];
    double x[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) n[u] = perm[q0 + 64 * u < un.end ? q0 + 64 * u : q0];
#pragma unroll
    for (int u = 0; u < U; ++u) { k[u] = P.obs_pt[n[u]]; z[u] = P.obs_z[n[u]]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { x[u][0] = X[3 * (size_t)k[u]]; x[u][1] = X[3 * (size_t)k[u] + 1]; x[u][2] = X[3 * (size_t)k[u] + 2]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (q0 + 64 * u < un.end) {
        double e[2], r[2], Jc[12], Jp[6];
        obs_linearize<true>(P.K, cm, x[u], z[u].x, z[u].y, P.sensor, e, r, Jc, Jp);
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int b = a; b < 6; ++b) acc[idx++] += Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += Jc[a] * r[0] + Jc[6 + a] * r[1];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = wave_sum(acc[i]);
  if (lane == 0) {
    double* hc = HCC + (size_t)un.cam * 36;
    int idx = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = a; b < 6; ++b) atomic_add_f64(hc + a * 6 + b, acc[idx++]);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) atomic_add_f64(bC + (size_t)un.cam * 6 + a, acc[21 + a]);
  }
}

// --------------------------------------------------------------------------
// backsubstitute (bundle_adjuster.py:316-331):
//   dP_k = HPPinv_k (bP_k - sum_i W_ik^T dC_i),  W^T dC = Jp^T (Jc dC).
// dC[nco*6] is indexed by optimised-camera position; frozen cameras contribute nothing.
// With cams_dst / X_dst given (ba_lm_trial) the kernel also writes the trial parameters
// R exp(sign dC), t + sign dt, x + sign dP (k_apply_update's work, one launch less).
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_backsub(DevProblem P, const double* __restrict__ cams,
                                                    const double* __restrict__ X, int glog,
                                                    const double* __restrict__ dC,
                                                    const double* __restrict__ HPPinv,
                                                    const double* __restrict__ bP, double* __restrict__ dP,
                                                    double sign, double* __restrict__ cams_dst,
                                                    double* __restrict__ X_dst) {
  const int G = 1 << glog;
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  // fused update_motion (bundle_adjuster.py:334-337): dC is complete before this kernel starts
  if (cams_dst) {
    const long long nthreads = (long long)gridDim.x * kBlock;
    for (long long i = tid; i < P.nc; i += nthreads) {
      double cm[12], out[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) cm[q] = cams[(size_t)i * 12 + q];
      const int pos = P.cam_opt_pos[i];
      if (pos >= 0) {
        double d[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) d[q] = sign * dC[(size_t)pos * 6 + q];
        camera_perturb(cm, d, out);
      } else {
#pragma unroll
        for (int q = 0; q < 12; ++q) out[q] = cm[q];
      }
#pragma unroll
      for (int q = 0; q < 12; ++q) cams_dst[(size_t)i * 12 + q] = out[q];
    }
  }
  const long long k = tid >> glog;
  const int l = (int)(tid & (G - 1));
  const bool valid = k < P.nt;
  int s = 0, e_ = 0;
  double x[3] = {0, 0, 0};
  if (valid) {
    s = P.pt_off[k]; e_ = P.pt_off[k + 1];
    x[0] = X[3 * k]; x[1] = X[3 * k + 1]; x[2] = X[3 * k + 2];
  }
  double acc[3] = {0, 0, 0};
  for (int n = s + l; n < e_; n += G) {
    const int c = P.obs_cam[n];
    const int pos = P.cam_opt_pos[c];
    if (pos < 0) continue;
    const double2 z = P.obs_z[n];
    double cm[12], e[2], r[2], Jc[12], Jp[6];
    load_cam(cams, c, cm);
    obs_linearize<true>(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
    const double* d = dC + (size_t)pos * 6;
    double v0 = 0.0, v1 = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) { v0 += Jc[a] * d[a]; v1 += Jc[6 + a] * d[a]; }
    acc[0] += Jp[0] * v0 + Jp[3] * v1;
    acc[1] += Jp[1] * v0 + Jp[4] * v1;
    acc[2] += Jp[2] * v0 + Jp[5] * v1;
  }
  for (int m = G >> 1; m >= 1; m >>= 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] += __shfl_xor(acc[i], m, 64);
  }
  if (valid && l == 0) {
    double A[6], v[3], out[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) A[i] = HPPinv[6 * k + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = bP[3 * k + i] - acc[i];
    sym3_apply(A, v, out);
#pragma unroll
    for (int i = 0; i < 3; ++i) dP[3 * k + i] = out[i];
    if (X_dst) {                      // fused update_structure (bundle_adjuster.py:340-343)
      const bool opt = P.pt_opt[k] != 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) X_dst[3 * k + i] = opt ? x[i] + sign * out[i] : x[i];
    }
  }
}

// --------------------------------------------------------------------------
// k_linearize / k_backsub for scenes whose points come in runs with identical camera lists (the groups
// of k_schur_groups: <= kGroupMaxPts points, L <= kPtGroupMaxL cameras).  The lanes-per-point kernels
// above give a point a power-of-two lane group (16 lanes for 10 observations: 10 of 16 busy); here
// one wavefront owns a group and lane = (point slot, observation): 64 / L points at a time, 60 of 64
// lanes busy at L = 10, the camera of a lane loaded once per group.  The per-point sums (9 values in
// k_linearize, 3 in k_backsub) go through LDS: every lane writes its terms, one lane per (point,
// value) adds the point's L entries in index order - deterministic, like the shuffle tree it replaces.
// --------------------------------------------------------------------------
// sum of the L <= 24 consecutive LDS values at p, in index order (deterministic): twelve per LDS round trip
__device__ __forceinline__ double lds_sum_in_order(const double* p, int L) {
  double v[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) v[j] = j < L ? p[j] : 0.0;
  double sum = 0.0;
#pragma unroll
  for (int j = 0; j < 12; ++j) sum += v[j];
  if (L > 12) {                                          // wave-uniform
#pragma unroll
    for (int j = 0; j < 12; ++j) v[j] = 12 + j < L ? p[12 + j] : 0.0;
#pragma unroll
    for (int j = 0; j < 12; ++j) sum += v[j];
  }
  return sum;
}
// Row length of the per-value LDS rows: lane (slot, oi) adds up value c = oi of its point from row c, so the lanes of a point read
// nine rows at the same offset - with rows of 64 doubles (512 bytes) all nine fall on the same banks (SQ_LDS_BANK_CONFLICT: 74 % of the
// kernel's LDS-active cycles, profiles/r05e_sq_counters.csv).  65: the rows two banks apart - measured 0.5 us of the lineariser's 20, within the noise
// of the trial (an LDS pass of nine values is a small part of a kernel that linearises 60 observations per wavefront and batch).
#ifndef BA_LIN_ROW
#define BA_LIN_ROW 65
#endif
// What a trial adds to its linearisation (ba_lm_trial with a matrix-core reduction, `inv.HPPinv` set): the point blocks are
// damped, inverted and factorised as soon as their sums exist - the lane that holds a point's first observation does what a
// lane of k_point_invert does, from LDS instead of from memory - and the workgroups behind the groups' clear [S | b]
// (schur_init_body; the camera blocks come out of the reduction): k_point_invert_schur_init's launch, 8 us at config 3, is gone.
struct FusedInvert {
  double damping, rcond;
  double* HPPinv;              // null: the linearisation alone
  double* fac;
  int* singular_count;
  int* next_count;
  int group_blocks;            // workgroups [0, group_blocks) linearise, the rest initialise [S | b]
  int n1, hb1;
  double* S;
  double* b;
};
template <bool FUSED>
__device__ __forceinline__ void linearize_groups_body(double (*sx)[9][BA_LIN_ROW], double (*si)[9][kGroupMaxPts + 1], int (*srange)[2], const DevProblem& P, const double* __restrict__ cams,
                                                      const double* __restrict__ X,
                                                      const SchurGroup* __restrict__ groups, int ngroups,
                                                      double* __restrict__ HCC, double* __restrict__ bC,
                                                      double* __restrict__ HPP, double* __restrict__ bP, const FusedInvert& inv) {
  if (FUSED && (int)blockIdx.x >= inv.group_blocks) {
    schur_init_body((long long)(blockIdx.x - inv.group_blocks) * kBlock + threadIdx.x, inv.n1, inv.hb1, nullptr, nullptr, nullptr, inv.damping,
                    inv.S, inv.b, 0);
    return;
  }
  if (FUSED && blockIdx.x == 0 && threadIdx.x == 0) *inv.next_count = 0;      // (the counter the NEXT inversion will use: point_invert_body)
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (HCC) {                                           // as k_linearize: the camera-block kernels accumulate with atomics
    const long long nthreads = (long long)(FUSED ? inv.group_blocks : gridDim.x) * kBlock;      // (the workgroups that are here)
    for (long long i = tid; i < (long long)P.nc * 36; i += nthreads) HCC[i] = 0.0;
    for (long long i = tid; i < (long long)P.nc * 6; i += nthreads) bC[i] = 0.0;
  }
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = blockIdx.x * (kBlock / kWave) + wv;
  if (g >= ngroups) {                                  // whole wavefront
    if constexpr (!FUSED) return;                      // (FUSED: an empty group - the inversion at the end is the workgroup's)
  }
  const SchurGroup gr = g < ngroups ? groups[g] : SchurGroup{0, 0, 1, 0};
  if (FUSED && lane == 0) { srange[wv][0] = gr.pt_begin; srange[wv][1] = gr.pt_end; }
  const int L = gr.L, NP = 64 / L;
  const int slot = lane / L, oi = lane - slot * L;
  const bool stager = lane < NP * L;
  const int n0 = P.pt_off[gr.pt_begin] + oi;
  double cm[12];
  load_cam(cams, P.obs_cam[stager ? n0 : P.pt_off[gr.pt_begin]], cm);
  double (*mx)[BA_LIN_ROW] = sx[wv];
  double (*mi)[kGroupMaxPts + 1] = si ? si[wv] : nullptr;
  for (int kb = gr.pt_begin; kb < gr.pt_end; kb += NP) {
    const int k = kb + slot;
    const bool live = stager && k < gr.pt_end;
    double loc[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) loc[c] = 0.0;
    if (live) {
      const double2 z = P.obs_z[n0 + (size_t)(k - gr.pt_begin) * L];
      const double x[3] = {X[3 * (size_t)k], X[3 * (size_t)k + 1], X[3 * (size_t)k + 2]};
      double e[2], r[2], Jc[12], Jp[6];
      obs_linearize(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
      loc[0] = Jp[0] * Jp[0] + Jp[3] * Jp[3]; loc[1] = Jp[0] * Jp[1] + Jp[3] * Jp[4]; loc[2] = Jp[0] * Jp[2] + Jp[3] * Jp[5];
      loc[3] = Jp[1] * Jp[1] + Jp[4] * Jp[4]; loc[4] = Jp[1] * Jp[2] + Jp[4] * Jp[5]; loc[5] = Jp[2] * Jp[2] + Jp[5] * Jp[5];
      loc[6] = Jp[0] * r[0] + Jp[3] * r[1]; loc[7] = Jp[1] * r[0] + Jp[4] * r[1]; loc[8] = Jp[2] * r[0] + Jp[5] * r[1];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) mx[c][lane] = loc[c];
    lds_wave_sync();
    if (live) {
      for (int c = oi; c < 9; c += L) {
        const double sum = lds_sum_in_order(&mx[c][slot * L], L);
        if (c < 6) HPP[6 * (size_t)k + c] = sum; else bP[3 * (size_t)k + c - 6] = sum;
        if (FUSED) mi[c][k - gr.pt_begin] = sum;
      }
    }
    lds_wave_sync();
  }
  if constexpr (FUSED) {
    // The point blocks of the workgroup's eight groups, one per thread (a group has at most kGroupMaxPts = 24 points): inverting
    // each batch's six points as they came kept 6 of 64 lanes busy with the heaviest part of the kernel (37 us against 20 + 9.6
    // for the two launches), each wavefront its own group's 24 of 64 lanes (28 us)
    __syncthreads();
    const int w = threadIdx.x / kGroupMaxPts, j = threadIdx.x - w * kGroupMaxPts;
    if (w < kBlock / kWave) {
      const int t = srange[w][0] + j;
      if (t < srange[w][1]) {
        double A[6], gv[3];
#pragma unroll
        for (int c = 0; c < 6; ++c) A[c] = si[w][c][j];
#pragma unroll
        for (int c = 0; c < 3; ++c) gv[c] = si[w][6 + c][j];
        point_invert_values((size_t)t, A, gv, inv.damping, inv.rcond, inv.HPPinv, inv.singular_count, inv.fac);
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_linearize_groups(DevProblem P, const double* __restrict__ cams,
                                                             const double* __restrict__ X,
                                                             const SchurGroup* __restrict__ groups, int ngroups,
                                                             double* __restrict__ HCC, double* __restrict__ bC,
                                                             double* __restrict__ HPP, double* __restrict__ bP) {
  __shared__ double sx[kBlock / kWave][9][BA_LIN_ROW];
  linearize_groups_body<false>(sx, nullptr, nullptr, P, cams, X, groups, ngroups, HCC, bC, HPP, bP, FusedInvert{});
}
// (four wavefronts a SIMD - all 4400 of config 3 at once: inlined, the inversion takes the kernel from 97 registers to 164;
//  held at 128 it spills some of them around the inversion)
__global__ __launch_bounds__(kBlock) void k_linearize_groups_trial(
    DevProblem P, const double* __restrict__ cams, const double* __restrict__ X, const SchurGroup* __restrict__ groups, int ngroups,
    double* __restrict__ HCC, double* __restrict__ bC, double* __restrict__ HPP, double* __restrict__ bP, FusedInvert inv) {
  __shared__ double sx[kBlock / kWave][9][BA_LIN_ROW];
  __shared__ double si[kBlock / kWave][9][kGroupMaxPts + 1];     // the sums of the groups' points, until the workgroup inverts them
  __shared__ int srange[kBlock / kWave][2];                      // the points of each wavefront's group
  linearize_groups_body<true>(sx, si, srange, P, cams, X, groups, ngroups, HCC, bC, HPP, bP, inv);
}

// With `host` given (ba_lm_trial: the trial parameter set is written here as well) the kernel also
// evaluates compute_cost of the TRIAL set (bundle_adjuster.py:165-171) - k_cost's work: the lanes of a
// point hold its observations and the old camera; the updated camera R exp(sign dC), t + sign dt is
// formed once per group per lane, the updated point comes from the point's first lane through LDS.
// Partials and status words go where k_cost puts them (one partial per workgroup, <= kCostBlocks).
#ifndef BA_BS_ROW
#define BA_BS_ROW 65
#endif
#ifndef BA_BACKSUB_WAVES
#define BA_BACKSUB_WAVES 3      // 168 VGPRs, 12 bytes of scratch per lane; with the next batch's inputs in flight: 27.5 us at config 3 (4 waves: 164 bytes of scratch per lane, 45 us; 2 waves: 35 us; without the prefetch 4 waves were best: 31.2 us)
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(BA_BACKSUB_WAVES, BA_BACKSUB_WAVES))) void k_backsub_groups(DevProblem P, const double* __restrict__ cams,
                                                           const double* __restrict__ X,
                                                           const SchurGroup* __restrict__ groups, int ngroups,
                                                           const double* __restrict__ dC,
                                                           const double* __restrict__ HPPinv,
                                                           const double* __restrict__ bP, double* __restrict__ dP,
                                                           double sign, double* __restrict__ cams_dst,
                                                           double* __restrict__ X_dst,
                                                           const int* __restrict__ singular_points,
                                                           const int* __restrict__ solve_info, HostResult* __restrict__ host,
                                                           double* __restrict__ dev_result) {
  __shared__ double sx[kBlock / kWave][3][BA_BS_ROW], sw[kBlock / kWave][3][BA_BS_ROW], wsum[kBlock / kWave];
  __shared__ double spx[kBlock / kWave][kGroupMaxPts][4];          // the group's updated points (x, y, z, optimised?) for the cost pass
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (cams_dst) {                                      // fused update_motion, as in k_backsub
    const long long nthreads = (long long)gridDim.x * kBlock;
    for (long long i = tid; i < P.nc; i += nthreads) {
      double cm[12], out[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) cm[q] = cams[(size_t)i * 12 + q];
      const int pos = P.cam_opt_pos[i];
      if (pos >= 0) {
        double d[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) d[q] = sign * dC[(size_t)pos * 6 + q];
        camera_perturb(cm, d, out);
      } else {
#pragma unroll
        for (int q = 0; q < 12; ++q) out[q] = cm[q];
      }
#pragma unroll
      for (int q = 0; q < 12; ++q) cams_dst[(size_t)i * 12 + q] = out[q];
    }
  }
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  double (*mx)[BA_BS_ROW] = sx[wv];
  double (*mw)[BA_BS_ROW] = sw[wv];
  const bool want_cost = host != nullptr && X_dst != nullptr;
  double cost_acc = 0.0;
#ifdef BA_BCR_PROFILE
  long long st[12]; int ns = 0;
#define BS_STAMP() do { if (ns < 12) st[ns++] = clock64(); } while (0)
  long long fs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BS_FINE(q) do { asm volatile("" ::: "memory"); fs[q] = clock64(); asm volatile("" ::: "memory"); } while (0)
  BS_STAMP();
#else
#define BS_STAMP()
#define BS_FINE(q)
#endif
  for (int g = blockIdx.x * (kBlock / kWave) + wv; g < ngroups; g += gridDim.x * (kBlock / kWave)) {   // wave-uniform
    const SchurGroup gr = groups[g];
    const int L = gr.L, NP = 64 / L;
    const int slot = lane / L, oi = lane - slot * L;
    const bool stager = lane < NP * L;
    const int n0 = P.pt_off[gr.pt_begin] + oi;
    const int c = P.obs_cam[stager ? n0 : P.pt_off[gr.pt_begin]];
    const int pos = stager ? P.cam_opt_pos[c] : -1;
    double cm[12], d[6];
    load_cam(cams, c, cm);
#pragma unroll
    for (int a = 0; a < 6; ++a) d[a] = pos >= 0 ? dC[(size_t)pos * 6 + a] : 0.0;
    double (*px)[4] = spx[wv];
    // the inputs of the NEXT batch are in flight while this one is worked on (a wavefront walks its group batch by batch,
    // about one wavefront per SIMD: without this every batch pays a trip to memory); what the end of a batch needs -
    // bP, HPPinv - is asked for at its beginning
    struct PointIn { double2 z; double x[3]; };
    auto fetch = [&](int kb_, PointIn& in) {
      const int k = kb_ + slot;
      if (stager && k < gr.pt_end) {
        in.z = P.obs_z[n0 + (size_t)(k - gr.pt_begin) * L];
#pragma unroll
        for (int q = 0; q < 3; ++q) in.x[q] = X[3 * (size_t)k + q];
      }
    };
    PointIn nxt;
    fetch(gr.pt_begin, nxt);
#ifdef BA_BCR_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    BS_STAMP();
    for (int kb = gr.pt_begin; kb < gr.pt_end; kb += NP) {
      const int k = kb + slot;
      const bool live = stager && k < gr.pt_end;
      BS_FINE(0);
      const PointIn cur = nxt;
      fetch(kb + NP, nxt);
      const double bpv = (live && oi < 3) ? bP[3 * (size_t)k + oi] : 0.0;
      double A[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) A[i] = (live && oi == 0) ? HPPinv[6 * (size_t)k + i] : 0.0;
      double x[3] = {0, 0, 0}, loc[3] = {0, 0, 0};
      if (live) {
        x[0] = cur.x[0]; x[1] = cur.x[1]; x[2] = cur.x[2];
        if (pos >= 0) {                                  // frozen cameras contribute nothing (bundle_adjuster.py:316-331)
          const double2 z = cur.z;
          double e[2], r[2], Jc[12], Jp[6];
          obs_linearize(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
          double v0 = 0.0, v1 = 0.0;
#pragma unroll
          for (int a = 0; a < 6; ++a) { v0 += Jc[a] * d[a]; v1 += Jc[6 + a] * d[a]; }
          loc[0] = Jp[0] * v0 + Jp[3] * v1;
          loc[1] = Jp[1] * v0 + Jp[4] * v1;
          loc[2] = Jp[2] * v0 + Jp[5] * v1;
        }
      }
      BS_FINE(1);
#pragma unroll
      for (int q = 0; q < 3; ++q) mx[q][lane] = loc[q];
      lds_wave_sync();
      BS_FINE(2);
      if (live) {                                        // lane q of a point adds component q of its L terms
        for (int q = oi; q < 3; q += L) {
          const double sum = lds_sum_in_order(&mx[q][slot * L], L);
          mw[q][slot] = (q == oi ? bpv : bP[3 * (size_t)k + q]) - sum;      // (L < 3: a lane adds more than one component)
        }
      }
      lds_wave_sync();
      BS_FINE(3);
      if (live && oi == 0) {
        double v[3], out[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) v[i] = mw[i][slot];
        sym3_apply(A, v, out);
#pragma unroll
        for (int i = 0; i < 3; ++i) dP[3 * (size_t)k + i] = out[i];
        if (X_dst) {                                     // fused update_structure (bundle_adjuster.py:340-343)
          const bool opt = P.pt_opt[k] != 0;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const double xn = opt ? x[i] + sign * out[i] : x[i];
            X_dst[3 * (size_t)k + i] = xn;
            px[k - gr.pt_begin][i] = xn;
          }
          px[k - gr.pt_begin][3] = opt ? 1.0 : 0.0;
        }
      }
      lds_wave_sync();
      BS_FINE(4);
      BS_STAMP();
    }
    // second pass over the group: compute_cost of the trial set (optimised camera AND optimised point).  Its
    // registers (updated camera, residual) replace the first pass's instead of adding to them.
    if (want_cost && pos >= 0) {
      double ds[6], cmn[12];
#pragma unroll
      for (int a = 0; a < 6; ++a) ds[a] = sign * d[a];
      camera_perturb(cm, ds, cmn);
      double2 zn = (stager && gr.pt_begin + slot < gr.pt_end) ? P.obs_z[n0 + (size_t)slot * L] : double2{0.0, 0.0};
      for (int kb = gr.pt_begin; kb < gr.pt_end; kb += NP) {
        const int k = kb + slot;
        const double2 z = zn;
        if (stager && k + NP < gr.pt_end) zn = P.obs_z[n0 + (size_t)(k + NP - gr.pt_begin) * L];
        if (stager && k < gr.pt_end && px[k - gr.pt_begin][3] != 0.0) {
          const double xn[3] = {px[k - gr.pt_begin][0], px[k - gr.pt_begin][1], px[k - gr.pt_begin][2]};
          double e[2], r[2];
          obs_residual(P.K, cmn, xn, z.x, z.y, P.sensor, e, r);
          cost_acc += r[0] * r[0] + r[1] * r[1];
        }
      }
    }
    lds_wave_sync();
    BS_STAMP();
  }
#ifdef BA_BCR_PROFILE
  if (blockIdx.x == 100 && threadIdx.x == 0) { printf("[k_backsub_groups wg 100 wave 0] cycles since start:"); for (int q = 1; q < ns; ++q) printf(" %lld", st[q] - st[0]); printf(" | last batch: inputs in hand .. linearised %lld, staged + sync %lld, sums (waits for bP) %lld, inverse applied + stores %lld\n", fs[1] - fs[0], fs[2] - fs[1], fs[3] - fs[2], fs[4] - fs[3]); }
#endif
  if (!host) return;
  cost_acc = wave_sum(cost_acc);
  if (lane == 0) wsum[wv] = cost_acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) s += wsum[w];
    host->partial[blockIdx.x] = s;
    if (dev_result) dev_result[blockIdx.x] = s;
    if (blockIdx.x == 0) {
      host->singular_points = *singular_points;
      host->solve_info = *solve_info;
      if (dev_result) { dev_result[kCostBlocks] = (double)*singular_points; dev_result[kCostBlocks + 1] = trial_status_word(*solve_info); }
    }
  }
  if (dev_result && blockIdx.x == 0)                     // the sharded adjuster sums ALL kCostBlocks entries
    for (int i = gridDim.x + threadIdx.x; i < kCostBlocks; i += kBlock) dev_result[i] = 0.0;
}

// --------------------------------------------------------------------------
// Bundle.triangulate_all (bundle.py:313-321; triangulate.algebraic_lsq, triangulate.py:6-18):
// per point the linear least-squares problem with two rows per observation,
//   A[2i]   = (K[0] - z0 K[2]) R_i ,   b[2i]   = (z0 K[2] - K[0]) . t_i
//   A[2i+1] = (K[1] - z1 K[2]) R_i ,   b[2i+1] = (z1 K[2] - K[1]) . t_i
// which the reference hands to numpy.linalg.lstsq (triangulate.py:17).  Solved here by QR, not through the 3 x 3 normal
// equations (round 2 did that: a track seen under little parallax - condition number 1e5 - lost ten digits to the squared
// condition number): every lane rotates its rows into a 3 x 3 upper triangle R and c = Q^T b (Givens row updates), the
// lanes of a point merge their triangles the same way (the partner's three rows are three more rows), R x = c by back-
// substitution.  A track whose system is rank deficient to working precision (|R_jj| <= rcond max|R_ii|: one observation,
// a point at infinity) takes lstsq's minimum-norm answer through the pseudo-inverse of A^T A = R^T R, as before.
// Same lanes-per-point mapping as k_linearize.
// --------------------------------------------------------------------------
__device__ __forceinline__ void tri_givens_row(double (&R)[6], double (&c)[3], double a0, double a1, double a2, double rhs) {
  // R = [r00 r01 r02; 0 r11 r12; 0 0 r22] as R[0..5]; rotate the row (a0 a1 a2 | rhs) into it
  {
    const double r = sqrt(R[0] * R[0] + a0 * a0);
    if (r > 0.0) {
      const double cs = R[0] / r, sn = a0 / r;
      const double t1 = cs * R[1] + sn * a1, t2 = cs * R[2] + sn * a2, tc = cs * c[0] + sn * rhs;
      a1 = cs * a1 - sn * R[1]; a2 = cs * a2 - sn * R[2]; rhs = cs * rhs - sn * c[0];
      R[0] = r; R[1] = t1; R[2] = t2; c[0] = tc;
    }
  }
  {
    const double r = sqrt(R[3] * R[3] + a1 * a1);
    if (r > 0.0) {
      const double cs = R[3] / r, sn = a1 / r;
      const double t2 = cs * R[4] + sn * a2, tc = cs * c[1] + sn * rhs;
      a2 = cs * a2 - sn * R[4]; rhs = cs * rhs - sn * c[1];
      R[3] = r; R[4] = t2; c[1] = tc;
    }
  }
  {
    const double r = sqrt(R[5] * R[5] + a2 * a2);
    if (r > 0.0) {
      const double cs = R[5] / r, sn = a2 / r;
      c[2] = cs * c[2] + sn * rhs;
      R[5] = r;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_triangulate(DevProblem P, const double* __restrict__ cams, int glog,
                                                        double rcond, double* __restrict__ Xout) {
  const int G = 1 << glog;
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long k = tid >> glog;
  const int l = (int)(tid & (G - 1));
  const bool valid = k < P.nt;
  int s = 0, e_ = 0;
  if (valid) { s = P.pt_off[k]; e_ = P.pt_off[k + 1]; }
  double R[6] = {0, 0, 0, 0, 0, 0}, c[3] = {0, 0, 0};
  for (int n = s + l; n < e_; n += G) {
    const int cam = P.obs_cam[n];
    const double2 z = P.obs_z[n];
    double cm[12];
    load_cam(cams, cam, cm);
    const double zz[2] = {z.x, z.y};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double k0 = P.K[3 * r] - zz[r] * P.K[6], k1 = P.K[3 * r + 1] - zz[r] * P.K[7],
                   k2 = P.K[3 * r + 2] - zz[r] * P.K[8];
      const double a0 = k0 * cm[0] + k1 * cm[3] + k2 * cm[6];
      const double a1 = k0 * cm[1] + k1 * cm[4] + k2 * cm[7];
      const double a2 = k0 * cm[2] + k1 * cm[5] + k2 * cm[8];
      const double rhs = -(k0 * cm[9] + k1 * cm[10] + k2 * cm[11]);
      tri_givens_row(R, c, a0, a1, a2, rhs);
    }
  }
  for (int m = G >> 1; m >= 1; m >>= 1) {          // merge with the partner's triangle: its three rows are three more rows
    double Rp[6], cp[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) Rp[i] = __shfl_xor(R[i], m, 64);
#pragma unroll
    for (int i = 0; i < 3; ++i) cp[i] = __shfl_xor(c[i], m, 64);
    tri_givens_row(R, c, Rp[0], Rp[1], Rp[2], cp[0]);
    tri_givens_row(R, c, 0.0, Rp[3], Rp[4], cp[1]);
    tri_givens_row(R, c, 0.0, 0.0, Rp[5], cp[2]);
  }
  if (valid && l == 0) {
    double x[3];
    const double d0 = fabs(R[0]), d1 = fabs(R[3]), d2 = fabs(R[5]);
    const double dmax = fmax(d0, fmax(d1, d2)), dmin = fmin(d0, fmin(d1, d2));
    if (dmin > rcond * dmax) {
      x[2] = c[2] / R[5];
      x[1] = (c[1] - R[4] * x[2]) / R[3];
      x[0] = (c[0] - R[1] * x[1] - R[2] * x[2]) / R[0];
    } else {
      // rank deficient: lstsq's minimum-norm solution, x = pinv(R^T R) R^T c
      const double ata[6] = {R[0] * R[0], R[0] * R[1], R[0] * R[2], R[1] * R[1] + R[3] * R[3], R[1] * R[2] + R[3] * R[4],
                             R[2] * R[2] + R[4] * R[4] + R[5] * R[5]};
      const double atb[3] = {R[0] * c[0], R[1] * c[0] + R[3] * c[1], R[2] * c[0] + R[4] * c[1] + R[5] * c[2]};
      double inv[6];
      sym3_pinv(ata, fmax(rcond * rcond, 1e-14), inv);
      sym3_apply(inv, atb, x);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) Xout[3 * k + i] = x[i];
  }
}

// --------------------------------------------------------------------------
// update_motion / update_structure (bundle_adjuster.py:334-343): dst = src (+) sign*delta
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_apply_update(int nc, int nt, const int* __restrict__ cam_opt_pos,
                                                         const unsigned char* __restrict__ pt_opt,
                                                         const double* __restrict__ cams_src,
                                                         const double* __restrict__ X_src,
                                                         const double* __restrict__ dC,
                                                         const double* __restrict__ dP, double sign,
                                                         double* __restrict__ cams_dst,
                                                         double* __restrict__ X_dst) {
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (tid < nc) {
    const int i = (int)tid;
    double cm[12], out[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) cm[q] = cams_src[(size_t)i * 12 + q];
    const int pos = cam_opt_pos[i];
    if (pos >= 0) {
      double d[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) d[q] = sign * dC[(size_t)pos * 6 + q];
      camera_perturb(cm, d, out);
    } else {
#pragma unroll
      for (int q = 0; q < 12; ++q) out[q] = cm[q];
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) cams_dst[(size_t)i * 12 + q] = out[q];
  } else if (tid < (long long)nc + nt) {
    const size_t k = (size_t)(tid - nc);
    const bool opt = pt_opt[k] != 0;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      X_dst[3 * k + q] = opt ? X_src[3 * k + q] + sign * dP[3 * k + q] : X_src[3 * k + q];
  }
}

}  // namespace ba
