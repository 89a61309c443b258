// ba_kernels.h - gfx950 (MI355X, CDNA4) kernels of the bundle-adjustment inner loop.
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

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ba_math.h"

namespace ba {

constexpr int kBlock = 256;       // 4 wavefronts
constexpr int kWave = 64;
constexpr int kTile = 16;         // observations of one track staged per Schur work unit

struct DevProblem {
  int nc, nt, nco;
  int hb;                   // block half-bandwidth of the reduced system (see reduced-system layout below)
  long long nobs;
  const int* obs_cam;
  const int* obs_pt;
  const double2* obs_z;
  const int* pt_off;        // [nt+1]
  const int* cam_opt_pos;   // [nc]
  const unsigned char* pt_opt;  // [nt]
  double K[9];
  Sensor sensor;
};

__device__ __forceinline__ void load_cam(const double* __restrict__ cams, int c, double cm[12]) {
  const double2* p = reinterpret_cast<const double2*>(cams + (size_t)c * 12);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double2 v = p[i];
    cm[2 * i] = v.x; cm[2 * i + 1] = v.y;
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Reduced-system layout ("block band"): S is symmetric with 6x6 blocks; only blocks
// (i, j) with i <= j <= i + hb can be non-zero, where hb = max over tracks of the spread
// of their optimised-camera positions.  Block (i, i+d) lives at ((i*(hb+1) + d)*36.
// A dense system is the special case hb = nco-1; a camera sequence with tracks of
// length 10 has hb = 9 and stores 5.5 MB instead of 288 MB at 1000 cameras.
__device__ __forceinline__ size_t band_block(int pi, int pj, int hb1) {
  return ((size_t)pi * hb1 + (pj - pi)) * 36;
}

// order LDS traffic between lanes of ONE wavefront (LDS executes a wavefront's
// instructions in order; this only stops the compiler from moving them and waits for
// the returns).  Deliberately no vmcnt: a fence would wait for global stores in flight.
__device__ __forceinline__ void lds_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Cross-lane exchange through DPP (VALU data path, no LDS round trip): value of the lane
// paired by the given DPP control.  0xB1 / 0x4E = quad_perm xor 1 / xor 2, 0x141 =
// row_half_mirror (i <-> 7-i), 0x140 = row_mirror (i <-> 15-i).
template <int CTRL>
__device__ __forceinline__ double dpp_pair(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sum over aligned groups of G lanes (G = 1, 2, 4, 8, 16, 32); every lane gets the sum
template <int G>
__device__ __forceinline__ double group_sum(double v) {
  if (G >= 2) v += dpp_pair<0xB1>(v);
  if (G >= 4) v += dpp_pair<0x4E>(v);
  if (G >= 8) v += dpp_pair<0x141>(v);
  if (G >= 16) v += dpp_pair<0x140>(v);
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

// cooperative copy global -> LDS of n doubles (n even, both 16-byte aligned): 16 B per lane
// and eight loads in flight per lane, so a chunk costs about one global round trip
template <int NT>
__device__ __forceinline__ void copy_to_lds(double* __restrict__ dst, const double* __restrict__ src, int n, int tid) {
  const double2* s2 = reinterpret_cast<const double2*>(src);
  double2* d2 = reinterpret_cast<double2*>(dst);
  const int n2 = n >> 1;
  for (int base = 0; base < n2; base += NT * 8) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * NT + tid;
      if (idx < n2) v[u] = s2[idx];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * NT + tid;
      if (idx < n2) d2[idx] = v[u];
    }
  }
}

// LDS-only workgroup barrier: waits for this wavefront's LDS traffic but leaves global
// loads / stores in flight (the prefetch of the next band row must not be drained at
// every barrier; cdna_hip_programming.md "raw s_barrier + lgkmcnt(0) only").
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// broadcast a double from a compile-time-constant lane through SGPRs (v_readlane_b32 x2):
// a few cycles, instead of the ~100-cycle LDS round trip of ds_bpermute behind __shfl.
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(x) to fp64 round-off: v_rsq_f64 seed (~2^-26) + two Newton steps.  The library
// sqrt + divide pair costs ~10x more on the serial critical path of the factorisation.
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}

// The same to the same accuracy with one cubic step, y (1 + e/2 + 3 e^2/8) with e = 1 - x y^2 (|e| ~ 2^-24 after the seed:
// the next term of the series is 5 e^3 / 16 ~ 2^-74): five fp64 instructions after the seed instead of seven, and a
// dependent chain of four instead of six - for the pivot chains, where one wavefront pays for every instruction it issues.
__device__ __forceinline__ double rsqrt_cubic(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}

// hardware fp64 atomic add (global_atomic_add_f64 / ds_add_f64 on gfx950)
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

// --------------------------------------------------------------------------
// compute_cost (bundle_adjuster.py:165-171): one observation per lane,
// wavefront + block reduction, one partial per block.  The partials go straight into a
// pinned host record (with the two status words of the trial) and the CPU adds them in
// index order after synchronising: deterministic, no second launch, no copy kernels.
// --------------------------------------------------------------------------
constexpr int kCostBlocks = 2048;
struct HostResult { int singular_points; int solve_info; double partial[kCostBlocks]; };
// The solver status as it travels in the shards' trial record, which is SUMMED over the ranks: a time-out (a fault, never a
// property of the matrix) must still be recognisable after the sum, so it weighs more than any sum of pivot indices can.
constexpr double kTrialTimedOutWord = 1099511627776.0;      // 2^40 > ranks * 2^31
__host__ __device__ inline double trial_status_word(int solve_info) { return solve_info == 0x7f000001 ? kTrialTimedOutWord : (double)solve_info; }
__host__ __device__ inline int trial_status_of_sum(double sum, int ranks, bool own_parts) {
  if (sum >= kTrialTimedOutWord) return 0x7f000001;
  if (sum == 0.0) return 0;
  // every rank solved the same system (status x ranks) - or, with the solve spread over the ranks, its own part of it: any non-zero = failed
  const double v = own_parts ? fabs(sum) : sum / ranks;
  const double c = v < 1.0 ? (own_parts ? 1.0 : v) : (v > 2.0e9 ? 2.0e9 : v);
  return (int)(c + (c >= 0 ? 0.5 : -0.5));
}

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
    obs_residual(P.K, cm, x, z.x, z.y, P.sensor, e, r);
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
  obs_linearize(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
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
  sensor_eval(s, e[2 * i], e[2 * i + 1], rr, JJ);
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
    obs_linearize(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
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
constexpr int kCamChunk = 2048;    // at most; the host shrinks it for scenes with few cameras (ba_set_problem)
struct CamUnit { int cam; int begin; int end; };

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
  for (int q = un.begin + lane; q < un.end; q += 64) {
    const int n = perm[q];
    const int k = P.obs_pt[n];
    const double2 z = P.obs_z[n];
    const double x[3] = {X[3 * (size_t)k], X[3 * (size_t)k + 1], X[3 * (size_t)k + 2]};
    double e[2], r[2], Jc[12], Jp[6];
    obs_linearize(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
    int idx = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = a; b < 6; ++b) acc[idx++] += Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] += Jc[a] * r[0] + Jc[6 + a] * r[1];
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
// apply_damping on HPP (bundle_adjuster.py:241-242, optimize.py:7-9) and the
// per-point inverse (bundle_adjuster.py:252-256).  One point per lane.
// --------------------------------------------------------------------------
__device__ __forceinline__ void point_invert_body(int k, int nt, const double* __restrict__ HPP, double damping, double rcond,
                                                  double* __restrict__ HPPinv, int* __restrict__ singular_count,
                                                  int* __restrict__ next_count, const double* __restrict__ bP = nullptr,
                                                  double* __restrict__ fac = nullptr) {
  if (k == 0) *next_count = 0;      // the counter the NEXT call will use (two counters alternate: no memset launch)
  if (k >= nt) return;
  double A[6], out[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) A[i] = HPP[6 * (size_t)k + i];
  const double f = 1.0 + damping;
  A[0] *= f; A[3] *= f; A[5] *= f;
  if (rcond >= 0.0) {
    sym3_pinv_fast(A, rcond, out);
  } else if (!sym3_inv(A, out)) {
    atomicAdd(singular_count, 1);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) HPPinv[6 * (size_t)k + i] = out[i];
  if (fac) {                                   // HPPinv = L D L^T and HPPinv bP for k_schur_groups_mfma2
    double f[9];
    sym3_ldl(out, sym3_ldl_tolerance(rcond), f, f + 3);
    const double g0 = bP[3 * (size_t)k], g1 = bP[3 * (size_t)k + 1], g2 = bP[3 * (size_t)k + 2];
    f[6] = out[0] * g0 + out[1] * g1 + out[2] * g2;
    f[7] = out[1] * g0 + out[3] * g1 + out[4] * g2;
    f[8] = out[2] * g0 + out[4] * g1 + out[5] * g2;
#pragma unroll
    for (int i = 0; i < 9; ++i) fac[9 * (size_t)k + i] = f[i];
  }
}

__global__ __launch_bounds__(kBlock) void k_point_invert(int nt, const double* __restrict__ HPP,
                                                         double damping, double rcond,
                                                         double* __restrict__ HPPinv,
                                                         int* __restrict__ singular_count,
                                                         int* __restrict__ next_count) {
  point_invert_body(blockIdx.x * kBlock + threadIdx.x, nt, HPP, damping, rcond, HPPinv, singular_count, next_count);
}

// --------------------------------------------------------------------------
// S[pos,pos] = damped HCC, b[pos] = bC for optimised cameras
// (bundle_adjuster.py:238-240, 263-265); every other block of the band is cleared in the
// same pass (one launch instead of two memsets + a scatter).  One thread per double of
// [S | b]; `opt_cam[pos]` is the camera at optimised position pos.
// --------------------------------------------------------------------------
__device__ __forceinline__ void schur_init_body(long long tid, int nco, int hb1, const int* __restrict__ opt_cam,
                                                const double* __restrict__ HCC, const double* __restrict__ bC,
                                                double damping, double* __restrict__ S, double* __restrict__ b, int use_hcc) {
  const long long nS = (long long)nco * hb1 * 36;
  if (tid < nS) {
    const int e = (int)(tid % 36);
    const long long blk = tid / 36;
    const int d = (int)(blk % hb1), pos = (int)(blk / hb1);
    double v = 0.0;
    if (d == 0 && use_hcc) {                     // (the MFMA reduction can add the camera blocks itself)
      const int a = e / 6, c = e % 6;
      const int lo = a < c ? a : c, hi = a < c ? c : a;
      v = HCC[(size_t)opt_cam[pos] * 36 + lo * 6 + hi];
      if (a == c) v *= (1.0 + damping);
    }
    S[tid] = v;
  } else if (tid < nS + (long long)nco * 6) {
    const long long q = tid - nS;
    b[q] = use_hcc ? bC[(size_t)opt_cam[q / 6] * 6 + q % 6] : 0.0;
  }
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
struct SchurUnit { int pt; int row0; int col0; };
struct SchurChunk { int begin; int end; int p0; };     // units [begin, end), window start p0
constexpr int kSchurChunkUnits = 64;                  // at most this many units per workgroup
constexpr int kSchurTileBytes = 48 * 1024;             // LDS budget of the accumulation tile

constexpr int kSchurBlock = 1024;                      // 16 wavefronts share one accumulation tile

// p-th pair (i <= j) of the upper triangle of an n x n grid, rows first
__device__ __forceinline__ void tri_decode(int p, int n, int& i, int& j) {
  const float t = 2.0f * n + 1.0f;
  int r = (int)((t - sqrtf(t * t - 8.0f * p)) * 0.5f);
  r = max(0, min(r, n - 1));
  while (r > 0 && r * (2 * n - r + 1) / 2 > p) --r;                 // first index of row r
  while ((r + 1) * (2 * n - r) / 2 <= p) ++r;
  i = r;
  j = r + (p - r * (2 * n - r + 1) / 2);
}

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
        obs_linearize(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
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
struct SchurGroup { int pt_begin; int pt_end; int L; int pad; };
constexpr int kGroupBlock = 512;                       // 8 wavefronts per workgroup
constexpr int kGroupMaxPts = 24;
constexpr int kGroupMaxL = 15;
constexpr int kGroupChunk = 8;                         // groups per workgroup

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
// The group reduction on the fp64 matrix cores, for track lengths <= 10 (a group's cameras span
// at most 60 rows of the reduced system: a 64 x 64 window = 4 x 4 MFMA tiles).
// For the points k of a group,
//     S_window -= sum_k Tstack_k Wstack_k^T ,   Tstack_k = [W_1k A_k; ...; W_Lk A_k]  (6L x 3)
// is ONE matrix product with inner dimension 3 * (#points): per batch of 6 points (K = 18, padded
// to 20) a wavefront
//   phase A  linearises one observation per lane (60 lanes), forms W and T = W HPPinv, stages
//            them K-major in LDS ([k][row], row = 6 * observation + a), and adds T bP to b;
//   phase B  runs 5 k-steps x 10 upper tiles of v_mfma_f64_16x16x4_f64 with the accumulators
//            (10 tiles x 4 doubles) living in registers across the whole group.
// The vector version spends 42 LDS reads and 162 FMA instructions per (pair, point); here a
// batch costs 40 LDS reads and 50 MFMAs per wavefront.  Epilogue and window logic as above.
// --------------------------------------------------------------------------
constexpr int kGmBlock = 256;                          // 4 wavefronts per workgroup (20 KB of staging each)
constexpr int kGmMaxL = 10;
constexpr int kGmChunk = 4;                            // groups per workgroup: one per wavefront
constexpr int kGmPts = 6;                              // points per batch
constexpr int kGmK = 20;                               // staged k rows: 3 per point, two zero rows
constexpr int kGmLd = 64;                              // staged row length (60 used)

template <bool FUSE_LIN>
__global__ __launch_bounds__(kGmBlock) void k_schur_groups_mfma(DevProblem P, const double* __restrict__ cams,
                                                                const double* __restrict__ X,
                                                                const SchurGroup* __restrict__ groups,
                                                                const SchurChunk* __restrict__ chunks, int wn,
                                                                double* __restrict__ HPPinv,
                                                                double* __restrict__ bP,
                                                                double* __restrict__ S, double* __restrict__ b,
                                                                double damping, int fuse_cam,
                                                                double* __restrict__ HPP, double rcond,
                                                                int* __restrict__ singular_count,
                                                                int* __restrict__ next_count) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  constexpr int NW = kGmBlock / kWave;
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* sT = dyn;                                   // [NW][kGmK][kGmLd]
  double* sWm = sT + NW * kGmK * kGmLd;               // [NW][kGmK][kGmLd]
  int* sPos = reinterpret_cast<int*>(sWm + NW * kGmK * kGmLd);   // [NW][16]
  double* sSum = reinterpret_cast<double*>(sPos + NW * 16);       // [NW][64]  FUSE_LIN: HPP | bP of the staged points
  double* tile = sSum + NW * 64;
  const int hb1 = P.hb + 1;
  const int rowlen = hb1 * 36;
  double* tb = tile + (size_t)wn * rowlen;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (FUSE_LIN && blockIdx.x == 0 && threadIdx.x == 0) *next_count = 0;   // as k_point_invert: the counter the NEXT inversion uses
  const int lr = lane & 15, lk = lane >> 4;
  const SchurChunk ck = chunks[blockIdx.x];
  const int p0 = ck.p0;
  for (int i = threadIdx.x; i < wn * (rowlen + 6); i += kGmBlock) tile[i] = 0.0;
  double* mT = sT + wv * kGmK * kGmLd;
  double* mW = sWm + wv * kGmK * kGmLd;
  int* mPos = sPos + wv * 16;
  double* mSum = sSum + wv * 64;
  for (int i = lane; i < kGmK * kGmLd; i += 64) { mT[i] = 0.0; mW[i] = 0.0; }     // incl. the two zero k rows
  __syncthreads();
  // column of this lane in each of the 4 column tiles: observation j = n / 6, entry c = n % 6;
  // row of accumulator register v of row tile ti: m = 16 ti + lane/16 + 4 v -> observation i, entry a
  int jn[4], cn[4], im[16], am[16];
#pragma unroll
  for (int t = 0; t < 4; ++t) { const int n = 16 * t + lr; jn[t] = n / 6; cn[t] = n - 6 * jn[t]; }
#pragma unroll
  for (int t = 0; t < 16; ++t) { const int m = 16 * (t >> 2) + lk + 4 * (t & 3); im[t] = m / 6; am[t] = m - 6 * im[t]; }

  for (int g = ck.begin + wv; g < ck.end; g += NW) {       // wave-uniform
    const SchurGroup gr = groups[g];
    const int L = gr.L;
    const int NP = 64 / L < kGmPts ? 64 / L : kGmPts;
    const int nts = (6 * L + 15) >> 4;                     // tiles per side that hold real rows
    if (lane < 16) mPos[lane] = lane < L ? P.cam_opt_pos[P.obs_cam[P.pt_off[gr.pt_begin] + lane]] : -1;
    lds_wave_sync();
    mfma_acc acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = mfma_acc{0.0, 0.0, 0.0, 0.0};
    double bacc[6] = {0, 0, 0, 0, 0, 0};
    double hc[21];                                            // fuse_cam: this lane's share of HCC (upper triangle)
#pragma unroll
    for (int q = 0; q < 21; ++q) hc[q] = 0.0;
    const int slot = lane / L, oi = lane - slot * L;          // phase A role: (staged point, observation)
    const bool stager = lane < NP * L;
    const int mypos = stager ? mPos[oi] : -1;
    // every point of the group sees the same cameras: this lane's camera is loaded once per group,
    // and its observations sit at a fixed stride (all tracks of the group have L observations)
    const int n0 = P.pt_off[gr.pt_begin] + oi;
    double cm[12];
    {
      const int c = P.obs_cam[stager ? n0 : P.pt_off[gr.pt_begin]];
      load_cam(cams, c, cm);
    }
    // per-point inputs of the NEXT batch are fetched while the matrix cores work on this one
    struct PointIn { double x[3], A[6], g[3]; double2 z; };
    auto fetch = [&](int kb_, PointIn& in) {
      const int k = kb_ + slot;
      if (stager && k < gr.pt_end) {
        in.z = P.obs_z[n0 + (size_t)(k - gr.pt_begin) * L];
#pragma unroll
        for (int q = 0; q < 3; ++q) in.x[q] = X[3 * (size_t)k + q];
        if (!FUSE_LIN) {                                       // otherwise both are formed right here, in phase A
#pragma unroll
          for (int q = 0; q < 3; ++q) in.g[q] = bP[3 * (size_t)k + q];
#pragma unroll
          for (int q = 0; q < 6; ++q) in.A[q] = HPPinv[6 * (size_t)k + q];
        }
      }
    };
    PointIn nxt;
    fetch(gr.pt_begin, nxt);

#ifdef BA_BCR_PROFILE
    long long tA = 0, tB = 0, ta0_first = 0; const long long tg0 = clock64();
#endif
    for (int kb = gr.pt_begin; kb < gr.pt_end; kb += NP) {
      const int np = min(NP, gr.pt_end - kb);
      const PointIn cur = nxt;
#ifdef BA_BCR_PROFILE
      const long long ta0 = clock64();
      if (kb == gr.pt_begin) ta0_first = ta0;
#endif
      // ---- phase A
      double e[2], r[2], Jc[12], Jp[6], Apt[6], gpt[3];
      const bool live = stager && slot < np;
      if (live) obs_linearize(P.K, cm, cur.x, cur.z.x, cur.z.y, P.sensor, e, r, Jc, Jp);
      if constexpr (FUSE_LIN) {
        // prepare_schur_complement for the staged points (k_linearize's and k_point_invert's work):
        // HPP_k = sum_i Jp^T Jp, bP_k = sum_i Jp^T r over the point's L lanes - 9 sums per point, each
        // done by one lane of the point through LDS - then every lane of the point inverts the damped
        // block for itself (same instruction count for the wavefront as one lane doing it).
        double loc[9];
        if (live) {
          loc[0] = Jp[0] * Jp[0] + Jp[3] * Jp[3]; loc[1] = Jp[0] * Jp[1] + Jp[3] * Jp[4]; loc[2] = Jp[0] * Jp[2] + Jp[3] * Jp[5];
          loc[3] = Jp[1] * Jp[1] + Jp[4] * Jp[4]; loc[4] = Jp[1] * Jp[2] + Jp[4] * Jp[5]; loc[5] = Jp[2] * Jp[2] + Jp[5] * Jp[5];
          loc[6] = Jp[0] * r[0] + Jp[3] * r[1]; loc[7] = Jp[1] * r[0] + Jp[4] * r[1]; loc[8] = Jp[2] * r[0] + Jp[5] * r[1];
        } else {
#pragma unroll
          for (int c = 0; c < 9; ++c) loc[c] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) mT[c * kGmLd + lane] = loc[c];       // rows 0..8 of the staging area are free here
        lds_wave_sync();
        if (live) {
          const int k = kb + slot;
          for (int c = oi; c < 9; c += L) {
            double v[kGmMaxL];
#pragma unroll
            for (int j = 0; j < kGmMaxL; ++j) v[j] = j < L ? mT[c * kGmLd + slot * L + j] : 0.0;   // one LDS round trip
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < kGmMaxL; ++j) sum += v[j];
            mSum[slot * 9 + c] = sum;
            if (c < 6) HPP[6 * (size_t)k + c] = sum; else bP[3 * (size_t)k + c - 6] = sum;
          }
        }
        lds_wave_sync();
        if (live) {
          double hp[6];
#pragma unroll
          for (int c = 0; c < 6; ++c) hp[c] = mSum[slot * 9 + c];
#pragma unroll
          for (int c = 0; c < 3; ++c) gpt[c] = mSum[slot * 9 + 6 + c];
          const double f = 1.0 + damping;
          hp[0] *= f; hp[3] *= f; hp[5] *= f;
          if (rcond >= 0.0) {
            sym3_pinv_fast(hp, rcond, Apt);
          } else if (!sym3_inv(hp, Apt) && oi == 0) {
            atomicAdd(singular_count, 1);
          }
          if (oi == 0) {
            const int k = kb + slot;
#pragma unroll
            for (int c = 0; c < 6; ++c) HPPinv[6 * (size_t)k + c] = Apt[c];
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) Apt[c] = cur.A[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) gpt[c] = cur.g[c];
      }
      if (stager) {
        double W[18], T[18];
        if (slot < np) {
          block_W(Jc, Jp, W);
          block_T(W, Apt, T);
          if (mypos >= 0) {                                   // b[i] -= T_i bP_k
#pragma unroll
            for (int a = 0; a < 6; ++a) bacc[a] -= T[a * 3] * gpt[0] + T[a * 3 + 1] * gpt[1] + T[a * 3 + 2] * gpt[2];
            if (fuse_cam) {                                   // HCC[i] += Jc^T Jc, b[i] += Jc^T r (k_camera_blocks' work)
              int idx = 0;
#pragma unroll
              for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int c2 = a; c2 < 6; ++c2) hc[idx++] += Jc[a] * Jc[c2] + Jc[6 + a] * Jc[6 + c2];
                bacc[a] += Jc[a] * r[0] + Jc[6 + a] * r[1];
              }
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < 18; ++q) { W[q] = 0.0; T[q] = 0.0; }     // a short last batch: zero k rows
        }
        const int so = 3 * slot * kGmLd + 6 * oi;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int d = 0; d < 3; ++d) { mT[so + d * kGmLd + a] = T[a * 3 + d]; mW[so + d * kGmLd + a] = W[a * 3 + d]; }
      }
      fetch(kb + NP, nxt);
      lds_wave_sync();
#ifdef BA_BCR_PROFILE
      const long long tb0 = clock64(); tA += tb0 - ta0;
#endif
      // ---- phase B: acc(ti, tj) += T[rows of ti][k] W[rows of tj][k]
#pragma unroll
      for (int s4 = 0; s4 < kGmK / 4; ++s4) {
        double ta[4], wb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ta[t] = mT[(4 * s4 + lk) * kGmLd + 16 * t + lr];
          wb[t] = mW[(4 * s4 + lk) * kGmLd + 16 * t + lr];
        }
        int q = 0;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
          for (int tj = ti; tj < 4; ++tj, ++q)
            if (tj < nts) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[ti], wb[tj], acc[q], 0, 0, 0);
      }
      lds_wave_sync();
#ifdef BA_BCR_PROFILE
      tB += clock64() - tb0;
#endif
    }
#ifdef BA_BCR_PROFILE
    const long long te0 = clock64();
#endif
    // ---- epilogue: C/D layout lane -> column n = 16 tj + lane%16, register v -> row m = 16 ti + lane/16 + 4 v.
    // Positions ascend along a track (checked on the host), so block (i, j), i <= j, lives at
    // row pos_i, offset (pos_j - pos_i) * 36 + a * 6 + c: the address splits into a row part and a
    // column part, and the LDS window test depends on the row alone.
    {
      int colpart[4], pjv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        pjv[t] = jn[t] < L ? mPos[jn[t]] : -1;
        colpart[t] = pjv[t] * 36 + cn[t];
      }
      int q = 0;
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        int rowpart[4], pim[4];
        bool inwin[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int i = im[4 * ti + v];
          const int pi = i < L ? mPos[i] : -1;
          const int wr = pi - p0;
          pim[v] = pi;
          inwin[v] = wr >= 0 && wr < wn;
          rowpart[v] = (inwin[v] ? wr * rowlen : pi * rowlen) - pi * 36 + am[4 * ti + v] * 6;
        }
#pragma unroll
        for (int tj = ti; tj < 4; ++tj, ++q) {
          if (tj >= nts) continue;
          const int j = jn[tj], c = cn[tj];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int i = im[4 * ti + v], a = am[4 * ti + v];
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
    if (mypos >= 0) {
      const int wr = mypos - p0;
      const bool in = wr >= 0 && wr < wn;
      // (separate LDS / global code paths: a pointer select would turn these into slow flat atomics)
      if (in) {
#pragma unroll
        for (int a = 0; a < 6; ++a) atomic_add_f64(tb + wr * 6 + a, bacc[a]);
      } else {
#pragma unroll
        for (int a = 0; a < 6; ++a) atomic_add_f64(b + (size_t)mypos * 6 + a, bacc[a]);
      }
      if (fuse_cam) {                                       // damped camera block onto the diagonal block (stored in full)
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
    lds_wave_sync();                                        // mPos is rewritten by the next group
#ifdef BA_BCR_PROFILE
    if (blockIdx.x == 100 && wv == 0 && lane == 0 && g == ck.begin) {
      long long* dbg = reinterpret_cast<long long*>(S + (size_t)P.nco * rowlen + 6 * P.nco);   // scratch behind [S | b] (profile builds only)
      (void)dbg;
      printf("[k_schur_groups_mfma] group of %d points: setup %lld, phase A %lld, phase B %lld, epilogue %lld cycles\n",
             gr.pt_end - gr.pt_begin, ta0_first - tg0, tA, tB, clock64() - te0);
    }
#endif
  }
  if (wn == 0) return;
  __syncthreads();
  for (int i = threadIdx.x; i < wn * rowlen; i += kGmBlock) {
    const double v = tile[i];
    const int wr = i / rowlen;
    if (v != 0.0 && p0 + wr < P.nco) atomic_add_f64(S + (size_t)(p0 + wr) * rowlen + (i - wr * rowlen), v);
  }
  for (int i = threadIdx.x; i < wn * 6; i += kGmBlock) {
    const double v = tb[i];
    if (v != 0.0 && p0 + i / 6 < P.nco) atomic_add_f64(b + (size_t)p0 * 6 + i, v);
  }
}

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
constexpr int kGm2Block = 512;
constexpr int kGm2Pairs = 4;
constexpr int kGm2DRows = 24;                          // D values per buffer (20 used)

__host__ __device__ inline size_t schur_mfma2_lds_bytes(int wn, int hb1) {
  return (size_t)kGm2Pairs * 2 * kGmK * kGmLd * sizeof(double) + (size_t)kGm2Pairs * 2 * kGm2DRows * sizeof(double) +
         (size_t)kGm2Pairs * (16 + 4) * sizeof(int) + 64 * sizeof(double) + (size_t)wn * ((size_t)hb1 * 36 + 6) * sizeof(double);
}

__device__ __forceinline__ void gm2_wait(int* flag, int need) {
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void gm2_post(int* flag, int value, int lane) {
  lds_wave_sync();                                         // my LDS reads / writes are done (in-order LDS: and visible)
  if (lane == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

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
              if (tj < nts) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[ti], wb[tj], acc[q], 0, 0, 0);
        }
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
  if (blockIdx.x == 100 && lane == 0 && (wv == 0 || wv == 4))
    printf("[k_schur_groups_mfma2 wg 100 %s] batches %lld: total %lld cycles, waiting for the partner %lld, epilogue %lld; workgroup setup %lld, tail (barrier + flush) %lld\n",
           wv == 0 ? "producer" : "consumer", pn, pk1 - pk0, pw, pe, pk0 - pkk, clock64() - pk1);
#endif
}

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
constexpr int kGm3MaxL = 24;                            // track length up to which the launches re-linearise at most four times
constexpr int kGm3MaxSpan = 40;                         // widest window: 15 tiles per side, the last tile COLUMN alone fills the 15 accumulator tiles of a launch
constexpr int kGm3PosLen = 64;                          // optimised positions of a group's cameras (40 used)
constexpr int kGm3MaxTiles = 15;                        // accumulator tiles of one launch

struct Gm3Params { int nts; int Ld; int Kbuf; int np_cap; int wn; int do_rhs; int wb1; };      // wb1: blocks per row of the LDS window (the widest group)
// A group of k_schur_groups_mfma3: consecutive points (internal order) whose optimised cameras all lie in the window of
// W <= 40 (kGm3MaxSpan) consecutive optimised positions starting at `lo`.  tab[(k - pt_begin) * W + w] = the observation of point k
// in the camera at position lo + w, or -1: the camera lists need NOT be identical, only close (tracks of different
// lengths, missing observations) - a run of points with one camera list is the special case of a full table.
struct WinGroup { int pt_begin; int pt_end; int W; int lo; int tab; int pad0; int pad1; int pad2; };

__host__ __device__ constexpr int gm3_ntiles(int tj0, int tj1) { return (tj1 * (tj1 + 1) - tj0 * (tj0 + 1)) / 2; }
__host__ __device__ inline int gm3_np(int L, int np_cap) { int np = 64 / L; if (np > kGmPts) np = kGmPts; if (np > np_cap) np = np_cap; return np; }
__host__ __device__ inline size_t schur_mfma3_lds_bytes(int Kbuf, int Ld, int wn, int hb1) {      // hb1: blocks per window row (Gm3Params::wb1)
  return (size_t)kGm2Pairs * 2 * Kbuf * Ld * sizeof(double) + (size_t)kGm2Pairs * 2 * kGm2DRows * sizeof(double) +
         (size_t)kGm2Pairs * (kGm3PosLen + 4) * sizeof(int) + 64 * sizeof(double) + (size_t)wn * ((size_t)hb1 * 36 + 6) * sizeof(double);
}

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
      const bool any = nts > TJ0;                                // this launch's tile columns exist for the group
      for (int ib = 0; ib < nb; ++ib) {
        gm2_wait(fStaged, nbatch + 1);
        const double* mU = sU + (pair * 2 + (nbatch & 1)) * BUF;
        const double* mD = sD + (pair * 2 + (nbatch & 1)) * kGm2DRows;
#pragma unroll KSC ? KSC : 1
        for (int s4 = 0; s4 < (KSC ? KSC : ks); ++s4) {
          double ta[TJ1], wb[TJ1];
          const double dk = mD[4 * s4 + lk];
          const double* row = mU + (4 * s4 + lk) * Ld + lr;
#pragma unroll
          for (int t = 0; t < TJ1; ++t) wb[t] = (t < nts && any) ? row[16 * t] : 0.0;
          if (s4 == ks - 1) { ++nbatch; gm2_post(fConsumed, nbatch, lane); }     // everything of this buffer is in registers
#pragma unroll
          for (int t = 0; t < TJ1; ++t) ta[t] = wb[t] * dk;
          int q = 0;
#pragma unroll
          for (int tj = TJ0; tj < TJ1; ++tj)
#pragma unroll
            for (int ti = 0; ti <= tj; ++ti, ++q)
              if (tj < nts) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[ti], wb[tj], acc[q], 0, 0, 0);
        }
      }
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
constexpr int kGw7 = 7;                                 // consumers
constexpr int kGwBlock = 64 * (1 + kGw7);
constexpr int kGwMinTiles = 11;                         // tiles per side from which a group comes here (10: five launches of k_schur_groups_mfma3 are faster, 0.68 against 0.90 ms at L = 25)
constexpr int kGwMaxTiles = 15;
constexpr int kGwLd = 16 * kGwMaxTiles;                 // staged row: 240 doubles
constexpr int kGwK = 8;                                 // k rows per buffer (two points: 6 + 2 zero rows)
__host__ __device__ constexpr int gw_own(int nts) { return (nts * (nts + 1) / 2 + kGw7 - 1) / kGw7; }      // tiles per consumer: 8 .. 18

__host__ __device__ inline size_t schur_wide_lds_bytes() {
  return (size_t)2 * kGwK * kGwLd * sizeof(double) + (size_t)2 * kGwK * sizeof(double) + (size_t)(64 + 8 + 128) * sizeof(int);
}

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
constexpr int kRectSeg = 32;                            // positions per segment
constexpr int kRectTiles = 6 * kRectSeg / 16;           // 12 tiles per side of a segment
constexpr int kRectLd = 2 * 6 * kRectSeg;               // staged row: [A | B], 384 doubles
constexpr int kRectGroupPts = 96;                       // listed points per group at most (the host halves it until the groups fill the chip)
constexpr int kRectConsumers = kRectTiles / 2;          // consumer wavefronts of a workgroup: two tile columns each
constexpr int kRectBlock = 64 * (2 + kRectConsumers);      // two producers (even / odd points of the group) + the consumers
struct RectGroup { int n; int pts; int tab; int loA; int loB; int WB; int pad0; int pad1; };      // points rtab[pts ...], table rtab[tab + q * 64 + column]

__host__ __device__ inline size_t schur_rect_lds_bytes() {
  return (size_t)2 * 4 * kRectLd * sizeof(double) + (size_t)2 * 4 * sizeof(double) + (size_t)(64 + 8) * sizeof(int);
}

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
  obs_linearize(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
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

constexpr int kSyrkTile = 64;                          // output tile edge
constexpr int kSyrkKc = 32;                            // rows of Ud per LDS panel

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
constexpr int kDenseRhsRows = 32;
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
    obs_linearize(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
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
constexpr int kPtGroupMaxL = 24;    // (= kGm3MaxL: every scene the matrix-core reduction takes also takes the group-packed point kernels)
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
__global__ __launch_bounds__(kBlock) void k_linearize_groups(DevProblem P, const double* __restrict__ cams,
                                                             const double* __restrict__ X,
                                                             const SchurGroup* __restrict__ groups, int ngroups,
                                                             double* __restrict__ HCC, double* __restrict__ bC,
                                                             double* __restrict__ HPP, double* __restrict__ bP) {
  __shared__ double sx[kBlock / kWave][9][64];
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (HCC) {                                           // as k_linearize: the camera-block kernels accumulate with atomics
    const long long nthreads = (long long)gridDim.x * kBlock;
    for (long long i = tid; i < (long long)P.nc * 36; i += nthreads) HCC[i] = 0.0;
    for (long long i = tid; i < (long long)P.nc * 6; i += nthreads) bC[i] = 0.0;
  }
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = blockIdx.x * (kBlock / kWave) + wv;
  if (g >= ngroups) return;                            // whole wavefront
  const SchurGroup gr = groups[g];
  const int L = gr.L, NP = 64 / L;
  const int slot = lane / L, oi = lane - slot * L;
  const bool stager = lane < NP * L;
  const int n0 = P.pt_off[gr.pt_begin] + oi;
  double cm[12];
  load_cam(cams, P.obs_cam[stager ? n0 : P.pt_off[gr.pt_begin]], cm);
  double (*mx)[64] = sx[wv];
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
      }
    }
    lds_wave_sync();
  }
}

// With `host` given (ba_lm_trial: the trial parameter set is written here as well) the kernel also
// evaluates compute_cost of the TRIAL set (bundle_adjuster.py:165-171) - k_cost's work: the lanes of a
// point hold its observations and the old camera; the updated camera R exp(sign dC), t + sign dt is
// formed once per group per lane, the updated point comes from the point's first lane through LDS.
// Partials and status words go where k_cost puts them (one partial per workgroup, <= kCostBlocks).
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
  __shared__ double sx[kBlock / kWave][3][64], sw[kBlock / kWave][3][64], wsum[kBlock / kWave];
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
  double (*mx)[64] = sx[wv];
  double (*mw)[64] = sw[wv];
  const bool want_cost = host != nullptr && X_dst != nullptr;
  double cost_acc = 0.0;
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
    for (int kb = gr.pt_begin; kb < gr.pt_end; kb += NP) {
      const int k = kb + slot;
      const bool live = stager && k < gr.pt_end;
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
#pragma unroll
      for (int q = 0; q < 3; ++q) mx[q][lane] = loc[q];
      lds_wave_sync();
      if (live) {                                        // lane q of a point adds component q of its L terms
        for (int q = oi; q < 3; q += L) {
          const double sum = lds_sum_in_order(&mx[q][slot * L], L);
          mw[q][slot] = (q == oi ? bpv : bP[3 * (size_t)k + q]) - sum;      // (L < 3: a lane adds more than one component)
        }
      }
      lds_wave_sync();
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
  }
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

// --------------------------------------------------------------------------
// solve_motion_normal_eqns, the flatten + mask step (bundle_adjuster.py:290-299):
// A[r,c] = S.transpose(0,2,1,3).reshape(6nco,6nco)[keep[r], keep[c]] from the block band.
// Used when the band is too wide for k_band_solve (dense LU on the GPU instead).
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_flatten(int nco, int hb, int nkeep, const int* __restrict__ keep,
                                                    const double* __restrict__ S,
                                                    const double* __restrict__ b, double* __restrict__ Aout,
                                                    double* __restrict__ rhs) {
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (tid >= (long long)nkeep * nkeep) return;
  const int r = (int)(tid / nkeep), c = (int)(tid % nkeep);
  const int p = keep[r], q = keep[c];
  const int i = p / 6, a = p % 6, j = q / 6, d = q % 6;
  double v = 0.0;
  if (i <= j) {
    if (j - i <= hb) v = S[band_block(i, j, hb + 1) + a * 6 + d];
  } else if (i - j <= hb) {
    v = S[band_block(j, i, hb + 1) + d * 6 + a];
  }
  Aout[tid] = v;
  if (c == 0) rhs[r] = b[p];
}

// --------------------------------------------------------------------------
// solve_motion_normal_eqns on the device (bundle_adjuster.py:281-312) for a block-banded
// reduced system: S x = b by block Cholesky S = U^T U, forward and backward
// substitution, all in ONE workgroup that slides an LDS window of the last hb block rows
// of U down the band (left-looking):
//   row j:  B[d] = S[j,j+d] - sum_{m=1..hb} U[j-m,j]^T U[j-m,j+d]      (d = 0..hb)
//           U[j,j] = chol(B[0]);  U[j,j+d] = U[j,j]^-T B[d];  y_j likewise from b
//   then    x_j = U[j,j]^-1 (y_j - sum_d U[j,j+d] x_{j+d})  for j = nco-1 .. 0.
// Masked camera parameters (param_mask) become identity rows/columns with zero rhs,
// which deletes them from the system exactly as the reference's row/column deletion
// does and leaves x = 0 there.  S is SPD whenever the reference's LU solve is
// meaningful; a non-positive pivot is reported through *info (caller falls back to
// the dense LU path, which reproduces the reference's LinAlgError semantics).
// LDS: hb*(hb+1)*288 B ring + one row; hb <= kMaxBandSolve.
// --------------------------------------------------------------------------
constexpr int kMaxBandSolve = 21;
constexpr int kSolveThreads = 256;

// LDS budget of k_band_solve: the U window, two row buffers and a staging area of `ch`
// band rows (S on the way down, U on the way back) so that global latency is paid once
// per chunk instead of once per row.
// LDS row stride (doubles) of the U window: padded so that the HB rows a wavefront reads
// together (one per lane of a group, `36` doubles further along in each older row) fall on
// distinct LDS banks for ds_read2_b64 (32 banks of 4 B): (stride - 36) % 16 == 2.
__host__ __device__ constexpr int band_ring_stride(int hb) {
  return (hb + 1) * 36 + ((2 - 36 * hb) % 16 + 16) % 16;
}
__host__ __device__ inline size_t band_solve_fixed_doubles(int hb) {
  const size_t hbm = hb > 0 ? hb : 1;
  return hbm * band_ring_stride(hb) + hbm * 6 * 2 + 2 * ((size_t)(hb + 1) * 36 + 6) + 16 * 6 + 8;
}
__host__ __device__ inline size_t band_solve_row_doubles(int hb) { return (size_t)(hb + 1) * 36 + 12; }
__host__ __device__ inline int band_solve_chunk(int hb, size_t lds_bytes) {
  const size_t fixed = band_solve_fixed_doubles(hb) * 8 + (size_t)(hb + 64) * 6 + 64;
  if (lds_bytes <= fixed) return 0;
  size_t ch = (lds_bytes - fixed) / (band_solve_row_doubles(hb) * 8 + 6);
  return (int)(ch > 32 ? 32 : ch);
}
__host__ __device__ inline size_t band_solve_lds_bytes(int hb, int ch) {
  return band_solve_fixed_doubles(hb) * 8 + (size_t)ch * band_solve_row_doubles(hb) * 8 + (size_t)(ch + hb) * 6 + 64;
}



// Pipelined left-looking schedule (HB = block half-bandwidth, compile-time so that the
// sums over the HB previous rows are fully unrolled and their LDS loads batched):
//   phase A  wavefront 0: chol + panel of row j (the serial critical path)
//            wavefronts 1..7: row j+1 minus the contributions of rows j+1-HB .. j-1
//   phase B  all: row j+1 minus the contribution of row j (which phase A just produced)
// so the O(HB^2) update of the next row hides behind the serial 6x6 factorisation.
template <int HB, bool MASKED>
__global__ __launch_bounds__(kSolveThreads) void k_band_solve(int nco, int ch, const double* __restrict__ S,
                                                              const double* __restrict__ b,
                                                              const unsigned char* __restrict__ mask,
                                                              double* __restrict__ U, double* __restrict__ y,
                                                              double* __restrict__ dinvg, double* __restrict__ x,
                                                              int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int HB1 = HB + 1, ROWLEN = HB1 * 36, NTASK = ROWLEN + 6, HBM = HB > 0 ? HB : 1;
  constexpr int RS = band_ring_stride(HB);            // padded LDS stride of a ring row
  const int tid = threadIdx.x;
  double* ring = sm;                                  // [HBM][RS]       rows j-HB .. j-1 of U (zero = no row)
  double* yring = ring + (size_t)HBM * RS;            // [HBM][6]
  double* xring = yring + HBM * 6;                    // [HBM][6]        (backward pass)
  double* Bbuf = xring + HBM * 6;                     // [2][NTASK]      row being factored / row being built
  double* part = Bbuf + 2 * NTASK;                    // [16][6]         partial sums (backward pass)
  int* bad = reinterpret_cast<int*>(part + 16 * 6);
  double* stage = part + 16 * 6 + 8;                  // [ch][ROWLEN]    chunk of S (forward) / U (backward)
  double* bstage = stage + (size_t)ch * ROWLEN;       // [ch][6]         chunk of b / y
  double* dstage = bstage + (size_t)ch * 6;           // [ch][6]         chunk of 1/diag (backward)
  unsigned char* mstage = reinterpret_cast<unsigned char*>(dstage + (size_t)ch * 6);   // [(ch+HB)*6]
  if (tid == 0) *bad = 0;
  for (int i = tid; i < HBM * RS + HBM * 6; i += kSolveThreads) ring[i] = 0.0;         // ring + yring
  long long t_c0 = 0, t_w0 = 0;
  if (tid == 0) { t_c0 = clock64(); t_w0 = wall_clock64(); }

  // stage `rows` contiguous band rows of S, b and the mask starting at row j0
  auto stage_chunk = [&](int j0) {
    const int rows = min(ch, nco - j0);
    copy_to_lds<kSolveThreads>(stage, S + (size_t)j0 * ROWLEN, rows * ROWLEN, tid);
    for (int i = tid; i < rows * 6; i += kSolveThreads) bstage[i] = b[(size_t)j0 * 6 + i];
    if (MASKED) {
      const int mc = (rows + HB) * 6;
      for (int i = tid; i < mc; i += kSolveThreads) mstage[i] = (j0 * 6 + i < nco * 6) ? mask[j0 * 6 + i] : 1;
    }
  };
  // entry tk of band row j (jj = j - chunk start), masked parameters replaced by identity rows
  auto staged = [&](int jj, int tk) -> double {
    if (tk < ROWLEN) {
      const int d = tk / 36, e = tk % 36, a = e / 6, c = e % 6;
      double v = stage[jj * ROWLEN + tk];
      if (MASKED && (!mstage[jj * 6 + a] || !mstage[(jj + d) * 6 + c])) v = (d == 0 && a == c) ? 1.0 : 0.0;
      return v;
    }
    const int a = tk - ROWLEN;
    return (MASKED && !mstage[jj * 6 + a]) ? 0.0 : bstage[jj * 6 + a];
  };
  // contribution of U row (slot) at distance m to entry tk of the row being built
  auto term = [&](int slot, int m, int tk) -> double {
    const double* row = ring + (size_t)slot * RS;
    double dot = 0.0;
    if (tk < ROWLEN) {
      const int d = tk / 36, e = tk % 36, a = e / 6, c = e % 6;
      const bool ok = m + d <= HB;
      const double* Uj = row + m * 36 + a;                          // U[r, j][:, a]
      const double* Ujd = row + (ok ? m + d : m) * 36 + c;          // U[r, j+d][:, c]
#pragma unroll
      for (int q = 0; q < 6; ++q) dot += Uj[q * 6] * Ujd[q * 6];
      return ok ? dot : 0.0;
    }
    const int a = tk - ROWLEN;
    const double* Uj = row + m * 36 + a;
    const double* yr = yring + slot * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) dot += Uj[q * 6] * yr[q];
    return dot;
  };

  stage_chunk(0);
  lds_barrier();
  for (int tk = tid; tk < NTASK; tk += kSolveThreads) Bbuf[tk] = staged(0, tk);   // row 0 has no predecessors
  int cur = 0;                                        // Bbuf[cur] = row j, Bbuf[1-cur] = row j+1
  int jslot = 0;                                      // j % HB
  int chunk0 = 0;                                     // first row of the staged chunk
  for (int j = 0; j < nco; ++j) {
    if (j + 1 < nco && j + 1 == chunk0 + ch) {        // row j+1 opens the next chunk: stage it now
      lds_barrier();
      if (*bad) {                                     // uniform (every thread reads the same LDS word);
        if (tid == 0) *info = *bad;                   // a failed pivot only produces NaNs until here
        return;
      }
      chunk0 = j + 1;
      stage_chunk(chunk0);
    }
    lds_barrier();                                    // Bbuf[cur], ring rows <= j-1 and the staged chunk are visible
    double* Brow = Bbuf + cur * NTASK;
    double* Bnext = Bbuf + (1 - cur) * NTASK;
    const int nslot = (HB > 0 && jslot + 1 == HB) ? 0 : jslot + 1;     // (j+1) % HB
    if (tid >= 64) {
      // ---- phase A, wavefronts 1..7: row j+1 from S minus the rows j+1-HB .. j-1 (m = 2..HB).
      // A group of G adjacent lanes shares one 3x3 sub-block of one 6x6 block: lane g of
      // the group owns the term m = g + 2 (36 LDS loads feed 54 FMAs, all loads in one
      // batch), then the G partial 3x3 blocks are summed with cross-lane shuffles.
      if (j + 1 < nco) {
        constexpr int NT = HB > 1 ? HB - 1 : 1;                          // number of terms m = 2..HB
        constexpr int G = NT <= 1 ? 1 : NT <= 2 ? 2 : NT <= 4 ? 4 : NT <= 8 ? 8 : NT <= 16 ? 16 : 32;
        const int jj1 = j + 1 - chunk0;
        for (int task = tid - 64; task < HB1 * 4 * G; task += kSolveThreads - 64) {
          const int g = task % G, blk = task / G;
          const int d = blk >> 2, a0 = (blk & 2) ? 3 : 0, c0 = (blk & 1) ? 3 : 0;
          const int m = g + 2;
          double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          if (HB >= 2 && m <= HB && m + d <= HB) {
            int slot = nslot - m;
            if (slot < 0) slot += HB;
            const double* row = ring + (size_t)slot * RS;
            const double* Uj = row + m * 36 + a0;                         // U[r, j+1][q][a0 .. a0+2]
            const double* Ujd = row + (m + d) * 36 + c0;                  // U[r, j+1+d][q][c0 .. c0+2]
            double ua[18], uc[18];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
#pragma unroll
              for (int i = 0; i < 3; ++i) { ua[q * 3 + i] = Uj[q * 6 + i]; uc[q * 3 + i] = Ujd[q * 6 + i]; }
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
#pragma unroll
              for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[i * 3 + k] += ua[q * 3 + i] * uc[q * 3 + k];
              }
            }
          }
#pragma unroll
          for (int i = 0; i < 9; ++i) acc[i] = group_sum<G>(acc[i]);
          if (g == 0) {
            double sv[9];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
              for (int k = 0; k < 3; ++k) sv[i * 3 + k] = staged(jj1, d * 36 + (a0 + i) * 6 + c0 + k);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
              for (int k = 0; k < 3; ++k) Bnext[d * 36 + (a0 + i) * 6 + c0 + k] = sv[i * 3 + k] - acc[i * 3 + k];
            }
          }
        }
        // right-hand side of row j+1: 6 entries, same split over m
        for (int task = tid - 64; task < 6 * G; task += kSolveThreads - 64) {
          const int g = task % G, a = task / G;
          const int m = g + 2;
          double acc = 0.0;
          if (HB >= 2 && m <= HB) {
            int slot = nslot - m;
            if (slot < 0) slot += HB;
            const double* Uj = ring + (size_t)slot * RS + m * 36 + a;
            const double* yr = yring + slot * 6;
#pragma unroll
            for (int q = 0; q < 6; ++q) acc += Uj[q * 6] * yr[q];
          }
          acc = group_sum<G>(acc);
          if (g == 0) Bnext[ROWLEN + a] = staged(jj1, ROWLEN + a) - acc;
        }
      }
    } else {
      // ---- phase A, wavefront 0: U[j,j] = chol(B[0]) (lane c owns column c), then the panel
      //      U[j,j+d] = U[j,j]^-T B[d], y_j = U[j,j]^-T rhs, and the stores of row j
      const int c = tid < 6 ? tid : 5;
      double col[6];
#pragma unroll
      for (int p = 0; p < 6; ++p) col[p] = Brow[p * 6 + c];
      int fail = 0;
      double dinv = 0.0;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const double piv = lane_bcast(col[q], q);
        if (!(piv > 0.0) && !fail) fail = q + 1;
        const double inv = rsqrt_nr(piv);
        const double uqq = piv * inv;
        if (c == q) dinv = inv;
        const double uqc = c == q ? uqq : (c > q ? col[q] * inv : 0.0);
        col[q] = uqc;
#pragma unroll
        for (int p = q + 1; p < 6; ++p) {
          const double uqp = lane_bcast(uqc, p);
          if (p <= c) col[p] -= uqp * uqc;
        }
      }
      if (fail && tid == 0) *bad = 6 * j + fail;
      // every lane needs the factor: Uf[p][q] (p < q) and 1/U[q][q], broadcast from lane q
      double Uf[15], di[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        di[q] = lane_bcast(dinv, q);
#pragma unroll
        for (int p = 0; p < q; ++p) Uf[q * (q - 1) / 2 + p] = lane_bcast(col[p], q);
      }
      if (tid < 6) {                                  // diagonal block of row j (upper triangle, zeros below)
        dinvg[6 * (size_t)j + c] = dinv;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
          const double v = p <= c ? col[p] : 0.0;
          if (HB > 0) ring[(size_t)jslot * RS + p * 6 + c] = v;
          U[(size_t)j * ROWLEN + p * 6 + c] = v;
        }
      }
      for (int tk = tid; tk < HB * 6 + 1; tk += 64) {
        const bool isrhs = tk == HB * 6;
        const int d = isrhs ? 0 : 1 + tk / 6, cc = isrhs ? 0 : tk % 6;
        double v[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = isrhs ? Brow[ROWLEN + q] : Brow[d * 36 + q * 6 + cc];
#pragma unroll
        for (int q = 0; q < 6; ++q) {                  // forward substitution with U[j,j]^T (lower)
          double t = v[q];
#pragma unroll
          for (int p = 0; p < q; ++p) t -= Uf[q * (q - 1) / 2 + p] * v[p];
          v[q] = t * di[q];
        }
        if (isrhs) {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            if (HB > 0) yring[jslot * 6 + q] = v[q];
            y[6 * (size_t)j + q] = v[q];
          }
        } else {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            ring[(size_t)jslot * RS + d * 36 + q * 6 + cc] = v[q];
            U[(size_t)j * ROWLEN + d * 36 + q * 6 + cc] = v[q];
          }
        }
      }
    }
    if (HB > 0 && j + 1 < nco) {
      lds_barrier();                                  // U row j (ring) and the partial row j+1 are visible
      // ---- phase B, all wavefronts: row j+1 minus the contribution of row j (m = 1)
      for (int tk = tid; tk < NTASK; tk += kSolveThreads) Bnext[tk] -= term(jslot, 1, tk);
    }
    cur = 1 - cur;
    jslot = nslot;
  }
  __syncthreads();     // full barrier: U, y (global) and *bad of this workgroup are visible
  if (*bad) {
    if (tid == 0) *info = *bad;
    return;
  }
  if (tid == 0) {      // instrumentation: shader cycles / 100 MHz wall ticks of the forward sweep
    info[2] = (int)(clock64() - t_c0);
    info[3] = (int)(wall_clock64() - t_w0);
  }

  // ---- backward substitution, rows nco-1 .. 0, again in chunks staged through LDS by the
  // whole workgroup; the recurrence itself runs on wavefront 0 alone (no barriers inside a
  // chunk).  lane (a = lane % 6, g = lane / 6) owns row a of blocks d = g, g + 10, g + 20.
  const int a_ = tid % 6, g_ = tid / 6;               // g_ in 0..10 for wavefront 0 (lanes 60..63 idle)
  constexpr int NG = HB1 < 10 ? HB1 : 10;
  int xslot = HB > 0 ? (nco - 1) % HB : 0;            // slot of row j in xring
  for (int jend = nco; jend > 0; jend -= ch) {
    const int jbeg = max(0, jend - ch), rows = jend - jbeg;
    lds_barrier();                                    // wavefront 0 is done with the previous chunk
    copy_to_lds<kSolveThreads>(stage, U + (size_t)jbeg * ROWLEN, rows * ROWLEN, tid);
    for (int i = tid; i < rows * 6; i += kSolveThreads) {
      bstage[i] = y[(size_t)jbeg * 6 + i];
      dstage[i] = dinvg[(size_t)jbeg * 6 + i];
    }
    lds_barrier();
    if (tid >= 64) continue;
    for (int jj = rows - 1; jj >= 0; --jj) {
      const int j = jbeg + jj;
      const double* urow = stage + (size_t)jj * ROWLEN;
      if (g_ < NG) {
        double sacc = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int d = g_ + 10 * r;
          if (d >= 1 && d <= HB && j + d < nco) {
            int sl = xslot + d;
            if (sl >= HB) sl -= HB;
            const double* xr = xring + sl * 6;
            const double* ur = urow + d * 36 + a_ * 6;
#pragma unroll
            for (int c = 0; c < 6; ++c) sacc += ur[c] * xr[c];
          }
        }
        part[g_ * 6 + a_] = sacc;
      }
      // lanes 0..5 fetch what the second half needs while the partial sums land
      double t = 0.0, dv = 1.0, ud[6] = {0, 0, 0, 0, 0, 0};
      if (g_ == 0) {
        t = bstage[jj * 6 + a_];
        dv = dstage[jj * 6 + a_];
#pragma unroll
        for (int c = 0; c < 6; ++c) ud[c] = urow[a_ * 6 + c];
      }
      lds_wave_sync();
      if (g_ == 0) {
        double pg[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) pg[g] = part[g * 6 + a_];
#pragma unroll
        for (int g = 0; g < NG; ++g) t -= pg[g];
      }
      double xs = 0.0;
#pragma unroll
      for (int q = 5; q >= 0; --q) {
        const double xq = lane_bcast(t * dv, q);        // x_q, final once rows > q were eliminated
        if (a_ == q) xs = xq;
        if (a_ < q) t -= ud[q] * xq;
      }
      if (g_ == 0) {
        if (HB > 0) xring[xslot * 6 + a_] = xs;
        x[6 * (size_t)j + a_] = xs;
      }
      lds_wave_sync();
      xslot = xslot == 0 ? (HB > 0 ? HB - 1 : 0) : xslot - 1;
    }
  }
  if (tid == 0) {
    info[4] = (int)(clock64() - t_c0);
    info[5] = (int)(wall_clock64() - t_w0);
    *info = 0;
  }
}

}  // namespace ba
