// ba_kernels.h - gfx950 (MI355X, CDNA4) kernels of the bundle-adjustment inner loop.
//
// Everything here is fp64 on blocks no larger than 6x6, so nothing is MFMA-shaped:
// the kernels are HBM / L2-atomic bound.  Layout rules used throughout:
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
constexpr int kTile = 32;         // observations of one track staged per Schur work unit

struct DevProblem {
  int nc, nt, nco;
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

// hardware fp64 atomic add (global_atomic_add_f64 / ds_add_f64 on gfx950)
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

// --------------------------------------------------------------------------
// compute_cost (bundle_adjuster.py:165-171): one observation per lane,
// wavefront + block reduction, one partial per block (second stage is
// deterministic: k_sum_partials).
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_cost(DevProblem P, const double* __restrict__ cams,
                                                 const double* __restrict__ X,
                                                 double* __restrict__ partial) {
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
    partial[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kBlock) void k_sum_partials(const double* __restrict__ partial, int n,
                                                         double* __restrict__ out) {
  __shared__ double wsum[kBlock / kWave];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) s += wsum[w];
    out[0] = s;
  }
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
// written once (deterministic); HCC / bC go to the camera records with fp64
// atomics (upper triangle of HCC only).  W is written only on request.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_linearize(DevProblem P, const double* __restrict__ cams,
                                                      const double* __restrict__ X, int glog,
                                                      double* __restrict__ HCC, double* __restrict__ bC,
                                                      double* __restrict__ HPP, double* __restrict__ bP,
                                                      double* __restrict__ Wout) {
  const int G = 1 << glog;
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
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
    double* hc = HCC + (size_t)c * 36;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = a; b < 6; ++b) atomic_add_f64(hc + a * 6 + b, Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]);
      atomic_add_f64(bC + (size_t)c * 6 + a, Jc[a] * r[0] + Jc[6 + a] * r[1]);
    }
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
  if (valid && l == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) HPP[6 * k + i] = hpp[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) bP[3 * k + i] = bp[i];
  }
}

// --------------------------------------------------------------------------
// apply_damping on HPP (bundle_adjuster.py:241-242, optimize.py:7-9) and the
// per-point inverse (bundle_adjuster.py:252-256).  One point per lane.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_point_invert(int nt, const double* __restrict__ HPP,
                                                         double damping, double rcond,
                                                         double* __restrict__ HPPinv,
                                                         int* __restrict__ singular_count) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= nt) return;
  double A[6], out[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) A[i] = HPP[6 * (size_t)k + i];
  const double f = 1.0 + damping;
  A[0] *= f; A[3] *= f; A[5] *= f;
  if (rcond >= 0.0) {
    sym3_pinv(A, rcond, out);
  } else if (!sym3_inv(A, out)) {
    atomicAdd(singular_count, 1);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) HPPinv[6 * (size_t)k + i] = out[i];
}

// --------------------------------------------------------------------------
// S[pos,pos] = damped HCC, b[pos] = bC for optimised cameras
// (bundle_adjuster.py:238-240, 263-265).  S was zero-filled by the caller.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_schur_init(int nc, int nco, const int* __restrict__ cam_opt_pos,
                                                       const double* __restrict__ HCC,
                                                       const double* __restrict__ bC, double damping,
                                                       double* __restrict__ S, double* __restrict__ b) {
  const int tid = blockIdx.x * kBlock + threadIdx.x;
  const int i = tid / 36, e = tid % 36;
  if (i >= nc) return;
  const int pos = cam_opt_pos[i];
  if (pos < 0) return;
  const int a = e / 6, c = e % 6;
  const int lo = a < c ? a : c, hi = a < c ? c : a;
  double v = HCC[(size_t)i * 36 + lo * 6 + hi];
  if (a == c) v *= (1.0 + damping);
  S[((size_t)pos * nco + pos) * 36 + e] = v;
  if (e < 6) b[(size_t)pos * 6 + e] = bC[(size_t)i * 6 + e];
}

// --------------------------------------------------------------------------
// compute_schur_complement, the reduction (bundle_adjuster.py:267-276):
//   b[i]   -= W_ik HPPinv_k bP_k
//   S[i,j] -= W_ik HPPinv_k W_jk^T     over the observation pairs of each point.
// One wavefront per work unit (point k, row tile r, col tile c >= r) of at most
// kTile x kTile observation pairs.  Phase A: lanes recompute W for the tile's
// observations and stage T = W HPPinv (rows) and W (cols) in LDS.  Phase B: the
// 64 lanes walk the (pair, entry) list so that 36 consecutive lanes hit the 36
// contiguous doubles of one 6x6 block: fp64 atomics on whole 288-byte blocks.
// Only the upper block triangle (pos_i <= pos_j) is accumulated; S is symmetric
// (ba_mirror_reduced / k_flatten fill the rest).
// --------------------------------------------------------------------------
struct SchurUnit { int pt; int row0; int col0; };

__global__ __launch_bounds__(kBlock) void k_schur_pairs(DevProblem P, const double* __restrict__ cams,
                                                        const double* __restrict__ X,
                                                        const SchurUnit* __restrict__ units, int nunits,
                                                        const double* __restrict__ HPPinv,
                                                        const double* __restrict__ bP,
                                                        double* __restrict__ S, double* __restrict__ b) {
  __shared__ double sT[kBlock / kWave][kTile][18];
  __shared__ double sW[kBlock / kWave][kTile][18];
  __shared__ int sPosR[kBlock / kWave][kTile];
  __shared__ int sPosC[kBlock / kWave][kTile];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int u = blockIdx.x * (kBlock / kWave) + wv;
  const bool active = u < nunits;     // wave-uniform; no early return (block barrier below)
  SchurUnit un = {0, 0, 0};
  int k = 0, s = 0, L = 0, nr = 0, ncol = 0;
  bool diag = false;
  if (active) {
    un = units[u];
    k = un.pt;
    s = P.pt_off[k];
    L = P.pt_off[k + 1] - s;
    nr = min(kTile, L - un.row0);
    ncol = min(kTile, L - un.col0);
    diag = un.row0 == un.col0;
  }

  // ---- phase A: lanes [0, nr) stage rows, lanes [32, 32+ncol) stage columns
  {
    const bool isRow = lane < kTile;
    const int idx = isRow ? lane : lane - kTile;
    const int cnt = isRow ? nr : ncol;
    const bool need = active && idx < cnt && !(diag && !isRow);   // diagonal tile: cols = rows
    if (need) {
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
        for (int i = 0; i < 18; ++i) sT[wv][idx][i] = T[i];
        sPosR[wv][idx] = pos;
        if (diag) {
#pragma unroll
          for (int i = 0; i < 18; ++i) sW[wv][idx][i] = W[i];
          sPosC[wv][idx] = pos;
          if (pos >= 0) {   // b[i] -= T_i bP_k, once per observation
            const double g0 = bP[3 * (size_t)k], g1 = bP[3 * (size_t)k + 1], g2 = bP[3 * (size_t)k + 2];
#pragma unroll
            for (int a = 0; a < 6; ++a)
              atomic_add_f64(b + (size_t)pos * 6 + a, -(T[a * 3] * g0 + T[a * 3 + 1] * g1 + T[a * 3 + 2] * g2));
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 18; ++i) sW[wv][idx][i] = W[i];
        sPosC[wv][idx] = pos;
      }
    }
  }
  __syncthreads();
  if (!active) return;

  // ---- phase B
  const size_t nco = (size_t)P.nco;
  for (int i = 0; i < nr; ++i) {
    const int pi = sPosR[wv][i];
    if (pi < 0) continue;                       // wave-uniform
    const int j0 = diag ? i : 0;
    const int items = (ncol - j0) * 36;
    for (int q = lane; q < items; q += 64) {
      const int j = j0 + q / 36, e = q % 36;
      const int pj = sPosC[wv][j];
      if (pj < 0) continue;
      const int a = e / 6, c = e % 6;
      const double v = sT[wv][i][a * 3] * sW[wv][j][c * 3] + sT[wv][i][a * 3 + 1] * sW[wv][j][c * 3 + 1] +
                       sT[wv][i][a * 3 + 2] * sW[wv][j][c * 3 + 2];
      // block (pi,pj) entry (a,c); keep the upper block triangle
      const size_t off = pi <= pj ? ((size_t)pi * nco + pj) * 36 + a * 6 + c
                                  : ((size_t)pj * nco + pi) * 36 + c * 6 + a;
      atomic_add_f64(S + off, -v);
    }
  }
}

// --------------------------------------------------------------------------
// backsubstitute (bundle_adjuster.py:316-331):
//   dP_k = HPPinv_k (bP_k - sum_i W_ik^T dC_i),  W^T dC = Jp^T (Jc dC).
// dCfull[nc*6] holds zeros for cameras that are not optimised.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_backsub(DevProblem P, const double* __restrict__ cams,
                                                    const double* __restrict__ X, int glog,
                                                    const double* __restrict__ dCfull,
                                                    const double* __restrict__ HPPinv,
                                                    const double* __restrict__ bP, double* __restrict__ dP) {
  const int G = 1 << glog;
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
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
    if (P.cam_opt_pos[c] < 0) continue;
    const double2 z = P.obs_z[n];
    double cm[12], e[2], r[2], Jc[12], Jp[6];
    load_cam(cams, c, cm);
    obs_linearize(P.K, cm, x, z.x, z.y, P.sensor, e, r, Jc, Jp);
    const double* d = dCfull + (size_t)c * 6;
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
  }
}

// --------------------------------------------------------------------------
// update_motion / update_structure (bundle_adjuster.py:334-343): dst = src (+) sign*delta
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_apply_update(int nc, int nt, const int* __restrict__ cam_opt_pos,
                                                         const unsigned char* __restrict__ pt_opt,
                                                         const double* __restrict__ cams_src,
                                                         const double* __restrict__ X_src,
                                                         const double* __restrict__ dCfull,
                                                         const double* __restrict__ dP, double sign,
                                                         double* __restrict__ cams_dst,
                                                         double* __restrict__ X_dst) {
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (tid < nc) {
    const int i = (int)tid;
    double cm[12], out[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) cm[q] = cams_src[(size_t)i * 12 + q];
    if (cam_opt_pos[i] >= 0) {
      double d[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) d[q] = sign * dCfull[(size_t)i * 6 + q];
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
// S is accumulated as an upper block triangle; fill the lower one in place.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_mirror(int nco, double* __restrict__ S) {
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long blk = tid / 36;
  const int e = (int)(tid % 36);
  if (blk >= (long long)nco * nco) return;
  const int i = (int)(blk / nco), j = (int)(blk % nco);
  if (i <= j) return;
  const int a = e / 6, c = e % 6;
  S[(size_t)blk * 36 + e] = S[((size_t)j * nco + i) * 36 + c * 6 + a];
}

// --------------------------------------------------------------------------
// solve_motion_normal_eqns, the flatten + mask step (bundle_adjuster.py:290-299):
// A[r,c] = S.transpose(0,2,1,3).reshape(6nco,6nco)[keep[r], keep[c]].
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_flatten(int nco, int nkeep, const int* __restrict__ keep,
                                                    const double* __restrict__ S,
                                                    const double* __restrict__ b, double* __restrict__ Aout,
                                                    double* __restrict__ rhs) {
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (tid >= (long long)nkeep * nkeep) return;
  const int r = (int)(tid / nkeep), c = (int)(tid % nkeep);
  const int p = keep[r], q = keep[c];
  const int i = p / 6, a = p % 6, j = q / 6, d = q % 6;
  const double v = i <= j ? S[((size_t)i * nco + j) * 36 + a * 6 + d]
                          : S[((size_t)j * nco + i) * 36 + d * 6 + a];
  Aout[tid] = v;
  if (c == 0) rhs[r] = b[p];
}

}  // namespace ba
