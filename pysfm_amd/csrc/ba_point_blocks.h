// ba_point_blocks.h - what happens to the point blocks and to [S | b] between the linearisation and the reduction of a trial
// (bundle_adjuster.py:238-256): damping + inversion of a point's 3 x 3 block, initialisation of the reduced system.  Device
// functions only: k_point_invert / k_schur_init / k_point_invert_schur_init (ba_schur_kernels.h) are launches of their own,
// k_linearize_groups (ba_obs_kernels.h) runs them at the end of a trial's linearisation - a launch less.  gfx950.
#pragma once

#include "ba_device.h"

namespace ba {

// --------------------------------------------------------------------------
// apply_damping on HPP (bundle_adjuster.py:241-242, optimize.py:7-9) and the
// per-point inverse (bundle_adjuster.py:252-256).  One point per lane.
// --------------------------------------------------------------------------
// A = the point's block (upper triangle, undamped), g = its right-hand side bP (only read with fac)
__device__ __forceinline__ void point_invert_values(size_t k, double (&A)[6], const double (&g)[3], double damping, double rcond,
                                                    double* __restrict__ HPPinv, int* __restrict__ singular_count,
                                                    double* __restrict__ fac) {
  double out[6];
  const double f = 1.0 + damping;
  A[0] *= f; A[3] *= f; A[5] *= f;
  if (rcond >= 0.0) {
    sym3_pinv_fast(A, rcond, out);
  } else if (!sym3_inv(A, out)) {
    atomicAdd(singular_count, 1);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) HPPinv[6 * k + i] = out[i];
  if (fac) {                                   // HPPinv = L D L^T and HPPinv bP for k_schur_groups_mfma2
    double ff[9];
    sym3_ldl(out, sym3_ldl_tolerance(rcond), ff, ff + 3);
    ff[6] = out[0] * g[0] + out[1] * g[1] + out[2] * g[2];
    ff[7] = out[1] * g[0] + out[3] * g[1] + out[4] * g[2];
    ff[8] = out[2] * g[0] + out[4] * g[1] + out[5] * g[2];
#pragma unroll
    for (int i = 0; i < 9; ++i) fac[9 * k + i] = ff[i];
  }
}

__device__ __forceinline__ void point_invert_body(int k, int nt, const double* __restrict__ HPP, double damping, double rcond,
                                                  double* __restrict__ HPPinv, int* __restrict__ singular_count,
                                                  int* __restrict__ next_count, const double* __restrict__ bP = nullptr,
                                                  double* __restrict__ fac = nullptr) {
  if (k == 0) *next_count = 0;      // the counter the NEXT call will use (two counters alternate: no memset launch)
  if (k >= nt) return;
  double A[6], g[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 6; ++i) A[i] = HPP[6 * (size_t)k + i];
  if (fac) {
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = bP[3 * (size_t)k + i];
  }
  point_invert_values((size_t)k, A, g, damping, rcond, HPPinv, singular_count, fac);
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

}  // namespace ba
