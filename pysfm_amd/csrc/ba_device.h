// ba_device.h - device-side helpers every kernel header uses: camera loads, wavefront / DPP reductions, LDS
// ordering, 1/sqrt, the fp64 atomic.
#pragma once

#include "ba_types.h"

namespace ba {

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

// hand-over between the wavefronts of a workgroup through a counter in LDS (producer / consumer pairs of the matrix-core
// reductions, the node kernels of the cyclic reduction)
__device__ __forceinline__ void gm2_wait(int* flag, int need) {
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void gm2_post(int* flag, int value, int lane) {
  lds_wave_sync();                                         // my LDS reads / writes are done (in-order LDS: and visible)
  if (lane == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace ba
