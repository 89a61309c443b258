// ba_setup_kernels.h - ba_set_problem on the device (set_bundle, bundle_adjuster.py:54-114: the reference's id / index
// bookkeeping, here also the internal order and the work lists of the kernels).  The observations arrive in any order;
// the device validates them, orders them by (track, camera rank), orders the tracks by (first optimised position, camera
// list), builds the CSR arrays and the per-point summaries the host plans the work lists from (O(points) on the host,
// O(observations) only here).  Integer work, bound by HBM and atomics; nothing here is on the trial's path.
#pragma once

#include "ba_device.h"

namespace ba {

typedef unsigned long long u64;

// words of the set-up status record
enum {
  SF_BAD = 0,        // smallest index of an observation whose camera / track is out of range (INT_MAX: none)
  SF_UNSORTED,       // 1: the observations are not ascending by (track, camera rank)
  SF_UNSORTED_PT,    // 1: ... not even grouped by ascending track
  SF_DUP,            // smallest (sorted) position of a (camera, track) pair observed twice (INT_MAX: none)
  SF_PERM,           // 1: the internal track order differs from the caller's
  SF_DESC,           // 1: first optimised positions descend somewhere along the caller's track order
  SF_RUNS_ORIG,      // places where the track key changes, caller's order
  SF_RUNS_SORTED,    // ... sorted order (equal to the former: tracks with one camera list are already adjacent)
  SF_NOT_ASC,        // 1: optimised positions do not ascend along some track (only without the internal sort)
  SF_OPERM,          // 1: the internal observation order differs from the caller's
  SF_MAXL,           // longest track
  SF_HB,             // widest spread of optimised positions inside a track
  SF_COUNT = 16
};

__global__ __launch_bounds__(256) void k_setup_init(int* flags) {
  if (threadIdx.x < SF_COUNT) flags[threadIdx.x] = (threadIdx.x == SF_BAD || threadIdx.x == SF_DUP) ? 0x7fffffff : 0;
}

__global__ __launch_bounds__(256) void k_iota(int n, int* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = i;
}

// key of observation n = (track << rank_bits) | rank of its camera; per-track counts; is the input already in key order?
__global__ __launch_bounds__(256) void k_setup_keys(long long N, int nc, int nt, const int* __restrict__ rc, const int* __restrict__ rp,
                                                    const int* __restrict__ crank, int rank_bits, u64* __restrict__ keys,
                                                    int* __restrict__ cnt, int* __restrict__ flags) {
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int c = rc[n], k = rp[n];
  if ((unsigned)c >= (unsigned)nc || (unsigned)k >= (unsigned)nt) {
    atomicMin(flags + SF_BAD, (int)n);
    keys[n] = ~0ull;
    return;
  }
  const u64 key = ((u64)k << rank_bits) | (u64)crank[c];
  keys[n] = key;
  atomicAdd(cnt + k, 1);
  if (n > 0) {
    const int c1 = rc[n - 1], k1 = rp[n - 1];
    if ((unsigned)c1 < (unsigned)nc && (unsigned)k1 < (unsigned)nt) {
      const u64 key1 = ((u64)k1 << rank_bits) | (u64)crank[c1];
      if (key1 > key) flags[SF_UNSORTED] = 1;
      if (k1 > k) flags[SF_UNSORTED_PT] = 1;
      if (key1 == key) atomicMin(flags + SF_DUP, (int)n);
    }
  }
}

// keys sorted: a (camera, track) pair observed twice shows as two equal neighbours
__global__ __launch_bounds__(256) void k_setup_dups(long long N, const u64* __restrict__ keys, int* __restrict__ flags) {
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n > 0 && n < N && keys[n] == keys[n - 1]) atomicMin(flags + SF_DUP, (int)n);
}

__global__ __launch_bounds__(256) void k_setup_track_keys_only(long long N, const int* __restrict__ rp, u64* __restrict__ keys) {
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n < N) keys[n] = (u64)rp[n];
}

// per caller track k (its observations: by_pt[coff[k] .. coff[k + 1]), by_pt == nullptr: the identity): the key the tracks
// are ordered by = (first optimised position, or nco if it has none) << 32 | 32 bits of a hash of its camera-rank list.
// Tracks with one camera list get one key: the stable sort by it makes them adjacent and lets the windows of the
// reduction slide along the band; the order among different lists that start at the same camera does not matter.
__global__ __launch_bounds__(256) void k_setup_track_keys(int nt, int nco, const int* __restrict__ coff, const int* __restrict__ by_pt,
                                                          const int* __restrict__ rc, const int* __restrict__ cam_opt_pos,
                                                          const int* __restrict__ crank, u64* __restrict__ tkey) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nt) return;
  const int b = coff[k], e = coff[k + 1];
  u64 hsh = 0x9E3779B97F4A7C15ull ^ (u64)(e - b);
  int minpos = nco;
  for (int q = b; q < e; ++q) {
    const int c = rc[by_pt ? by_pt[q] : q];
    const int p = cam_opt_pos[c];
    if (p >= 0 && p < minpos) minpos = p;
    hsh ^= (u64)crank[c] + 0x9E3779B97F4A7C15ull + (hsh << 6) + (hsh >> 2);
    hsh *= 0xD6E8FEB86659FD93ull;
  }
  hsh ^= hsh >> 32;
  tkey[k] = ((u64)minpos << 32) | (hsh & 0xffffffffull);
}

// is the caller's track order as good as the sorted one?  (first positions ascending, equal keys adjacent)
__global__ __launch_bounds__(256) void k_setup_order_check(int nt, const u64* __restrict__ tkey, const u64* __restrict__ tsorted,
                                                           int* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool ro = false, rs = false;
  if (i > 0 && i < nt) {
    if ((tkey[i] >> 32) < (tkey[i - 1] >> 32)) flags[SF_DESC] = 1;
    ro = tkey[i] != tkey[i - 1];
    rs = tsorted[i] != tsorted[i - 1];
  }
  const int no = __popcll(__ballot(ro)), ns = __popcll(__ballot(rs));
  if ((threadIdx.x & 63) == 0) {
    if (no) atomicAdd(flags + SF_RUNS_ORIG, no);
    if (ns) atomicAdd(flags + SF_RUNS_SORTED, ns);
  }
}

// the internal track order: the sorted one, or - when the caller's is as good - the caller's (host-facing arrays then need
// no permutation); track lengths in that order
__global__ __launch_bounds__(256) void k_setup_choose_order(int nt, int keep_callers, int* __restrict__ pperm, const int* __restrict__ cnt,
                                                            int* __restrict__ Lint, int* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nt) return;
  const bool ident = keep_callers || (flags[SF_DESC] == 0 && flags[SF_RUNS_ORIG] == flags[SF_RUNS_SORTED]);
  int k = pperm[i];
  if (ident) { k = i; pperm[i] = i; }
  else if (k != i) flags[SF_PERM] = 1;
  Lint[i] = cnt[k];
}

// observations into the internal order: internal point i = caller's track pperm[i], its observations in by_pt order
__global__ __launch_bounds__(256) void k_setup_gather(int nt, const int* __restrict__ pperm, const int* __restrict__ coff,
                                                      const int* __restrict__ off, const int* __restrict__ by_pt,
                                                      const int* __restrict__ rc, const double2* __restrict__ rz,
                                                      const unsigned char* __restrict__ rpo, int* __restrict__ obs_cam,
                                                      int* __restrict__ obs_pt, double2* __restrict__ obs_z, int* __restrict__ operm,
                                                      unsigned char* __restrict__ pt_opt, int* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nt) return;
  const int k = pperm[i];
  pt_opt[i] = rpo[k];
  const int src = coff[k], dst = off[i], L = off[i + 1] - dst;
  bool moved = false;
  for (int q = 0; q < L; ++q) {
    const int n = by_pt ? by_pt[src + q] : src + q;
    operm[dst + q] = n;
    obs_cam[dst + q] = rc[n];
    obs_pt[dst + q] = i;
    obs_z[dst + q] = rz[n];
    moved = moved || n != dst + q;
  }
  if (moved) flags[SF_OPERM] = 1;
}

// per internal point: lowest / highest optimised position among its cameras (INT_MAX / -1: none), whether its camera list
// equals the previous point's, and the scene-wide maxima
__global__ __launch_bounds__(256) void k_setup_point_summary(int nt, const int* __restrict__ off, const int* __restrict__ obs_cam,
                                                             const int* __restrict__ cam_opt_pos, int* __restrict__ plo,
                                                             int* __restrict__ phi, unsigned char* __restrict__ same,
                                                             int* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int L = 0, span = 0;
  if (i < nt) {
    const int b = off[i], e = off[i + 1];
    L = e - b;
    int lo = 0x7fffffff, hi = -1;
    bool asc = true;
    for (int n = b; n < e; ++n) {
      const int p = cam_opt_pos[obs_cam[n]];
      if (p < 0) continue;
      if (p <= hi) asc = false;
      lo = min(lo, p); hi = max(hi, p);
    }
    plo[i] = lo; phi[i] = hi;
    if (hi >= 0) span = hi - lo;
    if (!asc) flags[SF_NOT_ASC] = 1;
    bool s = false;
    if (i > 0) {
      const int pb = off[i - 1];
      s = b - pb == L;
      for (int q = 0; s && q < L; ++q) s = obs_cam[pb + q] == obs_cam[b + q];
    }
    same[i] = s ? 1 : 0;
  }
  // wavefront maxima, one atomic each
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { L = max(L, __shfl_xor(L, m, 64)); span = max(span, __shfl_xor(span, m, 64)); }
  if ((threadIdx.x & 63) == 0) { if (L) atomicMax(flags + SF_MAXL, L); if (span) atomicMax(flags + SF_HB, span); }
}

// (point, window column) -> observation tables of the window groups (k_schur_groups_mfma3, k_schur_wide_mfma): one workgroup
// per group; the table was cleared to -1
__global__ __launch_bounds__(256) void k_setup_fill_wtab(const WinGroup* __restrict__ groups, const int* __restrict__ off,
                                                         const int* __restrict__ obs_cam, const int* __restrict__ cam_opt_pos,
                                                         int* __restrict__ wtab) {
  const WinGroup g = groups[blockIdx.x];
  for (int q = g.pt_begin + threadIdx.x; q < g.pt_end; q += 256) {
    int* row = wtab + (size_t)g.tab + (size_t)(q - g.pt_begin) * g.W;
    for (int n = off[q]; n < off[q + 1]; ++n) {
      const int p = cam_opt_pos[obs_cam[n]];
      if (p >= 0) row[p - g.lo] = n;
    }
  }
}

// observations per camera (k_camera_blocks' units, built on first use)
__global__ __launch_bounds__(256) void k_setup_cam_hist(long long N, const int* __restrict__ obs_cam, int* __restrict__ cnt, u64* __restrict__ keys) {
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int c = obs_cam[n];
  atomicAdd(cnt + c, 1);
  keys[n] = (u64)c;
}

}  // namespace ba
