// ba_dist.h - the reduced camera solve SPREAD OVER THE RANKS of a sharded adjuster (one GPU each), for block-banded
// systems large enough that an all-reduce of the whole band and a replicated solve would be the end of the scaling
// (BASELINE config 5: 10 000 cameras, a 28.8 MB band, an 11-level cyclic reduction).  solve_motion_normal_eqns
// (bundle_adjuster.py:281-312) has no counterpart of this in the reference, which is single-process; the arithmetic is the
// cyclic reduction of ba_bcr.h, only its elimination tree is cut along the ranks:
//
//   * super-blocks ("nodes") of cb >= hb cameras, cb chosen so that the tree over the N = ceil(nco / cb) nodes is (nearly)
//     full: L levels, strides 1 .. 2^(L-1).  With G = 2^g ranks the nodes of the top g levels (strides >= P = 2^L / G) are
//     SEPARATORS: they cut the chain into G intervals of P - 1 nodes.  Rank r owns interval r (nodes r P .. r P + P - 2),
//     the separator to its right (node r P + P - 1) and the points whose first optimised camera lies in either.
//   * a rank's points only touch band rows of its own nodes, of its separator and of the first hb rows behind it (a track
//     spans at most hb + 1 cameras <= cb + 1).  Rows of separators and those first hb rows are SHARED ROWS: they are packed,
//     summed over the ranks (exchange 1, ~0.4 MB at config 5 / 8 ranks instead of the 28.8 MB band) and unpacked; after that
//     every rank holds the complete rows of its own interval and of all separators.
//   * local phase: every rank eliminates its own interval (strides 1 .. P/2: k_bcr_eliminate_fused over its nodes only).  The
//     updates its nodes make to the two separators next to it go to the rank's own copy of D_sep, f_sep (zero before, unless it
//     owns that separator).
//   * exchange 2: D, f of the separators (sums of the copies) and the factors P_j, Q_j of the G subtree roots (the couplings of
//     the first separator level are formed from them), ~0.6 MB.
//   * top phase: every rank eliminates the G - 1 separators (strides P .. 2^(L-1)), identically; back-substitution of the
//     separators and of the rank's own interval (k_bcr_backsolve_fused over those nodes).
//   * exchange 3: the solution dC (owners contribute, everybody else zero), 48 nco bytes: every rank moves all cameras.
//
// Three small all-reduces (and the 16 KB trial record) instead of one of the whole band, and log2(P) + g levels of which only
// the g top ones are replicated work.  The kernels here only move data between the library's arrays and the exchange buffer.
#pragma once

#include "ba_device.h"

namespace ba {

// band rows `rows[0..nrows)` (camera positions) of [S | b] <-> out[r][hb1 * 36 + 6]
__global__ __launch_bounds__(256) void k_dist_rows(int nrows, const int* __restrict__ rows, int hb1, double* __restrict__ S,
                                                    double* __restrict__ b, double* __restrict__ buf, int unpack) {
  const int w = hb1 * 36 + 6;
  const long long n = (long long)nrows * w;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    const int r = (int)(t / w), c = (int)(t - (long long)r * w);
    const int row = rows[r];
    double* p = c < hb1 * 36 ? S + (size_t)row * hb1 * 36 + c : b + (size_t)row * 6 + (c - hb1 * 36);
    if (unpack) *p = buf[t]; else buf[t] = *p;
  }
}

// D, f of the separators this rank does not own start from zero (its interval's updates add to them; the owner's copy carries
// the assembled block): sep[k] = node, owner[k] = owning rank
__global__ __launch_bounds__(256) void k_dist_zero_separators(int nsep, const int* __restrict__ sep, const int* __restrict__ owner, int rank,
                                                               int B, double* __restrict__ Dm, double* __restrict__ fm) {
  const int k = blockIdx.y;
  if (k >= nsep || owner[k] == rank) return;
  const size_t BB = (size_t)B * B;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < (int)BB + B; e += gridDim.x * 256) {
    if (e < (int)BB) Dm[(size_t)sep[k] * BB + e] = 0.0;
    else fm[(size_t)sep[k] * B + (e - (int)BB)] = 0.0;
  }
}

// exchange 2: [D_sep | f_sep] per separator, then [P_root | Q_root] per subtree root (zeros unless this rank owns the root)
__global__ __launch_bounds__(256) void k_dist_top(int nsep, const int* __restrict__ sep, int nroot, const int* __restrict__ root,
                                                   const int* __restrict__ root_owner, int rank, int B, double* __restrict__ Dm,
                                                   double* __restrict__ fm, double* __restrict__ Pm, double* __restrict__ Qm,
                                                   double* __restrict__ buf, int unpack) {
  const size_t BB = (size_t)B * B;
  const int k = blockIdx.y;
  if (k < nsep) {
    double* o = buf + (size_t)k * (BB + B);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < (int)BB + B; e += gridDim.x * 256) {
      double* p = e < (int)BB ? Dm + (size_t)sep[k] * BB + e : fm + (size_t)sep[k] * B + (e - (int)BB);
      if (unpack) *p = o[e]; else o[e] = *p;
    }
  } else if (k < nsep + nroot) {
    const int q = k - nsep;
    double* o = buf + (size_t)nsep * (BB + B) + (size_t)q * 2 * BB;
    const bool mine = root_owner[q] == rank;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 2 * (int)BB; e += gridDim.x * 256) {
      double* p = e < (int)BB ? Pm + (size_t)root[q] * BB + e : Qm + (size_t)root[q] * BB + (e - (int)BB);
      if (unpack) *p = o[e]; else o[e] = mine ? *p : 0.0;
    }
  }
}

// exchange 3: the solution entries this rank owns (cameras [lo, hi)), zeros elsewhere -> buf; unpack: buf -> dC
__global__ __launch_bounds__(256) void k_dist_solution(int n6, int lo6, int hi6, double* __restrict__ dC, double* __restrict__ buf, int unpack) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n6) return;
  if (unpack) dC[t] = buf[t];
  else buf[t] = (t >= lo6 && t < hi6) ? dC[t] : 0.0;
}

}  // namespace ba
