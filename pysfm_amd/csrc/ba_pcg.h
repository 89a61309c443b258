// ba_pcg.h - the reduced camera system of an UNORDERED scene: conjugate gradients with a block-Jacobi preconditioner over the
// blocks of S that the co-visibility graph defines (solve_motion_normal_eqns, bundle_adjuster.py:281-312; SURVEY 8f1 "block-Jacobi PCG").
//
// The reference solves a dense S whatever its pattern.  The direct solvers here want a band (ba_bcr*.h) or pay O(n^3)
// (ba_dense.h: 63 ms at 2000 cameras; beyond 2666 cameras only LU down the band is left: 34 s at 5000).  A photo collection in which
// every camera shares points with a few dozen others ANYWHERE has no narrow band under any order, but its S is sparse: 1-3 % of
// the blocks.  S stays where the reductions put it - the block band of the Cuthill-McKee order, mostly zeros - and a list of
// the blocks that can be non-zero (built once per problem from the tracks' camera lists, ba_pcg.hip) drives the product:
//     x = 0, r = b, z = M^-1 r, p = z          M = the 6 x 6 diagonal blocks of S (their inverses: k_pcg_minv)
//     q = S p, alpha = (r.z) / (p.q), x += alpha p, r -= alpha q, z = M^-1 r, beta = (r.z)' / (r.z), p = z + beta p
// Two launches per iteration, no grid barrier, no atomics: k_pcg_product forms p for the rows it owns and q = S p (a wavefront per
// block row, eight blocks in flight, 288 contiguous bytes each), k_pcg_update the rest; every scalar product leaves a partial sum
// per workgroup, and the NEXT launch's workgroups each add the partials up for themselves, in the same order - so all of them
// see the same alpha, beta and residual norm to the last bit and take the same decision to stop (a launch after convergence
// returns at once: the host enqueues iterations in batches and looks at the state in between).
// Masked camera parameters (param_mask) are rows and columns deleted: their entries of r, z, p, q stay exactly zero.
// A diagonal block that is not positive definite, or p.q <= 0 (S indefinite): status word > 0, as from the Cholesky solvers.
#pragma once

#include "ba_internal.h"
#include "ba_device.h"
#include "ba_math.h"

namespace ba {

constexpr int kPcgRowsPerBlock = 4;          // k_pcg_product: a wavefront per block row
constexpr int kPcgUnknownsPerBlock = 252;     // k_pcg_start / k_pcg_update: whole cameras per workgroup (42 of them), a thread per unknown, four threads idle

struct PcgState {                            // device memory, copied to the host between batches
  int last_iter;                             // the iteration the latest k_pcg_product saw
  int done_iter;                             // -1, or the iteration at which r.r <= tol^2 b.b
  int breakdown;                             // p.q <= 0 at this iteration + 1 (S is not positive definite), else 0
  int pad;
  double rr, bb;                             // r.r at last_iter, b.b
};

// sum of n partials, every thread of the workgroup gets it (same order in every workgroup: same bits)
__device__ __forceinline__ double pcg_block_sum(const double* __restrict__ part, int n, double* lds /*[kBlock]*/) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) s += part[i];
  lds[threadIdx.x] = s;
  __syncthreads();
  for (int w = kBlock / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) lds[threadIdx.x] += lds[threadIdx.x + w];
    __syncthreads();
  }
  const double out = lds[0];
  __syncthreads();
  return out;
}

// this workgroup's contribution to a scalar product -> part[blockIdx.x]
__device__ __forceinline__ void pcg_block_partial(double v, double* __restrict__ part, double* lds) {
  lds[threadIdx.x] = v;
  __syncthreads();
  for (int w = kBlock / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) lds[threadIdx.x] += lds[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = lds[0];
  __syncthreads();
}

// M^-1: inverses of the diagonal blocks (masked parameters: identity rows and columns), one thread per camera.
// A block that is not positive definite: status word = camera + 1 (smallest wins), its inverse = identity.
__global__ __launch_bounds__(kBlock) void k_pcg_minv(int nco, int hb1, const double* __restrict__ S, const int* __restrict__ udiag /* packed [S]: the list index of camera i's diagonal block */,
                                                     const unsigned char* __restrict__ mask,
                                                     double* __restrict__ minv, int* __restrict__ info) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nco) return;
  const double* D = udiag ? S + (size_t)udiag[i] * 36 : S + band_block(i, i, hb1);
  double A[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = a; b < 6; ++b) {
      const bool keep = !mask || (mask[6 * i + a] && mask[6 * i + b]);
      A[a][b] = keep ? D[a * 6 + b] : (a == b ? 1.0 : 0.0);      // (the band holds the diagonal blocks by their upper triangle)
    }
  // U^T U = A in place (upper), then W = U^-1 (upper), A^-1 = W W^T
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double d = A[k][k];
#pragma unroll
    for (int m = 0; m < k; ++m) d -= A[m][k] * A[m][k];
    ok = ok && d > 0.0;
    const double u = ok ? sqrt(d) : 1.0;
    A[k][k] = u;
#pragma unroll
    for (int c = k + 1; c < 6; ++c) {
      double v = A[k][c];
#pragma unroll
      for (int m = 0; m < k; ++m) v -= A[m][k] * A[m][c];
      A[k][c] = v / u;
    }
  }
  double W[6][6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
#pragma unroll
    for (int r = 5; r >= 0; --r) {
      if (r > c) { W[r][c] = 0.0; continue; }
      double v = r == c ? 1.0 : 0.0;
#pragma unroll
      for (int m = r + 1; m <= c; ++m) v -= A[r][m] * W[m][c];
      W[r][c] = v / A[r][r];
    }
  }
  double* out = minv + (size_t)i * 36;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) v += (m >= a && m >= b) ? W[a][m] * W[b][m] : 0.0;
      out[a * 6 + b] = ok ? v : (a == b ? 1.0 : 0.0);
    }
  if (!ok) {
    // the smallest failing camera + 1 wins (0 = fine): compare-and-swap loop on the status word
    int old = *info;
    while (old == 0 || old > i + 1) {
      const int seen = atomicCAS(info, old, i + 1);
      if (seen == old) break;
      old = seen;
    }
  }
}

// x = 0, r = b (masked), z = M^-1 r, partial sums of r.z and b.b; a thread per unknown.  Clears the state.
__global__ __launch_bounds__(kBlock) void k_pcg_start(int n, const double* __restrict__ b, const unsigned char* __restrict__ mask,
                                                      const double* __restrict__ minv, double* __restrict__ x, double* __restrict__ r,
                                                      double* __restrict__ z, double* __restrict__ part_rz, double* __restrict__ part_rr,
                                                      PcgState* __restrict__ st) {
  __shared__ double lds[kBlock];
  const int u = blockIdx.x * kPcgUnknownsPerBlock + threadIdx.x;
  double rz = 0.0, rr = 0.0;
  if (u < n && threadIdx.x < kPcgUnknownsPerBlock) {
    const int i = u / 6, a = u - 6 * i;
    double zv = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double rc = (!mask || mask[6 * i + c]) ? b[6 * i + c] : 0.0;
      zv += minv[(size_t)i * 36 + a * 6 + c] * rc;
    }
    const bool keep = !mask || mask[u];
    const double rv = keep ? b[u] : 0.0;
    zv = keep ? zv : 0.0;
    x[u] = 0.0; r[u] = rv; z[u] = zv;
    rz = rv * zv; rr = rv * rv;
  }
  pcg_block_partial(rz, part_rz, lds);
  pcg_block_partial(rr, part_rr, lds);
  if (blockIdx.x == 0 && threadIdx.x == 0) { st->last_iter = -1; st->done_iter = -1; st->breakdown = 0; st->rr = 0.0; st->bb = 0.0; }
}

// Iteration k, first half: beta from the partial sums of the two latest r.z, p = z + beta p_old for the rows this workgroup owns,
// q = S p for them, the partial sum of p.q.  nparts = partials of the update kernel's grid.
template <int WPR>      // wavefronts per block row: 1 (four rows a workgroup) or 4 (a workgroup a row: rows of dozens of blocks - 17 KB a row and iteration -
                        // are a chain of eight dependent trips to memory for one wavefront, two for four)
__global__ __launch_bounds__(kBlock) void k_pcg_product(int k, int nco, int hb1, const double* __restrict__ S /* the packed blocks */, const int* __restrict__ rowptr,
                                                        const int* __restrict__ col, const int* __restrict__ blk /* index of the (upper) block in the packed array */,
                                                        const double* __restrict__ z, const double* __restrict__ p_old,
                                                        double* __restrict__ p_new, double* __restrict__ q,
                                                        const double* __restrict__ part_rz /*[2][nparts]*/, const double* __restrict__ part_rr /*[2][nparts]*/,
                                                        int nparts, double tol2, double* __restrict__ part_pq, PcgState* __restrict__ st) {
  __shared__ double lds[kBlock];
  const double rz_new = pcg_block_sum(part_rz + (size_t)(k & 1) * nparts, nparts, lds);
  const double rr = pcg_block_sum(part_rr + (size_t)(k & 1) * nparts, nparts, lds);
  const double bb = k == 0 ? rr : st->bb;                 // (r = b at the start; written below by workgroup 0 of iteration 0, read from iteration 1 on)
  if (st->done_iter >= 0 || st->breakdown) return;        // (decided by an earlier launch: every workgroup reads the same words)
  const bool done = rr <= tol2 * bb;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->last_iter = k; st->rr = rr;
    if (k == 0) st->bb = rr;
  }
  if (done) {
    // (all workgroups take this branch together: the sums above are the same bits everywhere.  The word itself is written by the
    //  UPDATE launch of this iteration - after every workgroup of this launch has read it.)
    return;
  }
  double beta = 0.0;
  if (k > 0) {
    const double rz_old = pcg_block_sum(part_rz + (size_t)((k - 1) & 1) * nparts, nparts, lds);
    beta = rz_new / rz_old;
  }
  __shared__ double wsum[kBlock / 64][8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int RPB = kPcgRowsPerBlock / WPR;            // rows per workgroup
  const int i = blockIdx.x * RPB + wave / WPR, sub = wave % WPR;
  // lane = (column c of a block, one of eight blocks in flight): a lane fetches ITS entry of the neighbour's p and the six entries of
  // its column of the block - eight loads a block where a lane per row needed eighteen (the six entries of p twice over: z and
  // p_old): the kernel is bound by the number of load instructions (47 -> 2x us at 5000 cameras), not by the 87 MB they move
  const int c = lane & 7, ks = lane >> 3;
  const int cc = c < 6 ? c : 5;
  const bool active = c < 6;
  double y[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (i < nco) {
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    // two blocks a lane and round: the indices of both first, then all their loads, then the arithmetic - a round is two dependent
    // trips to memory, and a row of 60 blocks over four wavefronts is ONE round
    for (int e = e0 + 8 * sub + ks; e < e1; e += 16 * WPR) {
      const int eb = e + 8 * WPR;
      const bool two = eb < e1;
      const int ja = col[e], jb = two ? col[eb] : ja;
      const double* Ba = S + (size_t)blk[e] * 36;
      const double* Bb = S + (size_t)(two ? blk[eb] : blk[e]) * 36;
      double pa = z[6 * (size_t)ja + cc], pb = z[6 * (size_t)jb + cc];
      double oa = 0.0, ob = 0.0;
      if (k > 0) { oa = p_old[6 * (size_t)ja + cc]; ob = p_old[6 * (size_t)jb + cc]; }
      double sa[6], sb[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        // entry (a, c) of S_ij: stored as (a, c) of block (i, j) when i < j, as (c, a) of block (j, i) when j < i; the diagonal
        // blocks by their upper triangle
        const bool tra = ja < i || (ja == i && a > cc), trb = jb < i || (jb == i && a > cc);
        sa[a] = Ba[tra ? cc * 6 + a : a * 6 + cc];
        sb[a] = Bb[trb ? cc * 6 + a : a * 6 + cc];
      }
      pa = active ? pa + beta * oa : 0.0;
      pb = (active && two) ? pb + beta * ob : 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a) y[a] += sa[a] * pa + sb[a] * pb;
    }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double v = y[a];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    y[a] = v;
  }
  double acc = lane == 0 ? y[0] : lane == 1 ? y[1] : lane == 2 ? y[2] : lane == 3 ? y[3] : lane == 4 ? y[4] : y[5];      // lane a < 6: entry a of the row
  if (WPR > 1) {
    if (lane < 8) wsum[wave][lane] = acc;
    __syncthreads();
    if (sub == 0 && lane < 8) {
      acc = 0.0;
#pragma unroll
      for (int w = 0; w < WPR; ++w) acc += wsum[wave + w][lane];
    }
  }
  double pq = 0.0;
  if (i < nco && sub == 0 && lane < 6) {
    const size_t u = 6 * (size_t)i + lane;
    const double pv = z[u] + (k > 0 ? beta * p_old[u] : 0.0);
    p_new[u] = pv;
    q[u] = acc;
    pq = pv * acc;
  }
  pcg_block_partial(pq, part_pq, lds);
}

// Iteration k, second half: alpha = (r.z) / (p.q); x += alpha p, r -= alpha q (masked rows stay zero), z = M^-1 r; partial sums of
// the new r.z and r.r into the other parity.  A thread per unknown; nprod = partials of the product kernel's grid.
__global__ __launch_bounds__(kBlock) void k_pcg_update(int k, int n, const unsigned char* __restrict__ mask, const double* __restrict__ minv,
                                                       const double* __restrict__ p, const double* __restrict__ q, double* __restrict__ x,
                                                       double* __restrict__ r, double* __restrict__ z, double* __restrict__ part_rz,
                                                       double* __restrict__ part_rr, int nparts, const double* __restrict__ part_pq, int nprod,
                                                       double tol2, PcgState* __restrict__ st) {
  __shared__ double lds[kBlock];
  __shared__ double rs[kBlock];
  const double rz = pcg_block_sum(part_rz + (size_t)(k & 1) * nparts, nparts, lds);
  const double rr = pcg_block_sum(part_rr + (size_t)(k & 1) * nparts, nparts, lds);
  const double bb = st->bb;
  if (st->done_iter >= 0 || st->breakdown) return;
  if (rr <= tol2 * bb) {
    // converged at iteration k (the product launch of this iteration returned without touching anything): say so - the word is
    // only ever read at the START of a launch, and every workgroup of this one has read it above... not necessarily: a workgroup
    // that starts late would see it set and return, which is what it would do anyway (same branch, nothing to write).
    if (blockIdx.x == 0 && threadIdx.x == 0) st->done_iter = k;
    return;
  }
  const double pq = pcg_block_sum(part_pq, nprod, lds);
  if (!(pq > 0.0)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st->breakdown = k + 1;
    return;
  }
  const double alpha = rz / pq;
  const int u = blockIdx.x * kPcgUnknownsPerBlock + threadIdx.x;
  const bool in = u < n && threadIdx.x < kPcgUnknownsPerBlock;
  const bool keep = in && (!mask || mask[u]);
  double rv = 0.0;
  if (in) {
    x[u] += keep ? alpha * p[u] : 0.0;
    rv = keep ? r[u] - alpha * q[u] : 0.0;
    r[u] = rv;
  }
  rs[threadIdx.x] = rv;
  __syncthreads();
  double zv = 0.0;
  if (in) {
    // (the camera's six residual entries are this workgroup's: it owns whole cameras)
    const int i = u / 6, a = u - 6 * i, base = 6 * i - blockIdx.x * kPcgUnknownsPerBlock;
#pragma unroll
    for (int c = 0; c < 6; ++c) zv += minv[(size_t)i * 36 + a * 6 + c] * rs[base + c];
    zv = keep ? zv : 0.0;
  }
  if (in) z[u] = zv;
  pcg_block_partial(rv * zv, part_rz + (size_t)((k + 1) & 1) * nparts, lds);
  pcg_block_partial(rv * rv, part_rr + (size_t)((k + 1) & 1) * nparts, lds);
}

__global__ void k_pcg_set_status(int* info, int v) { *info = v; }

// The blocks of the pattern's upper triangle out of the band into ONE contiguous array, once per solve: in the band they lie 288 bytes
// here, 288 bytes there over gigabytes (5000 cameras: 7 GB) - every block of every product a page of its own; packed they are 43 MB
// that stay in the Infinity Cache from iteration to iteration.
__global__ __launch_bounds__(kBlock) void k_pcg_gather(long long nblocks, const long long* __restrict__ ublk, const double* __restrict__ S,
                                                       double* __restrict__ Sc) {
  const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (t < nblocks * 36) Sc[t] = S[ublk[t / 36] * 36 + t % 36];
}

// schur_init_body of ba_schur_kernels.h over a LIST of blocks (the upper triangle of the pattern) instead of the whole band
__global__ __launch_bounds__(kBlock) void k_schur_init_blocks(long long nblocks, const long long* __restrict__ ublk, int nco, int hb1,
                                                              const int* __restrict__ opt_cam, const double* __restrict__ HCC,
                                                              const double* __restrict__ bC, double damping, double* __restrict__ S,
                                                              double* __restrict__ b, int use_hcc, int packed) {
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (tid < nblocks * 36) {
    const int e = (int)(tid % 36);
    const long long blk = ublk[tid / 36];
    const int d = (int)(blk % hb1), pos = (int)(blk / hb1);
    double v = 0.0;
    if (d == 0 && use_hcc) {
      const int a = e / 6, c = e % 6;
      const int lo = a < c ? a : c, hi = a < c ? c : a;
      v = HCC[(size_t)opt_cam[pos] * 36 + lo * 6 + hi];
      if (a == c) v *= (1.0 + damping);
    }
    S[(packed ? tid / 36 : blk) * 36 + e] = v;
  } else if (tid < nblocks * 36 + (long long)nco * 6) {
    const long long q = tid - nblocks * 36;
    b[q] = use_hcc ? bC[(size_t)opt_cam[q / 6] * 6 + q % 6] : 0.0;
  }
}

// ---- the Schur reduction of a scene without a band, block by block (compute_schur_complement, bundle_adjuster.py:259-278):
//     S[i,j] -= sum_k W_ik HPPinv_k W_jk^T ,   b[i] -= sum_k W_ik HPPinv_k bP_k
// k_schur_pairs walks the points and scatters every product into the band with 36 global atomics (43 M of them at 5000 cameras / 600 k
// observations: 2.0 ms of a 2.6 ms trial, and a different sum order every run).  Here every block of the pattern OWNS its list of
// observation pairs (built with the pattern, ba_pcg.hip): eight lanes a block, a pair a lane and round - both observations linearised
// again, the 6 x 6 product formed in registers - the eight partial blocks added up across the lanes and subtracted from S once.  No
// atomics, the same bits every run.  A diagonal block's pairs are the camera's observations: they carry the right-hand side too.
constexpr int kSbLanes = 8;
constexpr int kSbChunk = 16;        // pairs per chunk: a camera's diagonal block has a pair per observation (120 of them where the others have 8): left whole, it was every wavefront's critical path
constexpr int kSbPartial = 48;      // doubles a chunk leaves behind: its 6 x 6 block and (diagonal blocks) 6 of the right-hand side
template <bool TABLE>
__global__ __launch_bounds__(kBlock) void k_schur_blocks(DevProblem P, const double* __restrict__ cams, const double* __restrict__ X,
                                                         const double* __restrict__ HPPinv, const double* __restrict__ bP, long long nchunks,
                                                         const int* __restrict__ cblk, const int* __restrict__ cptr,
                                                         const long long* __restrict__ ublk, const int* __restrict__ bptr,
                                                         const int2* __restrict__ pairs, int hb1, double* __restrict__ partial) {
  const long long g = ((long long)blockIdx.x * kBlock + threadIdx.x) / kSbLanes;
  const int q = threadIdx.x & (kSbLanes - 1);
  const bool valid = g < nchunks;
  const long long gg = valid ? g : nchunks - 1;
  const int u = cblk[gg];
  const int p0 = bptr[u] + kSbChunk * (int)(gg - cptr[u]);
  const int p1 = valid ? min(p0 + kSbChunk, bptr[u + 1]) : p0;
  const bool diag = ublk[u] % hb1 == 0;
  double acc[36], bacc[6];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = 0.0;
#pragma unroll
  for (int a = 0; a < 6; ++a) bacc[a] = 0.0;
  for (int e = p0 + q; e < p1; e += kSbLanes) {
    const int2 pr = pairs[e];
    const int k = P.obs_pt[pr.x];
    const double x[3] = {X[3 * (size_t)k], X[3 * (size_t)k + 1], X[3 * (size_t)k + 2]};
    double A[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) A[i] = HPPinv[6 * (size_t)k + i];
    double cm[12], er[2], r[2], Jc[12], Jp[6], Wa[18], Wb[18], T[18];
    {
      const double2 z = P.obs_z[pr.x];
      load_cam(cams, P.obs_cam[pr.x], cm);
      obs_linearize<TABLE>(P.K, cm, x, z.x, z.y, P.sensor, er, r, Jc, Jp);
      block_W(Jc, Jp, Wa);
    }
    block_T(Wa, A, T);
    if (pr.y != pr.x) {
      const double2 z = P.obs_z[pr.y];
      load_cam(cams, P.obs_cam[pr.y], cm);
      obs_linearize<TABLE>(P.K, cm, x, z.x, z.y, P.sensor, er, r, Jc, Jp);
      block_W(Jc, Jp, Wb);
    } else {
#pragma unroll
      for (int i = 0; i < 18; ++i) Wb[i] = Wa[i];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[a * 6 + c] += T[a * 3] * Wb[c * 3] + T[a * 3 + 1] * Wb[c * 3 + 1] + T[a * 3 + 2] * Wb[c * 3 + 2];
    if (diag) {
      const double g0 = bP[3 * (size_t)k], g1 = bP[3 * (size_t)k + 1], g2 = bP[3 * (size_t)k + 2];
#pragma unroll
      for (int a = 0; a < 6; ++a) bacc[a] += T[a * 3] * g0 + T[a * 3 + 1] * g1 + T[a * 3 + 2] * g2;
    }
  }
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = group_sum<kSbLanes>(acc[e]);
#pragma unroll
  for (int a = 0; a < 6; ++a) bacc[a] = group_sum<kSbLanes>(bacc[a]);
  if (!valid) return;
  double* out = partial + (size_t)gg * kSbPartial;
#pragma unroll
  for (int e = 0; e < 36; ++e)
    if ((e & (kSbLanes - 1)) == q) out[e] = acc[e];
#pragma unroll
  for (int a = 0; a < 6; ++a)
    if (a == q) out[36 + a] = bacc[a];
}

// The same in two steps, for the price of 288 bytes per observation: k_sparse_stage linearises every observation ONCE (a thread an
// observation, coalesced) and leaves T = W HPPinv and W behind; k_schur_blocks_staged is k_schur_blocks with two 144-byte loads per pair
// where that one gathers some forty scalars down a chain of indices (pair -> observation -> camera, point) and linearises twice:
// 0.35 -> 0.1x ms at 5000 cameras.  (72 instead of 228 VGPRs' worth of live values: the kernel is bound by its loads either way.)
template <bool TABLE>
__global__ __launch_bounds__(kBlock) void k_sparse_stage(DevProblem P, const double* __restrict__ cams, const double* __restrict__ X,
                                                         const double* __restrict__ HPPinv, double* __restrict__ TW) {
  const long long n = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (n >= P.nobs) return;
  const int c = P.obs_cam[n];
  if (P.cam_opt_pos[c] < 0) return;                       // (no pair of the lists names an observation of a camera that is not optimised)
  const int k = P.obs_pt[n];
  const double x[3] = {X[3 * (size_t)k], X[3 * (size_t)k + 1], X[3 * (size_t)k + 2]};
  double A[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) A[i] = HPPinv[6 * (size_t)k + i];
  const double2 z = P.obs_z[n];
  double cm[12], er[2], r[2], Jc[12], Jp[6], W[18], T[18];
  load_cam(cams, c, cm);
  obs_linearize<TABLE>(P.K, cm, x, z.x, z.y, P.sensor, er, r, Jc, Jp);
  block_W(Jc, Jp, W);
  block_T(W, A, T);
  double2* out = reinterpret_cast<double2*>(TW + (size_t)n * 36);
#pragma unroll
  for (int i = 0; i < 9; ++i) out[i] = double2{T[2 * i], T[2 * i + 1]};
#pragma unroll
  for (int i = 0; i < 9; ++i) out[9 + i] = double2{W[2 * i], W[2 * i + 1]};
}

__global__ __launch_bounds__(kBlock) void k_schur_blocks_staged(const int* __restrict__ obs_pt, const double* __restrict__ TW, const double* __restrict__ bP,
                                                                long long nchunks, const int* __restrict__ cblk, const int* __restrict__ cptr,
                                                                const long long* __restrict__ ublk, const int* __restrict__ bptr,
                                                                const int2* __restrict__ pairs, int hb1, double* __restrict__ partial) {
  const long long g = ((long long)blockIdx.x * kBlock + threadIdx.x) / kSbLanes;
  const int q = threadIdx.x & (kSbLanes - 1);
  const bool valid = g < nchunks;
  const long long gg = valid ? g : nchunks - 1;
  const int u = cblk[gg];
  const int p0 = bptr[u] + kSbChunk * (int)(gg - cptr[u]);
  const int p1 = valid ? min(p0 + kSbChunk, bptr[u + 1]) : p0;
  const bool diag = ublk[u] % hb1 == 0;
  double acc[36], bacc[6];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = 0.0;
#pragma unroll
  for (int a = 0; a < 6; ++a) bacc[a] = 0.0;
  for (int e = p0 + q; e < p1; e += kSbLanes) {
    const int2 pr = pairs[e];
    const double2* tp = reinterpret_cast<const double2*>(TW + (size_t)pr.x * 36);
    const double2* wp = reinterpret_cast<const double2*>(TW + (size_t)pr.y * 36 + 18);
    double T[18], Wb[18];
#pragma unroll
    for (int i = 0; i < 9; ++i) { const double2 t = tp[i], w = wp[i]; T[2 * i] = t.x; T[2 * i + 1] = t.y; Wb[2 * i] = w.x; Wb[2 * i + 1] = w.y; }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[a * 6 + c] += T[a * 3] * Wb[c * 3] + T[a * 3 + 1] * Wb[c * 3 + 1] + T[a * 3 + 2] * Wb[c * 3 + 2];
    if (diag) {
      const int k = obs_pt[pr.x];
      const double g0 = bP[3 * (size_t)k], g1 = bP[3 * (size_t)k + 1], g2 = bP[3 * (size_t)k + 2];
#pragma unroll
      for (int a = 0; a < 6; ++a) bacc[a] += T[a * 3] * g0 + T[a * 3 + 1] * g1 + T[a * 3 + 2] * g2;
    }
  }
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = group_sum<kSbLanes>(acc[e]);
#pragma unroll
  for (int a = 0; a < 6; ++a) bacc[a] = group_sum<kSbLanes>(bacc[a]);
  if (!valid) return;
  double* out = partial + (size_t)gg * kSbPartial;
#pragma unroll
  for (int e = 0; e < 36; ++e)
    if ((e & (kSbLanes - 1)) == q) out[e] = acc[e];
#pragma unroll
  for (int a = 0; a < 6; ++a)
    if (a == q) out[36 + a] = bacc[a];
}

// ... and a block's chunks added up in order and taken off [S | b]: a thread an entry
__global__ __launch_bounds__(kBlock) void k_schur_blocks_sum(long long nblocks, const int* __restrict__ cptr, const long long* __restrict__ ublk, int hb1,
                                                             const double* __restrict__ partial, double* __restrict__ S, double* __restrict__ b, int packed) {
  const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long u = t / kSbPartial;
  const int e = (int)(t % kSbPartial);
  if (u >= nblocks || e >= 42) return;
  const long long blk = ublk[u];
  const bool diag = blk % hb1 == 0;
  if (e >= 36 && !diag) return;
  double sum = 0.0;
  for (int c = cptr[u]; c < cptr[u + 1]; ++c) sum += partial[(size_t)c * kSbPartial + e];
  if (e < 36) S[(packed ? u : blk) * 36 + e] -= sum;
  else b[6 * (blk / hb1) + (e - 36)] -= sum;
}

static_assert(sizeof(PcgState) == sizeof(PcgStateRaw), "PcgState mirrors PcgStateRaw of ba_internal.h");

}  // namespace ba
