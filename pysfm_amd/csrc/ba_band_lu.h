// ba_band_lu.h - LU with partial pivoting of the (masked) reduced camera system, on the device, for any band width.
//
// The reference solves its reduced system with numpy.linalg.solve = LAPACK gesv (bundle_adjuster.py:302-305): LU with
// partial pivoting, which also solves systems that are NOT positive definite and fails only on an exactly zero pivot
// (LinAlgError -> NormalEquationsIllconditioned).  The Cholesky solvers of this library report such systems; this is where
// they go then, whatever the band width (nodes of up to 11 cameras keep their own LU form of the cyclic reduction,
// k_bcr_eliminate_lu).  The same pivot choices as gesv on the same matrix (largest magnitude in the column, first of equals:
// entries outside the band are zeros and never chosen), the same row operations: the solution is LAPACK's to round-off.
//
// Storage: the scalar half-bandwidth of S is bw = 6 hb + 5 (kl = ku = bw); pivoting widens U to kl + ku.  Row r keeps the
// columns [c0(r), c0(r) + W), W = min(3 bw + 1, n), c0(r) = clamp(r - bw, 0, n - W): a skewed band for narrow systems, the
// plain dense matrix once the band covers it.  Column j costs two small launches: the PIVOT step (one workgroup: arg-max of
// the column, the row exchange, the multipliers) and the UPDATE (rows j+1 .. j+kl minus multiplier x row j, over many
// workgroups); then one workgroup substitutes back.  A rare path (a not-positive-definite trial, option solver = lu):
// 2 n launches of a few microseconds each, 6000 unknowns ~ 30 ms.
#pragma once

#include "ba_device.h"

namespace ba {

constexpr int kLuThreads = 1024;
constexpr int kLuRowsPerBlock = 4;       // rows of the update per workgroup

struct LuShape { int n, bw, W; };
__host__ __device__ inline int lu_c0(const LuShape& s, int r) {
  int c = r - s.bw;
  if (c > s.n - s.W) c = s.n - s.W;
  return c < 0 ? 0 : c;
}

// band (+ mask) -> A (rows in the layout above) and the right-hand side; masked parameters become identity rows / columns
// with zero right-hand side (deleting them from the system, bundle_adjuster.py:290-299)
__global__ __launch_bounds__(256) void k_lu_assemble(LuShape s, int nco, int hb, const double* __restrict__ S, const double* __restrict__ b,
                                                     const unsigned char* __restrict__ mask, double* __restrict__ A, double* __restrict__ rhs,
                                                     int* __restrict__ info) {
  const int r = blockIdx.x;
  const int c0 = lu_c0(s, r);
  const int i = r / 6, a = r % 6;
  const bool rm = mask && !mask[r];
  for (int k = threadIdx.x; k < s.W; k += 256) {
    const int c = c0 + k;
    double v = 0.0;
    if (rm || (mask && !mask[c])) {
      v = r == c ? 1.0 : 0.0;
    } else {
      const int j = c / 6, d = c % 6;
      if (i <= j) { if (j - i <= hb) v = S[band_block(i, j, hb + 1) + a * 6 + d]; }
      else if (i - j <= hb) v = S[band_block(j, i, hb + 1) + d * 6 + a];
    }
    A[(size_t)r * s.W + k] = v;
  }
  if (threadIdx.x == 0) {
    rhs[r] = rm ? 0.0 : b[r];
    if (r == 0) *info = 0;
  }
}

// column j: pivot = the entry of largest magnitude among rows j .. j + kl (the first of equals), rows j and p exchanged over
// the columns j .. jmax (and in the right-hand side), multipliers of the rows below.  A zero pivot: *info = j + 1 (gesv's
// info), everything after it returns at once.
__global__ __launch_bounds__(kLuThreads) void k_lu_pivot(LuShape s, int j, double* __restrict__ A, double* __restrict__ rhs,
                                                         double* __restrict__ mult, int* __restrict__ info) {
  __shared__ double sv[kLuThreads / 64];
  __shared__ int si[kLuThreads / 64];
  __shared__ int sp;
  __shared__ double spiv, sajj;
  if (*info != 0) return;
  const int tid = threadIdx.x;
  const int rmax = min(s.n - 1, j + s.bw), jmax = min(s.n - 1, j + 2 * s.bw);
  double best = -1.0;
  int bi = 0x7fffffff;
  for (int r = j + tid; r <= rmax; r += kLuThreads) {
    const double v = fabs(A[(size_t)r * s.W + (j - lu_c0(s, r))]);
    if (v > best) { best = v; bi = r; }                     // (ascending r per thread: the first of equals stays)
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const double ov = __shfl_xor(best, m, 64);
    const int oi = __shfl_xor(bi, m, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    double v = sv[0]; int p = si[0];
    for (int w = 1; w < kLuThreads / 64; ++w)
      if (sv[w] > v || (sv[w] == v && si[w] < p)) { v = sv[w]; p = si[w]; }
    sp = p;
    spiv = A[(size_t)p * s.W + (j - lu_c0(s, p))];
    sajj = A[(size_t)j * s.W + (j - lu_c0(s, j))];
    if (!(v > 0.0)) *info = j + 1;                            // exactly singular (or NaN): the reference's LinAlgError
  }
  __syncthreads();
  const int p = sp;
  const double piv = spiv;
  if (!(fabs(piv) > 0.0)) return;
  // multipliers of the rows below (the row that WAS row j sits at p after the exchange)
  for (int r = j + 1 + tid; r <= rmax; r += kLuThreads) {
    const double a = r == p ? sajj : A[(size_t)r * s.W + (j - lu_c0(s, r))];
    mult[r - j - 1] = a / piv;
  }
  if (p != j) {
    double* rj = A + (size_t)j * s.W - lu_c0(s, j);
    double* rp = A + (size_t)p * s.W - lu_c0(s, p);
    for (int c = j + tid; c <= jmax; c += kLuThreads) { const double u = rp[c], a = rj[c]; rp[c] = a; rj[c] = u; }
    if (tid == 0) { const double u = rhs[p], a = rhs[j]; rhs[p] = a; rhs[j] = u; }
  }
}

// rows j + 1 .. j + kl: row r -= mult[r] x row j over the columns j + 1 .. jmax, the right-hand side with them
__global__ __launch_bounds__(256) void k_lu_update(LuShape s, int j, double* __restrict__ A, double* __restrict__ rhs,
                                                   const double* __restrict__ mult, const int* __restrict__ info) {
  if (*info != 0) return;
  const int rmax = min(s.n - 1, j + s.bw), jmax = min(s.n - 1, j + 2 * s.bw);
  const double* rj = A + (size_t)j * s.W - lu_c0(s, j);
  for (int q = 0; q < kLuRowsPerBlock; ++q) {
    const int r = j + 1 + blockIdx.x * kLuRowsPerBlock + q;
    if (r > rmax) return;
    const double l = mult[r - j - 1];
    if (l == 0.0) continue;
    double* rr = A + (size_t)r * s.W - lu_c0(s, r);
    for (int c = j + 1 + threadIdx.x; c <= jmax; c += 256) rr[c] = fma(-l, rj[c], rr[c]);
    if (threadIdx.x == 0) rhs[r] = fma(-l, rhs[j], rhs[r]);
  }
}

// U x = y from the last row up: one workgroup, a dot product of at most 2 bw terms per row
__global__ __launch_bounds__(kLuThreads) void k_lu_backsolve(LuShape s, const double* __restrict__ A, const double* __restrict__ rhs,
                                                             double* __restrict__ x, const int* __restrict__ info) {
  __shared__ double part[kLuThreads / 64];
  if (*info != 0) return;
  const int tid = threadIdx.x;
  for (int j = s.n - 1; j >= 0; --j) {
    const int jmax = min(s.n - 1, j + 2 * s.bw);
    const double* rj = A + (size_t)j * s.W - lu_c0(s, j);
    double acc = 0.0;
    for (int c = j + 1 + tid; c <= jmax; c += kLuThreads) acc = fma(rj[c], x[c], acc);
    acc = wave_sum(acc);
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < kLuThreads / 64; ++w) t += part[w];
      x[j] = (rhs[j] - t) / rj[j];
    }
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace ba
