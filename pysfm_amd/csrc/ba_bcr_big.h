// ba_bcr_big.h - block cyclic reduction of the reduced camera system for bands WIDER than an LDS-resident node
// (half-bandwidth hb > 23: tracks of 25 and more cameras - video features, loop closures).  Replaces
// solve_motion_normal_eqns' numpy.linalg.solve (bundle_adjuster.py:281-312) for those scenes, where the dense blocked
// Cholesky of ba_dense.h is a chain of n / 48 dependent panel steps (6000 unknowns: 125 steps, 3.4 ms).
//
// Nodes of cb >= hb cameras (B = 6 cb unknowns, 150 .. several hundred): the system is block tridiagonal over them.  One
// level with stride s eliminates the nodes i = s (2 k + 1) - 1; with a = i - s, c = i + s, U_a = T[a, i], U_i = T[i, c]:
//
//     | D_i    .     .     . |        the Schur complement of D_i in this symmetric matrix IS the level's whole update:
// K = | U_a    0     .     . |          K'[a,a] = -U_a G U_a^T   (added to D_a)        K'[c,a] = -U_i^T G U_a^T  (the new T[a, c]^T)
//     | U_i^T  0     0     . |          K'[c,c] = -U_i^T G U_i   (added to D_c)        K'[f,a], K'[f,c]: added to f_a, f_c
//     | f_i^T  0     0     0 |        with G = D_i^-1
//
// so a level is a PARTIAL dense Cholesky (the first B of 3 B columns) of one such matrix per eliminated node - exactly the
// panel / trailing-update kernels of ba_dense.h with a batch dimension (their layout already carries the right-hand side as
// one more row), between a gather and a scatter.  What the factorisation leaves in the first B columns is what the
// back-substitution needs:  rows 0..B-1 = L (D_i = L L^T), rows B..2B-1 = (L^-1 U_a^T)^T, rows 2B..3B-1 = (L^-1 U_i)^T,
// row 3B = (L^-1 f_i)^T;   x_i = L^-T (y_f - Y_a x_a - Y_c x_c).
//
// Launches per level: gather, B / 48 x (panel, update), scatter; back-substitution: one per level.  ceil(log2(N + 1)) levels.
#pragma once

#include "ba_dense.h"

namespace ba {

constexpr int kBigSlack = 16;        // entries right of the diagonal that the trailing update reads (16 x 16 sub-tiles)

__host__ __device__ inline size_t big_backsolve_lds_bytes(int B) { return dense_backsolve_lds_bytes(B) + (size_t)2 * B * sizeof(double); }

// K of every node eliminated at stride s: blockIdx.y = k (node i = s (2 k + 1) - 1), matrices batch_stride doubles apart.
// Every entry at most kBigSlack right of the diagonal is written (zeros where the picture above has them).
__global__ __launch_bounds__(256) void k_big_gather(int N, int B, int s, const double* __restrict__ Dm, const double* __restrict__ Um,
                                                    const double* __restrict__ fm, double* __restrict__ K, size_t batch_stride) {
  const int i = s * (2 * blockIdx.y + 1) - 1, a = i - s, c = i + s;
  const int n = 3 * B;
  const size_t BB = (size_t)B * B;
  double* Ki = K + (size_t)blockIdx.y * batch_stride;
  const double* Di = Dm + (size_t)i * BB;
  const double* Ua = a >= 0 ? Um + (size_t)a * BB : nullptr;
  const double* Ui = c < N ? Um + (size_t)i * BB : nullptr;
  for (int r = blockIdx.x; r <= n; r += gridDim.x) {
    double* row = Ki + (size_t)r * n;
    const int cend = r < n ? min(n, r + kBigSlack + 1) : n;
    for (int col = threadIdx.x; col < cend; col += 256) {
      double v = 0.0;
      if (col < B) {
        if (r < B) v = col <= r ? Di[(size_t)r * B + col] : 0.0;
        else if (r < 2 * B) v = Ua ? Ua[(size_t)(r - B) * B + col] : 0.0;
        else if (r < n) v = Ui ? Ui[(size_t)col * B + (r - 2 * B)] : 0.0;       // U_i^T
        else v = fm[(size_t)i * B + col];
      }
      row[col] = v;
    }
  }
}

// The trailing blocks of the factorised K's onto the surviving nodes m = 2 s (y + 1) - 1 (blockIdx.y = y): m is the right
// neighbour of the eliminated node in slot y and the left neighbour of the one in slot y + 1.  Only the lower triangle of D
// is kept up to date (k_big_gather reads nothing else).
__global__ __launch_bounds__(256) void k_big_scatter(int N, int B, int s, int cnt, double* __restrict__ Dm, double* __restrict__ Um,
                                                     double* __restrict__ fm, const double* __restrict__ K, size_t batch_stride) {
  const int y = blockIdx.y, m = 2 * s * (y + 1) - 1;
  if (m >= N) return;
  const int n = 3 * B;
  const size_t BB = (size_t)B * B;
  const double* Kl = K + (size_t)y * batch_stride;                              // node m - s: I am its c
  const double* Kr = (y + 1 < cnt && m + s < N) ? K + (size_t)(y + 1) * batch_stride : nullptr;      // node m + s: I am its a
  const bool next = Kr && m + 2 * s < N;                                        // ... and m + 2 s is its c: my next successor
  double* D = Dm + (size_t)m * BB;
  double* U = Um + (size_t)m * BB;
  for (int r = blockIdx.x; r <= B; r += gridDim.x) {
    if (r == B) {
      for (int col = threadIdx.x; col < B; col += 256)
        fm[(size_t)m * B + col] += Kl[(size_t)n * n + 2 * B + col] + (Kr ? Kr[(size_t)n * n + B + col] : 0.0);
      continue;
    }
    for (int col = threadIdx.x; col <= r; col += 256)
      D[(size_t)r * B + col] += Kl[(size_t)(2 * B + r) * n + 2 * B + col] + (Kr ? Kr[(size_t)(B + r) * n + B + col] : 0.0);
    // new T[m, m + 2 s] = (K'[c, a])^T of node m + s: U[p][q] = Kr[2B + q][B + p]; here r = q, so that the reads run along a row
    if (next)
      for (int p = threadIdx.x; p < B; p += 256) U[(size_t)p * B + r] = Kr[(size_t)(2 * B + r) * n + B + p];
  }
}

// t_i += (rows of [Y_a; Y_c] of this workgroup)^T (their part of [x_a; x_c]): what the neighbours' solutions take out of node i's
// right-hand side, spread over ceil(2 B / kBigMvRows) workgroups per node (round 5).  Inside k_big_backsolve ONE workgroup streamed
// the 2 B x B doubles of a node (3.7 MB at B = 480: 40 of the kernel's 92 us).  t is cleared once per solve.
constexpr int kBigMvRows = 64;
__global__ __launch_bounds__(1024) void k_big_backsolve_rhs(int N, int B, int s, const double* __restrict__ K, size_t batch_stride,
                                                            const double* __restrict__ x, double* __restrict__ t,
                                                            const int* __restrict__ info) {
  __shared__ double red[1024];
  __shared__ double xs[kBigMvRows];
  if (*info != 0) return;
  const int i = s * (2 * blockIdx.y + 1) - 1, a = i - s, c = i + s;
  const int n = 3 * B, tid = threadIdx.x;
  const double* Ki = K + (size_t)blockIdx.y * batch_stride;
  const int r0g = a >= 0 ? 0 : B, r1g = c < N ? 2 * B : B;          // (rows of a neighbour that does not exist are zeros)
  const int rb = r0g + blockIdx.x * kBigMvRows, re = min(r1g, rb + kBigMvRows);
  if (rb >= re) return;
  if (tid < re - rb) {
    const int r = rb + tid, node = r < B ? a : c;
    xs[tid] = x[(size_t)node * B + (r < B ? r : r - B)];
  }
  __syncthreads();
  const int cols = min((B + 63) & ~63, 1024), ng = max(1, 1024 / cols), g = tid / cols, k = tid - g * cols;
  for (int k0 = 0; k0 < B; k0 += cols) {
    double acc = 0.0;
    if (g < ng && k0 + k < B) {
      const double* p = Ki + (size_t)B * n + k0 + k;
      int r = rb + g;
      for (; r + 7 * ng < re; r += 8 * ng) {                         // eight loads in flight per thread
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(r + u * ng) * n];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u] * xs[r + u * ng - rb];
      }
      for (; r < re; r += ng) acc += p[(size_t)r * n] * xs[r - rb];
    }
    if (g < ng) red[g * cols + k] = acc;
    __syncthreads();
    if (tid < cols && k0 + tid < B) {
      double sum = 0.0;
      for (int q = 0; q < ng; ++q) sum += red[q * cols + tid];
      atomic_add_f64(t + (size_t)i * B + k0 + tid, sum);
    }
    __syncthreads();
  }
}

// x_i = L^-T (y_f - Y_a x_a - Y_c x_c) for the nodes eliminated at stride s; x is the solution vector itself (node i's
// unknowns are x[i B ..]: 6 cb cameras' worth).  tpre: Y_a x_a + Y_c x_c formed by k_big_backsolve_rhs (null: formed here).
__global__ __launch_bounds__(1024) void k_big_backsolve(int N, int B, int s, const double* __restrict__ K, size_t batch_stride,
                                                        double* __restrict__ x, const int* __restrict__ info,
                                                        const double* __restrict__ tpre = nullptr) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  if (*info != 0) return;
  const int i = s * (2 * blockIdx.x + 1) - 1, a = i - s, c = i + s;
  const int n = 3 * B, tid = threadIdx.x;
  const double* Ki = K + (size_t)blockIdx.x * batch_stride;
  double* w = sm;                                                    // [B]: the body's vector
  double* xn = sm + dense_backsolve_lds_bytes(B) / sizeof(double);   // [2 B]: x_a | x_c
  double* red = w + B + kDcNB * kDcLd + kDcNB;                       // [1024] (the body's own scratch, free until it starts)
  if (tpre) {
    for (int j = tid; j < B; j += 1024) w[j] = Ki[(size_t)n * n + j] - tpre[(size_t)i * B + j];
  } else {
  for (int j = tid; j < 2 * B; j += 1024) {
    const int node = j < B ? a : c;
    xn[j] = (node >= 0 && node < N) ? x[(size_t)node * B + (j < B ? j : j - B)] : 0.0;
  }
  __syncthreads();
  // w = y_f - [Y_a^T; Y_c^T]^T x: thread groups over the 2 B rows of K below L, lanes along a row
  // (B > 1024, half-bandwidths from 171: rounds of 1024 columns, one thread group)
  const int cols = min((B + 63) & ~63, 1024), ng = max(1, 1024 / cols), g = tid / cols, k = tid - g * cols;
  for (int k0 = 0; k0 < B; k0 += cols) {
    double acc = 0.0;
    const bool mine = g < ng && k0 + k < B;
    if (mine) {
      // (rows of a neighbour that does not exist are zeros, and so is its part of xn); eight loads in flight per thread
      const double* p = Ki + (size_t)B * n + k0 + k;
      const int r0 = a >= 0 ? 0 : B, r1 = c < N ? 2 * B : B;
      int r = r0 + g;
      for (; r + 7 * ng < r1; r += 8 * ng) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(r + u * ng) * n];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u] * xn[r + u * ng];
      }
      for (; r < r1; r += ng) acc += p[(size_t)r * n] * xn[r];
    }
    if (g < ng) red[g * cols + k] = acc;
    __syncthreads();
    if (tid < cols && k0 + tid < B) {
      double t = 0.0;
      for (int q = 0; q < ng; ++q) t += red[q * cols + tid];
      w[k0 + tid] = Ki[(size_t)n * n + k0 + tid] - t;
    }
    __syncthreads();
  }
  }
  dense_backsolve_body(B, n, B, Ki, sm);
  for (int j = tid; j < B; j += 1024) x[(size_t)i * B + j] = w[j];
}

}  // namespace ba
