// ba_solve_wide.hip - the reduced camera system for half-bandwidths beyond 11: cyclic reduction with three kernels per level (ba_bcr_wide.h), with nodes in device memory (ba_bcr_big.h), dense blocked Cholesky (ba_dense.h).
#include "ba_internal.h"

#include "ba_bcr_wide.h"
#include "ba_dense.h"
#include "ba_bcr_big.h"
#include "ba_band_lu.h"

using namespace ba;

namespace ba {

// factor + solve of one level (both are templates on the half-bandwidth)
template <int HB>
hipError_t launch_bcrw_factor_hb(ba_handle* h, int cnt, hipStream_t st, int N, int s, const double* D, double* L, double* Lv, const double* U,
                                 double* f, double* P, double* Q, double* G, int* info) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcrw_factor<HB>); e != hipSuccess) return e;
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcrw_solve_mfma<HB>); e != hipSuccess) return e;
  constexpr int B = 6 * HB;
  const int ntile = (3 * B + 1 + 15) / 16;
  if (h->opt.bcrw_merged) {
    // factor and solve in ONE kernel: every workgroup of a node's right-hand sides factors D_i for itself (k_bcrw_factor_solve)
    if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcrw_factor_solve<HB>); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bcrw_factor_solve<HB>, dim3(cnt, (ntile + 3) / 4), dim3(kBcrwFsThreads), bcrw_factor_solve_lds_bytes(B), st, N, s, D, U, f,
                       P, Q, G, info);
    return hipSuccess;
  }
  hipLaunchKernelGGL(k_bcrw_factor<HB>, dim3(cnt), dim3(kBcrElimThreads), bcrw_factor_lds_bytes(B), st, N, s, D, L, Lv, info);
  hipLaunchKernelGGL(k_bcrw_solve_mfma<HB>, dim3(cnt, (ntile + 3) / 4), dim3(1024), bcrw_solve_lds_bytes(B), st, N, s, L, Lv, U, f,
                     P, Q, G, info);
  return hipSuccess;
}

hipError_t launch_bcrw_factor(ba_handle* h, int hb, int cnt, hipStream_t st, int N, int s, const double* D, double* L, double* Lv, const double* U,
                              double* f, double* P, double* Q, double* G, int* info) {
#define BA_HB_CASE(K) case K: return launch_bcrw_factor_hb<K>(h, cnt, st, N, s, D, L, Lv, U, f, P, Q, G, info);
  switch (hb) {
    BA_HB_CASE(12) BA_HB_CASE(13) BA_HB_CASE(14) BA_HB_CASE(15) BA_HB_CASE(16) BA_HB_CASE(17) BA_HB_CASE(18) BA_HB_CASE(19)
    BA_HB_CASE(20) BA_HB_CASE(21) BA_HB_CASE(22) BA_HB_CASE(23)
    default: return hipErrorInvalidValue;
  }
#undef BA_HB_CASE
}

// Block cyclic reduction for half-bandwidths 12..21 (ba_bcr_wide.h): three kernels per level.
int solve_bcr_wide(ba_handle* h, const unsigned char* dmask) {
  const int hb = h->hb, B = 6 * hb, N = (h->nco + hb - 1) / hb;
  const size_t BB = (size_t)B * B;
  HIPCHECK(h, h->bcrD.resize(N * BB)); HIPCHECK(h, h->bcrU.resize(N * BB)); HIPCHECK(h, h->bcrP.resize(N * BB));
  HIPCHECK(h, h->bcrQ.resize(N * BB)); HIPCHECK(h, h->bcrG.resize(N * BB)); HIPCHECK(h, h->bcrL.resize(N * BB));
  HIPCHECK(h, h->bcrF.resize((size_t)N * B));
  HIPCHECK(h, h->bcrLv.resize((size_t)N * ((B + 11) / 12) * 144));
  HIPCHECK(h, h->dC.resize((size_t)N * B + 16));
  {
    ScopedTimer tm(h, BA_K_BCR_ASSEMBLE);       // also clears the status word flags[1]; marks the solution "not there yet" for the one-launch back-substitution
    launch_bcr_assemble(h, dim3(N), hb, dmask, h->dC.p, nullptr, nullptr);
  }
  std::vector<int> strides;
  for (int s = 1; (N / s + 1) / 2 > 0; s *= 2) strides.push_back(s);
  const int nt = (B + kBcrwPTile - 1) / kBcrwPTile, ntask = nt * (nt + 1) + nt * nt + 1;
  const bool back_fused = h->opt.fused_backsolve && hb <= kBcrwFusedBackMaxHB && N <= 8 * h->ncu;
  if (back_fused && h->bcr_order_n != N) {
    std::vector<int> order;                      // the nodes level by level from the root down (as solve_bcr's)
    for (int q = (int)strides.size() - 1; q >= 0; --q)
      for (int k = 0, cnt = (N / strides[q] + 1) / 2; k < cnt; ++k) {
        const int i = strides[q] * (2 * k + 1) - 1;
        if (i < N) order.push_back(i);
      }
    if ((int)order.size() != N) return h->fail(BA_ERR_STATE, "cyclic reduction: %d of %d nodes in the level lists", (int)order.size(), N);
    HIPCHECK(h, h->bcr_order.resize((size_t)N));
    HIPCHECK(h, hipMemcpyAsync(h->bcr_order.p, order.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));          // `order` goes out of scope
    h->bcr_order_n = N;
  }
  {
    ScopedTimer tm(h, BA_K_BCR_ELIMINATE, (h->opt.bcrw_merged ? 2 : 3) * (int)strides.size());
    for (int s : strides) {
      const int cnt = (N / s + 1) / 2;
      HIPCHECK(h, launch_bcrw_factor(h, hb, cnt, h->stream, N, s, h->bcrD.p, h->bcrL.p, h->bcrLv.p, h->bcrU.p, h->bcrF.p, h->bcrP.p,
                                     h->bcrQ.p, h->bcrG.p, h->flags.p + 1));
      hipLaunchKernelGGL(k_bcrw_products, dim3(cnt, ntask), dim3(1024), 0, h->stream, N, B, s, h->bcrD.p, h->bcrU.p, h->bcrF.p, h->bcrP.p, h->bcrQ.p, h->flags.p + 1);
    }
  }
  if (back_fused) {
    ScopedTimer tmb(h, BA_K_BCR_BACKSOLVE, 1);
    int* ticket = h->flags.p + 1 + kBcrTicketWord;
    switch (hb) {
#define BA_HB_CASE(K) case K: hipLaunchKernelGGL(k_bcrw_backsolve_fused<K>, dim3(N), dim3(1024), 0, h->stream, N, h->bcrF.p, h->bcrP.p, h->bcrQ.p, \
                                                 h->bcrG.p, h->dC.p, h->bcr_order.p, ticket); break;
      BA_HB_CASE(12) BA_HB_CASE(13) BA_HB_CASE(14) BA_HB_CASE(15) BA_HB_CASE(16) BA_HB_CASE(17) BA_HB_CASE(18) BA_HB_CASE(19)
      BA_HB_CASE(20) BA_HB_CASE(21)
#undef BA_HB_CASE
      default: return h->fail(BA_ERR_STATE, "k_bcrw_backsolve_fused: half-bandwidth %d", hb);
    }
    HIPCHECK(h, hipGetLastError());
    return BA_OK;
  }
  ScopedTimer tmb(h, BA_K_BCR_BACKSOLVE, (int)strides.size());
  for (int q = (int)strides.size() - 1; q >= 0; --q) {
    const int s = strides[q], cnt = (N / s + 1) / 2;
    hipLaunchKernelGGL(k_bcrw_backsolve, dim3(cnt), dim3(1024), 0, h->stream, N, B, s, h->bcrF.p, h->bcrP.p, h->bcrQ.p, h->bcrG.p,
                       h->dC.p);
  }
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

// Block cyclic reduction with nodes too large for LDS (ba_bcr_big.h): half-bandwidths beyond kBcrwMaxHB.  A level is a batched
// partial dense Cholesky of one 3B x 3B matrix per eliminated node.

int solve_bcr_big(ba_handle* h, const unsigned char* dmask) {
  const int hb = h->hb, cb = big_node_cameras(hb), B = 6 * cb, N = (h->nco + cb - 1) / cb, n = 3 * B;
  const size_t BB = (size_t)B * B, KS = big_matrix_doubles(B);
  HIPCHECK(h, h->bcrD.resize(N * BB)); HIPCHECK(h, h->bcrU.resize(N * BB)); HIPCHECK(h, h->bcrF.resize((size_t)N * B));
  HIPCHECK(h, h->bigK.resize((size_t)N * KS));
  HIPCHECK(h, h->bcrGv.resize((size_t)N * B));
  HIPCHECK(h, h->dC.resize((size_t)N * B + 16));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_dense_panel));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_dense_step));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_big_backsolve));
  int* info = h->flags.p + 1;
  {
    ScopedTimer tm(h, BA_K_BCR_ASSEMBLE);       // also clears the status word flags[1]
    const int rounds = (int)((BB + 3 * kBcrThreads - 1) / (3 * kBcrThreads));        // (three entries per thread and round)
    launch_bcr_assemble(h, dim3(N, std::max(1, std::min(rounds, 2048 / N))), cb, dmask, nullptr, nullptr, nullptr);
  }
  struct Level { int s, cnt; size_t base; };
  std::vector<Level> levels;
  size_t slots = 0;
  for (int s = 1; (N / s + 1) / 2 > 0; s *= 2) { levels.push_back({s, (N / s + 1) / 2, slots}); slots += (N / s + 1) / 2; }
  const int npanels = (B + kDcNB - 1) / kDcNB;
  {
    ScopedTimer tm(h, BA_K_BCR_ELIMINATE, (int)levels.size() * ((h->opt.dense_lookahead ? npanels + 1 : 2 * npanels) + 2));
    for (const Level& L : levels) {
      double* K = h->bigK.p + L.base * KS;
      hipLaunchKernelGGL(k_big_gather, dim3(std::min(n + 1, 128), L.cnt), dim3(256), 0, h->stream, N, B, L.s, h->bcrD.p, h->bcrU.p,
                         h->bcrF.p, K, KS);
      // the first panel step, then ONE launch per block column: panel step of the next column beside this column's trailing update
      // (k_dense_step); the last column's update on its own (the trailing 2 B x 2 B block is what the level is for)
      for (int k0 = 0; k0 < B; k0 += kDcNB) {
        const int nb = std::min(kDcNB, B - k0), kn = k0 + nb, total = n - kn + 1;      // every row below the block + the right-hand side row
        if (k0 == 0)
          hipLaunchKernelGGL(k_dense_panel, dim3((total + kDcRows - 1) / kDcRows, 1, L.cnt), dim3(1024), dense_panel_lds_bytes(), h->stream,
                             n, k0, nb, total, K, info, KS);
        if (kn < B && h->opt.dense_lookahead) {
          const int n1 = std::min(kDcNB, B - kn), total1 = n - (kn + n1) + 1;
          const int npanel = (total1 + kDcRows - 1) / kDcRows, T = (total1 + kDcTile - 1) / kDcTile;
          hipLaunchKernelGGL(k_dense_step, dim3(npanel + T * (T + 1) / 2, 1, L.cnt), dim3(1024), dense_step_lds_bytes(), h->stream, n, k0, nb,
                             kn, n1, total1, npanel, T, K, info, KS);
        } else {
          const int T = (total + kDcTile - 1) / kDcTile;
          hipLaunchKernelGGL(k_dense_update, dim3(T, T, L.cnt), dim3(1024), dense_update_lds_bytes(), h->stream, n, k0, nb, total, K, KS);
          if (kn < B) {
            const int n1 = std::min(kDcNB, B - kn), total1 = n - (kn + n1) + 1;
            hipLaunchKernelGGL(k_dense_panel, dim3((total1 + kDcRows - 1) / kDcRows, 1, L.cnt), dim3(1024), dense_panel_lds_bytes(), h->stream,
                               n, kn, n1, total1, K, info, KS);
          }
        }
      }
      const int nsurv = (N + 1) / (2 * L.s);       // nodes m = 2 s (y + 1) - 1 < N
      if (nsurv > 0)
        hipLaunchKernelGGL(k_big_scatter, dim3(std::min(B + 1, 64), nsurv), dim3(256), 0, h->stream, N, B, L.s, L.cnt, h->bcrD.p, h->bcrU.p,
                           h->bcrF.p, K, KS);
    }
  }
  // back-substitution, root down: what the neighbours' solutions take out of a node's right-hand side over many workgroups
  // (k_big_backsolve_rhs, into bcrGv), then one workgroup per node for the triangular solve
  const bool split_rhs = h->opt.dense_lookahead;
  if (split_rhs) HIPCHECK(h, hipMemsetAsync(h->bcrGv.p, 0, (size_t)N * B * sizeof(double), h->stream));
  ScopedTimer tmb(h, BA_K_BCR_BACKSOLVE, (int)levels.size() * (split_rhs ? 2 : 1));
  for (int q = (int)levels.size() - 1; q >= 0; --q) {
    const Level& L = levels[q];
    const bool has_nb = N > 1 && !(L.cnt == 1 && 2 * L.s - 1 >= N);      // (the root has no neighbours)
    if (split_rhs && has_nb)
      hipLaunchKernelGGL(k_big_backsolve_rhs, dim3((2 * B + kBigMvRows - 1) / kBigMvRows, L.cnt), dim3(1024), 0, h->stream, N, B, L.s,
                         h->bigK.p + L.base * KS, KS, h->dC.p, h->bcrGv.p, info);
    hipLaunchKernelGGL(k_big_backsolve, dim3(L.cnt), dim3(1024), big_backsolve_lds_bytes(B), h->stream, N, B, L.s,
                       h->bigK.p + L.base * KS, KS, h->dC.p, info, split_rhs ? h->bcrGv.p : nullptr);
  }
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

// Dense Cholesky of the whole reduced system (ba_dense.h): bands wider than the cyclic reduction's blocks.
int solve_dense_chol(ba_handle* h, const unsigned char* dmask) {
  const int n = 6 * h->nco;
  HIPCHECK(h, h->denseA.resize((size_t)(n + 1) * n));
  HIPCHECK(h, h->dC.resize((size_t)n + 16));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_dense_panel));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_dense_step));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_dense_backsolve));
  int* info = h->flags.p + 1;
  double* A = h->denseA.p;
  const int nsteps = (n + kDcNB - 1) / kDcNB;
  ScopedTimer tm(h, BA_K_DENSE_SOLVE, (h->opt.dense_lookahead ? nsteps : 2 * nsteps - 1) + 2);
  hipLaunchKernelGGL(k_dense_gather, dim3(n + 1), dim3(256), 0, h->stream, h->nco, h->hb, h->S, h->b, dmask, A, info);
  const int bw = std::min(n, 6 * (h->hb + 1) - 1);            // S[r][c] = 0 for |r - c| > bw
  for (int k0 = 0; k0 < n; k0 += kDcNB) {
    const int nb = std::min(kDcNB, n - k0), kn = k0 + nb;
    const int total = std::min(n, kn + bw) - kn + 1;           // rows below the block that can be non-zero + the rhs row
    if (k0 == 0)
      hipLaunchKernelGGL(k_dense_panel, dim3((total + kDcRows - 1) / kDcRows), dim3(1024), dense_panel_lds_bytes(), h->stream, n,
                         k0, nb, total, A, info);
    if (kn >= n) break;                                        // (the last block column: only the right-hand side row is behind it - done by its panel step)
    const int n1 = std::min(kDcNB, n - kn), kn1 = kn + n1, total1 = std::min(n, kn1 + bw) - kn1 + 1;
    if (h->opt.dense_lookahead) {
      // the next block column's panel step beside this one's trailing update (k_dense_step)
      const int npanel = (total1 + kDcRows - 1) / kDcRows, T = (total1 + kDcTile - 1) / kDcTile;
      hipLaunchKernelGGL(k_dense_step, dim3(npanel + T * (T + 1) / 2), dim3(1024), dense_step_lds_bytes(), h->stream, n, k0, nb, kn, n1, total1,
                         npanel, T, A, info);
    } else {
      const int T = (total + kDcTile - 1) / kDcTile;
      hipLaunchKernelGGL(k_dense_update, dim3(T, T), dim3(1024), dense_update_lds_bytes(), h->stream, n, k0, nb, total, A);
      hipLaunchKernelGGL(k_dense_panel, dim3((total1 + kDcRows - 1) / kDcRows), dim3(1024), dense_panel_lds_bytes(), h->stream, n,
                         kn, n1, total1, A, info);
    }
  }
  hipLaunchKernelGGL(k_dense_backsolve, dim3(1), dim3(1024), dense_backsolve_lds_bytes(n), h->stream, n, bw, A, h->dC.p, info);
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

// LU with partial pivoting of the (masked) reduced system (ba_band_lu.h): the reference's own factorisation, for systems the
// Cholesky solvers report as not positive definite and for option solver = lu.  Leaves the solution in h->dC and gesv's info
// (0, or the 1-based column of an exactly zero pivot) in flags[1].
int solve_band_lu(ba_handle* h, const unsigned char* dmask, int ncams, const double* rhs_in) {
  // (ncams, rhs_in: the band part of a bordered system and any right-hand side of it - ba_border.hip; default: the whole system, b)
  const int n1 = ncams >= 0 ? ncams : h->nco;
  if (!rhs_in) rhs_in = h->b;
  const int n = 6 * n1;
  LuShape s;
  s.n = n;
  s.bw = std::min(n - 1, 6 * h->hb + 5);
  s.W = (int)std::min<long long>(3ll * s.bw + 1, n);
  HIPCHECK(h, h->denseA.resize((size_t)n * s.W + (size_t)n + (size_t)s.bw + 8));
  HIPCHECK(h, h->dC.resize((size_t)n + 16));
  double* A = h->denseA.p;
  double* rhs = A + (size_t)n * s.W;
  double* mult = rhs + n;
  int* info = h->flags.p + 1;
  ScopedTimer tm(h, BA_K_DENSE_SOLVE, 2 * n + 2);
  hipLaunchKernelGGL(k_lu_assemble, dim3(n), dim3(256), 0, h->stream, s, n1, h->hb, h->S, rhs_in, dmask, A, rhs, info);
  for (int j = 0; j < n; ++j) {
    hipLaunchKernelGGL(k_lu_pivot, dim3(1), dim3(kLuThreads), 0, h->stream, s, j, A, rhs, mult, info);
    const int rows = std::min(n - 1, j + s.bw) - j;
    if (rows > 0)
      hipLaunchKernelGGL(k_lu_update, dim3((rows + kLuRowsPerBlock - 1) / kLuRowsPerBlock), dim3(256), 0, h->stream, s, j, A, rhs, mult, info);
  }
  hipLaunchKernelGGL(k_lu_backsolve, dim3(1), dim3(kLuThreads), 0, h->stream, s, A, rhs, h->dC.p, info);
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

}  // namespace ba
