// ba_pcg.hip - ba_solve_reduced for scenes whose co-visibility is no band: the list of the blocks of S that can be non-zero (once per
// problem, from the tracks' camera lists) and the preconditioned conjugate gradients over it (ba_pcg.h).
#include "ba_internal.h"

#include "ba_pcg.h"

#include <algorithm>
#include <chrono>

using namespace ba;

namespace ba {

// The pattern of S: block (i, j) can be non-zero iff some track is seen by the optimised cameras at positions i and j
// (bundle_adjuster.py:270-276 visits exactly those).  Rows of the FULL symmetric pattern, sorted by column; a block lives in the
// band at (min, max).  Built on the host from the observation list in the internal order (one download, 4 bytes per observation;
// a point with the camera list of its predecessor adds nothing), uploaded once per problem.
int pcg_build_pattern(ba_handle* h) {
  auto& g = h->pcg;
  if (g.built) return BA_OK;
  const auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (h->opt.solve_trace) fprintf(stderr, "[pcg_build_pattern] %s: %.2f ms since the call\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  };
  const int nco = h->nco, hb1 = h->hb + 1;
  std::vector<int> cam((size_t)h->nobs);
  if (h->nobs) HIPCHECK(h, hipMemcpyAsync(cam.data(), h->obs_cam.p, (size_t)h->nobs * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  std::vector<std::vector<int>> rows((size_t)nco);
  for (int i = 0; i < nco; ++i) rows[i].push_back(i);                       // the diagonal blocks always exist (HCC)
  std::vector<int> pos;
  if (g.shared_lists) {
    // a sharded scene: the lists of ALL its tracks (ba_set_pattern_lists), not only of this rank's
    for (size_t l = 0; l + 1 < g.shared_loff.size(); ++l) {
      pos.clear();
      for (int q = g.shared_loff[l]; q < g.shared_loff[l + 1]; ++q) {
        const int p0 = g.shared_lpos[q];
        if (p0 < 0 || p0 >= nco) return h->fail(BA_ERR_INVALID_ARG, "ba_set_pattern_lists: position %d of %d optimised cameras", p0, nco);
        pos.push_back(h->cpos_in.empty() ? p0 : h->cpos_in[p0]);
      }
      for (size_t a = 0; a < pos.size(); ++a)
        for (size_t b = 0; b < pos.size(); ++b)
          if (a != b) rows[pos[a]].push_back(pos[b]);
    }
  }
  for (int k = 0; k < h->nt && !g.shared_lists; ++k) {
    if (k > 0 && k < (int)h->h_same.size() && h->h_same[k]) continue;      // the camera list of the point before it
    pos.clear();
    for (int n = h->h_off[k]; n < h->h_off[k + 1]; ++n) {
      const int c = cam[n];
      const int q = c >= 0 && c < h->nc ? h->h_cam_opt_pos[c] : -1;
      if (q >= 0) pos.push_back(q);
    }
    for (size_t a = 0; a < pos.size(); ++a)
      for (size_t b = 0; b < pos.size(); ++b)
        if (a != b) rows[pos[a]].push_back(pos[b]);
  }
  lap("rows collected");
  std::vector<int> rowptr((size_t)nco + 1, 0), col;
  std::vector<long long> ublk;
  long long upper = 0;
  for (int i = 0; i < nco; ++i) {
    auto& r = rows[i];
    std::sort(r.begin(), r.end());
    r.erase(std::unique(r.begin(), r.end()), r.end());
    for (int j : r) {
      const int lo = std::min(i, j), hi = std::max(i, j);
      if (hi - lo > h->hb) return h->fail(BA_ERR_STATE, "pcg: cameras %d and %d share a track but the band is %d wide", lo, hi, h->hb);
      col.push_back(j);
      upper += j >= i;
      if (j >= i) ublk.push_back((long long)lo * hb1 + (hi - lo));
    }
    rowptr[i + 1] = (int)col.size();
    std::vector<int>().swap(r);
  }
  lap("rows sorted, pattern in CSR form");
  g.nnz = (long long)col.size();
  g.upper = upper;
  g.h_ublk = ublk;
  // ---- the upper blocks' lists of observation pairs (k_schur_blocks): block (i, j >= i) <- (observation of the camera at i, of the
  // camera at j) for every point both see.  Two passes over the points: count, fill.  Not built beyond 64 M pairs (0.5 GB; the
  // reduction then stays with k_schur_pairs).
  g.pairs_built = false;
  {
    std::vector<int> ufirst((size_t)nco + 1, 0);             // index of row i's first upper block in ublk
    {
      long long u = 0;
      for (int i = 0; i < nco; ++i) {
        ufirst[i] = (int)u;
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) u += col[e] >= i;
      }
      ufirst[nco] = (int)u;
    }
    auto block_of = [&](int i, int j) {                      // i <= j: rank of j among row i's columns >= i
      const int* b0 = col.data() + rowptr[i];
      const int* b1 = col.data() + rowptr[i + 1];
      const int* lo = std::lower_bound(b0, b1, i);
      const int* at = std::lower_bound(lo, b1, j);
      return ufirst[i] + (int)(at - lo);
    };
    {
      std::vector<int> udiag((size_t)std::max(1, nco));
      for (int i = 0; i < nco; ++i) udiag[i] = ufirst[i];      // (a row's columns are sorted and the diagonal block always exists: its first upper block)
      HIPCHECK(h, g.udiag.resize(udiag.size()));
      HIPCHECK(h, hipMemcpyAsync(g.udiag.p, udiag.data(), udiag.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
      HIPCHECK(h, hipStreamSynchronize(h->stream));
    }
    {
      // every entry of the full pattern -> its (upper) block in the packed array of a solve (k_pcg_gather)
      std::vector<int> cidx(col.size());
      for (int i = 0; i < nco; ++i)
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) cidx[e] = col[e] >= i ? block_of(i, col[e]) : block_of(col[e], i);
      HIPCHECK(h, g.cidx.resize(std::max<size_t>(1, cidx.size())));
      if (!cidx.empty()) HIPCHECK(h, hipMemcpyAsync(g.cidx.p, cidx.data(), cidx.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
      HIPCHECK(h, hipStreamSynchronize(h->stream));
    }
    std::vector<int> bptr((size_t)upper + 1, 0);
    std::vector<std::pair<int, int>> po;                      // (position, observation) of a point's optimised cameras
    std::vector<int> pair_block;
    size_t seen = 0;
    long long total = 0;
    for (int pass = 0; pass < 2 && total <= (64ll << 20); ++pass) {
      std::vector<int2> pr;
      std::vector<int> fill;
      if (pass == 1) {
        for (long long u = 0, s = 0; u <= upper; ++u) { const int c = u < upper ? bptr[u] : 0; bptr[u] = (int)s; s += c; }
        pr.resize((size_t)total);
        fill.assign(bptr.begin(), bptr.end() - 1);
      }
      for (int k = 0; k < h->nt; ++k) {
        po.clear();
        for (int n = h->h_off[k]; n < h->h_off[k + 1]; ++n) {
          const int c = cam[n];
          const int q = c >= 0 && c < h->nc ? h->h_cam_opt_pos[c] : -1;
          if (q >= 0) po.push_back({q, n});
        }
        for (size_t a = 0; a < po.size(); ++a)
          for (size_t b = 0; b < po.size(); ++b) {
            if (po[a].first > po[b].first || (po[a].first == po[b].first && a != b)) continue;      // (upper triangle; the diagonal: every observation with itself)
            // (the block of a pair is looked up once - two binary searches in its row - and remembered for the second pass)
            if (pass == 0) { const int u = block_of(po[a].first, po[b].first); pair_block.push_back(u); ++bptr[u]; ++total; }
            else pr[(size_t)fill[pair_block[seen++]]++] = int2{po[a].second, po[b].second};
          }
      }
      lap(pass == 0 ? "pairs counted" : "pairs filled");
      if (pass == 1) {
        HIPCHECK(h, g.bptr.resize(bptr.size()));
        HIPCHECK(h, g.pairs.resize(std::max<size_t>(1, pr.size())));
        HIPCHECK(h, hipMemcpyAsync(g.bptr.p, bptr.data(), bptr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        if (!pr.empty()) HIPCHECK(h, hipMemcpyAsync(g.pairs.p, pr.data(), pr.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
        // chunks of kSbChunk pairs: (block of chunk, first chunk of block)
        std::vector<int> cptr((size_t)upper + 1, 0), cblk;
        for (long long u = 0; u < upper; ++u) {
          const int np = bptr[u + 1] - bptr[u], nch = std::max(1, (np + kSbChunk - 1) / kSbChunk);
          cptr[u] = (int)cblk.size();
          for (int c = 0; c < nch; ++c) cblk.push_back((int)u);
        }
        cptr[upper] = (int)cblk.size();
        g.nchunks = (long long)cblk.size();
        HIPCHECK(h, g.cptr.resize(cptr.size()));
        HIPCHECK(h, g.cblk.resize(std::max<size_t>(1, cblk.size())));
        HIPCHECK(h, g.partial.resize(std::max<size_t>(1, cblk.size() * kSbPartial)));
        HIPCHECK(h, hipMemcpyAsync(g.cptr.p, cptr.data(), cptr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        if (!cblk.empty()) HIPCHECK(h, hipMemcpyAsync(g.cblk.p, cblk.data(), cblk.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));        // (the host vectors go out of scope)
        g.pairs_built = true;
      }
    }
  }
  HIPCHECK(h, g.rowptr.resize(rowptr.size()));
  HIPCHECK(h, g.col.resize(std::max<size_t>(1, col.size())));
  HIPCHECK(h, g.ublk.resize(std::max<size_t>(1, ublk.size())));
  HIPCHECK(h, hipMemcpyAsync(g.rowptr.p, rowptr.data(), rowptr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (!col.empty()) {
    HIPCHECK(h, hipMemcpyAsync(g.col.p, col.data(), col.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(g.ublk.p, ublk.data(), ublk.size() * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHECK(h, hipStreamSynchronize(h->stream));            // (the host vectors go out of scope)
  g.built = true;
  lap("uploaded");
  return BA_OK;
}

// Fraction of the band's blocks that can be non-zero (1 = a full band: nothing for an iterative solver to gain).
double pcg_band_fill(ba_handle* h) {
  if (pcg_build_pattern(h) != BA_OK || h->nco == 0) return 1.0;
  return (double)h->pcg.upper / ((double)h->nco * (h->hb + 1));
}

// Is this a scene for the sparse path - a wide band of mostly structural zeros, large enough for a dense factorisation to hurt, no
// border, the solver left to the library (or set to pcg)?  Builds the pattern on the first call of a problem.
bool sparse_layout(ba_handle* h) {
  if (h->pcg.packed) return true;                           // (decided by ba_set_problem: the problem's [S] has no band to go back to)
  if (h->nbc > 0 || h->dense_mode || h->nco == 0 || h->hb <= kBcrwMaxHB || (h->comm && !h->pcg.shared_lists)) return false;
  if (h->opt.solver == SOLVER_PCG) return pcg_build_pattern(h) == BA_OK;
  if (h->opt.solver != SOLVER_AUTO || h->nco < kPcgMinCams) return false;
  return pcg_band_fill(h) <= kPcgMaxFill;
}

// [S | b] initialised over the blocks of the pattern only (k_schur_init visits the whole band: 6.5 GB at 5000 cameras, 1.5 ms a
// trial): valid once everything outside the pattern is known to be zero (band_clean: after one full initialisation of this
// problem's band that nothing has scribbled over since).
int launch_schur_init_sparse(ba_handle* h, double damping, int use_hcc) {
  auto& g = h->pcg;
  const long long n = g.upper * 36 + (long long)h->nco * 6;
  hipLaunchKernelGGL(k_schur_init_blocks, dim3(blocks_for(n)), dim3(kBlock), 0, h->stream, g.upper, g.ublk.p, h->nco, h->hb + 1, h->opt_cam.p,
                     h->HCC.p, h->bC.p, damping, h->S, h->b, use_hcc, g.packed ? 1 : 0);
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

int launch_schur_blocks(ba_handle* h, int p) {
  auto& g = h->pcg;
  const unsigned grid = blocks_for(g.nchunks * kSbLanes);
  const DevProblem P = dev_problem_band(h);
  // every observation linearised once, T = W HPPinv and W left behind (288 bytes each), then the blocks sum their pairs from those -
  // when the device has the room for it (6 M observations: 1.7 GB); otherwise every pair linearises its two observations itself
  const size_t tw = (size_t)std::max<long long>(1, h->nobs) * 36;
  bool staged = h->opt.sparse_stage;
  if (staged && g.TW.n < tw) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || tw * sizeof(double) > free_b / 2) staged = false;
    else HIPCHECK(h, g.TW.resize(tw));
  }
  if (staged) {
    if (h->sensor.kind == SENSOR_TABLE)
      hipLaunchKernelGGL(k_sparse_stage<true>, dim3(blocks_for(h->nobs)), dim3(kBlock), 0, h->stream, P, h->cams[p].p, h->X[p].p, h->HPPinv.p, g.TW.p);
    else
      hipLaunchKernelGGL(k_sparse_stage<false>, dim3(blocks_for(h->nobs)), dim3(kBlock), 0, h->stream, P, h->cams[p].p, h->X[p].p, h->HPPinv.p, g.TW.p);
    hipLaunchKernelGGL(k_schur_blocks_staged, dim3(grid), dim3(kBlock), 0, h->stream, h->obs_pt.p, g.TW.p, h->bP.p, g.nchunks, g.cblk.p, g.cptr.p, g.ublk.p,
                       g.bptr.p, g.pairs.p, h->hb + 1, g.partial.p);
  } else if (h->sensor.kind == SENSOR_TABLE)
    hipLaunchKernelGGL(k_schur_blocks<true>, dim3(grid), dim3(kBlock), 0, h->stream, P, h->cams[p].p, h->X[p].p, h->HPPinv.p, h->bP.p, g.nchunks,
                       g.cblk.p, g.cptr.p, g.ublk.p, g.bptr.p, g.pairs.p, h->hb + 1, g.partial.p);
  else
    hipLaunchKernelGGL(k_schur_blocks<false>, dim3(grid), dim3(kBlock), 0, h->stream, P, h->cams[p].p, h->X[p].p, h->HPPinv.p, h->bP.p, g.nchunks,
                       g.cblk.p, g.cptr.p, g.ublk.p, g.bptr.p, g.pairs.p, h->hb + 1, g.partial.p);
  hipLaunchKernelGGL(k_schur_blocks_sum, dim3(blocks_for(g.upper * kSbPartial)), dim3(kBlock), 0, h->stream, g.upper, g.cptr.p, g.ublk.p, h->hb + 1, g.partial.p,
                     h->S, h->b, g.packed ? 1 : 0);
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

// Preconditioned conjugate gradients on [S | b] (mask: deleted parameters).  Leaves the solution in h->dC and the status word in
// flags[1]: 0 = converged to ||r|| <= tol ||b||; > 0 = a diagonal block or S itself is not positive definite (camera + 1, or nco + 1);
// kPcgStalled = no convergence within the iteration budget (the caller treats both like a failed Cholesky).
int solve_pcg(ba_handle* h, const unsigned char* dmask) {
  auto& g = h->pcg;
  if (int rc = pcg_build_pattern(h); rc != BA_OK) return rc;
  const int nco = h->nco, n = 6 * nco, hb1 = h->hb + 1;
  const int nparts = (n + kPcgUnknownsPerBlock - 1) / kPcgUnknownsPerBlock;
  // a workgroup per block row where the rows hold dozens of blocks (a chain of dependent trips to memory per eight of them), four rows a workgroup otherwise
  const bool wide_rows = g.nnz >= 24ll * nco;
  const int nprod = wide_rows ? nco : (nco + kPcgRowsPerBlock - 1) / kPcgRowsPerBlock;
  HIPCHECK(h, g.minv.resize((size_t)nco * 36));
  HIPCHECK(h, g.r.resize((size_t)n)); HIPCHECK(h, g.z.resize((size_t)n)); HIPCHECK(h, g.q.resize((size_t)n));
  HIPCHECK(h, g.p[0].resize((size_t)n)); HIPCHECK(h, g.p[1].resize((size_t)n));
  HIPCHECK(h, g.part.resize((size_t)4 * nparts + nprod));
  if (!g.packed) HIPCHECK(h, g.packed_blocks.resize(std::max<size_t>(1, (size_t)g.upper * 36)));
  const double* Sp = g.packed ? h->S : g.packed_blocks.p;      // the pattern's upper blocks, contiguous: [S] itself when it is stored that way
  HIPCHECK(h, g.state.resize(1));
  if (!g.host_state) HIPCHECK(h, hipHostMalloc((void**)&g.host_state, sizeof(PcgStateRaw), hipHostMallocDefault));
  double* part_rz = g.part.p, *part_rr = g.part.p + (size_t)2 * nparts, *part_pq = g.part.p + (size_t)4 * nparts;
  const double tol = h->opt.pcg_tol, tol2 = tol * tol;
  const int max_iter = h->opt.pcg_max_iter > 0 ? h->opt.pcg_max_iter : std::max(1000, std::min(20000, 4 * nco));
  ScopedTimer tm(h, BA_K_PCG_SOLVE);
  HIPCHECK(h, hipMemsetAsync(h->flags.p + 1, 0, sizeof(int), h->stream));
  hipLaunchKernelGGL(k_pcg_minv, dim3(blocks_for(nco)), dim3(kBlock), 0, h->stream, nco, hb1, h->S, g.packed ? g.udiag.p : (const int*)nullptr, dmask, g.minv.p, h->flags.p + 1);
  if (!g.packed) hipLaunchKernelGGL(k_pcg_gather, dim3(blocks_for(g.upper * 36)), dim3(kBlock), 0, h->stream, g.upper, g.ublk.p, h->S, g.packed_blocks.p);
  hipLaunchKernelGGL(k_pcg_start, dim3(nparts), dim3(kBlock), 0, h->stream, n, h->b, dmask, g.minv.p, h->dC.p, g.r.p, g.z.p, part_rz, part_rr, reinterpret_cast<PcgState*>(g.state.p));
  int k = 0, status = 0;
  double rr_checked = -1.0;
  int k_checked = 0;
  g.iterations = 0;
  while (true) {
    // (short batches first: a launch after convergence costs ~3 us, a look at the state a synchronisation - damped systems converge in 6 .. 30 iterations)
    // ... and the first batch is what the last solve of this problem needed plus one: consecutive trials of an LM walk need about the same
    const int first = g.prev_iterations > 0 ? std::min(h->opt.pcg_batch, g.prev_iterations + 1) : 8;
    const int batch = std::min(std::min(h->opt.pcg_batch, k == 0 ? first : std::max(8, k)), max_iter - k);
    for (int e = k + batch; k < e; ++k) {
      if (wide_rows)
        hipLaunchKernelGGL(k_pcg_product<4>, dim3(nprod), dim3(kBlock), 0, h->stream, k, nco, hb1, Sp, g.rowptr.p, g.col.p, g.cidx.p, g.z.p,
                           g.p[k & 1].p, g.p[(k + 1) & 1].p, g.q.p, part_rz, part_rr, nparts, tol2, part_pq, reinterpret_cast<PcgState*>(g.state.p));
      else
        hipLaunchKernelGGL(k_pcg_product<1>, dim3(nprod), dim3(kBlock), 0, h->stream, k, nco, hb1, Sp, g.rowptr.p, g.col.p, g.cidx.p, g.z.p,
                           g.p[k & 1].p, g.p[(k + 1) & 1].p, g.q.p, part_rz, part_rr, nparts, tol2, part_pq, reinterpret_cast<PcgState*>(g.state.p));
      hipLaunchKernelGGL(k_pcg_update, dim3(nparts), dim3(kBlock), 0, h->stream, k, n, dmask, g.minv.p, g.p[(k + 1) & 1].p, g.q.p, h->dC.p,
                         g.r.p, g.z.p, part_rz, part_rr, nparts, part_pq, nprod, tol2, reinterpret_cast<PcgState*>(g.state.p));
    }
    HIPCHECK(h, hipGetLastError());
    HIPCHECK(h, hipMemcpyAsync(g.host_state, g.state.p, sizeof(PcgStateRaw), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipMemcpyAsync(&g.host_status, h->flags.p + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    const PcgStateRaw& st = *g.host_state;
    g.iterations = st.done_iter >= 0 ? st.done_iter : st.last_iter + 1;
    g.rel_residual = st.bb > 0.0 ? sqrt(st.rr / st.bb) : 0.0;
    if (g.host_status > 0) { status = g.host_status; break; }                     // a diagonal block that is not positive definite
    if (st.breakdown) { status = nco + 1; break; }                              // p.S p <= 0: S is not positive definite
    if (st.done_iter >= 0) break;
    if (!(st.rr == st.rr)) { status = nco + 1; break; }                         // NaNs: nothing to wait for
    // no progress: the residual has not come down by a factor of ten over the last quarter of the budget
    if (rr_checked >= 0.0 && k - k_checked >= max_iter / 4) {
      if (st.rr > .01 * rr_checked) { status = kPcgStalled; break; }
      rr_checked = st.rr; k_checked = k;
    } else if (rr_checked < 0.0) { rr_checked = st.rr; k_checked = k; }
    if (k >= max_iter) { status = kPcgStalled; break; }
  }
  if (status != 0 && g.host_status == 0) hipLaunchKernelGGL(k_pcg_set_status, dim3(1), dim3(1), 0, h->stream, h->flags.p + 1, status);
  HIPCHECK(h, hipGetLastError());
  g.last_status = status;
  g.prev_iterations = status == 0 ? g.iterations : 0;
  return BA_OK;
}

}  // namespace ba

extern "C" {

int ba_set_pattern_lists(ba_handle* h, int32_t nlists, const int32_t* list_off, const int32_t* list_pos) {
  if (!h) return BA_ERR_INVALID_ARG;
  auto& g = h->pcg;
  g.shared_lists = false;
  g.shared_loff.clear(); g.shared_lpos.clear();
  g.built = false; g.pairs_built = false;
  if (nlists <= 0) return BA_OK;
  REQUIRE(h, list_off && list_pos && list_off[0] == 0, BA_ERR_INVALID_ARG, "ba_set_pattern_lists: NULL or malformed lists");
  for (int l = 0; l < nlists; ++l) REQUIRE(h, list_off[l + 1] >= list_off[l], BA_ERR_INVALID_ARG, "ba_set_pattern_lists: offsets must ascend");
  g.shared_loff.assign(list_off, list_off + nlists + 1);
  g.shared_lpos.assign(list_pos, list_pos + list_off[nlists]);
  g.shared_lists = true;
  return BA_OK;
}

int ba_pcg_info(ba_handle* h, int64_t* blocks, int32_t* iterations, double* rel_residual, double* band_fill) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_pcg_info: call ba_set_problem first");
  HIPCHECK(h, hipSetDevice(h->device));
  if (int rc = pcg_build_pattern(h); rc != BA_OK) return rc;
  if (blocks) *blocks = h->pcg.upper;
  if (iterations) *iterations = h->pcg.iterations;
  if (rel_residual) *rel_residual = h->pcg.rel_residual;
  if (band_fill) *band_fill = h->nco ? (double)h->pcg.upper / ((double)h->nco * (h->hb + 1)) : 1.0;
  return BA_OK;
}

}  // extern "C"
