// ba_border.hip - the reduced camera system as a band plus a border (ba_border.h): the border cameras' blocks, the solve.
#include "ba_internal.h"

#include "ba_bcr_blocks.h"
#include "ba_border.h"

using namespace ba;

namespace ba {

namespace {
inline int bcr_cams_per_node(const ba_handle* h) { const int n1 = h->band_cams(); return n1 <= kBcrMaxHB ? std::max(1, n1) : std::max(1, h->hb); }
// bordD = [D (ld x ld) | M (nb x nb, padded to ld x ld) | rv (ld) | x2 (ld) | status words (64 ints) | node workspace of the border solve]
inline double* bord_M(ba_handle* h) { return h->bordD.p + (size_t)h->bord_ld * h->bord_ld; }
inline double* bord_rv(ba_handle* h) { return bord_M(h) + (size_t)h->bord_ld * h->bord_ld; }
inline double* bord_x2(ba_handle* h) { return bord_rv(h) + h->bord_ld; }
inline int* bord_info(ba_handle* h) { return reinterpret_cast<int*>(bord_x2(h) + h->bord_ld); }
inline double* bord_node_ws(ba_handle* h) { return bord_x2(h) + h->bord_ld + 32; }

template <int NRT>
int launch_apply_levels(ba_handle* h, int N, int B, int ld) {
  const size_t lds = bord_apply_lds_bytes(16 * NRT);
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_bcr_apply<false, NRT>));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_bcr_apply<true, NRT>));
  const unsigned chunks = (unsigned)(ld / 16);
  int s_top = 1;
  for (int s = 1; N / (2 * s) >= 1; s *= 2) {
    hipLaunchKernelGGL((k_bcr_apply<false, NRT>), dim3((unsigned)(N / (2 * s)), chunks), dim3(kBordThreads), lds, h->stream, N, B, s, h->bcrP.p, h->bcrQ.p,
                       h->bcrG.p, h->bordF.p, ld);
    s_top = 2 * s;
  }
  while (s_top > 1 && (N / s_top + 1) / 2 == 0) s_top /= 2;
  for (int s = s_top; s >= 1; s /= 2) {
    const int cnt = (N / s + 1) / 2;
    if (cnt > 0)
      hipLaunchKernelGGL((k_bcr_apply<true, NRT>), dim3((unsigned)cnt, chunks), dim3(kBordThreads), lds, h->stream, N, B, s, h->bcrP.p, h->bcrQ.p, h->bcrG.p,
                         h->bordF.p, ld);
  }
  return BA_OK;
}
}  // namespace

// The border's work lists and buffers (once per problem): for every non-zero block (camera, border camera) the pairs of
// observations that add to it.  The internal observation order comes back from the device for it (4 bytes per observation).
int border_setup(ba_handle* h) {
  h->nbord_obs = 0;
  if (h->nbc <= 0) return BA_OK;
  const int n1 = h->band_cams(), nco = h->nco, cb = bcr_cams_per_node(h), B = 6 * cb, N = (n1 + cb - 1) / cb, nt = h->nt;
  const int ld = h->bord_ld = (6 * h->nbc + 15) / 16 * 16;
  h->bord_rows = (size_t)N * B;
  std::vector<int> hobs((size_t)h->nobs), pos((size_t)h->nc);
  HIPCHECK(h, hipMemcpyAsync(hobs.data(), h->obs_cam.p, (size_t)h->nobs * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipMemcpyAsync(pos.data(), h->cam_opt_pos.p, (size_t)h->nc * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  struct Entry { long long key; int na, nc_; };
  std::vector<Entry> ent;
  for (int i = 0; i < nt; ++i) {
    const int o0 = h->h_off[i], o1 = h->h_off[(size_t)i + 1];
    for (int na = o0; na < o1; ++na) {
      const int pa = pos[hobs[na]];
      if (pa < n1) continue;                               // (not a border camera)
      const int ja = pa - n1;
      for (int nc_ = o0; nc_ < o1; ++nc_) {
        const int pc = pos[hobs[nc_]];
        if (pc < 0) continue;                              // a camera that is not optimised adds nothing to S
        ent.push_back({(long long)ja * nco + pc, na, nc_});
      }
    }
  }
  std::stable_sort(ent.begin(), ent.end(), [](const Entry& a, const Entry& b) { return a.key < b.key; });
  std::vector<BorderBlock> blocks;
  std::vector<int2> pairs(ent.size());
  for (size_t e = 0; e < ent.size(); ++e) {
    pairs[e] = make_int2(ent[e].na, ent[e].nc_);
    if (e == 0 || ent[e].key != ent[e - 1].key) {
      if (!blocks.empty()) blocks.back().end = (int)e;
      blocks.push_back({(int)e, (int)ent.size(), (int)(ent[e].key % nco), (int)(ent[e].key / nco)});
    }
  }
  h->nbord_obs = (int)blocks.size();
  // chunks of a block's pairs (one wavefront each), where a block's chunks start, the band cameras that have a block
  std::vector<BorderChunk> chunks;
  std::vector<int> chunk_first(blocks.size() + 1, 0), rcams;
  for (size_t b = 0; b < blocks.size(); ++b) {
    chunk_first[b] = (int)chunks.size();
    for (int e = blocks[b].begin; e < blocks[b].end; e += kBordChunkPairs) chunks.push_back({(int)b, e, std::min(e + kBordChunkPairs, blocks[b].end), 0});
    if (blocks[b].pc < n1) rcams.push_back(blocks[b].pc);
  }
  chunk_first[blocks.size()] = (int)chunks.size();
  std::sort(rcams.begin(), rcams.end());
  rcams.erase(std::unique(rcams.begin(), rcams.end()), rcams.end());
  h->bord_nchunks = (int)chunks.size();
  h->bord_nrcams = (int)rcams.size();
  // bord_obs = [blocks | chunks | chunk_first | rcams | pairs]
  h->bord_off_chunks = blocks.size() * 4;
  h->bord_off_first = h->bord_off_chunks + chunks.size() * 4;
  h->bord_off_rcams = h->bord_off_first + chunk_first.size();
  h->bord_off_pairs = (h->bord_off_rcams + rcams.size() + 1) & ~(size_t)1;
  HIPCHECK(h, h->bord_obs.resize(std::max<size_t>(4, h->bord_off_pairs + pairs.size() * 2)));
  HIPCHECK(h, h->bord_partial.resize(std::max<size_t>(1, chunks.size() * kBordPartial)));
  if (!blocks.empty()) {
    HIPCHECK(h, hipMemcpyAsync(h->bord_obs.p, blocks.data(), blocks.size() * sizeof(BorderBlock), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(h->bord_obs.p + h->bord_off_chunks, chunks.data(), chunks.size() * sizeof(BorderChunk), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(h->bord_obs.p + h->bord_off_first, chunk_first.data(), chunk_first.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (!rcams.empty()) HIPCHECK(h, hipMemcpyAsync(h->bord_obs.p + h->bord_off_rcams, rcams.data(), rcams.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(h->bord_obs.p + h->bord_off_pairs, pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
  }
  HIPCHECK(h, h->bordC.resize(h->bord_rows * ld));
  HIPCHECK(h, h->bordF.resize(h->bord_rows * ld));
  const size_t Bb = (size_t)6 * h->nbc;
  HIPCHECK(h, h->bordD.resize((size_t)2 * ld * ld + 2 * ld + 32 + 5 * Bb * Bb + 2 * Bb + 64));
  // the blocks that exist are overwritten by every reduction; everything else of C and D stays zero from here on
  HIPCHECK(h, hipMemsetAsync(h->bordC.p, 0, h->bordC.n * sizeof(double), h->stream));
  HIPCHECK(h, hipMemsetAsync(h->bordD.p, 0, h->bordD.n * sizeof(double), h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));            // (the host vectors go out of scope)
  return BA_OK;
}

int border_join(ba_handle* h) {
  if (!h->bord_pending) return BA_OK;
  HIPCHECK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
  h->bord_pending = false;
  return BA_OK;
}

namespace {
// launches of the scope go to the side stream (the launch helpers and timers all use h->stream)
struct OnSideStream {
  ba_handle* h; hipStream_t keep;
  explicit OnSideStream(ba_handle* h_) : h(h_), keep(h_->stream) { if (h->opt.border_side_stream) h->stream = h->side; }
  ~OnSideStream() { h->stream = keep; }
};
int side_fork(ba_handle* h) {
  if (!h->opt.border_side_stream) return BA_OK;
  if (!h->side) {
    HIPCHECK(h, hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    HIPCHECK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHECK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  }
  HIPCHECK(h, hipEventRecord(h->ev_fork, h->stream));          // everything enqueued so far (the point inverses; the consumers of the last trial's C, D) ...
  HIPCHECK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));     // ... comes before what the side stream does next
  return BA_OK;
}
int side_mark(ba_handle* h) {
  if (!h->opt.border_side_stream) return BA_OK;
  HIPCHECK(h, hipEventRecord(h->ev_join, h->side));
  h->bord_pending = true;
  return BA_OK;
}
}  // namespace

// C, D and the border part of b: one wavefront per chunk of a block's pairs, every block written once (ba_border.h) - on the side
// stream: nothing of the band is read or written (the band's right-hand side ends at the band cameras)
int border_schur(ba_handle* h, int p, double damping) {
  if (h->nbc <= 0) return BA_OK;
  if (int rc = side_fork(h); rc != BA_OK) return rc;
  OnSideStream on_side(h);
  const int n1 = h->band_cams(), ld = h->bord_ld, nblocks = h->nbord_obs;
  ScopedTimer tm(h, BA_K_BORDER_SCHUR, 2);
  if (nblocks > 0) {
    const BorderBlock* blocks = reinterpret_cast<const BorderBlock*>(h->bord_obs.p);
    const BorderChunk* chunks = reinterpret_cast<const BorderChunk*>(h->bord_obs.p + h->bord_off_chunks);
    const int2* pairs = reinterpret_cast<const int2*>(h->bord_obs.p + h->bord_off_pairs);
    const int nchunks = h->bord_nchunks;
    const unsigned grid = (unsigned)((nchunks + kBlock / 64 - 1) / (kBlock / 64));
    if (h->sensor.kind == SENSOR_TABLE)
      hipLaunchKernelGGL(k_schur_border<true>, dim3(grid), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p, h->X[p].p, blocks, chunks, nchunks, pairs, n1,
                         damping, h->HPPinv.p, h->bP.p, h->bord_partial.p);
    else
      hipLaunchKernelGGL(k_schur_border<false>, dim3(grid), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p, h->X[p].p, blocks, chunks, nchunks, pairs, n1,
                         damping, h->HPPinv.p, h->bP.p, h->bord_partial.p);
    hipLaunchKernelGGL(k_border_sum, dim3(blocks_for((long long)nblocks * kBordPartial)), dim3(kBlock), 0, h->stream, blocks, nblocks, h->bord_obs.p + h->bord_off_first,
                       h->bord_partial.p, n1, h->bordC.p, h->bordD.p, ld, h->b + (size_t)6 * n1);
  }
  HIPCHECK(h, hipGetLastError());
  return side_mark(h);
}

// After solve_bcr (the factors of B in bcrP / bcrQ / bcrG, y = B^-1 b1 in dC): the columns of C through the same tree, the border
// system, the correction.  Status through flags[1] like every solver.
int border_solve(ba_handle* h, const unsigned char* dmask) {
  if (h->nbc <= 0) return BA_OK;
  const int n1 = h->band_cams(), cb = bcr_cams_per_node(h), B = 6 * cb, N = (n1 + cb - 1) / cb;
  const int ld = h->bord_ld, nb = 6 * h->nbc, rows1 = 6 * n1;
  if (h->bord_pending) {
    // the copy of C into the work array and the start of M, rv need C, D, b2 and the mask, nothing of the band's solve: still on the side stream
    OnSideStream on_side(h);
    if (dmask && h->opt.border_side_stream) { HIPCHECK(h, hipEventRecord(h->ev_fork, on_side.keep)); HIPCHECK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0)); }      // (the mask's upload is on the main stream)
    hipLaunchKernelGGL(k_border_prepare, dim3(blocks_for((long long)h->bord_rows * ld + (long long)nb * nb)), dim3(kBlock), 0, h->stream, (long long)h->bord_rows, rows1,
                       ld, nb, h->bordC.p, h->bordD.p, h->b + (size_t)rows1, dmask, h->bordF.p, bord_M(h), bord_rv(h), bord_info(h));
    if (int rc = side_mark(h); rc != BA_OK) return rc;
  } else {
    hipLaunchKernelGGL(k_border_prepare, dim3(blocks_for((long long)h->bord_rows * ld + (long long)nb * nb)), dim3(kBlock), 0, h->stream, (long long)h->bord_rows, rows1,
                       ld, nb, h->bordC.p, h->bordD.p, h->b + (size_t)rows1, dmask, h->bordF.p, bord_M(h), bord_rv(h), bord_info(h));
  }
  if (int rc = border_join(h); rc != BA_OK) return rc;
  ScopedTimer tm(h, BA_K_BORDER_SOLVE, 1);
  int rc = BA_OK;
  switch ((B + 15) / 16) {
    case 1: rc = launch_apply_levels<1>(h, N, B, ld); break;
    case 2: rc = launch_apply_levels<2>(h, N, B, ld); break;
    case 3: rc = launch_apply_levels<3>(h, N, B, ld); break;
    case 4: rc = launch_apply_levels<4>(h, N, B, ld); break;
    default: rc = launch_apply_levels<5>(h, N, B, ld); break;
  }
  if (rc != BA_OK) return rc;
  if (h->bord_nrcams > 0)
    hipLaunchKernelGGL(k_border_reduce, dim3((unsigned)((ld / 16) * (ld / 16 + 1) / 2 + 1), (unsigned)((h->bord_nrcams + 4 * kBordRedCams - 1) / (4 * kBordRedCams))),
                       dim3(kBordThreads), 0, h->stream, h->bord_nrcams,
                       h->bord_obs.p + h->bord_off_rcams, ld, nb, h->bordC.p, h->bordF.p, h->dC.p, dmask ? dmask + rows1 : nullptr, bord_M(h), bord_rv(h));
  if (h->nbc <= kBcrMaxHB) {
    // the border system as ONE node of the cyclic reduction (its root: factor, solve): the node kernel's pivot chain
    const size_t Bb = (size_t)nb, BB = Bb * Bb;
    double* ws = bord_node_ws(h);                          // U (never read: the node has no neighbours), P, Q, G^-1
    HIPCHECK(h, launch_bcr_eliminate(h, h->nbc, 1, bcr_lds_bytes(nb), h->stream, 1, 1, bord_M(h), ws, bord_rv(h), ws + BB, ws + 2 * BB, ws + 3 * BB, bord_info(h),
                                     bord_x2(h)));
  } else {
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_border_solve));
    hipLaunchKernelGGL(k_border_solve, dim3(1), dim3(kBordThreads), bord_solve_lds_bytes(nb), h->stream, nb, bord_M(h), bord_rv(h), bord_info(h), bord_x2(h));
  }
  hipLaunchKernelGGL(k_border_correct, dim3(blocks_for((long long)4 * (rows1 + nb))), dim3(kBlock), 0, h->stream, rows1, nb, ld, h->bordF.p, bord_x2(h), h->dC.p,
                     bord_info(h), h->flags.p + 1);
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

// The band + border system when it is NOT positive definite (the reference solves whatever is not singular: numpy.linalg.solve,
// bundle_adjuster.py:302-305), on the device: block elimination with LU everywhere a Cholesky factor stood -
//     Y = B^-1 C  column by column and  y = B^-1 b1  through the band's LU solver (the cyclic reduction with LU nodes up to 11
//     cameras a node, LU with partial pivoting down the band beyond: pivots stay inside the band part), then
//     (D - C^T Y) x2 = b2 - C^T y  by Gaussian elimination with partial pivoting in one workgroup,  x1 = y - Y x2.
// 6 k + 1 band solves (k <= 21 border cameras) instead of one factorisation whose factors would take the columns through: the LU
// nodes keep what the back-substitution needs, not what a second right-hand side's elimination would.  A rare path: 15 ms at 1000
// cameras with ten border cameras - the host LU it replaces took the whole matrix over PCIe (288 MB) and a second.
int border_solve_lu(ba_handle* h, const unsigned char* dmask) {
  if (h->nbc <= 0) return BA_OK;
  if (int rc = border_join(h); rc != BA_OK) return rc;
  const int n1 = h->band_cams(), ld = h->bord_ld, nb = 6 * h->nbc, rows1 = 6 * n1;
  const bool lu_nodes = bcr_cams_per_node(h) <= kBcrMaxHB;
  ScopedTimer tm(h, BA_K_BORDER_SOLVE, 1);
  hipLaunchKernelGGL(k_border_prepare, dim3(blocks_for((long long)h->bord_rows * ld + (long long)nb * nb)), dim3(kBlock), 0, h->stream, (long long)h->bord_rows, rows1,
                     ld, nb, h->bordC.p, h->bordD.p, h->b + (size_t)rows1, dmask, h->bordF.p, bord_M(h), bord_rv(h), bord_info(h));
  HIPCHECK(h, h->bordV.resize((size_t)rows1 + 64));
  int* acc = h->flags.p + 48;                               // the first failure among the band solves (each of them clears flags[1] when it starts)
  HIPCHECK(h, hipMemsetAsync(acc, 0, sizeof(int), h->stream));
  const unsigned gcol = blocks_for(rows1);
  for (int c = 0; c <= nb; ++c) {                            // the columns of C, then b1 (its solution stays in dC: y)
    const bool last = c == nb;
    if (!last) {
      hipLaunchKernelGGL(k_border_column, dim3(gcol), dim3(kBlock), 0, h->stream, rows1, ld, c, h->bordF.p, h->bordV.p, 0, h->flags.p + 1, (int*)nullptr);
    }
    const int rc = lu_nodes ? solve_bcr_lu(h, dmask, n1, last ? h->b : h->bordV.p) : solve_band_lu(h, dmask, n1, last ? h->b : h->bordV.p);
    if (rc != BA_OK) return rc;
    if (!last) hipLaunchKernelGGL(k_border_column, dim3(gcol), dim3(kBlock), 0, h->stream, rows1, ld, c, h->bordF.p, h->dC.p, 1, h->flags.p + 1, acc);
  }
  if (h->bord_nrcams > 0)
    hipLaunchKernelGGL(k_border_reduce, dim3((unsigned)((ld / 16) * (ld / 16 + 1) / 2 + 1), (unsigned)((h->bord_nrcams + 4 * kBordRedCams - 1) / (4 * kBordRedCams))),
                       dim3(kBordThreads), 0, h->stream, h->bord_nrcams,
                       h->bord_obs.p + h->bord_off_rcams, ld, nb, h->bordC.p, h->bordF.p, h->dC.p, dmask ? dmask + rows1 : nullptr, bord_M(h), bord_rv(h));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_border_solve_lu));
  hipLaunchKernelGGL(k_border_solve_lu, dim3(1), dim3(kBordThreads), ((size_t)nb * (nb + 3) + 8) * sizeof(double), h->stream, nb, bord_M(h), bord_rv(h), bord_info(h), bord_x2(h));
  // (a band solve that failed - an exactly singular band part - outranks the border's own status: the last solve's word is flags[1] itself)
  hipLaunchKernelGGL(k_border_column, dim3(1), dim3(kBlock), 0, h->stream, 0, ld, 0, h->bordF.p, h->bordV.p, 0, acc, h->flags.p + 1);
  hipLaunchKernelGGL(k_border_correct, dim3(blocks_for((long long)4 * (rows1 + nb))), dim3(kBlock), 0, h->stream, rows1, nb, ld, h->bordF.p, bord_x2(h), h->dC.p,
                     bord_info(h), h->flags.p + 1);
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

int border_get_dense(ba_handle* h, std::vector<double>& C, std::vector<double>& D) {
  if (int rc = border_join(h); rc != BA_OK) return rc;
  const size_t ld = h->bord_ld, rows1 = (size_t)6 * h->band_cams();
  C.resize(rows1 * ld); D.resize(ld * ld);
  if (rows1) HIPCHECK(h, hipMemcpyAsync(C.data(), h->bordC.p, C.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipMemcpyAsync(D.data(), h->bordD.p, D.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  return BA_OK;
}

int border_flatten(ba_handle* h, int nkeep, double* A_dev, double* rhs_dev) {
  if (int rc = border_join(h); rc != BA_OK) return rc;
  ScopedTimer tm(h, BA_K_FLATTEN);
  hipLaunchKernelGGL(k_flatten_bordered, dim3(blocks_for((long long)nkeep * nkeep)), dim3(kBlock), 0, h->stream, h->band_cams(), h->hb, nkeep, h->keep.p, h->S,
                     h->b, h->bordC.p, h->bordD.p, h->bord_ld, A_dev, rhs_dev);
  HIPCHECK(h, hipGetLastError());
  return BA_OK;
}

}  // namespace ba
