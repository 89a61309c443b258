// ba_problem.hip - ba_set_problem: the internal order of points and observations and the work lists of the kernels.
//
// The observations arrive in any order (the reference walks `tracks` and their `measurements` dicts as they come,
// bundle_adjuster.py:222-226).  Everything that touches every observation runs on the device (ba_setup_kernels.h): range
// checks, ordering by (track, camera rank) - skipped when the input is already in that order -, the duplicate check, the
// order of the tracks, CSR offsets, the gather into the internal arrays, the per-point summaries, the (point, window
// column) tables.  The host plans the work lists from the per-point summaries (a few passes over the POINTS) and uploads
// them.  Work lists of kernels that are not on the trial's path (k_camera_blocks, k_schur_pairs) are built on first use.
#include "ba_internal.h"

#include <chrono>

#include "ba_setup_kernels.h"

using namespace ba;

namespace {

inline int bits_for(unsigned long long count) {      // bits that hold 0 .. count - 1
  int b = 1;
  while (b < 63 && (1ull << b) < count) ++b;
  return b;
}

hipError_t pinned_staging(ba_handle* h, size_t bytes) {
  auto& su = h->su;
  if (su.host && su.host_bytes >= bytes) return hipSuccess;
  if (su.host) (void)hipHostFree(su.host);
  su.host = nullptr; su.host_bytes = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  hipError_t e = hipHostMalloc(&su.host, want, hipHostMallocDefault);
  if (e == hipSuccess) su.host_bytes = want;
  return e;
}

inline unsigned grid_for(long long n) { return (unsigned)std::max<long long>(1, (n + 255) / 256); }

// Small uploads go through a pinned arena: an H2D copy from pageable memory is staged synchronously by the runtime (10 - 15 us
// apiece, a dozen of them per problem - most of ba_set_problem for the sliding-window caller's 1000 observations); from pinned
// memory it is an asynchronous enqueue.  The arena is reset where nothing can be in flight from it (at the start of
// ba_set_problem and after its second synchronisation); what does not fit goes the ordinary way.
constexpr size_t kArenaBytes = 8u << 20, kArenaMaxItem = 2u << 20;
void arena_reset(ba_handle* h) {
  auto& su = h->su;
  if (!su.up && hipHostMalloc(&su.up, kArenaBytes, hipHostMallocDefault) == hipSuccess) su.up_bytes = kArenaBytes;
  if (su.up_pending) { (void)hipStreamSynchronize(h->stream); su.up_pending = false; }      // (a small problem leaves without waiting for its uploads)
  su.up_used = 0;
}
hipError_t stage_h2d(ba_handle* h, void* dst, const void* src, size_t bytes) {
  auto& su = h->su;
  const size_t aligned = (bytes + 63) & ~(size_t)63;
  if (bytes <= kArenaMaxItem && su.up && su.up_used + aligned <= su.up_bytes) {
    void* p = static_cast<char*>(su.up) + su.up_used;
    std::memcpy(p, src, bytes);
    su.up_used += aligned;
    src = p;
  } else {
    su.up_pageable = true;                       // (the caller's memory must outlive the copy: synchronise before returning)
  }
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream);
}


// ---- a dozen small uploads as ONE copy and ONE launch: the pieces are packed behind a table of (destination, offset, words)
// in the pinned arena, the block goes to a device mirror in one hipMemcpyAsync, and k_setup_scatter moves every piece to its
// buffer.  (An asynchronous copy costs 4 - 5 us of host time whatever its size; the sliding-window caller's problem has nine.)
constexpr int kPackMaxPieces = 30;
struct PackPiece { unsigned long long dst; unsigned off_words; unsigned words; };
struct PackTable { int n; int total_words; PackPiece piece[kPackMaxPieces]; };
constexpr size_t kPackHeaderBytes = 512;
static_assert(sizeof(PackTable) <= kPackHeaderBytes, "the table sits at the head of the block");

__global__ __launch_bounds__(256) void k_setup_scatter(const unsigned* __restrict__ blob) {
  const PackTable* tab = reinterpret_cast<const PackTable*>(blob);
  const unsigned w = blockIdx.x * 256 + threadIdx.x;
  if ((int)w >= tab->total_words) return;
  int d = 0;
  while (d + 1 < tab->n && w >= tab->piece[d + 1].off_words) ++d;
  const PackPiece pc = tab->piece[d];
  reinterpret_cast<unsigned*>(pc.dst)[w - pc.off_words] = blob[kPackHeaderBytes / 4 + w];
}

struct Packer {
  ba_handle* h;
  char* base = nullptr;          // the block in the arena
  PackTable* tab = nullptr;
  size_t cap = 0;
  bool ok = false;
  explicit Packer(ba_handle* h_, size_t payload_bytes) : h(h_) {
    auto& su = h->su;
    const size_t need = kPackHeaderBytes + payload_bytes + 64 * kPackMaxPieces;
    const size_t start = (su.up_used + 63) & ~(size_t)63;
    if (!su.up || start + need > su.up_bytes) return;           // (does not fit the arena: the caller copies piece by piece)
    base = static_cast<char*>(su.up) + start;
    su.up_used = start + need;
    tab = reinterpret_cast<PackTable*>(base);
    tab->n = 0; tab->total_words = 0;
    cap = payload_bytes / 4 + 16 * kPackMaxPieces;
    ok = true;
  }
  // dst must have room for the piece rounded up to 4 bytes
  void add(void* dst, const void* src, size_t bytes) {
    const unsigned words = (unsigned)((bytes + 3) / 4);
    if (!words) return;
    PackPiece& pc = tab->piece[tab->n++];
    pc.dst = reinterpret_cast<unsigned long long>(dst);
    pc.off_words = (unsigned)tab->total_words;
    pc.words = words;
    std::memcpy(base + kPackHeaderBytes + (size_t)tab->total_words * 4, src, bytes);
    tab->total_words += (int)words;
  }
  int flush() {
    if (!tab->n) return BA_OK;
    const size_t bytes = kPackHeaderBytes + (size_t)tab->total_words * 4;
    HIPCHECK(h, h->su.blob.resize((bytes + 3) / 4 + 64));
    HIPCHECK(h, hipMemcpyAsync(h->su.blob.p, base, bytes, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_setup_scatter, dim3((unsigned)((tab->total_words + 255) / 256)), dim3(256), 0, h->stream, h->su.blob.p);
    HIPCHECK(h, hipGetLastError());
    return BA_OK;
  }
};


// ---- ba_set_problem's front end for SMALL problems, on the host.  The device pipeline (a dozen launches, a radix sort, two
// synchronisations) costs 0.2 ms whatever the size; the sliding-window caller sets a problem of a thousand observations per
// frame, where the same work is a few microseconds of one core.  Same decisions, same internal order (the track key is the
// one of k_setup_track_keys, bit for bit), same staging layout for the planning code that follows.
constexpr long long kHostFrontMaxObs = 8192;

int host_front_end(ba_handle* h, int nc, int nt, long long N, int nco, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_z,
                   const int32_t* cam_opt_pos, const int32_t* cam_band_pos, const uint8_t* pt_opt, const std::vector<int>& crank, const std::vector<int>& opt_cam, int rank_bits, int* hflags, int* hoff,
                   int* hplo, int* hphi, int* hperm, unsigned char* same) {
  typedef unsigned long long u64;
  for (int i = 0; i < SF_COUNT; ++i) hflags[i] = (i == SF_BAD || i == SF_DUP) ? 0x7fffffff : 0;
  std::vector<u64> key((size_t)N);
  std::vector<int> cnt((size_t)nt + 1, 0), by((size_t)N);
  bool unsorted = false;
  for (long long n = 0; n < N; ++n) {
    const int c = obs_cam[n], k = obs_pt[n];
    if (c < 0 || c >= nc) return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: obs_cam[%d]=%d out of range", (int)n, c);
    if (k < 0 || k >= nt) return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: obs_pt[%d]=%d out of range", (int)n, k);
    key[n] = ((u64)k << rank_bits) | (u64)crank[c];
    ++cnt[k];
    if (n > 0 && key[n - 1] > key[n]) unsorted = true;
    by[n] = (int)n;
  }
  if (unsorted) std::stable_sort(by.begin(), by.end(), [&](int a, int b) { return key[a] < key[b]; });
  for (long long q = 1; q < N; ++q)
    if (key[by[q]] == key[by[q - 1]])
      return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: track %d has two observations in one camera", (int)(key[by[q]] >> rank_bits));
  std::vector<int> coff((size_t)nt + 1, 0);
  for (int k = 0; k < nt; ++k) coff[(size_t)k + 1] = coff[k] + cnt[k];
  // the tracks' order: (first optimised position, hash of the camera-rank list), stable
  std::vector<int> pperm((size_t)nt);
  for (int k = 0; k < nt; ++k) pperm[k] = k;
  if (nt > 1) {
    std::vector<u64> tkey((size_t)nt);
    for (int k = 0; k < nt; ++k) {
      const int b = coff[k], e = coff[(size_t)k + 1];
      u64 hsh = 0x9E3779B97F4A7C15ull ^ (u64)(e - b);
      int minpos = nco;
      for (int q = b; q < e; ++q) {
        const int c = obs_cam[by[q]];
        const int p = cam_band_pos[c];
        if (p >= 0 && p < minpos) minpos = p;
        hsh ^= (u64)crank[c] + 0x9E3779B97F4A7C15ull + (hsh << 6) + (hsh >> 2);
        hsh *= 0xD6E8FEB86659FD93ull;
      }
      hsh ^= hsh >> 32;
      tkey[k] = ((u64)minpos << 32) | (hsh & 0xffffffffull);
    }
    std::vector<int> sorted(pperm);
    std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b) { return tkey[a] < tkey[b]; });
    bool desc = false;
    int runs_orig = 0, runs_sorted = 0;
    for (int i = 1; i < nt; ++i) {
      if ((tkey[i] >> 32) < (tkey[i - 1] >> 32)) desc = true;
      runs_orig += tkey[i] != tkey[i - 1];
      runs_sorted += tkey[sorted[i]] != tkey[sorted[i - 1]];
    }
    hflags[SF_DESC] = desc; hflags[SF_RUNS_ORIG] = runs_orig; hflags[SF_RUNS_SORTED] = runs_sorted;
    if (desc || runs_orig != runs_sorted) {          // the caller's order is not as good as the sorted one
      pperm.swap(sorted);
      for (int i = 0; i < nt; ++i) if (pperm[i] != i) hflags[SF_PERM] = 1;
    }
  }
  // the internal arrays and the per-point summaries
  std::vector<int> icam((size_t)N), ipt((size_t)N), operm((size_t)N);
  std::vector<double> iz((size_t)2 * N);
  std::vector<unsigned char> iopt((size_t)nt);
  hoff[0] = 0;
  int maxL = 0, hbw = 0;
  for (int i = 0; i < nt; ++i) {
    const int k = pperm[i], src = coff[k], L = cnt[k], dst = hoff[i];
    hoff[i + 1] = dst + L;
    hperm[i] = k;
    iopt[i] = pt_opt[k];
    int lo = 0x7fffffff, hi = -1;
    bool asc = true;
    for (int q = 0; q < L; ++q) {
      const int n = by[(size_t)src + q];
      operm[(size_t)dst + q] = n;
      icam[(size_t)dst + q] = obs_cam[n];
      ipt[(size_t)dst + q] = i;
      iz[2 * ((size_t)dst + q)] = obs_z[2 * (size_t)n]; iz[2 * ((size_t)dst + q) + 1] = obs_z[2 * (size_t)n + 1];
      if (n != dst + q) hflags[SF_OPERM] = 1;
      const int p = cam_band_pos[obs_cam[n]];
      if (p < 0) continue;
      if (p <= hi) asc = false;
      lo = std::min(lo, p); hi = std::max(hi, p);
    }
    hplo[i] = lo; hphi[i] = hi;
    if (hi >= 0) hbw = std::max(hbw, hi - lo);
    if (!asc) hflags[SF_NOT_ASC] = 1;
    maxL = std::max(maxL, L);
    bool sm = false;
    if (i > 0) {
      const int pb = hoff[i - 1];
      sm = dst - pb == L;
      for (int q = 0; sm && q < L; ++q) sm = icam[(size_t)pb + q] == icam[(size_t)dst + q];
    }
    same[i] = sm ? 1 : 0;
  }
  hflags[SF_MAXL] = maxL; hflags[SF_HB] = hbw;
  Packer pk(h, (size_t)N * 32 + (size_t)nt * 16 + (size_t)nc * 8 + 256);
  auto put = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    if (!bytes) return hipSuccess;
    if (pk.ok && h->opt.packed_upload && pk.tab->n < kPackMaxPieces) { pk.add(dst, src, bytes); return hipSuccess; }
    return stage_h2d(h, dst, src, bytes);
  };
  HIPCHECK(h, put(h->cam_opt_pos.p, cam_opt_pos, (size_t)nc * sizeof(int)));
  if (cam_band_pos != cam_opt_pos) HIPCHECK(h, put(h->cam_band_pos.p, cam_band_pos, (size_t)nc * sizeof(int)));
  HIPCHECK(h, put(h->opt_cam.p, opt_cam.data(), opt_cam.size() * sizeof(int)));
  HIPCHECK(h, put(h->obs_cam.p, icam.data(), (size_t)N * sizeof(int)));
  HIPCHECK(h, put(h->obs_pt.p, ipt.data(), (size_t)N * sizeof(int)));
  HIPCHECK(h, put(h->obs_z.p, iz.data(), (size_t)N * sizeof(double2)));
  HIPCHECK(h, put(h->d_operm.p, operm.data(), (size_t)N * sizeof(int)));
  HIPCHECK(h, put(h->pt_off.p, hoff, ((size_t)nt + 1) * sizeof(int)));
  HIPCHECK(h, put(h->pt_opt.p, iopt.data(), (size_t)nt));
  HIPCHECK(h, put(h->d_pperm.p, hperm, (size_t)nt * sizeof(int)));
  if (pk.ok && h->opt.packed_upload) { const int rc = pk.flush(); if (rc != BA_OK) return rc; }
  return BA_OK;
}


}  // namespace

namespace ba {

// camera-ordered view of the observations for k_camera_blocks (only when HCC / bC are asked for through the API or a
// reduction kernel that does not form the camera blocks itself is in use): a stable sort by camera on the device, units on the host
int ensure_cam_units(ba_handle* h) {
  if (h->cam_units_built) return BA_OK;
  const int nc = h->nc;
  const long long N = h->nobs;
  auto& su = h->su;
  std::vector<CamUnit> cam_units;
  if (N > 0 && nc > 0) {
    HIPCHECK(h, su.cnt.resize((size_t)nc + 1));
    HIPCHECK(h, su.key.resize((size_t)N)); HIPCHECK(h, su.key2.resize((size_t)N)); HIPCHECK(h, su.vals.resize((size_t)N));
    HIPCHECK(h, h->cam_perm.resize((size_t)N));
    HIPCHECK(h, hipMemsetAsync(su.cnt.p, 0, ((size_t)nc + 1) * sizeof(int), h->stream));
    hipLaunchKernelGGL(k_setup_cam_hist, dim3(grid_for(N)), dim3(256), 0, h->stream, N, h->obs_cam.p, su.cnt.p, su.key.p);
    hipLaunchKernelGGL(k_iota, dim3(grid_for(N)), dim3(256), 0, h->stream, (int)N, su.vals.p);
    HIPCHECK(h, sort_pairs_u64(h, su.key.p, su.key2.p, su.vals.p, h->cam_perm.p, (size_t)N, bits_for((unsigned long long)nc)));
    std::vector<int> cnt((size_t)nc);
    HIPCHECK(h, hipMemcpyAsync(cnt.data(), su.cnt.p, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    // one wavefront per unit: few cameras with long observation lists (dense visibility) would leave the chip
    // empty at kCamChunk observations per unit, so shrink the chunk until there are about 500 units (measured: 100 000 observations of 100 cameras: 28 us at 2048 per unit, 14 at 256, 19 at 64); many
    // cameras with ~1000 observations each keep one unit per camera (one atomic result per camera)
    const int cam_chunk = (int)std::min<int64_t>(kCamChunk, std::max<int64_t>(64, (N / 512 + 63) / 64 * 64));
    int begin = 0;
    for (int i = 0; i < nc; ++i) {
      const int end = begin + cnt[i];
      for (int s = begin; s < end; s += cam_chunk) cam_units.push_back({i, s, std::min(s + cam_chunk, end)});
      begin = end;
    }
  }
  HIPCHECK(h, h->cam_units.resize(std::max<size_t>(1, cam_units.size())));
  if (!cam_units.empty()) {
    HIPCHECK(h, hipMemcpyAsync(h->cam_units.p, cam_units.data(), cam_units.size() * sizeof(CamUnit), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
  }
  h->ncam_units = (int)cam_units.size();
  h->cam_units_built = true;
  return BA_OK;
}

// Schur work units (point, row tile, col tile >= row tile) of k_schur_pairs and their chunks: consecutive units whose
// optimised-camera positions fit a window of wn band rows, so that a workgroup can accumulate them in an LDS tile
int ensure_pair_units(ba_handle* h) {
  if (h->pair_units_built) return BA_OK;
  const int nt = h->nt, wn = h->schur_wn;
  const std::vector<int>&off = h->h_off, &plo = h->h_plo, &phi = h->h_phi;
  std::vector<SchurUnit> units;
  units.reserve((size_t)nt);
  for (int k = 0; k < nt; ++k) {
    const int L = off[(size_t)k + 1] - off[k];
    for (int r = 0; r < L; r += kTile)
      for (int c = r; c < L; c += kTile) units.push_back({k, r, c});
  }
  std::vector<SchurChunk> chunks;
  {
    int begin = 0, lo = INT32_MAX, hi = -1;
    for (int u = 0; u < (int)units.size(); ++u) {
      const int k = units[u].pt;
      const int nlo = std::min(lo, plo[k]), nhi = std::max(hi, phi[k]);
      const bool fits = wn == 0 || nhi < 0 || nhi - nlo + 1 <= wn;
      if (u > begin && (!fits || u - begin >= kSchurChunkUnits)) {
        chunks.push_back({begin, u, lo == INT32_MAX ? 0 : lo});
        begin = u; lo = plo[k]; hi = phi[k];
      } else {
        lo = nlo; hi = nhi;
      }
    }
    if (!units.empty()) chunks.push_back({begin, (int)units.size(), lo == INT32_MAX ? 0 : lo});
  }
  HIPCHECK(h, h->units.resize(std::max<size_t>(1, units.size())));
  HIPCHECK(h, h->chunks.resize(std::max<size_t>(1, chunks.size())));
  if (!units.empty()) {
    HIPCHECK(h, hipMemcpyAsync(h->units.p, units.data(), units.size() * sizeof(SchurUnit), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(h->chunks.p, chunks.data(), chunks.size() * sizeof(SchurChunk), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
  }
  h->nunits = (int)units.size();
  h->nchunks = (int)chunks.size();
  h->pair_units_built = true;
  return BA_OK;
}


// ---- the work lists of the general kernels (groups, chunks, windows, segment pairs) from what ba_set_problem kept of the scene
int ensure_plan(ba_handle* h) {
  if (!h->plan_pending) return BA_OK;
  const int nt = h->nt, nco = h->nco, hb = h->hb, wn = h->schur_wn;
  const long long nobs = h->nobs, maxL = h->group_maxL;
  const int* off = h->h_off.data();
  const int* plo = h->h_plo.data();
  const int* phi = h->h_phi.data();
  const unsigned char* same = h->h_same.data();
  const int* flags = h->plan_flags.data();
  const int* cam_opt_pos = h->h_cam_opt_pos.data();
  const int* obs_cam = nullptr;                   // (fetched from the device by the one branch that looks at observations)
  (void)nobs; (void)hb; (void)nco; (void)plo; (void)obs_cam;
  HIPCHECK(h, hipSetDevice(h->device));
  // Groups: runs of consecutive points (internal order) with identical observation lists.
  //   groups / gchunks   <= kGroupMaxPts points each: k_schur_groups (vector kernel, track length <= 15) and the
  //                      group-packed point kernels k_linearize_groups / k_backsub_groups (<= kGm3MaxL)
  //   mgroups / mchunks  longer runs for the matrix-core reductions; mchunks under the LDS window `wn` of the older
  //                      kernels (track length <= 10), m3chunks under k_schur_groups_mfma3's own window
  std::vector<SchurGroup> groups, mgroups;
  std::vector<SchurChunk> gchunks, mchunks, m3chunks;
  int group_rounds = 0;
  // points per run of identical camera lists (runs are cut at kGroupMaxPts = 24) from which the kernels that work run by run pay:
  // the lineariser / back-substitution a wavefront per run (below, their lanes idle), the matrix-core reduction a run per
  // wavefront pair with an epilogue of its own (every extra run costs a pair ~20 us: only scenes that are ALL long runs)
  constexpr double kGroupsWorthMean = 9.0, kMfmaGroupsWorthMean = 20.0;
  bool groups_worth = false, mgroups_worth = false, mgroups_any = false;
  const bool groups_ascending = flags[SF_NOT_ASC] == 0;      // optimised positions ascend along every track (always, with the internal sort)
  Gm3Params gm3{0, 0, 0, 0, 0, 1, 1};
  if (maxL >= 1 && maxL <= kGm3MaxL) {
    auto build_groups = [&](int max_pts, std::vector<SchurGroup>& gs, std::vector<int>& glo, std::vector<int>& ghi) {
      for (int k = 0; k < nt;) {
        const int L = off[(size_t)k + 1] - off[k];
        if (L == 0) { ++k; continue; }
        int e = k + 1;
        while (e < nt && e - k < max_pts && same[e]) ++e;      // (same[e]: the camera list of point e equals that of point e - 1)
        gs.push_back({k, e, L, 0});
        glo.push_back(plo[k]); ghi.push_back(phi[k]);
        k = e;
      }
    };
    // consecutive groups whose optimised positions fit a window of `win` band rows (win == 0: no window, by count only)
    auto chunk_groups = [&](int limit, int win, const std::vector<SchurGroup>& gs, const std::vector<int>& glo, const std::vector<int>& ghi,
                            std::vector<SchurChunk>& out) {
      int begin = 0, lo = INT32_MAX, hi = -1;
      for (int g = 0; g < (int)gs.size(); ++g) {
        const int nlo = std::min(lo, glo[g]), nhi = std::max(hi, ghi[g]);
        const bool fits = win == 0 || nhi < 0 || nhi - nlo + 1 <= win;
        if (g > begin && (!fits || g - begin >= limit)) {
          out.push_back({begin, g, lo == INT32_MAX ? 0 : lo});
          begin = g; lo = glo[g]; hi = ghi[g];
        } else {
          lo = nlo; hi = nhi;
        }
      }
      if (!gs.empty()) out.push_back({begin, (int)gs.size(), lo == INT32_MAX ? 0 : lo});
    };
    std::vector<int> glo, ghi, mlo, mhi;
    build_groups(kGroupMaxPts, groups, glo, ghi);
    if (maxL <= kGroupMaxL) chunk_groups(kGroupChunk, wn, groups, glo, ghi, gchunks);
    // MFMA kernels: one group per wavefront pair, and the epilogue is expensive, so runs are cut
    // only where the chip would otherwise idle: about one group per wavefront-pair slot (4 per CU)
    const int ncu = h->ncu;
    const int slots = std::max(1, ncu * (kGmBlock / kWave));
    // The kernel lasts as long as its longest group (one round of workgroups), so: natural runs of points with
    // identical camera lists, runs longer than `cap` cut into EQUAL parts of whole batches, and the smallest cap
    // for which the groups still fit the wavefront-pair slots of one round (config 3: 991 runs of 101 +- 23
    // points, cap 120 -> 1021 groups in 256 workgroups; a fixed 1.25 x mean cap gave groups of 126).
    std::vector<SchurGroup> runs;
    std::vector<int> rlo, rhi;
    build_groups(INT32_MAX, runs, rlo, rhi);
    auto split_runs = [&](int cap, bool emit) -> size_t {
      size_t count = 0;
      for (size_t r = 0; r < runs.size(); ++r) {
        const int n = runs[r].pt_end - runs[r].pt_begin;
        const int k = (n + cap - 1) / cap;
        const int part = ((n + k - 1) / k + kGmPts - 1) / kGmPts * kGmPts;
        for (int b = runs[r].pt_begin; b < runs[r].pt_end; b += part) {
          ++count;
          if (emit) {
            mgroups.push_back({b, std::min(b + part, runs[r].pt_end), runs[r].L, 0});
            mlo.push_back(rlo[r]); mhi.push_back(rhi[r]);
          }
        }
      }
      return count;
    };
    int cap = std::max(kGroupMaxPts, (int)(((nt + slots - 1) / slots + kGmPts - 1) / kGmPts * kGmPts));
    int longest = 0;
    for (const SchurGroup& r : runs) longest = std::max(longest, r.pt_end - r.pt_begin);
    while (cap < longest && split_runs(cap, false) > (size_t)slots) cap += kGmPts;     // (beyond the longest run nothing changes)
    // More runs than slots (config 5; a shard of it; any scene a little larger than one round): several workgroups per compute
    // unit, one after the other, and the cap above has grown past every run - the 500-point runs at the two ends of a camera
    // track then last four times as long as everybody else.  Measured (scripts/gm_cap_sweep.sh: a workgroup costs 13.4 us +
    // 2.0 us per batch of its longest group): parts of at most 60 points are the best size for every multi-round scene tried
    // (1250 cameras x 125k points: 278 -> 101 us; config 5: 722 -> 658 us).
    const bool multi_round = split_runs(cap, false) > (size_t)slots;
    if (multi_round) cap = kGmMultiRoundCap;
    if (h->opt.gm_cap > 0) cap = std::max(kGmPts, h->opt.gm_cap);     // tuning aid (ba_set_option "gm_cap")
    split_runs(cap, true);
    // Groups per workgroup: one per wavefront pair when everything runs at once.  With several rounds, one, two or three per
    // pair: a workgroup's set-up and flush (about 5 of its 13.4 us of fixed cost) are then paid once for two or three times the
    // work - but fewer, longer workgroups quantise worse over the compute units, so: the count that minimises
    // rounds x (5 + groups per pair x (8.4 + 2.0 x batches)) us, rounds rounded up while there are fewer than three
    // (scripts/gm_cap_sweep.sh, profiles/r04_gm_cap_sweep.txt: predicted / measured 100 / 102, 124 / 121, 90 / 93 us for 4, 8, 12
    // groups per workgroup at 1250 cameras x 125k points, 676 / 658, 641 / 597, 639 / 596 us at config 5).
    int gm_chunk = kGmChunk;
    if (multi_round) {
      const double G = (double)mgroups.size(), batches = std::ceil(std::min<double>(cap, 1.15 * nt / std::max(1.0, G)) / kGmPts);
      double best = 0.0;
      for (int c = kGmChunk; c <= 3 * kGmChunk; c += kGmChunk) {
        const double wgs = std::ceil(G / c) / h->ncu, rounds = wgs <= 1.0 ? 1.0 : wgs < 3.0 ? std::ceil(wgs) : wgs + 0.5;
        const double t = rounds * (5.0 + (c / kGmChunk) * (8.4 + 2.0 * batches));
        if (c == kGmChunk || t < best) { best = t; gm_chunk = c; }
      }
    }
    if (h->opt.gm_chunk > 0) gm_chunk = h->opt.gm_chunk;
    if (maxL <= kGmMaxL && wn > 0) chunk_groups(gm_chunk, wn, mgroups, mlo, mhi, mchunks);
    // worth it only when points really share camera lists
    const double mean_group = groups.empty() ? 0.0 : (double)nt / groups.size();
    // (a run of identical camera lists is worked on six points at a time and has an epilogue of its own: with runs of two or three
    //  points - a fifth of the observations missing - the run kernels took 2.5 times what the window groups and the point-per-
    //  lane-group kernels take: 0.73 against 0.29 ms per trial)
    groups_worth = mean_group >= kGroupsWorthMean;
    mgroups_worth = mean_group >= kMfmaGroupsWorthMean;
    mgroups_any = mean_group >= 2.0;               // (still better than the vector kernels where there are no window groups)
    if (groups_worth && maxL <= kGroupMaxL && wn > 0) group_rounds = (int)((maxL * (maxL + 1) / 2 + 63) / 64);
  }
  // Window groups for k_schur_groups_mfma3: consecutive points (internal order: by first optimised position) whose
  // optimised cameras all lie within `wmax` consecutive positions - identical camera lists are NOT required, so tracks
  // of different lengths, tracks with missing observations and tracks that start anywhere all join.  wmax = the
  // widest window that costs no more 16-row tiles than the widest track needs.
  std::vector<WinGroup> wgroups;
  int gm3_chunk = kGmChunk;                          // window groups per workgroup
  size_t wtab_size = 0;
  std::vector<int> hobs;
  bool wgroups_worth = false;
  std::vector<RectGroup> rgroups;
  std::vector<int> rtab, wide_list;
  int wide_begin[kGwMaxTiles - kGwMinTiles + 2] = {0};
  int nlong_points = 0;
  {
    int maxspan = 0;
    for (int k = 0; k < nt; ++k)
      if (phi[k] >= 0) maxspan = std::max(maxspan, phi[k] - plo[k] + 1);
    bool sorted_by_lo = true;                          // (the internal sort guarantees it; "sort_points" = 0 may not)
    for (int k = 1, last = -1; k < nt && sorted_by_lo; ++k) {
      if (phi[k - 1] >= 0) last = plo[k - 1];
      if (phi[k] >= 0 && plo[k] < last) sorted_by_lo = false;
    }
    // Points whose optimised cameras span MORE than the widest window (features that survive for a long stretch of a video):
    // their cameras are cut along a grid of SEGMENTS of kRectSeg positions, and what a point adds to S is a sum over the pairs
    // (A <= B) of segments it touches, S[A, B] -= U_A^T D U_B: k_schur_rect_mfma over the points that touch both (rgroups).
    // A == B also carries the right-hand side and the camera blocks - every observation lies in exactly one segment.
    auto is_long = [&](int k) { return phi[k] >= 0 && phi[k] - plo[k] + 1 > kGm3MaxSpan; };
    long long nlong = 0;
    int shortspan = 0;
    for (int k = 0; k < nt; ++k) {
      if (phi[k] < 0) continue;
      if (is_long(k)) ++nlong; else shortspan = std::max(shortspan, phi[k] - plo[k] + 1);
    }
    bool hybrid = nlong > 0 && sorted_by_lo && nco > 0;
    if (hybrid) {
      // Long tracks that are SCATTERED rather than long (an unordered photo collection: three cameras anywhere among five thousand)
      // touch pairs of segments all over the matrix with one or two observations each: the segment kernel would linearise table
      // rows of 64 entries that are 97 % empty, behind a table of 300 MB that takes 240 ms of every ba_set_problem to build (5000
      // cameras).  When the long tracks fill less than a tenth of the positions they span, they are left to the general kernels
      // (k_schur_pairs, or k_schur_blocks on the sparse path) and no table is built.
      long long filled = 0, spanned = 0;
      for (int k = 0; k < nt; ++k)
        if (is_long(k)) { filled += off[(size_t)k + 1] - off[k]; spanned += phi[k] - plo[k] + 1; }
      if (10 * filled < spanned) hybrid = false;
    }
    struct SegTask { int qa, qb; std::vector<int> pts; };
    std::vector<SegTask> rect_tasks;
    if (hybrid) {
      maxspan = shortspan;
      nlong_points = (int)nlong;
      // (the one place where the host looks at observations: the few long tracks' cameras)
      hobs.resize((size_t)nobs);
      HIPCHECK(h, hipMemcpyAsync(hobs.data(), h->obs_cam.p, (size_t)nobs * sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(h, hipStreamSynchronize(h->stream));
      obs_cam = hobs.data();
      std::map<long long, int> rect_id;
      for (int k = 0; k < nt; ++k) {
        if (!is_long(k)) continue;
        std::vector<int> segs;
        for (int n = off[k]; n < off[(size_t)k + 1]; ++n) {
          const int p = cam_opt_pos[obs_cam[n]];
          if (p >= 0) segs.push_back(p / kRectSeg);
        }
        std::sort(segs.begin(), segs.end());
        segs.erase(std::unique(segs.begin(), segs.end()), segs.end());
        for (size_t x = 0; x < segs.size(); ++x) {
          for (size_t y = x; y < segs.size(); ++y) {
            const long long key = ((long long)segs[x] << 32) | (unsigned)segs[y];
            auto jt = rect_id.find(key);
            if (jt == rect_id.end()) { jt = rect_id.emplace(key, (int)rect_tasks.size()).first; rect_tasks.push_back({segs[x], segs[y], {}}); }
            rect_tasks[jt->second].pts.push_back(k);
          }
        }
      }
    }
    if ((maxspan >= 1 || hybrid) && maxspan <= kGm3MaxSpan && sorted_by_lo && nco > 0) {
      gm3.nts = (6 * maxspan + 15) / 16;
      gm3.Ld = 16 * gm3.nts;
      const int wmax = std::max(1, std::min(kGm3MaxSpan, gm3.Ld / 6));
      // natural groups: extend while the window still holds everybody
      struct Run { int b, e, lo, hi; };
      std::vector<Run> runs;
      for (int k = 0; k < nt;) {
        if (phi[k] < 0 || (hybrid && is_long(k))) { ++k; continue; }      // (points without an optimised camera add nothing to S or b; long ones are members of their segments' groups)
        int e = k + 1, lo = plo[k], hi = phi[k];
        while (e < nt && !(hybrid && is_long(e)) && (phi[e] < 0 || std::max(hi, phi[e]) - lo + 1 <= wmax)) {
          if (phi[e] >= 0) hi = std::max(hi, phi[e]);
          ++e;
        }
        while (e > k + 1 && phi[e - 1] < 0) --e;       // no trailing points without cameras
        runs.push_back({k, e, lo, hi});
        k = e;
      }
      // cut long groups into equal parts so that one round of workgroups holds them all and none lasts much longer
      // than the rest (as for the identical-list groups above)
      const int ncu = h->ncu;
      const int slots = std::max(1, ncu * kGm2Pairs);
      auto parts_of = [&](const Run& r, int cap) { return (r.e - r.b + cap - 1) / cap; };
      int cap = std::max(kGroupMaxPts, (int)(((nt + slots - 1) / slots + kGmPts - 1) / kGmPts * kGmPts));
      int longest = 0;
      for (const Run& r : runs) longest = std::max(longest, r.e - r.b);
      auto count = [&](int c) { size_t n = 0; for (const Run& r : runs) n += parts_of(r, c); return n; };
      while (cap < longest && count(cap) > (size_t)slots) cap += kGmPts;
      // several rounds of workgroups (see the identical-list groups above): parts of 60 points while that is a matter of two or
      // three rounds (1250 cameras / 125k points, 30 % of the observations dropped: 120 -> 111 us; track length 16 at 2000 cameras:
      // 402 -> 342 us), of 120 beyond (a window group pays more per group than a run: config 5 with 30 % dropped: 628 against 722 us)
      const bool multi_round = count(cap) > (size_t)slots;
      if (multi_round) cap = count(2 * kGmMultiRoundCap) > (size_t)4 * slots ? 2 * kGmMultiRoundCap : kGmMultiRoundCap;
      gm3_chunk = h->opt.gm_chunk > 0 ? h->opt.gm_chunk : kGmChunk;       // (window groups: more than one per pair was slower wherever measured)
      if (h->opt.gm_cap > 0) cap = std::max(kGmPts, h->opt.gm_cap);
      std::vector<int> wlo, whi;
      for (const Run& r : runs) {
        const int n = r.e - r.b, k = parts_of(r, cap);
        const int part = ((n + k - 1) / k + kGmPts - 1) / kGmPts * kGmPts;
        for (int b0 = r.b; b0 < r.e; b0 += part) {
          const int e0 = std::min(b0 + part, r.e);
          int lo = INT32_MAX, hi = -1;
          for (int q = b0; q < e0; ++q)
            if (phi[q] >= 0) { lo = std::min(lo, plo[q]); hi = std::max(hi, phi[q]); }
          if (hi < 0) continue;
          const int W = hi - lo + 1;
          if (wtab_size + (size_t)(e0 - b0) * W > (size_t)INT32_MAX) return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: the window-group tables exceed 2^31 entries");
          WinGroup g{b0, e0, W, lo, (int)wtab_size, 0, 0, 0};
          wtab_size += (size_t)(e0 - b0) * W;               // (filled on the device: k_setup_fill_wtab)
          wgroups.push_back(g);
          wlo.push_back(lo); whi.push_back(hi);
        }
      }
      // groups of 27 .. 40 cameras go through k_schur_wide_mfma (every observation linearised once): narrow ones first.  When the
      // narrow ones are few beside them (tracks cut short by the end of the sequence), they go the same way - their own two to
      // five launches of k_schur_groups_mfma3 would each last as long as one group
      {
        auto tiles = [&](int g) { return (6 * wgroups[g].W + 15) >> 4; };
        long long pn = 0, pw = 0;
        for (size_t g = 0; g < wgroups.size(); ++g) (tiles((int)g) >= kGwMinTiles ? pw : pn) += wgroups[g].pt_end - wgroups[g].pt_begin;
        const bool all_wide = pw > 0 && 4 * pn <= pw;
        auto cls = [&](int g) { return (tiles(g) >= kGwMinTiles || all_wide) ? std::max(tiles(g), kGwMinTiles) : 0; };      // 0: narrow
        std::vector<int> order(wgroups.size());
        for (size_t g = 0; g < order.size(); ++g) order[g] = (int)g;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cls(x) < cls(y); });
        std::vector<WinGroup> wg2(wgroups.size());
        std::vector<int> lo2(wgroups.size()), hi2(wgroups.size()), cl2(wgroups.size());
        for (size_t g = 0; g < order.size(); ++g) { wg2[g] = wgroups[order[g]]; lo2[g] = wlo[order[g]]; hi2[g] = whi[order[g]]; cl2[g] = cls(order[g]); }
        wgroups.swap(wg2); wlo.swap(lo2); whi.swap(hi2);
        for (size_t g = 0; g < cl2.size(); ++g)
          if (cl2[g] > 0) wide_list.push_back((int)g);
        for (int t = kGwMinTiles; t <= kGwMaxTiles + 1; ++t) {      // wide_list is sorted by class: where class t begins
          int n = 0;
          for (size_t g = 0; g < cl2.size(); ++g) n += (cl2[g] > 0 && cl2[g] < t) ? 1 : 0;
          wide_begin[t - kGwMinTiles] = n;
        }
      }
      const int nshort_groups = (int)wgroups.size() - (int)wide_list.size();
      if (nshort_groups > 0) {                              // the narrow groups' launches: as many tiles as THEY need
        int wn = 1;
        for (int g = 0; g < nshort_groups; ++g) wn = std::max(wn, wgroups[g].W);
        gm3.nts = (6 * wn + 15) / 16;
        gm3.Ld = 16 * gm3.nts;
      }
      // points per group: a group is one serial chain of batches on one workgroup, and there are few long tracks - halve the
      // groups until they fill the chip twice (24 points at least: an epilogue of up to 144 tiles is paid per group)
      int rect_pts = kRectGroupPts;
      for (; rect_pts > 24; rect_pts /= 2) {
        size_t ngr = 0;
        for (const SegTask& t : rect_tasks) ngr += (t.pts.size() + rect_pts - 1) / rect_pts;
        if (ngr >= (size_t)2 * ncu) break;                 // one workgroup per group, one workgroup per compute unit (its consumers' registers)
      }
      // groups of the pairs of segments: table rows of 2 kRectSeg columns [A | B] (A == B: the B half stays empty)
      for (const SegTask& t : rect_tasks) {
        const int loA = t.qa * kRectSeg, loB = t.qb * kRectSeg, WB = std::min(kRectSeg, nco - loB);
        const size_t parts = (t.pts.size() + rect_pts - 1) / rect_pts, part = (t.pts.size() + parts - 1) / parts;
        for (size_t b0 = 0; b0 < t.pts.size(); b0 += part) {
          const int cnt = (int)std::min<size_t>(part, t.pts.size() - b0);
          RectGroup g{cnt, (int)rtab.size(), 0, loA, loB, WB, 0, 0};
          rtab.insert(rtab.end(), t.pts.begin() + b0, t.pts.begin() + b0 + cnt);
          g.tab = (int)rtab.size();
          rtab.resize(rtab.size() + (size_t)cnt * 2 * kRectSeg, -1);
          for (int q = 0; q < cnt; ++q) {
            const int k = t.pts[b0 + q];
            for (int n2 = off[k]; n2 < off[(size_t)k + 1]; ++n2) {
              const int p = cam_opt_pos[obs_cam[n2]];
              if (p >= loA && p < loA + kRectSeg) rtab[(size_t)g.tab + (size_t)q * 2 * kRectSeg + (p - loA)] = n2;
              else if (t.qa != t.qb && p >= loB && p < loB + WB) rtab[(size_t)g.tab + (size_t)q * 2 * kRectSeg + kRectSeg + (p - loB)] = n2;
            }
          }
          rgroups.push_back(g);
        }
      }
      // staging: four wavefront pairs with two buffers each must fit in LDS next to the (optional) accumulation window
      auto finish_set = [&](int g0, int g1, Gm3Params& G, std::vector<SchurChunk>& out) {
        if (g1 <= g0) return;
        const size_t lds_total = 160 * 1024, fixed = schur_mfma3_lds_bytes(0, 0, 0, hb + 1) + 1024;
        G.wb1 = 1;                                        // rows of the LDS window: as many blocks as the widest group spans
        for (int g = g0; g < g1; ++g) G.wb1 = std::max(G.wb1, std::min(hb + 1, wgroups[g].W));
        for (G.np_cap = kGmPts; G.np_cap >= 1; --G.np_cap) {
          int kmax = 4;
          for (int g = g0; g < g1; ++g) kmax = std::max(kmax, (3 * gm3_np(wgroups[g].W, G.np_cap) + 3) / 4 * 4);
          G.Kbuf = kmax;
          if ((size_t)kGm2Pairs * 2 * G.Kbuf * G.Ld * sizeof(double) <= 96 * 1024) break;
        }
        G.np_cap = std::max(1, G.np_cap);
        const size_t staging = (size_t)kGm2Pairs * 2 * G.Kbuf * G.Ld * sizeof(double);
        const size_t rowbytes = ((size_t)G.wb1 * 36 + 6) * sizeof(double);
        int w3 = (int)((lds_total - fixed - staging) / rowbytes);
        w3 = std::min(w3, std::max(16, G.wb1 + 6));
        if (w3 < G.wb1 + 1 || !h->opt.lds_window) w3 = 0;
        G.wn = w3;
        // chunks of <= gm3_chunk groups under the LDS window.  A chunk may reach `over` rows past the window (the kernel adds what
        // falls outside straight to S): when the strict rule makes chunks of three groups and with them more workgroups than the
        // chip runs at once (tracks of 16 cameras: 334 chunks of 3 on 256 units = two rounds, 262 us against 138 for tracks
        // of 15), a row or two of global atomics per chunk is the cheaper price.
        auto build = [&](int over, std::vector<SchurChunk>& dst) {
          int begin = g0, lo = INT32_MAX, hi = -1;
          for (int g = g0; g < g1; ++g) {
            const int nlo = std::min(lo, wlo[g]), nhi = std::max(hi, whi[g]);
            const bool fits = w3 == 0 || nhi - nlo + 1 <= w3 + over;
            if (g > begin && (!fits || g - begin >= gm3_chunk)) {
              dst.push_back({begin, g, lo});
              begin = g; lo = wlo[g]; hi = whi[g];
            } else {
              lo = nlo; hi = nhi;
            }
          }
          dst.push_back({begin, g1, lo});
        };
        std::vector<SchurChunk> strict;
        build(0, strict);
        if (w3 > 0 && (int)strict.size() > ncu) {
          for (int over = 1; over <= std::max(1, G.wb1 / 4); ++over) {
            std::vector<SchurChunk> relaxed;
            build(over, relaxed);
            if ((relaxed.size() + ncu - 1) / ncu < (strict.size() + ncu - 1) / ncu) { strict.swap(relaxed); break; }
          }
        }
        out.insert(out.end(), strict.begin(), strict.end());
      };
      finish_set(0, nshort_groups, gm3, m3chunks);
      // an epilogue per >= 12 points (the short tracks' groups decide; a scene of nothing but long tracks: its segment groups)
      long long covered = 0;
      for (const WinGroup& g : wgroups) covered += g.pt_end - g.pt_begin;
      wgroups_worth = wgroups.empty() ? !rgroups.empty() : covered >= 12ll * (long long)wgroups.size();
    }
  }
  h->ngchunks = (int)gchunks.size();
  h->nmchunks = (int)mchunks.size();
  h->nm3chunks = (int)m3chunks.size();
  h->nwgroups = (int)wgroups.size();
  h->gm3_uniform_ks = !m3chunks.empty();
  for (const SchurChunk& c : m3chunks)
    for (int g = c.begin; g < c.end; ++g) h->gm3_uniform_ks = h->gm3_uniform_ks && gm3_np(wgroups[g].W, gm3.np_cap) == kGmPts;
  h->wgroups_worth = wgroups_worth;
  h->nrgroups = (int)rgroups.size();
  h->nwide = (int)wide_list.size();
  std::memcpy(h->wide_begin, wide_begin, sizeof wide_begin);
  h->nlong_points = rgroups.empty() ? 0 : nlong_points;
  h->gm3 = gm3;
  h->groups_worth = groups_worth;
  h->mgroups_worth = mgroups_worth;
  h->mgroups_any = mgroups_any;
  h->nmgroups_total = (int)mgroups.size();
  h->groups_ascending = groups_ascending;
  h->ngroups = (int)groups.size();
  {
    long long covered = 0;
    for (const SchurGroup& g : groups) covered += g.pt_end - g.pt_begin;
    h->point_groups = groups_worth && covered == nt;        // (points without observations are in no group)
  }
  h->group_rounds = group_rounds;
  HIPCHECK(h, h->wide_list.resize(std::max<size_t>(1, wide_list.size())));
  if (!wide_list.empty())
    HIPCHECK(h, stage_h2d(h, h->wide_list.p, wide_list.data(), wide_list.size() * sizeof(int)));
  HIPCHECK(h, h->rgroups.resize(std::max<size_t>(1, rgroups.size())));
  HIPCHECK(h, h->rtab.resize(std::max<size_t>(1, rtab.size())));
  if (!rgroups.empty()) {
    HIPCHECK(h, stage_h2d(h, h->rgroups.p, rgroups.data(), rgroups.size() * sizeof(RectGroup)));
    HIPCHECK(h, stage_h2d(h, h->rtab.p, rtab.data(), rtab.size() * sizeof(int)));
  }
  HIPCHECK(h, h->groups.resize(std::max<size_t>(1, groups.size())));
  HIPCHECK(h, h->gchunks.resize(std::max<size_t>(1, gchunks.size())));
  HIPCHECK(h, h->mchunks.resize(std::max<size_t>(1, mchunks.size())));
  HIPCHECK(h, h->m3chunks.resize(std::max<size_t>(1, m3chunks.size())));
  if (!m3chunks.empty())
    HIPCHECK(h, stage_h2d(h, h->m3chunks.p, m3chunks.data(), m3chunks.size() * sizeof(SchurChunk)));
  HIPCHECK(h, h->wgroups.resize(std::max<size_t>(1, wgroups.size())));
  HIPCHECK(h, h->wtab.resize(std::max<size_t>(1, wtab_size)));
  if (!wgroups.empty()) {
    HIPCHECK(h, stage_h2d(h, h->wgroups.p, wgroups.data(), wgroups.size() * sizeof(WinGroup)));
    HIPCHECK(h, hipMemsetAsync(h->wtab.p, 0xff, wtab_size * sizeof(int), h->stream));        // -1: "the point does not see this camera"
    hipLaunchKernelGGL(k_setup_fill_wtab, dim3((unsigned)wgroups.size()), dim3(256), 0, h->stream, h->wgroups.p, h->pt_off.p, h->obs_cam.p,
                       h->nbc > 0 ? h->cam_band_pos.p : h->cam_opt_pos.p, h->wtab.p);
  }
  HIPCHECK(h, h->mgroups.resize(std::max<size_t>(1, mgroups.size())));
  if (!mgroups.empty())
    HIPCHECK(h, stage_h2d(h, h->mgroups.p, mgroups.data(), mgroups.size() * sizeof(SchurGroup)));
  if (!mchunks.empty())
    HIPCHECK(h, stage_h2d(h, h->mchunks.p, mchunks.data(), mchunks.size() * sizeof(SchurChunk)));
  if (!groups.empty()) {
    HIPCHECK(h, stage_h2d(h, h->groups.p, groups.data(), groups.size() * sizeof(SchurGroup)));
    if (!gchunks.empty())
      HIPCHECK(h, stage_h2d(h, h->gchunks.p, gchunks.data(), gchunks.size() * sizeof(SchurChunk)));
  }
  h->plan_pending = false;
  if (h->su.up_pageable) { HIPCHECK(h, hipStreamSynchronize(h->stream)); h->su.up_pageable = false; }
  else h->su.up_pending = true;
  return BA_OK;
}

}  // namespace ba

namespace {

// the problem with the optimised cameras at the positions given (ba_set_problem below chooses them)
// cam_band_pos (nbc > 0 border cameras, ba_border.h): the positions as the band sees them - a border camera (true position >=
// nco - nbc) is -1 there, like a camera that is not optimised; nullptr = no border
int set_problem_impl(ba_handle* h, int32_t nc, int32_t nt, int64_t nobs, const int32_t* obs_cam,
                     const int32_t* obs_pt, const double* obs_z, const double* K,
                     const int32_t* cam_opt_pos, const uint8_t* pt_opt, const int32_t* cam_band_pos = nullptr, int nbc = 0) {
  if (!cam_band_pos) { cam_band_pos = cam_opt_pos; nbc = 0; }
  REQUIRE(h, nc >= 0 && nt >= 0 && nobs >= 0, BA_ERR_INVALID_ARG, "ba_set_problem: negative size");
  REQUIRE(h, nobs < (1ll << 31) - 64, BA_ERR_INVALID_ARG, "ba_set_problem: nobs must fit int32");
  REQUIRE(h, K && (nc == 0 || cam_opt_pos) && (nt == 0 || pt_opt), BA_ERR_INVALID_ARG,
          "ba_set_problem: NULL argument");
  REQUIRE(h, nobs == 0 || (obs_cam && obs_pt && obs_z), BA_ERR_INVALID_ARG, "ba_set_problem: NULL observation array");
  HIPCHECK(h, hipSetDevice(h->device));
  h->have_problem = false;
  h->res_out_phys = -1;

  // ---- cameras (host, O(nc)): optimised-camera positions must be a permutation of 0..nco-1; the RANK of a camera orders a
  // track's observations: frozen cameras by index, then the optimised ones by position
  int nco = 0;
  for (int i = 0; i < nc; ++i) if (cam_opt_pos[i] >= 0) ++nco;
  std::vector<int> crank((size_t)std::max(1, nc)), opt_cam((size_t)std::max(1, nco), 0);
  {
    std::vector<char> seen((size_t)nco, 0);
    int f = 0;
    const int nfrozen = nc - nco;
    for (int i = 0; i < nc; ++i) {
      const int p = cam_opt_pos[i];
      if (p < 0) { crank[i] = f++; continue; }
      if (p >= nco || seen[p]) return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: cam_opt_pos is not a permutation of 0..nco-1");
      seen[p] = 1;
      crank[i] = nfrozen + p;
      opt_cam[p] = i;
    }
  }
  const bool sort_points = h->opt.sort_points;
  const long long N = nobs;
  auto& su = h->su;
  const int rank_bits = bits_for((unsigned long long)std::max(1, nc)), track_bits = bits_for((unsigned long long)std::max(1, nt));

  // ---- the caller's arrays to the device
  HIPCHECK(h, su.rc.resize(std::max<size_t>(1, N))); HIPCHECK(h, su.rp.resize(std::max<size_t>(1, N))); HIPCHECK(h, su.rz.resize(std::max<size_t>(1, N)));
  HIPCHECK(h, su.key.resize(std::max<size_t>(1, N)));
  HIPCHECK(h, su.cnt.resize((size_t)nt + 2)); HIPCHECK(h, su.coff.resize((size_t)nt + 2)); HIPCHECK(h, su.Lint.resize((size_t)nt + 2));
  HIPCHECK(h, su.crank.resize(crank.size())); HIPCHECK(h, su.rpo.resize(std::max(1, nt))); HIPCHECK(h, su.flags.resize(SF_COUNT));
  HIPCHECK(h, su.plo.resize(std::max(1, nt))); HIPCHECK(h, su.phi.resize(std::max(1, nt))); HIPCHECK(h, su.same.resize(std::max(1, nt)));
  HIPCHECK(h, su.tkey.resize(std::max(1, nt))); HIPCHECK(h, su.tkey2.resize(std::max(1, nt))); HIPCHECK(h, su.iota.resize(std::max(1, nt)));
  HIPCHECK(h, h->d_pperm.resize(std::max(1, nt))); HIPCHECK(h, h->d_operm.resize(std::max<size_t>(1, N)));
  HIPCHECK(h, h->obs_cam.resize(std::max<size_t>(1, N))); HIPCHECK(h, h->obs_pt.resize(std::max<size_t>(1, N))); HIPCHECK(h, h->obs_z.resize(std::max<size_t>(1, N)));
  HIPCHECK(h, h->pt_off.resize((size_t)nt + 2)); HIPCHECK(h, h->cam_opt_pos.resize(std::max(1, nc))); HIPCHECK(h, h->pt_opt.resize((size_t)nt + 4));
  HIPCHECK(h, h->opt_cam.resize(opt_cam.size()));
  if (nbc > 0) HIPCHECK(h, h->cam_band_pos.resize(std::max(1, nc)));
  h->nbc = nbc;
  const int* dev_band_pos = nbc > 0 ? h->cam_band_pos.p : h->cam_opt_pos.p;
  const size_t staging = ((size_t)3 * nt + 8) * sizeof(int) + (size_t)nt + SF_COUNT * sizeof(int) + (size_t)nt * sizeof(int) + 64;
  HIPCHECK(h, pinned_staging(h, staging));
  arena_reset(h);
  su.up_pageable = false;
  int* hflags = static_cast<int*>(su.host);
  int* hoff = hflags + SF_COUNT;
  int* hplo = hoff + nt + 2;
  int* hphi = hplo + nt;
  int* hperm = hphi + nt;
  unsigned char* same = reinterpret_cast<unsigned char*>(hperm + nt);
  const unsigned long long* sorted_keys = nullptr;
  const bool host_front = sort_points && N <= kHostFrontMaxObs && h->opt.host_setup;
  if (host_front) {
    const int rc = host_front_end(h, nc, nt, N, nco, obs_cam, obs_pt, obs_z, cam_opt_pos, cam_band_pos, pt_opt, crank, opt_cam, rank_bits, hflags, hoff, hplo, hphi, hperm, same);
    if (rc != BA_OK) return rc;
  } else {
  if (nc) HIPCHECK(h, stage_h2d(h, h->cam_opt_pos.p, cam_opt_pos, (size_t)nc * sizeof(int)));
  if (nc && nbc > 0) HIPCHECK(h, stage_h2d(h, h->cam_band_pos.p, cam_band_pos, (size_t)nc * sizeof(int)));
  HIPCHECK(h, stage_h2d(h, h->opt_cam.p, opt_cam.data(), opt_cam.size() * sizeof(int)));
  if (N) {
    HIPCHECK(h, stage_h2d(h, su.rc.p, obs_cam, (size_t)N * sizeof(int)));
    HIPCHECK(h, stage_h2d(h, su.rp.p, obs_pt, (size_t)N * sizeof(int)));
    HIPCHECK(h, stage_h2d(h, su.rz.p, obs_z, (size_t)N * sizeof(double2)));
  }
  if (nc) HIPCHECK(h, stage_h2d(h, su.crank.p, crank.data(), (size_t)nc * sizeof(int)));
  if (nt) HIPCHECK(h, stage_h2d(h, su.rpo.p, pt_opt, (size_t)nt));
  HIPCHECK(h, hipMemsetAsync(su.cnt.p, 0, ((size_t)nt + 2) * sizeof(int), h->stream));
  HIPCHECK(h, hipMemsetAsync(su.Lint.p, 0, ((size_t)nt + 2) * sizeof(int), h->stream));
  hipLaunchKernelGGL(k_setup_init, dim3(1), dim3(256), 0, h->stream, su.flags.p);
  // ---- validate, count per track, and find out whether the observations already come ordered by (track, camera rank)
  if (N) hipLaunchKernelGGL(k_setup_keys, dim3(grid_for(N)), dim3(256), 0, h->stream, N, nc, nt, su.rc.p, su.rp.p, su.crank.p, rank_bits,
                            su.key.p, su.cnt.p, su.flags.p);
  HIPCHECK(h, hipMemcpyAsync(hflags, su.flags.p, SF_COUNT * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));        // (obs_cam / obs_pt / obs_z / pt_opt are caller memory: not read after this point)
  if (hflags[SF_BAD] != 0x7fffffff) {
    const int n = hflags[SF_BAD], c = obs_cam[n], k = obs_pt[n];
    if (c < 0 || c >= nc) return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: obs_cam[%d]=%d out of range", n, c);
    return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: obs_pt[%d]=%d out of range", n, k);
  }
  // ---- order by (track, rank): a stable radix sort of (key, index) pairs, only if needed
  const int* by_pt = nullptr;                        // position in the sorted order -> caller's observation index (nullptr: identity)
  sorted_keys = su.key.p;
  if (hflags[SF_UNSORTED]) {
    HIPCHECK(h, su.key2.resize((size_t)N)); HIPCHECK(h, su.vals.resize((size_t)N)); HIPCHECK(h, su.by_pt.resize((size_t)N));
    hipLaunchKernelGGL(k_iota, dim3(grid_for(N)), dim3(256), 0, h->stream, (int)N, su.vals.p);
    HIPCHECK(h, sort_pairs_u64(h, su.key.p, su.key2.p, su.vals.p, su.by_pt.p, (size_t)N, rank_bits + track_bits));
    hipLaunchKernelGGL(k_setup_dups, dim3(grid_for(N)), dim3(256), 0, h->stream, N, su.key2.p, su.flags.p);
    sorted_keys = su.key2.p;
    if (sort_points) {
      by_pt = su.by_pt.p;
    } else if (hflags[SF_UNSORTED_PT]) {
      // without the internal sort a track keeps its observations in the caller's order: grouped by track, nothing else
      hipLaunchKernelGGL(k_setup_track_keys_only, dim3(grid_for(N)), dim3(256), 0, h->stream, N, su.rp.p, su.key.p);
      HIPCHECK(h, sort_pairs_u64(h, su.key.p, su.key2.p, su.vals.p, su.by_pt.p, (size_t)N, track_bits));
      by_pt = su.by_pt.p;
      sorted_keys = nullptr;                         // (the duplicate's key is gone: the message names no track)
    }
  }
  // ---- CSR by caller track, the tracks' order, CSR in that order, the internal arrays
  HIPCHECK(h, exclusive_scan_i32(h, su.cnt.p, su.coff.p, (size_t)nt + 1));
  if (nt) {
    hipLaunchKernelGGL(k_iota, dim3(grid_for(nt)), dim3(256), 0, h->stream, nt, su.iota.p);
    if (sort_points && nt > 1) {
      hipLaunchKernelGGL(k_setup_track_keys, dim3(grid_for(nt)), dim3(256), 0, h->stream, nt, nco, su.coff.p, by_pt, su.rc.p, dev_band_pos,
                         su.crank.p, su.tkey.p);
      HIPCHECK(h, sort_pairs_u64(h, su.tkey.p, su.tkey2.p, su.iota.p, h->d_pperm.p, (size_t)nt, 32 + bits_for((unsigned long long)nco + 1)));
      hipLaunchKernelGGL(k_setup_order_check, dim3(grid_for(nt)), dim3(256), 0, h->stream, nt, su.tkey.p, su.tkey2.p, su.flags.p);
    }
    hipLaunchKernelGGL(k_setup_choose_order, dim3(grid_for(nt)), dim3(256), 0, h->stream, nt, (sort_points && nt > 1) ? 0 : 1, h->d_pperm.p,
                       su.cnt.p, su.Lint.p, su.flags.p);
  }
  HIPCHECK(h, exclusive_scan_i32(h, su.Lint.p, h->pt_off.p, (size_t)nt + 1));
  if (nt) {
    hipLaunchKernelGGL(k_setup_gather, dim3(grid_for(nt)), dim3(256), 0, h->stream, nt, h->d_pperm.p, su.coff.p, h->pt_off.p, by_pt, su.rc.p,
                       su.rz.p, su.rpo.p, h->obs_cam.p, h->obs_pt.p, h->obs_z.p, h->d_operm.p, h->pt_opt.p, su.flags.p);
    hipLaunchKernelGGL(k_setup_point_summary, dim3(grid_for(nt)), dim3(256), 0, h->stream, nt, h->pt_off.p, h->obs_cam.p, dev_band_pos,
                       su.plo.p, su.phi.p, su.same.p, su.flags.p);
  }
  HIPCHECK(h, hipGetLastError());
  // ---- the per-point summaries come back: everything below is O(points)
  HIPCHECK(h, hipMemcpyAsync(hflags, su.flags.p, SF_COUNT * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipMemcpyAsync(hoff, h->pt_off.p, ((size_t)nt + 1) * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (nt) {
    HIPCHECK(h, hipMemcpyAsync(hplo, su.plo.p, (size_t)nt * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipMemcpyAsync(hphi, su.phi.p, (size_t)nt * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipMemcpyAsync(hperm, h->d_pperm.p, (size_t)nt * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipMemcpyAsync(same, su.same.p, (size_t)nt, hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  arena_reset(h);                                    // (everything uploaded so far has arrived)
  }
  const int* flags = hflags;
  if (flags[SF_DUP] != 0x7fffffff) {                 // each (camera, track) pair at most once (bundle.py: a dict per track)
    unsigned long long key = 0;
    if (sorted_keys) {
      HIPCHECK(h, hipMemcpy(&key, sorted_keys + flags[SF_DUP], sizeof key, hipMemcpyDeviceToHost));
      return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: track %d has two observations in one camera", (int)(key >> rank_bits));
    }
    return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: a track has two observations in one camera");
  }
  h->h_off.assign(hoff, hoff + nt + 1);
  h->h_plo.assign(hplo, hplo + nt);
  h->h_phi.assign(hphi, hphi + nt);
  if (flags[SF_PERM]) h->pperm.assign(hperm, hperm + nt); else h->pperm.clear();
  h->operm_identity = flags[SF_OPERM] == 0;
  const int* off = h->h_off.data();
  const long long maxL = flags[SF_MAXL];
  long long nunits = 0;                              // work units of k_schur_pairs (built on first use: ensure_pair_units)
  for (int k = 0; k < nt; ++k) {
    const long long T = (off[(size_t)k + 1] - off[k] + kTile - 1) / kTile;
    nunits += T * (T + 1) / 2;
  }
  // block half-bandwidth of the reduced system: widest spread of optimised-camera positions within one track
  int hb = flags[SF_HB];
  hb = std::max(hb, std::min(h->min_hb, std::max(0, nco - 1)));   // sharded adjuster: every rank uses the widest band
  // LDS window of the older reduction kernels: wn band rows
  int wn = (int)(kSchurTileBytes / (((size_t)(hb + 1) * 36 + 6) * sizeof(double)));
  wn = std::min(wn, 64);
  if (wn < hb + 2 || nco == 0) wn = 0;                 // band too wide for an LDS tile: global atomics only
  // lanes per point: smallest power of two >= mean track length, in [1, 64]
  int glog = 0;
  const double meanL = nt > 0 ? (double)nobs / nt : 1.0;
  while ((1 << glog) < meanL && glog < 6) ++glog;

  h->nc = nc; h->nt = nt; h->nco = nco; h->hb = hb; h->nobs = nobs; h->glog = glog;
  std::memcpy(h->K, K, sizeof h->K);
  h->nunits = (int)std::min<long long>(nunits, INT32_MAX);
  h->nchunks = 0;
  h->pair_units_built = h->cam_units_built = false;
  h->schur_wn = wn;
  h->group_maxL = maxL;
  h->ncam_units = 0;
  h->plan_flags.assign(flags, flags + SF_COUNT);
  h->h_cam_opt_pos.assign(cam_band_pos, cam_band_pos + nc);
  h->h_same.assign(same, same + nt);
  h->plan_pending = true;      // (ba_set_problem builds them once the order of the cameras is settled)
  for (int i = 0; i < 2; ++i) {
    HIPCHECK(h, h->cams[i].resize(std::max<size_t>(1, (size_t)nc * 12)));
    HIPCHECK(h, h->X[i].resize(std::max<size_t>(1, (size_t)nt * 3)));
  }
  HIPCHECK(h, h->HCC.resize(std::max<size_t>(1, (size_t)nc * 36)));
  HIPCHECK(h, h->bC.resize(std::max<size_t>(1, (size_t)nc * 6)));
  HIPCHECK(h, h->HPP.resize(std::max<size_t>(1, (size_t)nt * 6)));
  HIPCHECK(h, h->bP.resize(std::max<size_t>(1, (size_t)nt * 3)));
  HIPCHECK(h, h->HPPinv.resize(std::max<size_t>(1, (size_t)nt * 6)));
  HIPCHECK(h, h->dC.resize(((size_t)nco + 16) * 6));     // padded: the cyclic-reduction solve writes whole super-blocks
  HIPCHECK(h, h->ysol.resize(std::max<size_t>(1, (size_t)nco * 6)));
  HIPCHECK(h, h->dinv.resize(std::max<size_t>(1, (size_t)nco * 6)));
  HIPCHECK(h, h->mask.resize(std::max<size_t>(1, (size_t)nco * 6)));
  HIPCHECK(h, h->dP.resize(std::max<size_t>(1, (size_t)nt * 3)));
  HIPCHECK(h, h->flags.resize(64));
  HIPCHECK(h, hipMemsetAsync(h->flags.p, 0, 64 * sizeof(int), h->stream));
  // a reduced system bound for another problem size is no longer valid
  h->S = nullptr; h->b = nullptr;
  h->have_problem = true;
  h->dist.on = false;                 // (a cut of the solve over the ranks belongs to the problem it was made for: ba_dist_enable)
  h->have_params[0] = h->have_params[1] = false;
  h->have_linearization = h->have_schur = h->have_backsub = h->have_solution = false;
  h->cur = 0;
  // host vectors go out of scope: what went through the pinned arena needs no wait (the next arena_reset waits if it must)
  if (su.up_pageable) HIPCHECK(h, hipStreamSynchronize(h->stream));
  else su.up_pending = true;
  return BA_OK;
}

// The internal layout of the optimised cameras: their ORDER (ba_order.hip) and, for scenes that are a sequence plus a few
// long-range tracks, a BORDER (ba_border.h).  The problem has been set up in the caller's order; when that order
// is not provably as narrow as an order can be (a track of L optimised cameras spreads over at least L - 1 positions), the
// distinct camera lists come back from the device and the candidates are ranked by what their solve would cost: the caller's
// order and the Cuthill-McKee order, each as it is and - when its band is wider than the narrow cyclic reduction takes - with
// the cameras that make it so moved to a border (plan_camera_layout, ba_order.hip).  If anything beats the caller's order, the
// problem is set up again with the cameras at their new positions.  Costs nothing for a scene that arrives in sequence order.
int choose_camera_order(ba_handle* h, int32_t nc, int32_t nt, int64_t nobs, const int32_t* obs_cam, const int32_t* obs_pt,
                        const double* obs_z, const double* K, const int32_t* cam_opt_pos, const uint8_t* pt_opt) {
  const int nco = h->nco, hb0 = h->plan_flags[SF_HB];
  h->caller_hb = hb0;
  if (!h->forced_pos.empty()) {
    // the caller's layout (ba_set_camera_layout: the ranks of a sharded adjuster share one)
    if ((int)h->forced_pos.size() != nco) return h->fail(BA_ERR_INVALID_ARG, "ba_set_problem: a layout for %d optimised cameras is imposed (ba_set_camera_layout), the problem has %d", (int)h->forced_pos.size(), nco);
    bool ident = true;
    for (int p = 0; p < nco; ++p) ident = ident && h->forced_pos[p] == p;
    if (ident) return BA_OK;
    std::vector<int32_t> cop((size_t)nc);
    for (int i = 0; i < nc; ++i) cop[i] = cam_opt_pos[i] >= 0 ? h->forced_pos[cam_opt_pos[i]] : -1;
    const int rc = set_problem_impl(h, nc, nt, nobs, obs_cam, obs_pt, obs_z, K, cop.data(), pt_opt);
    if (rc != BA_OK) return rc;
    h->cpos_in = h->forced_pos;
    h->cpos_out.assign((size_t)nco, 0);
    for (int p = 0; p < nco; ++p) h->cpos_out[h->forced_pos[p]] = p;
    h->caller_hb = hb0;
    return BA_OK;
  }
  if (h->opt.camera_order == CAMORDER_OFF || h->min_hb > 0 || h->comm || nco < 3 || nobs == 0) return BA_OK;      // (sharded: the ranks must agree on one layout)
  if (h->opt.camera_order != CAMORDER_ALWAYS && (hb0 <= std::max<long long>(1, h->group_maxL - 1) || resident_shape(h))) return BA_OK;
  // the distinct camera lists, as optimised positions in the caller's order
  std::vector<int> hobs((size_t)nobs);
  HIPCHECK(h, hipMemcpyAsync(hobs.data(), h->obs_cam.p, (size_t)nobs * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  std::vector<int> loff(1, 0), lpos;
  for (int i = 0; i < nt; ++i) {
    if (i > 0 && h->h_same[i]) continue;
    const size_t before = lpos.size();
    for (int n = h->h_off[i]; n < h->h_off[(size_t)i + 1]; ++n) {
      const int p = cam_opt_pos[hobs[n]];
      if (p >= 0) lpos.push_back(p);
    }
    if (lpos.size() - before >= 2) loff.push_back((int)lpos.size()); else lpos.resize(before);
  }
  // ... and how many points have each of them (a list that ONE point has and that ties far-apart cameras is a loop closure)
  std::vector<int> lmult;
  {
    int count = 0;
    bool open = false;
    for (int i = 0; i < nt; ++i) {
      if (!(i > 0 && h->h_same[i])) {
        if (open) lmult.push_back(count);
        int nopt = 0;
        for (int n = h->h_off[i]; n < h->h_off[(size_t)i + 1]; ++n) nopt += cam_opt_pos[hobs[n]] >= 0 ? 1 : 0;
        open = nopt >= 2;
        count = 0;
      }
      ++count;
    }
    if (open) lmult.push_back(count);
  }
  std::vector<int> newpos;
  int n1 = nco, hb_planned = hb0;
  if (!plan_camera_layout(nco, loff, lpos, lmult, h->opt.border, newpos, &n1, &hb_planned)) return BA_OK;      // the caller's order stays
  const int k_border = nco - n1;
  std::vector<int32_t> cop((size_t)nc), cband((size_t)nc);
  for (int i = 0; i < nc; ++i) {
    cop[i] = cam_opt_pos[i] >= 0 ? newpos[cam_opt_pos[i]] : -1;
    cband[i] = cop[i] >= n1 ? -1 : cop[i];
  }
  const int rc = set_problem_impl(h, nc, nt, nobs, obs_cam, obs_pt, obs_z, K, cop.data(), pt_opt, k_border > 0 ? cband.data() : nullptr, k_border);
  if (rc != BA_OK) return rc;
  h->cpos_in = newpos;
  h->cpos_out.assign((size_t)nco, 0);
  for (int p = 0; p < nco; ++p) h->cpos_out[newpos[p]] = p;
  h->caller_hb = hb0;
  return border_setup(h);
}

}  // namespace

extern "C" {

int ba_set_problem(ba_handle* h, int32_t nc, int32_t nt, int64_t nobs, const int32_t* obs_cam,
                   const int32_t* obs_pt, const double* obs_z, const double* K,
                   const int32_t* cam_opt_pos, const uint8_t* pt_opt) {
  if (!h) return BA_ERR_INVALID_ARG;
  h->cpos_in.clear(); h->cpos_out.clear();
  h->lin_reused = 0;
  h->refined = 0;
  h->pcg.built = false;
  h->pcg.pairs_built = false;
  h->pcg.band_clean = false;
  h->pcg.packed = false;
  h->pcg.prev_iterations = 0;
  // (option solve_trace: where the set-up's time goes, on stderr)
  const auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (h->opt.solve_trace) fprintf(stderr, "[ba_set_problem] %s: %.2f ms since the call\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  };
  int rc = set_problem_impl(h, nc, nt, nobs, obs_cam, obs_pt, obs_z, K, cam_opt_pos, pt_opt);
  if (rc != BA_OK) return rc;
  lap("internal order of tracks and observations (set_problem_impl)");
  rc = choose_camera_order(h, nc, nt, nobs, obs_cam, obs_pt, obs_z, K, cam_opt_pos, pt_opt);
  if (rc != BA_OK) { h->have_problem = false; return rc; }
  lap("layout of the optimised cameras (choose_camera_order)");
  // The work lists of the general kernels: at once - or, for a problem the resident loop takes whole (ba_resident.h), when a
  // general kernel first asks for them (ensure_plan): the sliding-window caller sets a problem per frame and never does
  if (!resident_shape(h)) {
    rc = ensure_plan(h);
    if (rc != BA_OK) { h->have_problem = false; return rc; }
    lap("work lists of the general kernels (ensure_plan)");
    // A scene the sparse path takes whole (ba_pcg.h: a wide band of mostly structural zeros, the solver left to the library or set to
    // pcg BEFORE this call, no group kernel worth its while): [S] is the list of the pattern's blocks and nothing else
    // (a sharded adjuster: only with the lists of ALL the scene's tracks - ba_set_pattern_lists - and every rank deciding alike: the caller sees to that)
    const bool whole = !h->comm && h->min_hb == 0 && h->forced_pos.empty();
    if (h->opt.packed_store && (whole || h->pcg.shared_lists) && h->nbc == 0 && sparse_layout(h) && h->pcg.pairs_built && pick_schur_kernel(h) == KERN_PAIRS) {
      h->pcg.packed = true;
      lap("list of the blocks of S and their observation pairs (packed store)");
    }
  }
  return BA_OK;
}


int ba_problem_info(ba_handle* h, int64_t* out, int32_t n) {
  if (h && h->have_problem) { const int rc = ensure_plan(h); if (rc != BA_OK) return rc; }
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_problem_info: call ba_set_problem first");
  REQUIRE(h, out && n >= 1, BA_ERR_INVALID_ARG, "ba_problem_info: bad argument");
  const int kern = pick_schur_kernel(h);
  const int64_t v[BA_INFO_COUNT] = {
      h->pperm.empty() ? 0 : 1, h->operm_identity ? 0 : 1, h->ngroups, (int64_t)(kern == KERN_MFMA3 ? h->nwgroups : h->nmgroups_total), h->point_groups ? 1 : 0,
      h->group_maxL, h->hb, kern_is_mfma(kern) ? 1 : 0, kern != KERN_PAIRS && kern != KERN_DENSE ? 1 : 0,
      kern == KERN_MFMA3 ? h->gm3.wn : h->schur_wn, h->nunits, kern, h->gm3.np_cap, h->gm3.Kbuf, h->cpos_in.empty() ? 0 : 1, h->caller_hb, h->nbc, h->lin_reused, h->refined, h->pcg.packed ? 1 : 0};
  for (int i = 0; i < n && i < BA_INFO_COUNT; ++i) out[i] = v[i];
  return BA_OK;
}

}  // extern "C"
