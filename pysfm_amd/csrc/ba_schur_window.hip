// ba_schur_window.hip - launches of the window-group reductions (k_schur_groups_mfma3 per range of tile columns, k_schur_wide_mfma per tile count, k_schur_rect_mfma).
#include "ba_internal.h"

#include "ba_schur_window_kernels.h"

using namespace ba;

namespace ba {

// k_schur_groups_mfma3 over the tile columns [TJ0, TJ1) of every group's window; one set of groups (chunks) with its parameters
struct M3Launch { Gm3Params G; const SchurChunk* chunks; int nchunks; bool uniform_ks; };
template <int TJ0, int TJ1, int LDC = 0, int KSC = 0>
int launch_mfma3(ba_handle* h, const M3Launch& L, int p, double damping, bool fuse_cam, bool first) {
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_groups_mfma3<TJ0, TJ1, LDC, KSC>));
  Gm3Params G = L.G;
  G.do_rhs = first ? 1 : 0;
  hipLaunchKernelGGL((k_schur_groups_mfma3<TJ0, TJ1, LDC, KSC>), dim3(L.nchunks), dim3(kGm2Block),
                     schur_mfma3_lds_bytes(G.Kbuf, G.Ld, G.wn, G.wb1), h->stream, dev_problem_band(h), h->cams[p].p, h->X[p].p,
                     h->wgroups.p, h->wtab.p, h->opt_cam.p, L.chunks, G, h->fac.p, h->S, h->b, damping, fuse_cam ? 1 : 0);
  return BA_OK;
}

int launch_mfma3_set(ba_handle* h, const M3Launch& L, int p, double damping, bool fuse_cam) {
  if (L.nchunks <= 0) return BA_OK;
  const int nts = L.G.nts;          // tiles per side of the widest window; at most 15 accumulator tiles per launch
  if (nts < 1 || nts > 15) return h->fail(BA_ERR_STATE, "k_schur_groups_mfma3: %d tiles per side", nts);
  // windows of at most 10 cameras with 6 points per batch everywhere (the north-star scenes): row length and k-steps fixed
  if (nts == 4 && L.G.np_cap == kGmPts && L.G.Kbuf == kGmK && L.uniform_ks) return launch_mfma3<0, 4, 64, 5>(h, L, p, damping, fuse_cam, true);
  // Windows of six tiles a side and more (round 5; option six_tile_launch).  A launch is bound by its producers, who linearise every
  // observation again - so the fewer launches the better, spilled registers and all: tile columns 0 .. 5 in one launch (21 accumulator
  // tiles, 104 spilled registers) are 109 us where <0, 4> + <4, 6> were 140 (tracks of 14 cameras); columns 0 .. 6 (28 tiles, 378
  // spilled) 204 us where <0, 6> + <6, 7> were 226 (17 cameras).  Eight and nine columns in one launch are where it ends (36 / 45
  // tiles: 3.7 / 1.4 ms).  Measured splits:   7 tiles a side: <0,7>   8: <0,6> + <6,8> (278 us; <0,7> + <7,8>: 287)
  //                                             9: <0,7> + <7,9> (367 us; <0,6> + <6,8> + <8,9>: 402)
  const bool six = nts >= 6 && h->opt.six_tile_launch;
  const bool seven = six && (nts == 7 || nts >= 9);          // the first launch takes seven columns
  int rc = seven ? launch_mfma3<0, 7>(h, L, p, damping, fuse_cam, true)
         : six ? launch_mfma3<0, 6>(h, L, p, damping, fuse_cam, true)
         // (windows of 11 .. 13 cameras: the row length fixed as well - addresses of the staged rows become immediates)
         : nts == 5 ? (L.G.Ld == 80 ? launch_mfma3<0, 5, 80, 0>(h, L, p, damping, fuse_cam, true) : launch_mfma3<0, 5>(h, L, p, damping, fuse_cam, true))
                    : launch_mfma3<0, 4>(h, L, p, damping, fuse_cam, true);
  if (seven) {
    if (rc == BA_OK && nts >= 9) rc = launch_mfma3<7, 9>(h, L, p, damping, fuse_cam, false);
  } else {
    if (rc == BA_OK && nts >= 6 && !six) rc = launch_mfma3<4, 6>(h, L, p, damping, fuse_cam, false);
    if (rc == BA_OK && nts == 7) rc = launch_mfma3<6, 7>(h, L, p, damping, fuse_cam, false);
    if (rc == BA_OK && nts >= 8) rc = launch_mfma3<6, 8>(h, L, p, damping, fuse_cam, false);
    if (rc == BA_OK && nts >= 9) rc = launch_mfma3<8, 9>(h, L, p, damping, fuse_cam, false);
  }
  // windows of 25 .. 40 cameras (tracks that long: video): one launch per further tile column (tj + 1 <= 15 tiles each)
  if (rc == BA_OK && nts >= 10) rc = launch_mfma3<9, 10>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 11) rc = launch_mfma3<10, 11>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 12) rc = launch_mfma3<11, 12>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 13) rc = launch_mfma3<12, 13>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 14) rc = launch_mfma3<13, 14>(h, L, p, damping, fuse_cam, false);
  if (rc == BA_OK && nts >= 15) rc = launch_mfma3<14, 15>(h, L, p, damping, fuse_cam, false);
  return rc;
}
// k_schur_wide_mfma<NT> over the window groups of NT tiles per side (25 .. 40 cameras)
template <int NT>
int launch_wide(ba_handle* h, int p, double damping, bool fuse_cam) {
  const int g0 = h->wide_begin[NT - kGwMinTiles], n = h->wide_begin[NT - kGwMinTiles + 1] - g0;
  if (n <= 0) return BA_OK;
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_wide_mfma<NT>));
  hipLaunchKernelGGL(k_schur_wide_mfma<NT>, dim3(n), dim3(kGwBlock), schur_wide_lds_bytes(), h->stream, dev_problem_band(h), h->cams[p].p,
                     h->X[p].p, h->wgroups.p, h->wide_list.p + g0, h->wtab.p, h->opt_cam.p, h->fac.p, h->S, h->b, damping, fuse_cam ? 1 : 0);
  return BA_OK;
}
int launch_wide_all(ba_handle* h, int p, double damping, bool fuse_cam) {
  int rc = launch_wide<11>(h, p, damping, fuse_cam);
  if (rc == BA_OK) rc = launch_wide<12>(h, p, damping, fuse_cam);
  if (rc == BA_OK) rc = launch_wide<13>(h, p, damping, fuse_cam);
  if (rc == BA_OK) rc = launch_wide<14>(h, p, damping, fuse_cam);
  if (rc == BA_OK) rc = launch_wide<15>(h, p, damping, fuse_cam);
  return rc;
}
int wide_launches(const ba_handle* h) {
  int n = 0;
  for (int t = 0; t <= kGwMaxTiles - kGwMinTiles; ++t) n += h->wide_begin[t + 1] > h->wide_begin[t] ? 1 : 0;
  return n;
}

int launch_mfma3_all(ba_handle* h, int p, double damping, bool fuse_cam) {
  return launch_mfma3_set(h, M3Launch{h->gm3, h->m3chunks.p, h->nm3chunks, h->gm3_uniform_ks}, p, damping, fuse_cam);
}
int mfma3_launches(int nts) { return nts <= 5 ? 1 : nts == 6 ? 2 : nts <= 8 ? 3 : nts - 5; }      // (one less from six tiles a side on with option six_tile_launch)

// k_schur_rect_mfma: the products between the segments of tracks that span more than 40 cameras
int launch_rect(ba_handle* h, int p, double damping, bool fuse_cam) {
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_rect_mfma));
  hipLaunchKernelGGL(k_schur_rect_mfma, dim3(h->nrgroups), dim3(kRectBlock), schur_rect_lds_bytes(), h->stream, dev_problem_band(h), h->cams[p].p,
                     h->X[p].p, h->rgroups.p, h->nrgroups, h->rtab.p, h->opt_cam.p, h->fac.p, h->S, h->b, damping, fuse_cam ? 1 : 0);
  return BA_OK;
}

}  // namespace ba
