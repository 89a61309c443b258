// ba_schur.hip - ba_schur: damping, point-block inversion and the Schur reduction into the reduced camera system.
#include "ba_internal.h"

#include "ba_schur_kernels.h"

using namespace ba;

namespace ba {

int pick_schur_kernel(const ba_handle* h) {
  if (h->dense_mode && h->nt > 0 && h->nco > 0) return KERN_DENSE;
  if (h->sensor.kind == SENSOR_TABLE) return KERN_PAIRS;      // (a caller-defined sensor model: the kernels compiled with the table evaluation)
  const bool asc = h->groups_ascending && h->group_maxL >= 1;
  const bool m3 = (h->nm3chunks > 0 && h->nwgroups > 0) || h->nrgroups > 0 || h->nwide > 0;           // window groups: no identical camera lists needed
  const bool m12 = asc && h->nmchunks > 0 && h->schur_wn > 0 && h->group_maxL <= kGmMaxL;      // the L <= 10 kernels
  const bool vec = h->ngchunks > 0 && h->schur_wn > 0 && h->group_maxL <= kGroupMaxL;
  switch (h->opt.schur) {
    case SCHUR_PAIRS: return KERN_PAIRS;
    case SCHUR_GROUPS: return vec ? KERN_GROUPS : KERN_PAIRS;
    case SCHUR_MFMA2: return m12 ? KERN_MFMA2 : KERN_PAIRS;
    case SCHUR_MFMA: return m3 ? KERN_MFMA3 : KERN_PAIRS;
    default: break;
  }
  if (h->mgroups_worth && m12) return KERN_MFMA2;          // runs of identical camera lists, track length <= 10: the fixed-shape kernel (with the
                                                          // camera blocks folded in it is 6 % faster than the general one's <0, 4, 64, 5> instance)
  if (m3 && h->wgroups_worth) return KERN_MFMA3;
  if (h->mgroups_any && m12) return KERN_MFMA2;
  if (h->groups_worth && vec && h->group_rounds >= 1 && h->group_rounds <= 2) return KERN_GROUPS;
  return KERN_PAIRS;
}


}  // namespace ba

extern "C" {

int ba_schur(ba_handle* h, int which, double damping, double pinv_rcond) {
  if (h) { const int rc = ensure_plan(h); if (rc != BA_OK) return rc; }
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_schur: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_linearization && h->have_params[p], BA_ERR_STATE, "ba_schur: call ba_linearize first");
  h->schur_damping = damping;
  HIPCHECK(h, hipSetDevice(h->device));
  if (int rcj = border_join(h); rcj != BA_OK) return rcj;      // (border kernels of an earlier ba_schur may still be reading what this call writes: side stream, ba_border.hip)
  int rc = ensure_reduced(h);
  if (rc != BA_OK) return rc;
  const int kern = pick_schur_kernel(h);         // (ba_set_option "schur" forces one: tests)
  const bool dense = kern == KERN_DENSE;
  const bool use_mfma = kern_is_mfma(kern);
  const bool use_groups = kern == KERN_GROUPS;
  // point blocks and camera blocks: normally in HPP / bP (k_linearize) and HCC / bC (k_camera_blocks);
  // ba_lm_trial leaves both to the MFMA reduction, which linearises every observation anyway
  if (!h->point_blocks_valid) {
    rc = launch_point_blocks(h, h->lin_phys, nullptr);
    if (rc != BA_OK) return rc;
  }
  const bool hybrid = kern == KERN_MFMA3 && h->nrgroups > 0;       // long tracks: rectangular groups between their segments
  const bool fuse_cam = use_mfma && !h->cam_blocks_valid;
  if (!h->cam_blocks_valid && !fuse_cam) {
    rc = launch_camera_blocks(h, h->lin_phys, true);
    if (rc != BA_OK) return rc;
  }
  // the producer / consumer reductions and the dense one work from the factorised point inverses, which the merged
  // inversion + initialisation launch below writes (or has written, for the same damping)
  const bool want_fac = kern == KERN_MFMA2 || kern == KERN_MFMA3 || dense;
  const bool have_inv = h->inv_valid && h->inv_damping == damping && h->inv_rcond == pinv_rcond && (!want_fac || h->fac_valid);
  h->inv_valid = false;
  if (want_fac) HIPCHECK(h, h->fac.resize((size_t)9 * std::max(1, h->nt)));
  if (!have_inv) h->fac_valid = false;
  // (with a border, ba_border.h, the band and its right-hand side end at the band cameras: border_schur below clears and fills the rest)
  const int n1 = h->band_cams();
  const long long ninit = (long long)n1 * (h->hb + 1) * 36 + (long long)n1 * 6;
  REQUIRE(h, !h->pcg.packed || (kern == KERN_PAIRS && h->pcg.pairs_built), BA_ERR_STATE, "ba_schur: this problem's reduced system is stored as the list of its blocks (packed store): only the block-wise reduction applies (options schur / packed_store are read by ba_set_problem)");
  const bool sparse_init = sparse_layout(h) && (h->pcg.packed || h->pcg.band_clean);      // (a band: false the first time - that call clears the whole band)
  if (sparse_init) {
    if (have_inv) h->inv_valid = true;
    else if (h->nt > 0) {
      h->sing_epoch ^= 1;
      ScopedTimer tm(h, BA_K_POINT_INVERT);
      hipLaunchKernelGGL(k_point_invert, dim3(blocks_for(h->nt)), dim3(kBlock), 0, h->stream, h->nt, h->HPP.p,
                         damping, pinv_rcond, h->HPPinv.p, h->sing_counter(), h->flags.p + 40 + ((h->sing_epoch ^ 1) & 1), h->bP.p, want_fac ? h->fac.p : (double*)nullptr);
      h->inv_valid = true; h->inv_damping = damping; h->inv_rcond = pinv_rcond;
      h->fac_valid = want_fac;
    } else {
      HIPCHECK(h, hipMemsetAsync(h->flags.p + 40, 0, 2 * sizeof(int), h->stream));
    }
    ScopedTimer tm(h, BA_K_SCHUR_INIT);
    rc = launch_schur_init_sparse(h, damping, fuse_cam ? 0 : 1);
    if (rc != BA_OK) return rc;
  } else
  if (have_inv && h->trial_init_done && fuse_cam) {
    h->inv_valid = true;          // the trial's linearisation has done both (k_linearize_groups, launch_point_blocks): nothing to launch
  } else
  if (!have_inv && h->nt > 0 && h->nco > 0) {
    // point inverses and the initialisation of [S | b] are independent: one launch for both
    h->sing_epoch ^= 1;     // this call counts singular blocks in sing_counter(); the kernel clears the other one
    ScopedTimer tm(h, BA_K_POINT_INVERT);
    const unsigned nbi = blocks_for(h->nt);
    hipLaunchKernelGGL(k_point_invert_schur_init, dim3(nbi + blocks_for(ninit)), dim3(kBlock), 0, h->stream, (int)nbi, h->nt,
                       h->HPP.p, damping, pinv_rcond, h->HPPinv.p, h->sing_counter(),
                       h->flags.p + 40 + ((h->sing_epoch ^ 1) & 1), n1, h->hb + 1, h->opt_cam.p, h->HCC.p, h->bC.p, h->S,
                       h->b, fuse_cam ? 0 : 1, h->bP.p, want_fac ? h->fac.p : (double*)nullptr);
    h->inv_valid = true; h->inv_damping = damping; h->inv_rcond = pinv_rcond;
    h->fac_valid = want_fac;
  } else {
    if (have_inv) {
      h->inv_valid = true;        // already inverted for this (damping, rcond)
    } else if (h->nt > 0) {
      h->sing_epoch ^= 1;
      ScopedTimer tm(h, BA_K_POINT_INVERT);
      hipLaunchKernelGGL(k_point_invert, dim3(blocks_for(h->nt)), dim3(kBlock), 0, h->stream, h->nt, h->HPP.p,
                         damping, pinv_rcond, h->HPPinv.p, h->sing_counter(), h->flags.p + 40 + ((h->sing_epoch ^ 1) & 1));
      h->inv_valid = true; h->inv_damping = damping; h->inv_rcond = pinv_rcond;
    } else {
      HIPCHECK(h, hipMemsetAsync(h->flags.p + 40, 0, 2 * sizeof(int), h->stream));
    }
    if (h->nco > 0) {
      ScopedTimer tm(h, BA_K_SCHUR_INIT);       // clears the band and writes the damped diagonal + b in one pass
      hipLaunchKernelGGL(k_schur_init, dim3(blocks_for(ninit)), dim3(kBlock), 0, h->stream, n1, h->hb + 1, h->opt_cam.p,
                         h->HCC.p, h->bC.p, damping, h->S, h->b, fuse_cam ? 0 : 1);
    }
  }
  h->trial_init_done = false;
  if (!sparse_init && h->nco > 0) h->pcg.band_clean = !dense;      // (the whole band has just been initialised; the dense reduction writes all of it)
  if (dense) {
    // dense visibility: the reduction is one symmetric matrix product over all points (the kernels below
    // would do 36 global atomics per (pair, point): 258 M of them at 100 cameras x 1000 tracks)
    const int M = 6 * h->nco, R = 3 * h->nt;
    const int T = (M + kSyrkTile - 1) / kSyrkTile, pairs = T * (T + 1) / 2;
    int nsplit = std::max(1, std::min(16, (768 + pairs - 1) / pairs));
    const int chunk = ((R + nsplit - 1) / nsplit + kSyrkKc - 1) / kSyrkKc * kSyrkKc;
    nsplit = (R + chunk - 1) / chunk;
    HIPCHECK(h, h->dUd.resize((size_t)R * M)); HIPCHECK(h, h->dDd.resize(R)); HIPCHECK(h, h->dyd.resize(R));
    HIPCHECK(h, h->dpart.resize((size_t)nsplit * M * M));
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS, 5);
    HIPCHECK(h, hipMemsetAsync(h->dUd.p, 0, (size_t)R * M * sizeof(double), h->stream));
    const long long n = std::max<long long>(h->nobs, (long long)h->nt);
    hipLaunchKernelGGL(k_dense_stage, dim3(blocks_for(n)), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p, h->X[p].p,
                       h->fac.p, h->bP.p, M, h->dUd.p, h->dDd.p, h->dyd.p);
    hipLaunchKernelGGL(k_dense_syrk, dim3(T, T, nsplit), dim3(1024), 0, h->stream, M, R, chunk, h->dUd.p, h->dDd.p, h->dpart.p);
    hipLaunchKernelGGL(k_dense_apply, dim3(blocks_for(reduced_doubles(h))), dim3(kBlock), 0, h->stream, h->nco, h->hb + 1, M, nsplit,
                       h->dpart.p, h->S);
    hipLaunchKernelGGL(k_dense_rhs, dim3((M + kBlock - 1) / kBlock, (R + kDenseRhsRows - 1) / kDenseRhsRows), dim3(kBlock), 0,
                       h->stream, M, R, h->dUd.p, h->dyd.p, h->b);
  } else if (kern == KERN_MFMA3) {
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS, (h->nm3chunks ? mfma3_launches(h->gm3.nts) : 0) + (hybrid ? 1 : 0) + wide_launches(h));
    rc = launch_mfma3_all(h, p, damping, fuse_cam);
    if (rc != BA_OK) return rc;
    if (h->nwide > 0) {
      rc = launch_wide_all(h, p, damping, fuse_cam);
      if (rc != BA_OK) return rc;
    }
    if (hybrid) {
      rc = launch_rect(h, p, damping, fuse_cam);
      if (rc != BA_OK) return rc;
    }
  } else if (kern == KERN_MFMA2) {
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS);
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_groups_mfma2));
    hipLaunchKernelGGL(k_schur_groups_mfma2, dim3(h->nmchunks), dim3(kGm2Block), schur_mfma2_lds_bytes(h->schur_wn, h->hb + 1), h->stream,
                       dev_problem_band(h), h->cams[p].p, h->X[p].p, h->mgroups.p, h->mchunks.p, h->schur_wn, h->fac.p, h->S, h->b, damping,
                       fuse_cam ? 1 : 0);
  } else if (use_groups) {
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS);
    const int NW = kGroupBlock / kWave;
    const size_t lds = (size_t)NW * 64 * 24 * sizeof(double) + (size_t)NW * 16 * sizeof(int) +
                       (size_t)h->schur_wn * ((size_t)(h->hb + 1) * 36 + 6) * sizeof(double);
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_groups<1>));
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_groups<2>));
    const int maxpairs_rounds = h->group_rounds >= 1 ? h->group_rounds : 2;
    if (maxpairs_rounds == 1)
      hipLaunchKernelGGL(k_schur_groups<1>, dim3(h->ngchunks), dim3(kGroupBlock), lds, h->stream, dev_problem_band(h), h->cams[p].p,
                         h->X[p].p, h->groups.p, h->gchunks.p, h->schur_wn, h->HPPinv.p, h->bP.p, h->S, h->b);
    else
      hipLaunchKernelGGL(k_schur_groups<2>, dim3(h->ngchunks), dim3(kGroupBlock), lds, h->stream, dev_problem_band(h), h->cams[p].p,
                         h->X[p].p, h->groups.p, h->gchunks.p, h->schur_wn, h->HPPinv.p, h->bP.p, h->S, h->b);
  } else if (kern == KERN_PAIRS && !fuse_cam && sparse_layout(h) && h->pcg.pairs_built) {
    // a scene without a band: every block of the pattern sums its own list of observation pairs (k_schur_blocks, ba_pcg.h) - no atomics
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS);
    rc = launch_schur_blocks(h, p);
    if (rc != BA_OK) return rc;
  } else if (h->nunits > 0) {
    rc = ensure_pair_units(h);                          // (the pair kernel's work list is built on first use)
    if (rc != BA_OK) return rc;
    ScopedTimer tm(h, BA_K_SCHUR_PAIRS);
    const int NW = kSchurBlock / kWave;
    const size_t lds = (size_t)NW * kTile * 18 * 2 * sizeof(double) + (size_t)NW * kTile * 2 * sizeof(int) +
                       (size_t)h->schur_wn * ((size_t)(h->hb + 1) * 36 + 6) * sizeof(double);
    HIPCHECK(h, ensure_lds_attr(h, (const void*)k_schur_pairs));
    hipLaunchKernelGGL(k_schur_pairs, dim3(h->nchunks), dim3(kSchurBlock), lds, h->stream, dev_problem_band(h), h->cams[p].p,
                       h->X[p].p, h->units.p, h->chunks.p, h->schur_wn, h->HPPinv.p, h->bP.p, h->S, h->b);
  }
  HIPCHECK(h, hipGetLastError());
  if (h->nbc > 0) {                                     // every block that involves a border camera (ba_border.h)
    rc = border_schur(h, p, damping);
    if (rc != BA_OK) return rc;
  }
#ifdef BA_BCR_PROFILE
  if (h->opt.solve_trace && kern == KERN_MFMA2 && !h->defer) {
    // when the workgroups of the reduction started and ended, relative to the first start (us)
    const int n = std::min(h->nmchunks, kSchurTraceMax);
    std::vector<long long> tr((size_t)2 * n);
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    HIPCHECK(h, hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_schur_trace), tr.size() * sizeof(long long)));
    long long t0 = LLONG_MAX;
    for (int w = 0; w < n; ++w) t0 = std::min(t0, tr[2 * w]);
    std::vector<double> st(n), en(n);
    for (int w = 0; w < n; ++w) { st[w] = (tr[2 * w] - t0) * 0.01; en[w] = (tr[2 * w + 1] - t0) * 0.01; }
    std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
    auto q = [&](const std::vector<double>& v, double f) { return v[(size_t)std::min<double>(v.size() - 1, f * v.size())]; };
    fprintf(stderr, "[k_schur_groups_mfma2: %d workgroups] start: last %.1f us | end: first %.1f, 10 %% %.1f, median %.1f, 90 %% %.1f, last %.1f us after the first start\n",
            n, st.back(), en.front(), q(en, .1), q(en, .5), q(en, .9), en.back());
  }
#endif
  h->have_schur = true;
  h->have_backsub = h->have_solution = false;
  if (pinv_rcond < 0.0 && !h->defer) {   // plain-inverse mode must report singular blocks (numpy.linalg.inv raises)
    int nsing = 0;
    HIPCHECK(h, hipMemcpyAsync(&nsing, h->sing_counter(), sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    if (nsing > 0) return h->fail(BA_ERR_SINGULAR, "ba_schur: %d singular 3x3 point block(s) in plain-inverse mode", nsing);
  }
  return BA_OK;
}

int ba_reduced_layout(ba_handle* h, int32_t* nco, int32_t* half_bandwidth, int64_t* S_doubles) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_reduced_layout: call ba_set_problem first");
  if (nco) *nco = h->nco;
  if (half_bandwidth) *half_bandwidth = h->hb;
  if (S_doubles) *S_doubles = (int64_t)reduced_doubles(h);
  return BA_OK;
}

int ba_get_reduced(ba_handle* h, double* S, double* b) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_schur, BA_ERR_STATE, "ba_get_reduced: call ba_schur first");
  HIPCHECK(h, hipSetDevice(h->device));
  const int nco = h->nco, hb1 = h->hb + 1;
  if (int rcj = border_join(h); rcj != BA_OK) return rcj;      // (the border part of b comes from the side stream)
  std::vector<double> band(S ? reduced_doubles(h) : 0);
  if (S && nco) HIPCHECK(h, hipMemcpyAsync(band.data(), h->S, band.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (b && nco) HIPCHECK(h, hipMemcpyAsync(b, h->b, (size_t)nco * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  if (S && nco && h->pcg.packed) {   // the list of the pattern's upper blocks -> the reference's dense (nco, nco, 6, 6)
    std::memset(S, 0, (size_t)nco * nco * 36 * sizeof(double));
    const int* out = h->cpos_out.empty() ? nullptr : h->cpos_out.data();
    for (size_t u = 0; u < h->pcg.h_ublk.size(); ++u) {
      const int i = (int)(h->pcg.h_ublk[u] / hb1), d = (int)(h->pcg.h_ublk[u] % hb1);
      const double* src = &band[u * 36];
      const int r = out ? out[i] : i, q = out ? out[i + d] : i + d;
      std::memcpy(S + ((size_t)r * nco + q) * 36, src, 36 * sizeof(double));
      if (d > 0) {
        double* lo = S + ((size_t)q * nco + r) * 36;
        for (int a = 0; a < 6; ++a)
          for (int c = 0; c < 6; ++c) lo[c * 6 + a] = src[a * 6 + c];
      }
    }
    if (b) cam_rows_out(h, b, 6);
    return BA_OK;
  }
  if (S && nco) {   // expand the block band to the reference's dense (nco,nco,6,6), mirroring the upper triangle
    std::memset(S, 0, (size_t)nco * nco * 36 * sizeof(double));
    const int* out = h->cpos_out.empty() ? nullptr : h->cpos_out.data();      // internal position -> the caller's
    for (int i = 0; i < nco; ++i)
      for (int d = 0; d < hb1 && i + d < nco; ++d) {
        const double* src = &band[((size_t)i * hb1 + d) * 36];
        const int r = out ? out[i] : i, q = out ? out[i + d] : i + d;
        double* up = S + ((size_t)r * nco + q) * 36;
        std::memcpy(up, src, 36 * sizeof(double));
        if (d > 0) {
          double* lo = S + ((size_t)q * nco + r) * 36;
          for (int a = 0; a < 6; ++a)
            for (int c = 0; c < 6; ++c) lo[c * 6 + a] = src[a * 6 + c];
        }
      }
  }
  if (S && nco && h->nbc > 0) {   // ... and the blocks of the border cameras (ba_border.h): C [6 n1][ld], D [ld][ld]
    std::vector<double> C, D;
    const int rcb = border_get_dense(h, C, D);
    if (rcb != BA_OK) return rcb;
    const int* out = h->cpos_out.empty() ? nullptr : h->cpos_out.data();
    const int n1 = h->band_cams(), ld = h->bord_ld;
    auto put = [&](int i, int j, int u, int v, double val) {      // entry (u, v) of block (internal i, internal j)
      const int r = out ? out[i] : i, q = out ? out[j] : j;
      S[((size_t)r * nco + q) * 36 + u * 6 + v] = val;
    };
    for (int jb = 0; jb < h->nbc; ++jb) {
      for (int i = 0; i < n1; ++i)
        for (int u = 0; u < 6; ++u)
          for (int v = 0; v < 6; ++v) {
            const double val = C[(size_t)(6 * i + u) * ld + 6 * jb + v];
            put(i, n1 + jb, u, v, val);
            put(n1 + jb, i, v, u, val);
          }
      for (int ib = 0; ib < h->nbc; ++ib)
        for (int u = 0; u < 6; ++u)
          for (int v = 0; v < 6; ++v) put(n1 + ib, n1 + jb, u, v, D[(size_t)(6 * ib + u) * ld + 6 * jb + v]);
    }
  }
  cam_rows_out(h, b, 6);
  return BA_OK;
}

int ba_get_point_inverses(ba_handle* h, double* out) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_schur && out, BA_ERR_STATE, "ba_get_point_inverses: call ba_schur first");
  HIPCHECK(h, hipSetDevice(h->device));
  std::vector<double> s6((size_t)h->nt * 6);
  if (h->nt) HIPCHECK(h, hipMemcpyAsync(s6.data(), h->HPPinv.p, s6.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  for (int k = 0; k < h->nt; ++k) {
    const double* s = &s6[(size_t)k * 6];
    double* d = out + (size_t)(h->pperm.empty() ? k : h->pperm[k]) * 9;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[1]; d[4] = s[3]; d[5] = s[4]; d[6] = s[2]; d[7] = s[4]; d[8] = s[5];
  }
  return BA_OK;
}

int ba_reduced_device_ptrs(ba_handle* h, void** S_blocks, void** b) {
  if (!h) return BA_ERR_INVALID_ARG;
  if (int rcj = border_join(h); rcj != BA_OK) return rcj;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_reduced_device_ptrs: call ba_set_problem first");
  HIPCHECK(h, hipSetDevice(h->device));
  int rc = ensure_reduced(h);
  if (rc != BA_OK) return rc;
  if (S_blocks) *S_blocks = h->S;
  if (b) *b = h->b;
  return BA_OK;
}

int ba_bind_reduced_buffers(ba_handle* h, void* S_blocks_dev, void* b_dev) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_bind_reduced_buffers: call ba_set_problem first");
  h->S = (double*)S_blocks_dev;
  h->b = (double*)b_dev;
  h->have_schur = false;
  h->pcg.band_clean = false;
  return BA_OK;
}

int ba_set_dense_visibility(ba_handle* h, int32_t on) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, !(on && h->pcg.packed), BA_ERR_STATE, "ba_set_dense_visibility: this problem's reduced system is stored as the list of its blocks (packed store)");
  h->dense_mode = on != 0;
  h->inv_valid = false;
  if (!on) { h->dUd.release(); h->dDd.release(); h->dyd.release(); h->dpart.release(); }
  return BA_OK;
}


}  // extern "C"
