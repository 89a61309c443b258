// ba_points.hip - the per-observation passes: cost, evaluation, linearisation (point and camera blocks), back-substitution, parameter update, triangulation.
#include "ba_internal.h"

#include "ba_obs_kernels.h"

using namespace ba;

namespace ba {

bool trial_init_fusable(ba_handle* h) {
  return h->opt.fuse_invert && h->nt > 0 && h->nco > 0 && h->point_groups && !h->opt.point_kernels_v1 && h->sensor.kind != SENSOR_TABLE &&
         !h->pcg.packed && !sparse_layout(h) && ensure_reduced(h) == BA_OK;
}

// fused (ba_lm_trial_begin): the group lineariser also inverts the point blocks for `damping` and clears [S | b] - what the
// ba_schur call that follows would launch k_point_invert_schur_init for (ba_schur.hip reads h->trial_init_done)
int launch_point_blocks(ba_handle* h, int p, double* Wd, bool fused, double damping, double rcond) {
  h->trial_init_done = false;
  if (h->nt > 0) {
    ScopedTimer tm(h, BA_K_LINEARIZE);
    const long long threads = (long long)h->nt << h->glog;
    if (!Wd && h->point_groups && !h->opt.point_kernels_v1 && h->sensor.kind != SENSOR_TABLE) {
      const int per_block = kBlock / kWave;
      const int gblocks = (h->ngroups + per_block - 1) / per_block;
      FusedInvert inv{};
      unsigned nblocks = gblocks;
      if (fused) {
        HIPCHECK(h, h->fac.resize((size_t)9 * std::max(1, h->nt)));
        const int n1 = h->band_cams();
        const long long ninit = (long long)n1 * (h->hb + 1) * 36 + (long long)n1 * 6;
        h->sing_epoch ^= 1;        // (as ba_schur does: this inversion counts singular blocks in sing_counter(), the kernel clears the other counter)
        inv = FusedInvert{damping, rcond, h->HPPinv.p, h->fac.p, h->sing_counter(), h->flags.p + 40 + ((h->sing_epoch ^ 1) & 1), gblocks, n1, h->hb + 1, h->S, h->b};
        nblocks += blocks_for(ninit);
      }
      if (fused)
        hipLaunchKernelGGL(k_linearize_groups_trial, dim3(nblocks), dim3(kBlock), 0, h->stream,
                           dev_problem(h), h->cams[p].p, h->X[p].p, h->groups.p, h->ngroups, h->HCC.p, h->bC.p, h->HPP.p, h->bP.p, inv);
      else
        hipLaunchKernelGGL(k_linearize_groups, dim3(nblocks), dim3(kBlock), 0, h->stream,
                           dev_problem(h), h->cams[p].p, h->X[p].p, h->groups.p, h->ngroups, h->HCC.p, h->bC.p, h->HPP.p, h->bP.p);
      if (fused) {
        h->inv_valid = true; h->inv_damping = damping; h->inv_rcond = rcond;
        h->fac_valid = true;
        h->trial_init_done = true;
      }
    } else {
      hipLaunchKernelGGL(k_linearize, dim3(blocks_for(threads)), dim3(kBlock), 0, h->stream, dev_problem(h),
                         h->cams[p].p, h->X[p].p, h->glog, h->HCC.p, h->bC.p, h->HPP.p, h->bP.p, Wd,
                         0.0, 0.0, (double*)nullptr, (int*)nullptr, (int*)nullptr);
    }
  }
  h->point_blocks_valid = true;
  h->cam_blocks_valid = false;          // k_linearize cleared HCC / bC
  return BA_OK;
}

int launch_camera_blocks(ba_handle* h, int p, bool clear) {
  if (clear) {
    HIPCHECK(h, hipMemsetAsync(h->HCC.p, 0, (size_t)h->nc * 36 * sizeof(double), h->stream));
    HIPCHECK(h, hipMemsetAsync(h->bC.p, 0, (size_t)h->nc * 6 * sizeof(double), h->stream));
  }
  if (int rc = ensure_cam_units(h); rc != BA_OK) return rc;      // (the camera-ordered observation lists are built on first use)
  if (h->ncam_units > 0) {
    ScopedTimer tm(h, BA_K_CAMERA_BLOCKS);
    const int per_block = kBlock / kWave;
    hipLaunchKernelGGL(k_camera_blocks, dim3((h->ncam_units + per_block - 1) / per_block), dim3(kBlock), 0, h->stream,
                       dev_problem(h), h->cams[p].p, h->X[p].p, h->cam_perm.p, h->cam_units.p, h->ncam_units, h->HCC.p,
                       h->bC.p);
  }
  h->cam_blocks_valid = true;
  return BA_OK;
}

// ba_linearize; with fuse (ba_lm_trial + a matrix-core reduction) the camera blocks are left to the reduction kernel
int linearize_impl(ba_handle* h, int which, int store_W, bool fuse, double damping, double rcond) {
  { const int rc = ensure_plan(h); if (rc != BA_OK) return rc; }
  const int p = h->phys(which);
  HIPCHECK(h, hipSetDevice(h->device));
  if (int rcj = border_join(h); rcj != BA_OK) return rcj;      // (border kernels of an earlier ba_schur may still be reading what this call writes: side stream, ba_border.hip)
  double* Wd = nullptr;
  if (store_W) {
    HIPCHECK(h, h->W.resize(std::max<size_t>(1, (size_t)h->nobs * 18)));
    Wd = h->W.p;
  }
  fuse = fuse && h->nt > 0 && !store_W;
  h->inv_valid = false;
  h->fac_valid = false;
  h->cam_blocks_valid = false;
  h->point_blocks_valid = false;
  // fuse (ba_lm_trial with a matrix-core reduction): the reduction kernel linearises every observation anyway and
  // adds the camera blocks on the way, so k_camera_blocks is skipped (ba_schur / ba_get_blocks run it lazily
  // if another path asks for HCC / bC)
  int rc = launch_point_blocks(h, p, Wd, fuse && trial_init_fusable(h), damping, rcond);
  if (rc == BA_OK && !fuse) rc = launch_camera_blocks(h, p, h->nt == 0);      // k_linearize cleared HCC / bC otherwise
  if (rc != BA_OK) return rc;
  HIPCHECK(h, hipGetLastError());
  h->have_linearization = true;
  h->lin_phys = p;
  h->have_schur = h->have_backsub = false;
  return BA_OK;
}

}  // namespace ba

extern "C" {

int ba_cost(ba_handle* h, int which, double* cost_out) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, cost_out, BA_ERR_INVALID_ARG, "ba_cost: cost_out is NULL");
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_cost: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_problem && h->have_params[p], BA_ERR_STATE, "ba_cost: set problem and parameters first");
  HIPCHECK(h, hipSetDevice(h->device));
  const int nb = (int)std::max<long long>(1, std::min<long long>(kCostBlocks, blocks_for(h->nobs)));
  {
    ScopedTimer tm(h, BA_K_COST);
    hipLaunchKernelGGL(k_cost, dim3(nb), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p, h->X[p].p,
                       (const int*)h->sing_counter(), (const int*)(h->flags.p + 1), h->host_result, h->trial_result_dev);
  }
  h->cost_blocks = nb;
  HIPCHECK(h, hipGetLastError());
  if (h->defer) return BA_OK;
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  *cost_out = h->host_cost();
  return BA_OK;
}

int ba_eval_observations(ba_handle* h, int which, double* e, double* r, double* Jc, double* Jp) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_eval_observations: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_problem && h->have_params[p], BA_ERR_STATE, "ba_eval_observations: set problem and parameters first");
  if (h->nobs == 0) return BA_OK;
  HIPCHECK(h, hipSetDevice(h->device));
  const size_t N = (size_t)h->nobs;
  const size_t need = (e ? 2 * N : 0) + (r ? 2 * N : 0) + (Jc ? 12 * N : 0) + (Jp ? 6 * N : 0);
  if (!need) return BA_OK;
  HIPCHECK(h, h->scratch.resize(need));
  double* d = h->scratch.p;
  double* de = nullptr; double* dr = nullptr; double* dJc = nullptr; double* dJp = nullptr;
  if (e) { de = d; d += 2 * N; }
  if (r) { dr = d; d += 2 * N; }
  if (Jc) { dJc = d; d += 12 * N; }
  if (Jp) { dJp = d; d += 6 * N; }
  {
    ScopedTimer tm(h, BA_K_EVAL);
    hipLaunchKernelGGL(k_eval, dim3(blocks_for(h->nobs)), dim3(kBlock), 0, h->stream, dev_problem(h),
                       h->cams[p].p, h->X[p].p, de, dr, dJc, dJp);
  }
  HIPCHECK(h, hipGetLastError());
  int rc = BA_OK;
  if (e && rc == BA_OK) rc = download_rows(h, h->operm_identity ? nullptr : h->d_operm.p, de, e, N, 2);
  if (r && rc == BA_OK) rc = download_rows(h, h->operm_identity ? nullptr : h->d_operm.p, dr, r, N, 2);
  if (Jc && rc == BA_OK) rc = download_rows(h, h->operm_identity ? nullptr : h->d_operm.p, dJc, Jc, N, 12);
  if (Jp && rc == BA_OK) rc = download_rows(h, h->operm_identity ? nullptr : h->d_operm.p, dJp, Jp, N, 6);
  if (rc != BA_OK) return rc;
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  return BA_OK;
}

int ba_eval_sensor(ba_handle* h, int64_t n, const double* e, double* r, double* J) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, n >= 0 && (n == 0 || e), BA_ERR_INVALID_ARG, "ba_eval_sensor: bad arguments");
  if (n == 0 || (!r && !J)) return BA_OK;
  HIPCHECK(h, hipSetDevice(h->device));
  const size_t N = (size_t)n;
  HIPCHECK(h, h->scratch.resize(8 * N));
  double* de = h->scratch.p;
  double* dr = de + 2 * N;
  double* dJ = dr + 2 * N;
  HIPCHECK(h, hipMemcpyAsync(de, e, 2 * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  {
    ScopedTimer tm(h, BA_K_EVAL);
    hipLaunchKernelGGL(k_eval_sensor, dim3(blocks_for(n)), dim3(kBlock), 0, h->stream, h->sensor, (long long)n, de,
                       r ? dr : nullptr, J ? dJ : nullptr);
  }
  HIPCHECK(h, hipGetLastError());
  if (r) HIPCHECK(h, hipMemcpyAsync(r, dr, 2 * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (J) HIPCHECK(h, hipMemcpyAsync(J, dJ, 4 * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  return BA_OK;
}

int ba_linearize(ba_handle* h, int which, int store_W) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_linearize: bad parameter set");
  REQUIRE(h, h->have_problem && h->have_params[h->phys(which)], BA_ERR_STATE, "ba_linearize: set problem and parameters first");
  return linearize_impl(h, which, store_W, false, 0.0, 0.0);
}

int ba_get_blocks(ba_handle* h, double* HCC, double* bC, double* HPP, double* bP, double* W) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_linearization, BA_ERR_STATE, "ba_get_blocks: call ba_linearize first");
  HIPCHECK(h, hipSetDevice(h->device));
  if ((HPP || bP) && !h->point_blocks_valid) {
    int rc = launch_point_blocks(h, h->lin_phys, nullptr);
    if (rc != BA_OK) return rc;
  }
  if ((HCC || bC) && !h->cam_blocks_valid) {       // ba_lm_trial left them to the reduction kernel
    int rc = launch_camera_blocks(h, h->lin_phys, true);
    if (rc != BA_OK) return rc;
  }
  std::vector<double> hpp6;
  if (HCC && h->nc) HIPCHECK(h, hipMemcpyAsync(HCC, h->HCC.p, (size_t)h->nc * 36 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (bC && h->nc) HIPCHECK(h, hipMemcpyAsync(bC, h->bC.p, (size_t)h->nc * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (HPP && h->nt) {
    hpp6.resize((size_t)h->nt * 6);
    HIPCHECK(h, hipMemcpyAsync(hpp6.data(), h->HPP.p, hpp6.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  if (bP) { const int rc = download_rows(h, h->pperm.empty() ? nullptr : h->d_pperm.p, h->bP.p, bP, (size_t)h->nt, 3); if (rc != BA_OK) return rc; }
  if (W && h->nobs) {
    REQUIRE(h, h->W.p && h->W.n >= (size_t)h->nobs * 18, BA_ERR_STATE, "ba_get_blocks: W was not stored (ba_linearize store_W=0)");
    const int rc = download_rows(h, h->operm_identity ? nullptr : h->d_operm.p, h->W.p, W, (size_t)h->nobs, 18);
    if (rc != BA_OK) return rc;
  }
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  if (HCC) {   // device keeps the upper triangle only
    for (int i = 0; i < h->nc; ++i)
      for (int a = 0; a < 6; ++a)
        for (int c = 0; c < a; ++c) HCC[(size_t)i * 36 + a * 6 + c] = HCC[(size_t)i * 36 + c * 6 + a];
  }
  if (HPP) {
    for (int k = 0; k < h->nt; ++k) {
      const double* s = &hpp6[(size_t)k * 6];
      double* d = HPP + (size_t)(h->pperm.empty() ? k : h->pperm[k]) * 9;
      d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[1]; d[4] = s[3]; d[5] = s[4]; d[6] = s[2]; d[7] = s[4]; d[8] = s[5];
    }
  }
  return BA_OK;
}

int ba_backsubstitute(ba_handle* h, int which, const double* dC, double* dP) {
  if (h) { const int rc = ensure_plan(h); if (rc != BA_OK) return rc; }
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_backsubstitute: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_schur && h->have_params[p], BA_ERR_STATE, "ba_backsubstitute: call ba_schur first");
  REQUIRE(h, dC || h->have_solution || h->nco == 0, BA_ERR_STATE,
          "ba_backsubstitute: dC is NULL and no device solution exists (ba_solve_reduced)");
  HIPCHECK(h, hipSetDevice(h->device));
  if (dC && h->nco) {
    const double* src = cam_rows_in(h, dC, h->rows_host, 6);
    HIPCHECK(h, hipMemcpyAsync(h->dC.p, src, (size_t)h->nco * 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->have_solution = true;
  }
  // inside ba_lm_trial the update of the trial parameter set rides along (one launch less)
  const bool fuse_update = h->defer && which == BA_PARAMS_CUR && h->nt > 0;
  if (fuse_update) h->params_written(1 - p);
  if (h->nt > 0) {
    ScopedTimer tm(h, BA_K_BACKSUB);
    const long long threads = (long long)h->nt << h->glog;
    h->cost_fused = false;
    if (h->point_groups && !h->opt.point_kernels_v1 && h->sensor.kind != SENSOR_TABLE) {
      // inside ba_lm_trial the cost of the trial set rides along as well (k_cost's work)
      const int per_block = kBlock / kWave;
      const int nblk = std::min(kCostBlocks, (h->ngroups + per_block - 1) / per_block);
      const bool fuse_cost = fuse_update && h->opt.fuse_cost;
      hipLaunchKernelGGL(k_backsub_groups, dim3(nblk), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p, h->X[p].p,
                         h->groups.p, h->ngroups, h->dC.p, h->HPPinv.p, h->bP.p, h->dP.p, -1.0,
                         fuse_update ? h->cams[1 - p].p : (double*)nullptr, fuse_update ? h->X[1 - p].p : (double*)nullptr,
                         (const int*)h->sing_counter(), (const int*)(h->flags.p + 1),
                         fuse_cost ? h->host_result : (HostResult*)nullptr, fuse_cost ? h->trial_result_dev : (double*)nullptr);
      if (fuse_cost) { h->cost_fused = true; h->cost_blocks = nblk; }
    } else {
      hipLaunchKernelGGL(k_backsub, dim3(blocks_for(threads)), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p,
                         h->X[p].p, h->glog, h->dC.p, h->HPPinv.p, h->bP.p, h->dP.p, -1.0,
                         fuse_update ? h->cams[1 - p].p : (double*)nullptr, fuse_update ? h->X[1 - p].p : (double*)nullptr);
    }
  }
  HIPCHECK(h, hipGetLastError());
  if (dP) { const int rc = download_rows(h, h->pperm.empty() ? nullptr : h->d_pperm.p, h->dP.p, dP, (size_t)h->nt, 3); if (rc != BA_OK) return rc; }
  if (dC || dP) HIPCHECK(h, hipStreamSynchronize(h->stream));   // dC is caller memory
  h->have_backsub = true;
  return BA_OK;
}

int ba_apply_update(ba_handle* h, int src, int dst, const double* motion, const double* structure) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, (src == 0 || src == 1) && (dst == 0 || dst == 1), BA_ERR_INVALID_ARG, "ba_apply_update: bad parameter set");
  const int ps = h->phys(src), pd = h->phys(dst);
  h->params_written(pd);
  REQUIRE(h, h->have_problem && h->have_params[ps], BA_ERR_STATE, "ba_apply_update: source parameter set is empty");
  REQUIRE(h, (motion == nullptr) == (structure == nullptr), BA_ERR_INVALID_ARG,
          "ba_apply_update: give both motion and structure, or neither");
  HIPCHECK(h, hipSetDevice(h->device));
  if (int rcj = border_join(h); rcj != BA_OK) return rcj;      // (border kernels of an earlier ba_schur may still be reading what this call writes: side stream, ba_border.hip)

  double sign = -1.0;
  if (motion) {
    sign = 1.0;
    const double* src = cam_rows_in(h, motion, h->rows_host, 6);
    if (h->nco) HIPCHECK(h, hipMemcpyAsync(h->dC.p, src, (size_t)h->nco * 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->nt) { const int rc = upload_rows(h, h->pperm.empty() ? nullptr : h->d_pperm.p, structure, h->dP.p, (size_t)h->nt, 3); if (rc != BA_OK) return rc; }
    h->have_backsub = h->have_solution = false;   // dC / dP now hold the caller's update
  } else {
    REQUIRE(h, h->have_backsub, BA_ERR_STATE, "ba_apply_update: no update on the device (call ba_backsubstitute)");
  }
  if (h->nc + h->nt > 0) {
    ScopedTimer tm(h, BA_K_UPDATE);
    hipLaunchKernelGGL(k_apply_update, dim3(blocks_for((long long)h->nc + h->nt)), dim3(kBlock), 0, h->stream, h->nc,
                       h->nt, h->cam_opt_pos.p, h->pt_opt.p, h->cams[ps].p, h->X[ps].p, h->dC.p, h->dP.p, sign,
                       h->cams[pd].p, h->X[pd].p);
  }
  HIPCHECK(h, hipGetLastError());
  if (motion) HIPCHECK(h, hipStreamSynchronize(h->stream));
  h->have_params[pd] = true;
  if (dst == BA_PARAMS_CUR) h->have_linearization = h->have_schur = false;
  return BA_OK;
}

int ba_triangulate(ba_handle* h, int which, double rcond, double* X) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_triangulate: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_problem && h->have_params[p], BA_ERR_STATE, "ba_triangulate: set problem and parameters first");
  h->params_written(p);
  HIPCHECK(h, hipSetDevice(h->device));
  if (rcond < 0) rcond = 2.220446049250313e-16 * std::max<double>(3.0, 2.0 * 64);   // numpy's default scale
  // QR of the 2L x 3 system (k_triangulate): full-rank systems are solved to cond(A) * eps like lstsq's; the rank decision
  // (|R_jj| <= rcond * max |R_ii|, never below 1e-13) sends what is rank deficient to working precision to the minimum-norm answer
  if (h->nt > 0) {
    ScopedTimer tm(h, BA_K_TRIANGULATE);
    const long long threads = (long long)h->nt << h->glog;
    hipLaunchKernelGGL(k_triangulate, dim3(blocks_for(threads)), dim3(kBlock), 0, h->stream, dev_problem(h), h->cams[p].p,
                       h->glog, std::max(rcond, 1e-13), h->X[p].p);
  }
  HIPCHECK(h, hipGetLastError());
  if (which == BA_PARAMS_CUR) h->have_linearization = h->have_schur = h->have_backsub = h->have_solution = false;
  if (X && h->nt) {
    const int rc = download_rows(h, h->pperm.empty() ? nullptr : h->d_pperm.p, h->X[p].p, X, (size_t)h->nt, 3);
    if (rc != BA_OK) return rc;
    HIPCHECK(h, hipStreamSynchronize(h->stream));
  }
  return BA_OK;
}

}  // extern "C"
