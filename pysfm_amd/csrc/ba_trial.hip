// ba_trial.hip - ba_lm_trial: one Levenberg-Marquardt trial as one batch of launches with one synchronisation.
#include "ba_internal.h"


using namespace ba;

extern "C" {

int ba_lm_trial_begin(ba_handle* h, double damping, double pinv_rcond) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem && h->have_params[h->phys(BA_PARAMS_CUR)], BA_ERR_STATE, "ba_lm_trial: set problem and parameters first");
  { const int rc = ensure_plan(h); if (rc != BA_OK) return rc; }
  h->defer = true;
  // with the MFMA reduction the camera blocks come out of the reduction itself: one launch and one pass
  // over the observations less
  // ... up to six tiles a side of the widest window (tracks of 16 cameras): beyond, the reduction is several launches over
  // the observations and a separate k_camera_blocks is the cheaper way (tracks of 18 .. 40: 7 - 22 % of the trial)
  const int kern = pick_schur_kernel(h);
  const bool fuse = h->opt.schur == SCHUR_AUTO && h->opt.fuse_cam && kern_is_mfma(kern) && (kern != KERN_MFMA3 || h->gm3.nts <= 6);
  // After a REJECTED trial the current set is what it was: its point blocks (and, where the reduction does not form them itself,
  // its camera blocks) are still on the device - the reference recomputes identical blocks there (bundle_adjuster.py:132-140,
  // SURVEY 3.1).  Everything that writes the current set or changes the model clears have_linearization.
  const bool reuse = h->opt.reuse_linearization && h->have_linearization && h->lin_phys == h->phys(BA_PARAMS_CUR) && h->point_blocks_valid &&
                     (fuse || h->cam_blocks_valid);
  int rc = BA_OK;
  if (reuse) { h->have_schur = h->have_backsub = false; ++h->lin_reused; }
  else rc = linearize_impl(h, BA_PARAMS_CUR, 0, fuse, damping, pinv_rcond);
  if (rc == BA_OK) rc = ba_schur(h, BA_PARAMS_CUR, damping, pinv_rcond);
  h->defer = false;
  h->trial_rcond = pinv_rcond;
  return rc;
}

// the tail of a trial once the solution is on the device: back-substitution, trial parameter set, trial cost - nothing read back
int ba_lm_trial_finish(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_schur && h->have_solution, BA_ERR_STATE, "ba_lm_trial_finish: no solution on the device");
  h->defer = true;
  double unused = 0.0;
  int rc = ba_backsubstitute(h, BA_PARAMS_CUR, nullptr, nullptr);
  if (rc == BA_OK) {
    if (h->nt > 0) h->have_params[h->phys(BA_PARAMS_TRIAL)] = true;      // k_backsub wrote the trial set
    else rc = ba_apply_update(h, BA_PARAMS_CUR, BA_PARAMS_TRIAL, nullptr, nullptr);
  }
  if (rc == BA_OK && !(h->cost_fused && h->nt > 0)) rc = ba_cost(h, BA_PARAMS_TRIAL, &unused);
  h->defer = false;
  return rc;
}

int ba_lm_trial_end(ba_handle* h, const uint8_t* cam_param_mask, int32_t* pre_info) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, pre_info, BA_ERR_INVALID_ARG, "ba_lm_trial_end: NULL output");
  REQUIRE(h, h->have_schur, BA_ERR_STATE, "ba_lm_trial_end: call ba_lm_trial_begin first");
  *pre_info = 0;
  h->defer = true;
  int rc = ba_solve_reduced(h, cam_param_mask, pre_info);
  h->defer = false;
  if (rc == BA_OK && *pre_info != 0) return BA_OK;   // band too wide: caller takes the dense path
  if (rc == BA_OK) rc = ba_lm_trial_finish(h);
  return rc;
}

int ba_lm_trial(ba_handle* h, double damping, double pinv_rcond, const uint8_t* cam_param_mask, double* next_cost,
                int32_t* info) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, next_cost && info, BA_ERR_INVALID_ARG, "ba_lm_trial: NULL output");
  *info = 0;
  int32_t pre = 0;
  int rc = ba_lm_trial_begin(h, damping, pinv_rcond);
  const bool dist = h->comm && h->dist.on && h->hb <= kBcrMaxHB;
  if (dist) {
    // the solve spread over the ranks (ba_dist.h): three small sums instead of one of the whole band
    for (int stage = 1; stage <= 4 && rc == BA_OK; ++stage) {
      size_t count = 0;
      rc = dist_stage(h, stage, cam_param_mask, &count);
      if (rc == BA_OK && count) RCCLCHECK(h, g_rccl.AllReduce(h->dist.xbuf, h->dist.xbuf, count, ncclFloat64, ncclSum, h->comm, h->stream));
    }
    if (rc == BA_OK) rc = ba_lm_trial_finish(h);
  } else {
    if (rc == BA_OK && h->comm) rc = comm_allreduce_reduced(h);      // sharded: the one data-path collective
    if (rc == BA_OK) rc = ba_lm_trial_end(h, cam_param_mask, &pre);
  }
  if (rc != BA_OK) return rc;
  if (pre != 0) { *info = pre; return BA_OK; }
  int st[2];
  if (h->comm) {
    // the shards' trial records (cost partials | singular blocks | solver status) are summed in place - 16 KB, the
    // latency of 8 bytes - and come back with one copy; the partials are added on the host in index order
    RCCLCHECK(h, g_rccl.AllReduce(h->comm_dev.p, h->comm_dev.p, (size_t)kCostBlocks + 2, ncclFloat64, ncclSum, h->comm, h->stream));
    // (a kernel storing into the pinned record: a 16 KB hipMemcpyAsync goes through the DMA engine and costs more)
    launch_copy_doubles(h, h->comm_dev.p, h->comm_host, kCostBlocks + 2);
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    double sum = 0.0;
    for (int i = 0; i < kCostBlocks; ++i) sum += h->comm_host[i];
    *next_cost = sum;
    st[0] = (int)std::llround(h->comm_host[kCostBlocks]);                          // over all shards
    st[1] = trial_status_of_sum(h->comm_host[kCostBlocks + 1], h->comm_ranks, dist);      // (a time-out on any rank stays a time-out)
  } else {
    HIPCHECK(h, hipStreamSynchronize(h->stream));    // k_cost left the cost partials + status words in pinned memory
    st[0] = h->host_result->singular_points; st[1] = h->host_result->solve_info;
    *next_cost = h->host_cost();
  }
  if (pinv_rcond < 0.0 && st[0] > 0)
    return h->fail(BA_ERR_SINGULAR, "ba_lm_trial: %d singular 3x3 point block(s) in plain-inverse mode", st[0]);
  *info = st[1];
  if (st[1] != 0) h->have_solution = h->have_backsub = false;
  return BA_OK;
}

}  // extern "C"
