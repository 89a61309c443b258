// ba_resident.hip - ba_lm_resident: the Levenberg-Marquardt loop of a small problem as one resident launch of a few
// workgroups (ba_resident.h).
#include "ba_internal.h"
#include "ba_resident.h"

using namespace ba;

static_assert(sizeof(ba_resident_log) == sizeof(ResidentLog), "the log of include/pysfm_ba.h and ba_resident.h must agree");
static_assert(BA_RESIDENT_MAX_TRIALS == kResMaxTrials, "the log of include/pysfm_ba.h and ba_resident.h must agree");

namespace ba {

// the shape of the problem (not yet its sensor model or its communicator) fits the resident loop: ba_set_problem leaves the
// work lists of the general kernels for later then (ensure_plan)
bool resident_shape(const ba_handle* h) {
  if (!h->opt.resident || h->comm) return false;
  if (h->nco < 1 || h->nco > kResMaxNco || h->nc > kResMaxNc || h->nt < 1 || h->nt > kResMaxNt) return false;
  if (h->group_maxL < 1 || h->group_maxL > kResMaxL || h->nobs < 1 || h->nobs > (1 << 20)) return false;
  // its workgroups wait for each other and each fills the LDS of a compute unit: all of them must be resident at once
  if ((h->nt + kResP - 1) / kResP > h->ncu) return false;
  return resident_lds(h->nc, h->nco, h->group_maxL).bytes <= 157 * 1024;
}

}  // namespace ba

namespace {

// every cost word says "not yet" (at allocation, and after a launch that lost its workgroups)
int resident_reset_cost_words(ba_handle* h) {
  long long init[2 * kResMaxGroups];
  for (long long& w : init) w = kResNotYet;
  HIPCHECK(h, hipMemcpy(h->res_cost.p, init, sizeof init, hipMemcpyHostToDevice));
  h->res_parity = 0;
  return BA_OK;
}

// ... and its sensor model, its state and the handle's mode allow it too
bool resident_fits(const ba_handle* h, ResidentLds* lds_out) {
  if (!h->have_problem || h->dense_mode || !resident_shape(h)) return false;
  if (lds_out) *lds_out = resident_lds(h->nc, h->nco, h->group_maxL);
  return true;
}

}  // namespace

extern "C" {

int ba_lm_resident_fits(ba_handle* h) {
  if (!h) return 0;
  return resident_fits(h, nullptr) ? 1 : 0;
}

// the launch, and nothing that waits for it
int ba_lm_resident_begin(ba_handle* h, int32_t max_steps, int32_t steps_taken, int32_t in_step, int32_t converged, double damping,
                         double improvement_threshold, double pinv_rcond, double cur_cost, const uint8_t* cam_param_mask) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, !h->res_inflight, BA_ERR_STATE, "ba_lm_resident_begin: the last launch has not been collected (ba_lm_resident_end)");
  REQUIRE(h, h->have_problem && h->have_params[h->phys(BA_PARAMS_CUR)], BA_ERR_STATE, "ba_lm_resident: set problem and parameters first");
  ResidentLds lds;
  REQUIRE(h, resident_fits(h, &lds), BA_ERR_STATE, "ba_lm_resident: not a problem for the resident loop (ba_lm_resident_fits)");
  HIPCHECK(h, hipSetDevice(h->device));
  if (int rcj = border_join(h); rcj != BA_OK) return rcj;      // (border kernels of an earlier ba_schur may still be reading what this call writes: side stream, ba_border.hip)
  const int G = (h->nt + kResP - 1) / kResP;
  if (h->res_xb.n < (size_t)(G + 1) * kResRec || h->res_epoch.n < (size_t)2 * G) {
    HIPCHECK(h, h->res_xb.resize((size_t)(kResMaxGroups + 1) * kResRec));      // + the record of sums
    HIPCHECK(h, h->res_epoch.resize((size_t)2 * kResMaxGroups));
    HIPCHECK(h, hipMemsetAsync(h->res_epoch.p, 0, h->res_epoch.n * sizeof(long long), h->stream));
    h->res_epoch0 = 0;
    HIPCHECK(h, h->res_cost.resize((size_t)2 * kResMaxGroups));
    const int rc = resident_reset_cost_words(h);
    if (rc != BA_OK) return rc;
  }
  if (!h->res_log) HIPCHECK(h, hipHostMalloc(&h->res_log, sizeof(ResidentLog), hipHostMallocDefault));
  if (!h->res_exit) HIPCHECK(h, hipHostMalloc((void**)&h->res_exit, 2 * kResMaxGroups * sizeof(int), hipHostMallocDefault));
  HIPCHECK(h, h->res_stage.resize((size_t)h->nc * 12 + (size_t)h->nt * 3));
  for (int i = 0; i < 2 * kResMaxGroups; ++i) h->res_exit[i] = -1;
  {
    const size_t need = (size_t)h->nc * 12 + (size_t)h->nt * 3;
    if (h->res_out_doubles < need) {
      if (h->res_out) (void)hipHostFree(h->res_out);
      h->res_out = nullptr; h->res_out_doubles = 0;
      HIPCHECK(h, hipHostMalloc((void**)&h->res_out, (need + 1024) * sizeof(double), hipHostMallocDefault));
      h->res_out_doubles = need + 1024;
    }
    h->res_out_phys = -1;
  }
  {
    // (the kernel has a few static LDS words of its own: ask for less than the whole 160 KB)
    const void* fn = h->sensor.kind == SENSOR_TABLE ? (const void*)k_resident_lm<true> : (const void*)k_resident_lm<false>;
    if (std::find(h->lds_attr_done.begin(), h->lds_attr_done.end(), fn) == h->lds_attr_done.end()) {
      HIPCHECK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
      h->lds_attr_done.push_back(fn);
    }
  }
  const int p = h->phys(BA_PARAMS_CUR);
  ResidentArgs a;
  a.nc = h->nc; a.nt = h->nt; a.nco = h->nco; a.maxL = h->group_maxL; a.nobs = (int)h->nobs;
  a.ngroups = G;
  a.obs_cam = h->obs_cam.p; a.obs_z = h->obs_z.p; a.pt_off = h->pt_off.p; a.cam_opt_pos = h->cam_opt_pos.p; a.pt_opt = h->pt_opt.p;
  for (int i = 0; i < 9; ++i) a.K[i] = h->K[i];
  a.sensor = h->sensor;
  if (!h->opt.fast_paths) a.sensor.fast = 0;
  a.cams = h->cams[p].p; a.X = h->X[p].p; a.stage = h->res_stage.p; a.group_exit = h->res_exit; a.fault_group = h->opt.resident_fault; a.out = h->res_out; a.xb = h->res_xb.p; a.epoch = h->res_epoch.p; a.epoch0 = h->res_epoch0;
  a.cost_slots = h->res_cost.p; a.parity0 = h->res_parity;
  a.scatter_min = h->opt.resident_scatter_min;
  a.max_steps = max_steps; a.max_trials = kResMaxTrials; a.nsteps = steps_taken; a.in_step = in_step ? 1 : 0; a.converged = converged ? 1 : 0;
  a.damping = damping; a.improvement_threshold = improvement_threshold; a.rcond = pinv_rcond; a.cur_cost = cur_cost;
  a.have_mask = cam_param_mask ? 1 : 0;
  memset(a.mask, 1, sizeof a.mask);
  if (cam_param_mask)
    for (int i = 0; i < 6 * h->nco; ++i) a.mask[h->cpos_in.empty() ? i : h->cpos_in[i / 6] * 6 + i % 6] = cam_param_mask[i] ? 1 : 0;
  a.log = static_cast<ResidentLog*>(h->res_log);
  a.trace = nullptr;
  a.dbg = nullptr;
  if (h->opt.solve_trace) {
    if (!h->res_trace) HIPCHECK(h, hipHostMalloc(&h->res_trace, 64 * 16 * sizeof(long long) + ((kResMaxN + 2) * kResSLd + 128) * sizeof(double), hipHostMallocDefault));
    memset(h->res_trace, 0, 64 * 16 * sizeof(long long));
    a.trace = static_cast<long long*>(h->res_trace);
    a.dbg = reinterpret_cast<double*>(a.trace + 64 * 16);
  }
  a.log->ntrials = -1;
  if (h->sensor.kind == SENSOR_TABLE) hipLaunchKernelGGL(k_resident_lm<true>, dim3(G), dim3(kResThreads), lds.bytes, h->stream, a);
  else hipLaunchKernelGGL(k_resident_lm<false>, dim3(G), dim3(kResThreads), lds.bytes, h->stream, a);
  HIPCHECK(h, hipGetLastError());
  h->res_inflight = true;
  h->res_inflight_phys = p;
  h->res_inflight_groups = G;
  return BA_OK;
}

// ... and the wait: the log, the state of the handle after the run
int ba_lm_resident_end(ba_handle* h, ba_resident_log* log) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, log, BA_ERR_INVALID_ARG, "ba_lm_resident_end: NULL log");
  REQUIRE(h, h->res_inflight, BA_ERR_STATE, "ba_lm_resident_end: no launch to collect (ba_lm_resident_begin)");
  h->res_inflight = false;
  const int p = h->res_inflight_phys;
  struct { ResidentLog* log; } a = {static_cast<ResidentLog*>(h->res_log)};
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  REQUIRE(h, a.log->ntrials >= 0, BA_ERR_HIP, "ba_lm_resident: the kernel left no log");
  h->res_epoch0 += 2 * ((long long)a.log->ntrials + 2);
  h->res_parity = (h->res_parity + a.log->ntrials) & 1;
  // every workgroup must have ended the same way after the same number of trials: anything else (one that gave up waiting for the
  // others, or - same fault, seen from the other side - one that finished while another gave up) is a launch that did not happen
  bool agreed = a.log->exit_reason != RES_TIMED_OUT;
  for (int g = 0; g < h->res_inflight_groups; ++g)
    agreed = agreed && h->res_exit[2 * g] == a.log->exit_reason && h->res_exit[2 * g + 1] == a.log->ntrials;
  if (agreed) h->res_out_phys = p;      // (the kernel left the set it ended on in pinned memory)
  if (!agreed) {
    // a workgroup gave up waiting for the others: a fault of the kernel or of the GPU, never a property of the problem
    (void)hipMemsetAsync(h->res_epoch.p, 0, h->res_epoch.n * sizeof(long long), h->stream);
    h->res_epoch0 = 0;
    (void)resident_reset_cost_words(h);
    // the kernel writes to a staging copy only (ba_resident.h): the current set is the one the launch was given, and the log says so -
    // no trial, nothing accepted, the schedule where it was.  The caller goes on through ba_lm_trial.
    h->err = "ba_lm_resident: the workgroups of the resident loop lost each other (timed out); nothing was changed";
    memset(log, 0, offsetof(ResidentLog, trial_damping));
    log->exit_reason = RES_TIMED_OUT;
    log->exit_info = a.log->ntrials;
    return BA_OK;
  }
  {
    // the header and what the trials wrote (the log is 20 KB, a window's run fills a fiftieth of it)
    const ResidentLog* src = a.log;
    const size_t nt = (size_t)std::min(src->ntrials, kResMaxTrials);
    memcpy(log, src, offsetof(ResidentLog, trial_damping));
    memcpy(log->trial_damping, src->trial_damping, nt * sizeof(double));
    memcpy(log->trial_cost, src->trial_cost, nt * sizeof(double));
    memcpy(log->trial_accepted, src->trial_accepted, nt * sizeof(int));
  }
  // the current set has moved (or not): the staging copy becomes the current set (two small copies behind the kernel, nobody waits
  // for them), and nothing that was derived from the old one on the device is valid any more
  if (log->accepted) {
    HIPCHECK(h, hipMemcpyAsync(h->cams[p].p, h->res_stage.p, (size_t)h->nc * 12 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(h->X[p].p, h->res_stage.p + (size_t)h->nc * 12, (size_t)h->nt * 3 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    h->have_linearization = h->have_schur = h->have_backsub = h->have_solution = false;
    h->point_blocks_valid = h->cam_blocks_valid = h->inv_valid = h->fac_valid = false;
  }
  return BA_OK;
}

int ba_lm_resident(ba_handle* h, int32_t max_steps, int32_t steps_taken, int32_t in_step, int32_t converged, double damping,
                   double improvement_threshold, double pinv_rcond, double cur_cost, const uint8_t* cam_param_mask,
                   ba_resident_log* log) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, log, BA_ERR_INVALID_ARG, "ba_lm_resident: NULL log");
  const int rc = ba_lm_resident_begin(h, max_steps, steps_taken, in_step, converged, damping, improvement_threshold, pinv_rcond, cur_cost, cam_param_mask);
  return rc != BA_OK ? rc : ba_lm_resident_end(h, log);
}

int ba_lm_resident_trace(ba_handle* h, int64_t* out) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, out && h->res_trace, BA_ERR_STATE, "ba_lm_resident_trace: set option solve_trace and run ba_lm_resident first");
  memcpy(out, h->res_trace, 64 * 16 * sizeof(long long));
  return BA_OK;
}

int ba_lm_resident_debug(ba_handle* h, double* S_out, double* b_out, double* dC_out) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, S_out && b_out && dC_out && h->res_trace, BA_ERR_STATE, "ba_lm_resident_debug: set option solve_trace and run ba_lm_resident first");
  const double* d = reinterpret_cast<const double*>(static_cast<const long long*>(h->res_trace) + 64 * 16);
  const int n = 6 * h->nco;
  auto at = [&](int i) { return h->cpos_out.empty() ? i : h->cpos_out[i / 6] * 6 + i % 6; };      // internal parameter index -> the caller's
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) S_out[(size_t)at(i) * n + at(j)] = i >= j ? d[i * kResSLd + j] : d[j * kResSLd + i];
  for (int i = 0; i < n; ++i) { b_out[at(i)] = d[n * kResSLd + i]; dC_out[at(i)] = d[(kResMaxN + 2) * kResSLd + i]; }
  return BA_OK;
}

}  // extern "C"
