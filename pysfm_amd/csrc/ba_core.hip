// ba_core.hip - the handle of the C ABI (include/pysfm_ba.h): life cycle, options, stream, parameters, cost / evaluation, RCCL, timing.
#include "ba_internal.h"

#include <dlfcn.h>

using namespace ba;

namespace ba {

thread_local std::string g_create_error;
RcclApi g_rccl;

hipEvent_t get_event(ba_handle* h) {
  if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);      // (timing only: nobody reads memory on the strength of these events; the system-scope fence of a default event costs the stream ~20 us)
  return e;
}

void resolve_timings(ba_handle* h) {
  if (h->pending.empty()) return;
  (void)hipStreamSynchronize(h->stream);
  for (auto& t : h->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { h->ms[t.id] += ms; h->launches[t.id] += t.count; }
    h->ev_pool.push_back(t.a);
    h->ev_pool.push_back(t.b);
  }
  h->pending.clear();
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: remembered per handle (one handle = one device)
hipError_t ensure_lds_attr(ba_handle* h, const void* fn) {
  for (const void* f : h->lds_attr_done) if (f == fn) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) h->lds_attr_done.push_back(fn);
  return e;
}

DevProblem dev_problem(const ba_handle* h) {
  DevProblem P;
  P.nc = h->nc; P.nt = h->nt; P.nco = h->nco; P.hb = h->hb; P.nobs = h->nobs;
  P.obs_cam = h->obs_cam.p; P.obs_pt = h->obs_pt.p; P.obs_z = h->obs_z.p; P.pt_off = h->pt_off.p;
  P.cam_opt_pos = h->cam_opt_pos.p; P.pt_opt = h->pt_opt.p;
  std::memcpy(P.K, h->K, sizeof P.K);
  P.sensor = h->sensor;
  const double* K = h->K;
  if (K[0] == 1.0 && K[4] == 1.0 && K[8] == 1.0 && K[1] == 0.0 && K[2] == 0.0 && K[3] == 0.0 && K[5] == 0.0 && K[6] == 0.0 && K[7] == 0.0)
    P.sensor.fast |= FAST_K_IDENTITY;
  if (!h->opt.fast_paths) P.sensor.fast = 0;
  return P;
}

DevProblem dev_problem_band(const ba_handle* h) {
  DevProblem P = dev_problem(h);
  if (h->nbc > 0) P.cam_opt_pos = h->cam_band_pos.p;
  return P;
}

int ensure_reduced(ba_handle* h) {
  if (!h->S) {
    HIPCHECK(h, h->S_own.resize(std::max<size_t>(1, reduced_doubles(h))));
    h->S = h->S_own.p;
  }
  if (!h->b) {
    HIPCHECK(h, h->b_own.resize(std::max<size_t>(1, (size_t)h->nco * 6)));
    h->b = h->b_own.p;
  }
  return BA_OK;
}

// Host-facing per-point / per-observation arrays go through the internal order of ba_set_problem on the DEVICE
// (k_rows_permute: rows of w doubles, perm[i] = the caller's index of internal row i; nullptr = identity).
__global__ __launch_bounds__(256) void k_rows_permute(long long n, int w, const int* __restrict__ perm, const double* __restrict__ src,
                                                      double* __restrict__ dst, int to_internal) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * w) return;
  const long long i = t / w;
  const int a = (int)(t - i * w);
  const long long j = perm[i];
  if (to_internal) dst[i * w + a] = src[j * w + a];
  else dst[j * w + a] = src[i * w + a];
}

int download_rows(ba_handle* h, const int* dev_perm, const double* dev, double* host, size_t n, int w) {
  if (n == 0) return BA_OK;
  if (dev_perm) {
    HIPCHECK(h, h->scratch2.resize(n * w));
    hipLaunchKernelGGL(k_rows_permute, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, h->stream, (long long)n, w, dev_perm, dev, h->scratch2.p, 0);
    dev = h->scratch2.p;
  }
  HIPCHECK(h, hipMemcpyAsync(host, dev, n * w * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (dev_perm) HIPCHECK(h, hipStreamSynchronize(h->stream));      // (scratch2 is free again)
  return BA_OK;
}

int upload_rows(ba_handle* h, const int* dev_perm, const double* host, double* dev, size_t n, int w) {
  if (n == 0) return BA_OK;
  if (!dev_perm) {
    HIPCHECK(h, hipMemcpyAsync(dev, host, n * w * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return BA_OK;
  }
  HIPCHECK(h, h->scratch2.resize(n * w));
  HIPCHECK(h, hipMemcpyAsync(h->scratch2.p, host, n * w * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_rows_permute, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, h->stream, (long long)n, w, dev_perm, h->scratch2.p, dev, 1);
  return BA_OK;
}

// in-place sum over the shards of the band-stored [S | b] (contiguous), on the handle's stream
int comm_allreduce_reduced(ba_handle* h) {
  if (h->pcg.packed && !h->pcg.shared_lists) return h->fail(BA_ERR_STATE, "the reduced system of this problem is stored as the list of ITS tracks' blocks: the shards of a sharded adjuster cannot add such lists up (set the communicator before ba_set_problem)");
  const size_t nS = reduced_doubles(h), nb = (size_t)h->nco * 6;
  if (nS + nb == 0) return BA_OK;
  if (h->b == h->S + nS) {                            // the usual case: one contiguous [S | b]
    RCCLCHECK(h, g_rccl.AllReduce(h->S, h->S, nS + nb, ncclFloat64, ncclSum, h->comm, h->stream));
  } else {
    RCCLCHECK(h, g_rccl.AllReduce(h->S, h->S, nS, ncclFloat64, ncclSum, h->comm, h->stream));
    RCCLCHECK(h, g_rccl.AllReduce(h->b, h->b, nb, ncclFloat64, ncclSum, h->comm, h->stream));
  }
  return BA_OK;
}

}  // namespace ba

__global__ __launch_bounds__(256) void k_copy_doubles(const double* __restrict__ src, double* __restrict__ dst, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// The achievable HBM rate of this box (SURVEY 8d asks for the roofline fraction against it as well as
// against the 8 TB/s of the data sheet): a streaming copy, ONE 16-byte element per lane, no loop, non-temporal
// loads and stores.  tools/copy_probe.hip compares the forms on an MI355X (read + write, 1 GiB): this one
// 6.5 TB/s; the grid-stride loop it replaces 4.5 - 5.0; four elements per lane in flight 5.6 - 6.2;
// hipMemcpyAsync device-to-device 5.2.  (The guide quotes 6.29 TB/s for its float4 copy.)
typedef float copy_vec __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_copy(const copy_vec* __restrict__ src, copy_vec* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// test aid (ba_debug_poison): every LDS word of a compute unit / every double of a workspace buffer becomes a NaN
__global__ __launch_bounds__(1024) void k_poison_lds(int ndoubles) {
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  for (int i = threadIdx.x; i < ndoubles; i += 1024) dyn[i] = __longlong_as_double(0x7FF8DEADDEADDEADll);
  __syncthreads();
  if (dyn[(threadIdx.x * 7) % ndoubles] == 0.0) dyn[0] = 1.0;      // (keeps the stores)
}
__global__ __launch_bounds__(256) void k_poison_doubles(double* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = __longlong_as_double(0x7FF8DEADDEADDEADll);
}

namespace ba {

void launch_copy_doubles(ba_handle* h, const double* src, double* dst, int n) {
  hipLaunchKernelGGL(k_copy_doubles, dim3((n + 255) / 256), dim3(256), 0, h->stream, src, dst, n);
}

}  // namespace ba

extern "C" {

const char* ba_version(void) { return "pysfm_ba 0.1 (gfx950)"; }

const char* ba_kernel_name(int id) {
  static const char* names[BA_K_COUNT] = {"k_cost", "k_linearize", "k_point_invert", "k_schur_init",
                                          "k_schur_pairs", "k_backsub", "k_apply_update", "k_flatten",
                                          "k_band_solve", "k_eval", "k_camera_blocks", "k_triangulate",
                                          "k_bcr_assemble", "k_bcr_eliminate", "k_bcr_backsolve", "k_dense_solve", "k_schur_border", "k_border_solve", "k_bcr_refine", "k_pcg"};
  return (id >= 0 && id < BA_K_COUNT) ? names[id] : "?";
}

const char* ba_last_error(const ba_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ba_create(int device_id, ba_handle** out) {
  if (!out) { g_create_error = "ba_create: out is NULL"; return BA_ERR_INVALID_ARG; }
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = std::string("ba_create: no HIP device available (") +
                     (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") + ")";
    return BA_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= ndev) {
    g_create_error = "ba_create: device_id out of range";
    return BA_ERR_INVALID_ARG;
  }
  if ((e = hipSetDevice(device_id)) != hipSuccess) {
    g_create_error = std::string("ba_create: hipSetDevice failed: ") + hipGetErrorString(e);
    return BA_ERR_HIP;
  }
  ba_handle* h = new ba_handle();
  h->device = device_id;
  (void)hipDeviceGetAttribute(&h->ncu, hipDeviceAttributeMultiprocessorCount, device_id);
  if (h->ncu <= 0) h->ncu = 256;
  if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) {
    g_create_error = std::string("ba_create: hipStreamCreate failed: ") + hipGetErrorString(e);
    delete h;
    return BA_ERR_HIP;
  }
  h->own_stream = true;
  if ((e = hipHostMalloc((void**)&h->host_result, sizeof(HostResult), hipHostMallocDefault)) != hipSuccess) {
    g_create_error = std::string("ba_create: hipHostMalloc failed: ") + hipGetErrorString(e);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return BA_ERR_HIP;
  }
  *out = h;
  return BA_OK;
}

int ba_destroy(ba_handle* h) {
  if (!h) return BA_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)ba_comm_destroy(h);
  for (auto& t : h->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (auto e : h->ev_pool) (void)hipEventDestroy(e);
  h->obs_cam.release(); h->obs_pt.release(); h->pt_off.release(); h->cam_opt_pos.release(); h->opt_cam.release();
  h->keep.release(); h->obs_z.release(); h->pt_opt.release(); h->units.release(); h->chunks.release(); h->groups.release(); h->mgroups.release(); h->gchunks.release(); h->mchunks.release(); h->m3chunks.release(); h->wide_list.release(); h->wgroups.release(); h->wtab.release(); h->cam_perm.release(); h->cam_units.release();
  for (int i = 0; i < 2; ++i) { h->cams[i].release(); h->X[i].release(); }
  h->HCC.release(); h->bC.release(); h->HPP.release(); h->bP.release(); h->HPPinv.release();
  h->W.release(); h->S_own.release(); h->b_own.release(); h->dC.release(); h->Ufac.release(); h->ysol.release(); h->dinv.release();
  h->bcrD.release(); h->bcrU.release(); h->bcrF.release(); h->bcrP.release(); h->bcrQ.release(); h->bcrG.release(); h->bcrGv.release(); h->bcr_order.release(); h->bcr_work.release(); h->bcr_done.release(); h->bcr_trace.release(); h->dist.rows.release(); h->dist.sep.release(); h->dist.sep_owner.release(); h->dist.root.release(); h->dist.root_owner.release(); h->dist.work.release(); h->rgroups.release(); h->rtab.release(); h->dist.order.release(); h->dist.asm_nodes.release(); h->dist.xown.release(); h->bcrL.release(); h->bcrLv.release(); h->denseA.release(); h->bigK.release(); h->fac.release(); h->dUd.release(); h->dDd.release(); h->dyd.release(); h->dpart.release(); h->mask.release(); h->dP.release();
  h->scratch.release(); h->scratch2.release(); h->sensor_table.release(); h->flags.release(); h->d_pperm.release(); h->d_operm.release();
  {
    auto& su = h->su;
    su.rc.release(); su.rp.release(); su.by_pt.release(); su.cnt.release(); su.coff.release(); su.Lint.release(); su.plo.release(); su.phi.release();
    su.iota.release(); su.crank.release(); su.flags.release(); su.vals.release(); su.key.release(); su.key2.release(); su.tkey.release(); su.tkey2.release();
    su.rz.release(); su.rpo.release(); su.same.release(); su.tmp.release(); su.blob.release();
    if (su.host) (void)hipHostFree(su.host);
    if (su.up) (void)hipHostFree(su.up);
  }
  if (h->host_result) (void)hipHostFree(h->host_result);
  if (h->res_log) (void)hipHostFree(h->res_log);
  if (h->res_exit) (void)hipHostFree(h->res_exit);
  if (h->pcg.host_state) (void)hipHostFree(h->pcg.host_state);
  if (h->res_out) (void)hipHostFree(h->res_out);
  if (h->io) (void)hipHostFree(h->io);
  if (h->res_trace) (void)hipHostFree(h->res_trace);
  h->res_xb.release(); h->res_epoch.release(); h->res_cost.release();
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  delete h;
  return BA_OK;
}

int ba_debug_poison(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, ensure_lds_attr(h, (const void*)k_poison_lds));
  hipLaunchKernelGGL(k_poison_lds, dim3(8 * h->ncu), dim3(1024), 160 * 1024, h->stream, 160 * 1024 / 8);
  DevBuf<double>* bufs[] = {&h->HCC, &h->bC, &h->HPP, &h->bP, &h->HPPinv, &h->W, &h->dC, &h->dP, &h->scratch, &h->Ufac, &h->ysol, &h->dinv,
                            &h->bcrD, &h->bcrU, &h->bcrF, &h->bcrP, &h->bcrQ, &h->bcrG, &h->bcrGv, &h->bcrL, &h->bcrLv, &h->denseA, &h->bigK, &h->fac,
                            &h->dUd, &h->dDd, &h->dyd, &h->dpart, &h->cams[1 - h->cur], &h->X[1 - h->cur], &h->bordF, &h->bord_partial};
  for (DevBuf<double>* b : bufs)
    if (b->p && b->n) hipLaunchKernelGGL(k_poison_doubles, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, h->stream, b->p, b->n);
  if (h->bcr_done.p && h->bcr_done.n >= 2)          // the "handed on" words of k_bcr_eliminate_fused: garbage that reads as "done" unless k_bcr_assemble clears it
    hipLaunchKernelGGL(k_poison_doubles, dim3((unsigned)((h->bcr_done.n / 2 + 255) / 256)), dim3(256), 0, h->stream, reinterpret_cast<double*>(h->bcr_done.p), h->bcr_done.n / 2);
  h->pcg.band_clean = false;
  if (h->S) hipLaunchKernelGGL(k_poison_doubles, dim3((unsigned)((reduced_doubles(h) + 255) / 256)), dim3(256), 0, h->stream, h->S, reduced_doubles(h));
  if (h->b && h->nco) hipLaunchKernelGGL(k_poison_doubles, dim3((unsigned)(((size_t)h->nco * 6 + 255) / 256)), dim3(256), 0, h->stream, h->b, (size_t)h->nco * 6);
  HIPCHECK(h, hipGetLastError());
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  h->have_params[1 - h->cur] = false;
  h->params_written(1 - h->cur);
  h->have_linearization = h->have_schur = h->have_backsub = h->have_solution = false;
  h->inv_valid = h->fac_valid = h->point_blocks_valid = h->cam_blocks_valid = false;
  return BA_OK;
}

int ba_set_option(ba_handle* h, const char* name, const char* value) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, name && value, BA_ERR_INVALID_ARG, "ba_set_option: NULL argument");
  const std::string n(name), v(value);
  auto choice = [&](std::initializer_list<const char*> names, int& out) {
    int i = 0;
    for (const char* c : names) { if (v == c) { out = i; return true; } ++i; }
    return false;
  };
  auto flag = [&](bool& out) {
    if (v == "1" || v == "on" || v == "true") { out = true; return true; }
    if (v == "0" || v == "off" || v == "false") { out = false; return true; }
    return false;
  };
  bool ok = false;
  if (n == "schur") ok = choice({"auto", "pairs", "groups", "mfma2", "mfma"}, h->opt.schur);
  else if (n == "solver") ok = choice({"auto", "bcr", "band", "dense", "lu", "bcr1", "pcg"}, h->opt.solver);
  else if (n == "pcg_tol") { char* end = nullptr; const double c = strtod(value, &end); ok = end && *end == 0 && c > 0.0 && c < 1.0; if (ok) h->opt.pcg_tol = c; }
  else if (n == "pcg_max_iter") { char* end = nullptr; const long c = strtol(value, &end, 10); ok = end && *end == 0 && c >= 0 && c <= 10000000; if (ok) h->opt.pcg_max_iter = (int)c; }
  else if (n == "pcg_batch") { char* end = nullptr; const long c = strtol(value, &end, 10); ok = end && *end == 0 && c >= 1 && c <= 100000; if (ok) h->opt.pcg_batch = (int)c; }
  else if (n == "point_kernels") { int c = 0; ok = choice({"auto", "v1"}, c); if (ok) h->opt.point_kernels_v1 = c == 1; }
  else if (n == "fuse_cost") ok = flag(h->opt.fuse_cost);
  else if (n == "fuse_invert") ok = flag(h->opt.fuse_invert);
  else if (n == "fuse_cam") ok = flag(h->opt.fuse_cam);
  else if (n == "sort_points") ok = flag(h->opt.sort_points);
  else if (n == "solve_trace") ok = flag(h->opt.solve_trace);
  else if (n == "lds_window") ok = flag(h->opt.lds_window);
  else if (n == "fused_backsolve") ok = flag(h->opt.fused_backsolve);
  else if (n == "dense_lookahead") ok = flag(h->opt.dense_lookahead);
  else if (n == "six_tile_launch") ok = flag(h->opt.six_tile_launch);
  else if (n == "bcrw_merged") ok = flag(h->opt.bcrw_merged);
  else if (n == "fused_eliminate") ok = flag(h->opt.fused_eliminate);
  else if (n == "device_lu") ok = flag(h->opt.device_lu);
  else if (n == "fast_paths") ok = flag(h->opt.fast_paths);
  else if (n == "resident") ok = flag(h->opt.resident);
  else if (n == "resident_scatter_min") { char* end = nullptr; const long c = strtol(value, &end, 10); ok = end && *end == 0 && c >= 2 && c <= 1000; if (ok) h->opt.resident_scatter_min = (int)c; }
  else if (n == "resident_fault") { char* end = nullptr; const long c = strtol(value, &end, 10); ok = end && *end == 0 && c >= -1 && c < 64; if (ok) h->opt.resident_fault = (int)c; }
  else if (n == "host_setup") ok = flag(h->opt.host_setup);
  else if (n == "camera_order") ok = choice({"auto", "off", "always"}, h->opt.camera_order);
  else if (n == "packed_store") ok = flag(h->opt.packed_store);
  else if (n == "sparse_stage") ok = flag(h->opt.sparse_stage);
  else if (n == "refine") ok = choice({"auto", "1", "0"}, h->opt.refine);
  else if (n == "refine_debug") { char* end = nullptr; const long c = strtol(value, &end, 10); ok = end && *end == 0 && c >= 0 && c <= 7; if (ok) h->opt.refine_debug = (int)c; }
  else if (n == "border") ok = flag(h->opt.border);
  else if (n == "border_side_stream") ok = flag(h->opt.border_side_stream);
  else if (n == "reuse_linearization") ok = flag(h->opt.reuse_linearization);
  else if (n == "packed_upload") ok = flag(h->opt.packed_upload);
  else if (n == "gm_chunk") { char* end = nullptr; const long c = strtol(value, &end, 10); ok = end && *end == 0 && c >= 0 && c <= 64; if (ok) h->opt.gm_chunk = (int)c; }
  else if (n == "gm_cap") { char* end = nullptr; const long c = strtol(value, &end, 10); ok = end && *end == 0 && c >= 0; if (ok) h->opt.gm_cap = (int)c; }
  else return h->fail(BA_ERR_INVALID_ARG, "ba_set_option: unknown option '%s'", name);
  if (!ok) return h->fail(BA_ERR_INVALID_ARG, "ba_set_option: bad value '%s' for option '%s'", value, name);
  h->inv_valid = h->fac_valid = false;          // a different kernel family may need different by-products
  return BA_OK;
}

int ba_set_stream(ba_handle* h, void* hip_stream) {
  if (!h) return BA_ERR_INVALID_ARG;
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  resolve_timings(h);
  if (h->own_stream) { (void)hipStreamDestroy(h->stream); h->own_stream = false; }
  if (hip_stream) {
    h->stream = (hipStream_t)hip_stream;
  } else {
    HIPCHECK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  return BA_OK;
}

int ba_synchronize(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  HIPCHECK(h, hipSetDevice(h->device));
  if (int rcj = border_join(h); rcj != BA_OK) return rcj;      // (the side stream's border kernels are the handle's work too: the main stream waits for them first)
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  return BA_OK;
}

int ba_set_sensor(ba_handle* h, int kind, const double* params, int nparams) {
  if (!h) return BA_ERR_INVALID_ARG;
  Sensor s{kind, {1, 0, 0, 1}, 1.0, 1.0, 0};
  switch (kind) {
    case BA_SENSOR_GAUSS:
      REQUIRE(h, params && nparams == 4, BA_ERR_INVALID_ARG, "ba_set_sensor: Gaussian needs 4 params (L row-major)");
      for (int i = 0; i < 4; ++i) s.L[i] = params[i];
      if (s.L[0] == 1.0 && s.L[1] == 0.0 && s.L[2] == 0.0 && s.L[3] == 1.0) s.fast |= FAST_UNIT_GAUSS;
      break;
    case BA_SENSOR_CAUCHY:
      REQUIRE(h, params && nparams == 1 && params[0] > 0, BA_ERR_INVALID_ARG, "ba_set_sensor: Cauchy needs sigma > 0");
      s.sigma = params[0];
      break;
    case BA_SENSOR_HUBER:
      REQUIRE(h, params && nparams == 1 && params[0] > 0, BA_ERR_INVALID_ARG, "ba_set_sensor: Huber needs k > 0");
      s.k = params[0];
      break;
    case BA_SENSOR_TABLE: {
      // params = [log2(rho_0), nodes per octave, n, h_0, dh/du_0, h_1, dh/du_1, ...]
      REQUIRE(h, params && nparams >= 3 + 4, BA_ERR_INVALID_ARG, "ba_set_sensor: a table needs [log2 rho_0, nodes per octave, n, 2 n values]");
      const int n = (int)params[2];
      REQUIRE(h, n >= 2 && nparams == 3 + 2 * n && params[1] > 0, BA_ERR_INVALID_ARG, "ba_set_sensor: table size does not match its header");
      for (int i = 0; i < 2 * n; ++i)
        if (!std::isfinite(params[3 + i])) return h->fail(BA_ERR_INVALID_ARG, "ba_set_sensor: table entry %d is not finite", i);
      HIPCHECK(h, hipSetDevice(h->device));
      HIPCHECK(h, hipStreamSynchronize(h->stream));       // (kernels in flight may still read the previous table)
      HIPCHECK(h, h->sensor_table.resize((size_t)2 * n));
      HIPCHECK(h, hipMemcpy(h->sensor_table.p, params + 3, (size_t)2 * n * sizeof(double), hipMemcpyHostToDevice));
      {      // (address and shape in the fields the other kinds use: ba_math.h Sensor)
        const unsigned long long bits = reinterpret_cast<unsigned long long>(h->sensor_table.p);
        std::memcpy(&s.L[0], &bits, 8);
        s.L[1] = params[0]; s.L[2] = params[1]; s.L[3] = (double)n;
      }
      break;
    }
    default:
      return h->fail(BA_ERR_INVALID_ARG, "ba_set_sensor: unknown kind %d", kind);
  }
  h->sensor = s;
  h->have_linearization = h->have_schur = h->have_backsub = false;
  return BA_OK;
}

// Small parameter sets travel through a pinned buffer of the handle: a copy from / to pageable memory is staged synchronously by
// the runtime (10 - 15 us apiece; the sliding-window caller uploads and fetches a parameter set per frame).
static hipError_t io_buffer(ba_handle* h, size_t bytes) {
  if (h->io_pending) { (void)hipStreamSynchronize(h->stream); h->io_pending = false; }      // (the last upload may still be reading it)
  if (h->io && h->io_bytes >= bytes) return hipSuccess;
  if (h->io) (void)hipHostFree(h->io);
  h->io = nullptr; h->io_bytes = 0;
  const size_t want = std::max<size_t>(bytes, 64 << 10);
  const hipError_t e = hipHostMalloc(&h->io, want, hipHostMallocDefault);
  if (e == hipSuccess) h->io_bytes = want;
  return e;
}
constexpr size_t kIoMaxBytes = 4u << 20;

int ba_set_params(ba_handle* h, int which, const double* R, const double* t, const double* X) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_set_params: call ba_set_problem first");
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_set_params: bad parameter set");
  REQUIRE(h, (h->nc == 0 || (R && t)) && (h->nt == 0 || X), BA_ERR_INVALID_ARG, "ba_set_params: NULL argument");
  HIPCHECK(h, hipSetDevice(h->device));
  if (int rcj = border_join(h); rcj != BA_OK) return rcj;      // (border kernels of an earlier ba_schur may still be reading what this call writes: side stream, ba_border.hip)
  const int p = h->phys(which);
  h->params_written(p);
  const size_t ncam = (size_t)h->nc * 12, nx = (size_t)h->nt * 3;
  const bool small = (ncam + nx) * sizeof(double) <= kIoMaxBytes;
  std::vector<double> pageable;
  double* packed;
  if (small) {
    HIPCHECK(h, io_buffer(h, (ncam + nx) * sizeof(double)));
    packed = static_cast<double*>(h->io);
    if (nx) std::memcpy(packed + ncam, X, nx * sizeof(double));
    X = packed + ncam;
  } else {
    pageable.resize(ncam);
    packed = pageable.data();
  }
  for (int i = 0; i < h->nc; ++i) {
    std::memcpy(packed + (size_t)i * 12, R + (size_t)i * 9, 9 * sizeof(double));
    std::memcpy(packed + (size_t)i * 12 + 9, t + (size_t)i * 3, 3 * sizeof(double));
  }
  if (h->nc) HIPCHECK(h, hipMemcpyAsync(h->cams[p].p, packed, ncam * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (h->nt) { const int rc = upload_rows(h, h->pperm.empty() ? nullptr : h->d_pperm.p, X, h->X[p].p, (size_t)h->nt, 3); if (rc != BA_OK) return rc; }
  if (small) h->io_pending = true;                 // (everything later is ordered behind the copies on the handle's stream)
  else HIPCHECK(h, hipStreamSynchronize(h->stream));
  h->have_params[p] = true;
  if (which == BA_PARAMS_CUR) h->have_linearization = h->have_schur = h->have_backsub = false;
  return BA_OK;
}

int ba_get_params(ba_handle* h, int which, double* R, double* t, double* X) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, which == 0 || which == 1, BA_ERR_INVALID_ARG, "ba_get_params: bad parameter set");
  const int p = h->phys(which);
  REQUIRE(h, h->have_problem && h->have_params[p], BA_ERR_STATE, "ba_get_params: parameter set is empty");
  if (h->res_out_phys == p && h->res_out) {
    // the resident loop left this very set in pinned memory (internal point order): no copy, no synchronisation
    const double* src = h->res_out;
    for (int i = 0; i < h->nc; ++i) {
      if (R) std::memcpy(R + (size_t)i * 9, src + (size_t)i * 12, 9 * sizeof(double));
      if (t) std::memcpy(t + (size_t)i * 3, src + (size_t)i * 12 + 9, 3 * sizeof(double));
    }
    if (X) {
      const double* xs = src + (size_t)h->nc * 12;
      if (h->pperm.empty()) std::memcpy(X, xs, (size_t)h->nt * 3 * sizeof(double));
      else for (int i = 0; i < h->nt; ++i) std::memcpy(X + (size_t)h->pperm[i] * 3, xs + (size_t)i * 3, 3 * sizeof(double));
    }
    return BA_OK;
  }
  HIPCHECK(h, hipSetDevice(h->device));
  const size_t ncam = (size_t)h->nc * 12, nx = X ? (size_t)h->nt * 3 : 0;
  const bool small = (ncam + nx) * sizeof(double) <= kIoMaxBytes;
  std::vector<double> pageable;
  double* packed;
  double* xdst = X;
  if (small) {
    HIPCHECK(h, io_buffer(h, (ncam + nx) * sizeof(double)));
    packed = static_cast<double*>(h->io);
    xdst = packed + ncam;
  } else {
    pageable.resize(ncam);
    packed = pageable.data();
  }
  if (h->nc) HIPCHECK(h, hipMemcpyAsync(packed, h->cams[p].p, ncam * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (X) { const int rc = download_rows(h, h->pperm.empty() ? nullptr : h->d_pperm.p, h->X[p].p, xdst, (size_t)h->nt, 3); if (rc != BA_OK) return rc; }
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  if (small && nx) std::memcpy(X, xdst, nx * sizeof(double));
  for (int i = 0; i < h->nc; ++i) {
    if (R) std::memcpy(R + (size_t)i * 9, packed + (size_t)i * 12, 9 * sizeof(double));
    if (t) std::memcpy(t + (size_t)i * 3, packed + (size_t)i * 12 + 9, 3 * sizeof(double));
  }
  return BA_OK;
}

int ba_swap_params(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem && h->have_params[1 - h->cur], BA_ERR_STATE, "ba_swap_params: trial set is empty");
  h->cur = 1 - h->cur;
  h->have_linearization = h->have_schur = h->have_backsub = false;
  return BA_OK;
}

int ba_get_camera_layout(ba_handle* h, int32_t* new_pos, int32_t* band_cameras) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_problem, BA_ERR_STATE, "ba_get_camera_layout: call ba_set_problem first");
  REQUIRE(h, new_pos || h->nco == 0, BA_ERR_INVALID_ARG, "ba_get_camera_layout: new_pos is NULL");
  for (int p = 0; p < h->nco; ++p) new_pos[p] = h->cpos_in.empty() ? p : h->cpos_in[p];
  if (band_cameras) *band_cameras = h->band_cams();
  return BA_OK;
}

int ba_comm_load(const char* librccl_path) {
  if (g_rccl.ok()) return BA_OK;
  void* lib = dlopen(librccl_path && *librccl_path ? librccl_path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) { g_create_error = std::string("ba_comm_load: ") + dlerror(); return BA_ERR_HIP; }
  g_rccl.lib = lib;
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(lib, "ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!g_rccl.ok()) { g_create_error = "ba_comm_load: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce"; return BA_ERR_HIP; }
  return BA_OK;
}

int ba_comm_unique_id(void* id128) {
  if (!id128 || !g_rccl.ok()) return BA_ERR_STATE;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return BA_ERR_HIP;
  std::memcpy(id128, &id, sizeof id);
  return BA_OK;
}

int ba_comm_init(ba_handle* h, const void* id128, int32_t rank, int32_t nranks) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, g_rccl.ok(), BA_ERR_STATE, "ba_comm_init: call ba_comm_load first");
  REQUIRE(h, id128 && nranks >= 1 && rank >= 0 && rank < nranks, BA_ERR_INVALID_ARG, "ba_comm_init: bad argument");
  REQUIRE(h, !h->comm, BA_ERR_STATE, "ba_comm_init: communicator already attached");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  HIPCHECK(h, hipSetDevice(h->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  RCCLCHECK(h, g_rccl.CommInitRank(&h->comm, nranks, id, rank));
  h->comm_ranks = nranks;
  HIPCHECK(h, h->comm_dev.resize(kCostBlocks + 2));
  HIPCHECK(h, hipMemsetAsync(h->comm_dev.p, 0, (kCostBlocks + 2) * sizeof(double), h->stream));
  HIPCHECK(h, hipHostMalloc((void**)&h->comm_host, (kCostBlocks + 2) * sizeof(double), hipHostMallocDefault));
  h->trial_result_dev = h->comm_dev.p;               // k_cost / k_backsub_groups leave the trial record here
  return BA_OK;
}

int ba_comm_destroy(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  if (h->comm) {
    (void)hipStreamSynchronize(h->stream);
    (void)g_rccl.CommDestroy(h->comm);
    h->comm = nullptr; h->comm_ranks = 0;
    if (h->trial_result_dev == h->comm_dev.p) h->trial_result_dev = nullptr;
    if (h->comm_host) { (void)hipHostFree(h->comm_host); h->comm_host = nullptr; }
    h->comm_dev.release();
  }
  return BA_OK;
}

int ba_comm_allreduce_reduced(ba_handle* h) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->comm, BA_ERR_STATE, "ba_comm_allreduce_reduced: no communicator (ba_comm_init)");
  REQUIRE(h, h->have_schur, BA_ERR_STATE, "ba_comm_allreduce_reduced: call ba_schur first");
  HIPCHECK(h, hipSetDevice(h->device));
  return comm_allreduce_reduced(h);
}

int ba_comm_allreduce_sum(ba_handle* h, double* values, int32_t n) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->comm, BA_ERR_STATE, "ba_comm_allreduce_sum: no communicator (ba_comm_init)");
  REQUIRE(h, values && n >= 1 && n <= kCostBlocks, BA_ERR_INVALID_ARG, "ba_comm_allreduce_sum: bad argument");
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, h->scratch.resize((size_t)n));
  HIPCHECK(h, hipMemcpyAsync(h->scratch.p, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RCCLCHECK(h, g_rccl.AllReduce(h->scratch.p, h->scratch.p, (size_t)n, ncclFloat64, ncclSum, h->comm, h->stream));
  HIPCHECK(h, hipMemcpyAsync(values, h->scratch.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  return BA_OK;
}

int ba_set_camera_layout(ba_handle* h, const int32_t* new_pos, int32_t nco) {
  if (!h) return BA_ERR_INVALID_ARG;
  h->forced_pos.clear();
  if (!new_pos) return BA_OK;
  REQUIRE(h, nco >= 0, BA_ERR_INVALID_ARG, "ba_set_camera_layout: negative size");
  std::vector<char> seen((size_t)nco, 0);
  for (int p = 0; p < nco; ++p) {
    if (new_pos[p] < 0 || new_pos[p] >= nco || seen[new_pos[p]]) return h->fail(BA_ERR_INVALID_ARG, "ba_set_camera_layout: new_pos is not a permutation of 0..nco-1");
    seen[new_pos[p]] = 1;
  }
  h->forced_pos.assign(new_pos, new_pos + nco);
  return BA_OK;
}

int ba_set_min_half_bandwidth(ba_handle* h, int32_t min_hb) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, min_hb >= 0, BA_ERR_INVALID_ARG, "ba_set_min_half_bandwidth: negative");
  h->min_hb = min_hb;
  return BA_OK;
}

int ba_bind_trial_result(ba_handle* h, void* result_dev) {
  if (!h) return BA_ERR_INVALID_ARG;
  static_assert(BA_TRIAL_PARTIALS == kCostBlocks, "header and kernel disagree on the number of cost partials");
  h->trial_result_dev = static_cast<double*>(result_dev);
  return BA_OK;
}

int ba_enable_timing(ba_handle* h, int on) {
  if (!h) return BA_ERR_INVALID_ARG;
  if (!on) resolve_timings(h);
  h->timing = on != 0;
  return BA_OK;
}

int ba_set_timing_mask(ba_handle* h, uint64_t kernel_id_mask) {
  if (!h) return BA_ERR_INVALID_ARG;
  h->timing_mask = kernel_id_mask;
  return BA_OK;
}

int ba_set_timing_stride(ba_handle* h, int32_t stride) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, stride >= 1, BA_ERR_INVALID_ARG, "ba_set_timing_stride: stride must be >= 1");
  h->timing_stride = stride;
  for (auto& c : h->timing_seen) c = 0;
  return BA_OK;
}

int ba_measure_copy_bandwidth(ba_handle* h, int64_t bytes, int32_t repeats, double* gbytes_per_s) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, bytes >= 4096 && repeats >= 1 && gbytes_per_s, BA_ERR_INVALID_ARG, "ba_measure_copy_bandwidth: bad argument");
  HIPCHECK(h, hipSetDevice(h->device));
  const size_t n = (size_t)bytes / sizeof(copy_vec);
  REQUIRE(h, (n + 255) / 256 < (1ull << 31), BA_ERR_INVALID_ARG, "ba_measure_copy_bandwidth: too large");
  DevBuf<copy_vec> src, dst;
  HIPCHECK(h, src.resize(n)); HIPCHECK(h, dst.resize(n));
  HIPCHECK(h, hipMemsetAsync(src.p, 0, n * sizeof(copy_vec), h->stream));
  hipEvent_t a, b;
  HIPCHECK(h, hipEventCreate(&a)); HIPCHECK(h, hipEventCreate(&b));
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, h->stream, src.p, dst.p, n);     // warm-up
  HIPCHECK(h, hipEventRecord(a, h->stream));
  for (int r = 0; r < repeats; ++r) hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, h->stream, src.p, dst.p, n);
  HIPCHECK(h, hipEventRecord(b, h->stream));
  HIPCHECK(h, hipEventSynchronize(b));
  float ms = 0.f;
  HIPCHECK(h, hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  src.release(); dst.release();
  *gbytes_per_s = 2.0 * (double)(n * sizeof(copy_vec)) * repeats / (ms * 1e-3) / 1e9;           // read + write
  return BA_OK;
}

int ba_get_timings(ba_handle* h, double* ms, int64_t* launches, int reset) {
  if (!h) return BA_ERR_INVALID_ARG;
  HIPCHECK(h, hipSetDevice(h->device));
  resolve_timings(h);
  for (int i = 0; i < BA_K_COUNT; ++i) {
    if (ms) ms[i] = h->ms[i];
    if (launches) launches[i] = h->launches[i];
    if (reset) { h->ms[i] = 0; h->launches[i] = 0; }
  }
  return BA_OK;
}

}  // extern "C"
