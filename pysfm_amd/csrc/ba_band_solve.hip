// ba_band_solve.hip - the reduced camera system as one workgroup sees it: k_band_solve (instantiated per half-bandwidth 0..21, with and without masked parameters) and ba_flatten_reduced.
#include "ba_internal.h"

#include "ba_band.h"

using namespace ba;

namespace ba {

// k_band_solve is instantiated per block half-bandwidth (compile-time unrolling); the instances without masked camera parameters
namespace {      // (a template of the same name lives in the other band-solve unit: internal linkage)
template <int HB>
hipError_t launch_band_solve_hb(ba_handle* h, size_t lds, hipStream_t stream, int nco, int ch, const double* S, const double* b,
                                const unsigned char* mask, double* U, double* y, double* dinv, double* x, int* info) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_band_solve<HB, false>); e != hipSuccess) return e;
  hipLaunchKernelGGL((k_band_solve<HB, false>), dim3(1), dim3(kSolveThreads), lds, stream, nco, ch, S, b, mask, U, y, dinv, x, info);
  return hipGetLastError();
}
}  // namespace

hipError_t launch_band_solve(ba_handle* h, int hb, size_t lds, hipStream_t stream, int nco, int ch, const double* S, const double* b,
                             const unsigned char* mask, double* U, double* y, double* dinv, double* x, int* info) {
  if (mask) return launch_band_solve_masked(h, hb, lds, stream, nco, ch, S, b, mask, U, y, dinv, x, info);      // (ba_band_solve_masked.hip)
#define BA_HB_CASE(N) case N: return launch_band_solve_hb<N>(h, lds, stream, nco, ch, S, b, mask, U, y, dinv, x, info);
  switch (hb) {
    BA_HB_CASE(0) BA_HB_CASE(1) BA_HB_CASE(2) BA_HB_CASE(3) BA_HB_CASE(4) BA_HB_CASE(5) BA_HB_CASE(6) BA_HB_CASE(7)
    BA_HB_CASE(8) BA_HB_CASE(9) BA_HB_CASE(10) BA_HB_CASE(11) BA_HB_CASE(12) BA_HB_CASE(13) BA_HB_CASE(14)
    BA_HB_CASE(15) BA_HB_CASE(16) BA_HB_CASE(17) BA_HB_CASE(18) BA_HB_CASE(19) BA_HB_CASE(20) BA_HB_CASE(21)
    default: return hipErrorInvalidValue;
  }
#undef BA_HB_CASE
}

}  // namespace ba

extern "C" {

int ba_flatten_reduced(ba_handle* h, const int32_t* keep, int32_t nkeep, void* A_dev, void* rhs_dev) {
  if (!h) return BA_ERR_INVALID_ARG;
  REQUIRE(h, h->have_schur, BA_ERR_STATE, "ba_flatten_reduced: call ba_schur first");
  REQUIRE(h, !h->pcg.packed, BA_ERR_STATE, "ba_flatten_reduced: this problem's reduced system is stored as the list of its blocks (option packed_store = 0 before ba_set_problem keeps the band)");
  REQUIRE(h, nkeep >= 0 && (nkeep == 0 || (keep && A_dev && rhs_dev)), BA_ERR_INVALID_ARG, "ba_flatten_reduced: NULL argument");
  for (int i = 0; i < nkeep; ++i)
    if (keep[i] < 0 || keep[i] >= h->nco * 6) return h->fail(BA_ERR_INVALID_ARG, "ba_flatten_reduced: keep[%d] out of range", i);
  if (nkeep == 0) return BA_OK;
  HIPCHECK(h, hipSetDevice(h->device));
  HIPCHECK(h, h->keep.resize(nkeep));
  std::vector<int> ikeep;
  if (!h->cpos_in.empty()) {                     // flat parameter indices by the caller's camera positions -> the internal ones
    ikeep.resize(nkeep);
    for (int i = 0; i < nkeep; ++i) ikeep[i] = h->cpos_in[keep[i] / 6] * 6 + keep[i] % 6;
    keep = ikeep.data();
  }
  HIPCHECK(h, hipMemcpyAsync(h->keep.p, keep, nkeep * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (h->nbc > 0) {                               // band + border (ba_border.h)
    const int rc = border_flatten(h, nkeep, (double*)A_dev, (double*)rhs_dev);
    if (rc != BA_OK) return rc;
  } else {
    ScopedTimer tm(h, BA_K_FLATTEN);
    hipLaunchKernelGGL(k_flatten, dim3(blocks_for((long long)nkeep * nkeep)), dim3(kBlock), 0, h->stream, h->nco, h->hb,
                       nkeep, h->keep.p, h->S, h->b, (double*)A_dev, (double*)rhs_dev);
  }
  HIPCHECK(h, hipGetLastError());
  HIPCHECK(h, hipStreamSynchronize(h->stream));   // `keep` is caller memory
  return BA_OK;
}

}  // extern "C"
