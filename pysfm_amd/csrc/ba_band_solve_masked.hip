// ba_band_solve_masked.hip - the instances of k_band_solve for systems with masked camera parameters (a unit of their own: 22 more kernels).
#include "ba_internal.h"

#define BA_BAND_TEMPLATES_ONLY 1
#include "ba_band.h"

using namespace ba;

namespace ba {

// k_band_solve is instantiated per block half-bandwidth (compile-time unrolling); the instances with masked camera parameters
namespace {      // (a template of the same name lives in the other band-solve unit: internal linkage)
template <int HB>
hipError_t launch_band_solve_hb(ba_handle* h, size_t lds, hipStream_t stream, int nco, int ch, const double* S, const double* b,
                                const unsigned char* mask, double* U, double* y, double* dinv, double* x, int* info) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_band_solve<HB, true>); e != hipSuccess) return e;
  hipLaunchKernelGGL((k_band_solve<HB, true>), dim3(1), dim3(kSolveThreads), lds, stream, nco, ch, S, b, mask, U, y, dinv, x, info);
  return hipGetLastError();
}
}  // namespace

hipError_t launch_band_solve_masked(ba_handle* h, int hb, size_t lds, hipStream_t stream, int nco, int ch, const double* S, const double* b,
                                    const unsigned char* mask, double* U, double* y, double* dinv, double* x, int* info) {
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
