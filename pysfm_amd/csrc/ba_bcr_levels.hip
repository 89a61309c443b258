// ba_bcr_levels.hip - one workgroup per node and level: the Cholesky node (k_bcr_eliminate, levels wider than the chip) and the LU node for systems that are not positive definite (k_bcr_eliminate_lu); instantiated per cameras-per-node 1..11.
#include "ba_internal.h"

#define BA_BCR_TEMPLATES_ONLY 1
#include "ba_bcr.h"

using namespace ba;

namespace ba {

template <int HB>
hipError_t launch_bcr_eliminate_hb(ba_handle* h, int cnt, size_t lds, hipStream_t st, int N, int s, double* D, double* U, double* f,
                                   double* P, double* Q, double* G, int* info, double* x) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcr_eliminate<HB>); e != hipSuccess) return e;
  hipLaunchKernelGGL(k_bcr_eliminate<HB>, dim3(cnt), dim3(kBcrElimThreads), lds, st, N, s, D, U, f, P, Q, G, info, x);
  return hipSuccess;
}

hipError_t launch_bcr_eliminate(ba_handle* h, int hb, int cnt, size_t lds, hipStream_t st, int N, int s, double* D, double* U, double* f,
                                double* P, double* Q, double* G, int* info, double* x) {
#define BA_HB_CASE(K) case K: return launch_bcr_eliminate_hb<K>(h, cnt, lds, st, N, s, D, U, f, P, Q, G, info, x);
  switch (hb) {
    BA_HB_CASE(1) BA_HB_CASE(2) BA_HB_CASE(3) BA_HB_CASE(4) BA_HB_CASE(5) BA_HB_CASE(6) BA_HB_CASE(7) BA_HB_CASE(8)
    BA_HB_CASE(9) BA_HB_CASE(10) BA_HB_CASE(11)
    default: return hipErrorInvalidValue;
  }
#undef BA_HB_CASE
}

template <int HB>
hipError_t launch_bcr_lu_hb(ba_handle* h, int cnt, hipStream_t st, int N, int s, double* D, double* U, double* f, double* P, double* Q,
                            double* G, int* info, double* x) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcr_eliminate_lu<HB>); e != hipSuccess) return e;
  hipLaunchKernelGGL(k_bcr_eliminate_lu<HB>, dim3(cnt), dim3(kBcrElimThreads), bcr_lu_lds_bytes(6 * HB), st, N, s, D, U, f, P, Q, G, info, x);
  return hipSuccess;
}

hipError_t launch_bcr_lu(ba_handle* h, int hb, int cnt, hipStream_t st, int N, int s, double* D, double* U, double* f, double* P, double* Q,
                         double* G, int* info, double* x) {
#define BA_HB_CASE(K) case K: return launch_bcr_lu_hb<K>(h, cnt, st, N, s, D, U, f, P, Q, G, info, x);
  switch (hb) {
    BA_HB_CASE(1) BA_HB_CASE(2) BA_HB_CASE(3) BA_HB_CASE(4) BA_HB_CASE(5) BA_HB_CASE(6) BA_HB_CASE(7) BA_HB_CASE(8)
    BA_HB_CASE(9) BA_HB_CASE(10) BA_HB_CASE(11)
    default: return hipErrorInvalidValue;
  }
#undef BA_HB_CASE
}

}  // namespace ba
