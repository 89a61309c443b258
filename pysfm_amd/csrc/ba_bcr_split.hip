// ba_bcr_split.hip - the node of the cyclic reduction spread over three workgroups: one launch per level (k_bcr_eliminate_split) or all levels and the back-substitution in one launch (k_bcr_eliminate_fused); instantiated per cameras-per-node 1..13.
#include "ba_internal.h"

#define BA_BCR_TEMPLATES_ONLY 1
#include "ba_bcr.h"

using namespace ba;

namespace ba {

template <int HB>
hipError_t launch_bcr_split_hb(ba_handle* h, int cnt, hipStream_t st, int N, int s, double* D, const double* U, double* f,
                               double* P, double* Q, double* G, double* gv, int* info, double* x) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcr_eliminate_split<HB>); e != hipSuccess) return e;
  hipLaunchKernelGGL(k_bcr_eliminate_split<HB>, dim3(cnt, 3), dim3(kBcrElimThreads), bcr_split_lds_bytes(6 * HB), st, N, s, D, U, f, P,
                     Q, G, gv, info, x);
  return hipSuccess;
}

template <int HB>
hipError_t launch_bcr_fused_hb(ba_handle* h, int nwork, hipStream_t st, int N, int s_first, double* D, const double* U, double* f,
                               double* P, double* Q, double* G, double* gv, int* info, double* x, const int* work, int* done) {
  if (hipError_t e = ensure_lds_attr(h, (const void*)k_bcr_eliminate_fused<HB>); e != hipSuccess) return e;
  long long* trace = nullptr;
#ifdef BA_BCR_PROFILE
  if (h->opt.solve_trace) {
    if (hipError_t e = h->bcr_trace.resize((size_t)8 * nwork); e != hipSuccess) return e;
    trace = h->bcr_trace.p;
    h->bcr_trace_n = nwork;
  }
#endif
  hipLaunchKernelGGL(k_bcr_eliminate_fused<HB>, dim3(nwork), dim3(kBcrElimThreads), bcr_split_lds_bytes(6 * HB), st, N, s_first, D, U, f,
                     P, Q, G, gv, info, x, work, done, trace);
  return hipSuccess;
}

hipError_t launch_bcr_fused(ba_handle* h, int hb, int nwork, hipStream_t st, int N, int s_first, double* D, const double* U, double* f,
                            double* P, double* Q, double* G, double* gv, int* info, double* x, const int* work, int* done) {
#define BA_HB_CASE(K) case K: return launch_bcr_fused_hb<K>(h, nwork, st, N, s_first, D, U, f, P, Q, G, gv, info, x, work, done);
  switch (hb) {
    BA_HB_CASE(1) BA_HB_CASE(2) BA_HB_CASE(3) BA_HB_CASE(4) BA_HB_CASE(5) BA_HB_CASE(6) BA_HB_CASE(7) BA_HB_CASE(8)
    BA_HB_CASE(9) BA_HB_CASE(10) BA_HB_CASE(11) BA_HB_CASE(12) BA_HB_CASE(13)
    default: return hipErrorInvalidValue;
  }
#undef BA_HB_CASE
}

hipError_t launch_bcr_split(ba_handle* h, int hb, int cnt, hipStream_t st, int N, int s, double* D, const double* U, double* f,
                            double* P, double* Q, double* G, double* gv, int* info, double* x) {
#define BA_HB_CASE(K) case K: return launch_bcr_split_hb<K>(h, cnt, st, N, s, D, U, f, P, Q, G, gv, info, x);
  switch (hb) {
    BA_HB_CASE(1) BA_HB_CASE(2) BA_HB_CASE(3) BA_HB_CASE(4) BA_HB_CASE(5) BA_HB_CASE(6) BA_HB_CASE(7) BA_HB_CASE(8)
    BA_HB_CASE(9) BA_HB_CASE(10) BA_HB_CASE(11) BA_HB_CASE(12) BA_HB_CASE(13)
    default: return hipErrorInvalidValue;
  }
#undef BA_HB_CASE
}

}  // namespace ba
