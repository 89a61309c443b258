// ba_sort.hip - the two device primitives ba_set_problem needs that are not worth writing by hand: a stable LSD radix sort of
// (64-bit key, 32-bit value) pairs and an exclusive prefix sum, both from rocPRIM (header-only, part of ROCm).  A unit of its
// own: rocPRIM's dispatch instantiates many kernel configurations (20 s of compile time that nothing else should wait for).
#include "ba_internal.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace ba {

// keys_out / values_out = the pairs ordered by the low `end_bit` bits of the key, equal keys in input order (stable)
hipError_t sort_pairs_u64(ba_handle* h, const unsigned long long* keys_in, unsigned long long* keys_out, const int* values_in,
                          int* values_out, size_t n, int end_bit) {
  if (n == 0) return hipSuccess;
  end_bit = std::max(1, std::min(64, end_bit));
  size_t bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, values_in, values_out, n, 0u, (unsigned)end_bit, h->stream);
  if (e != hipSuccess) return e;
  if ((e = h->su.tmp.resize(std::max<size_t>(bytes, 16))) != hipSuccess) return e;
  return rocprim::radix_sort_pairs(h->su.tmp.p, bytes, keys_in, keys_out, values_in, values_out, n, 0u, (unsigned)end_bit, h->stream);
}

// out[i] = in[0] + ... + in[i - 1], i < n
hipError_t exclusive_scan_i32(ba_handle* h, const int* in, int* out, size_t n) {
  if (n == 0) return hipSuccess;
  size_t bytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, 0, n, rocprim::plus<int>(), h->stream);
  if (e != hipSuccess) return e;
  if ((e = h->su.tmp.resize(std::max<size_t>(bytes, 16))) != hipSuccess) return e;
  return rocprim::exclusive_scan(h->su.tmp.p, bytes, in, out, 0, n, rocprim::plus<int>(), h->stream);
}

}  // namespace ba
