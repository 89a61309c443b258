// copy_probe.hip - which streaming-copy kernel reaches the guide's ~6.3 TB/s (read + write) on this box?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/copy_probe tools/copy_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const vf4* __restrict__ src, vf4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
    vf4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 256;
      if (i < n) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 256;
      if (i < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
    }
  }
}

// one element per thread, no loop: the grid covers the array
template <bool NT>
__global__ __launch_bounds__(256) void k_copy_flat(const vf4* __restrict__ src, vf4* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i); else dst[i] = src[i]; }
}

template <typename F>
double time_it(F launch, size_t bytes, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(); launch();
  hipEventRecord(a, 0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return 2.0 * bytes * reps / (ms * 1e-3) / 1e9;
}

int main() {
  for (size_t mb : {256, 1024, 4096}) {
    const size_t bytes = mb << 20, n = bytes / sizeof(vf4);
    vf4 *src, *dst;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
    hipMemset(src, 1, bytes);
    printf("%zu MiB:", mb);
    for (int g : {2048, 8192, 32768}) {
      printf("  [grid %d] u1 %.0f", g, time_it([&] { k_copy<1, false><<<g, 256>>>(src, dst, n); }, bytes, 10));
      printf(" u4 %.0f", time_it([&] { k_copy<4, false><<<g, 256>>>(src, dst, n); }, bytes, 10));
      printf(" u4nt %.0f", time_it([&] { k_copy<4, true><<<g, 256>>>(src, dst, n); }, bytes, 10));
      printf(" u8nt %.0f", time_it([&] { k_copy<8, true><<<g, 256>>>(src, dst, n); }, bytes, 10));
    }
    printf("  flat %.0f flat-nt %.0f", time_it([&] { k_copy_flat<false><<<(unsigned)((n + 255) / 256), 256>>>(src, dst, n); }, bytes, 10),
           time_it([&] { k_copy_flat<true><<<(unsigned)((n + 255) / 256), 256>>>(src, dst, n); }, bytes, 10));
    float ms; hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(a, 0);
    for (int r = 0; r < 10; ++r) hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    printf("  hipMemcpyD2D %.0f GB/s\n", 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
    hipFree(src); hipFree(dst);
  }
  return 0;
}
