// gridsync_probe: cost of a grid-wide barrier (cooperative launch) vs a dependent kernel launch.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ void k_sync(int n, double* out) {
  cg::grid_group grid = cg::this_grid();
  double a = threadIdx.x;
  for (int i = 0; i < n; ++i) { a = a * 1.0000001 + 1.0; grid.sync(); }
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}
// hand-rolled barrier: one arrival counter, monotonically increasing target
__global__ void k_sync2(int n, double* out, unsigned int* ctr) {
  double a = threadIdx.x;
  unsigned int target = 0;
  for (int i = 0; i < n; ++i) {
    a = a * 1.0000001 + 1.0;
    __syncthreads();
    target += gridDim.x;
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(ctr, 1u);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __threadfence();
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}
__global__ void k_empty(double* out) { if (threadIdx.x == 0) out[blockIdx.x] += 1.0; }
int main() {
  double* d; unsigned int* c; hipMalloc(&d, 1 << 16); hipMalloc(&c, 64);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {1, 8, 56, 256}) {
    int n = 200; float ms;
    void* args[] = {&n, &d};
    hipLaunchCooperativeKernel((const void*)k_sync, dim3(blocks), dim3(1024), args, 0, st);   // warm
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    hipError_t e = hipLaunchCooperativeKernel((const void*)k_sync, dim3(blocks), dim3(1024), args, 0, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("blocks=%3d  cg grid.sync: %.2f us each (%s)", blocks, ms * 1e3 / n, hipGetErrorString(e));
    hipMemsetAsync(c, 0, 4, st);
    hipLaunchKernelGGL(k_sync2, dim3(blocks), dim3(1024), 0, st, n, d, c); hipStreamSynchronize(st);
    hipMemsetAsync(c, 0, 4, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(k_sync2, dim3(blocks), dim3(1024), 0, st, n, d, c);
    hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf(" | hand-rolled: %.2f us each", ms * 1e3 / n);
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(1024), 0, st, d);
    hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf(" | dependent launch: %.2f us each\n", ms * 1e3 / n);
  }
  return 0;
}
