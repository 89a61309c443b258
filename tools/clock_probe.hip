// clock_probe: effective shader clock and dependent-op latencies under light / heavy load.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fma_chain(double* out, int n, long long* cyc) {
  double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
  long long t0 = clock64(); long long w0 = wall_clock64();
  for (int i = 0; i < n; ++i) { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }
  long long t1 = clock64(); long long w1 = wall_clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = a;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
__global__ void k_lds_chain(double* out, int n, long long* cyc) {
  __shared__ double sm[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = (double)((i * 7 + 1) % 1024);
  __syncthreads();
  int idx = threadIdx.x;
  long long t0 = clock64(); long long w0 = wall_clock64();
  for (int i = 0; i < n; ++i) { idx = (int)sm[idx]; idx = (int)sm[idx]; idx = (int)sm[idx]; idx = (int)sm[idx]; }
  long long t1 = clock64(); long long w1 = wall_clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = idx;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[2] = t1 - t0; cyc[3] = w1 - w0; }
}
__global__ void k_barrier_chain(double* out, int n, long long* cyc) {
  long long t0 = clock64(); long long w0 = wall_clock64();
  for (int i = 0; i < n; ++i) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
  long long t1 = clock64(); long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[4] = t1 - t0; cyc[5] = w1 - w0; }
}
int main() {
  double* d; long long* c; hipMalloc(&d, 1 << 24); hipMalloc(&c, 64); hipMemset(d, 0, 1 << 24);
  long long h[8];
  for (int blocks : {1, 256, 2048}) {
    for (int threads : {64, 512}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int n = 100000;
      float ms[3];
      hipEventRecord(e0); hipLaunchKernelGGL(k_fma_chain, dim3(blocks), dim3(threads), 0, 0, d, n, c); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
      hipEventRecord(e0); hipLaunchKernelGGL(k_lds_chain, dim3(blocks), dim3(threads), 0, 0, d, n, c); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
      hipEventRecord(e0); hipLaunchKernelGGL(k_barrier_chain, dim3(blocks), dim3(threads), 0, 0, d, n, c); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[2], e0, e1);
      hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
      printf("blocks=%4d threads=%3d | fma: %.3f ms, %.1f clk/op (clock64), %.2f ns/op, clock64 rate %.0f MHz | lds: %.3f ms %.1f clk/op %.2f ns/op | barrier: %.3f ms %.1f clk/op %.2f ns/op\n",
             blocks, threads, ms[0], h[0] / (4.0 * n), ms[0] * 1e6 / (4.0 * n), h[0] / (h[1] * 10e-9) / 1e6,
             ms[1], h[2] / (4.0 * n), ms[1] * 1e6 / (4.0 * n), ms[2], h[4] / (4.0 * n), ms[2] * 1e6 / (4.0 * n));
    }
  }
  return 0;
}
