// fma_probe: fp64 throughput of one CU (one workgroup), wall cycles until the LAST wavefront finishes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k_probe(double* out, int n, long long* cyc) {
  __shared__ long long tend[16];
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = out[threadIdx.x + i];
  double b = out[threadIdx.x + 9], c = out[threadIdx.x + 10];
  double4_t acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = double4_t{a[0], a[1], a[2], a[3]};
  __syncthreads();
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = fma(b, c, a[u]);
    }
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[u]) : "v"(b), "v"(c));
    }
    if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc[u & 3], 0, 0, 0);
    }
    if (MODE == 4) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc[0], 0, 0, 0);
    }
    if (MODE == 5) {      // two dependent MFMAs, then the result is consumed by a VALU op (as in a tile task)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(c, b, acc[u], 0, 0, 0);
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a[u], 0, 0, 0);
    }
  }
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) tend[threadIdx.x >> 6] = t1 - t0;
  __syncthreads();
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    long long m = 0, mn = 1ll << 60;
    for (int w = 0; w < (int)blockDim.x / 64; ++w) { m = tend[w] > m ? tend[w] : m; mn = tend[w] < mn ? tend[w] : mn; }
    cyc[2 * MODE] = m; cyc[2 * MODE + 1] = mn;
  }
}
int main() {
  double* d; long long* c; hipMalloc(&d, 1 << 20); hipMalloc(&c, 128); hipMemset(d, 0, 1 << 20);
  const int n = 20000;
  long long h[12];
  const char* names[6] = {"v_fma_f64", "v_fmac_f64_dpp", "mfma_f64_16x16x4", "mfma_f64_4x4x4", "mfma16 dependent chain", "mfma16 dependent pairs"};
  for (int threads : {64, 256, 512, 1024}) {
    hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(threads), 0, 0, d, n, c);
    hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(threads), 0, 0, d, n, c);
    hipLaunchKernelGGL(k_probe<2>, dim3(1), dim3(threads), 0, 0, d, n, c);
    hipLaunchKernelGGL(k_probe<3>, dim3(1), dim3(threads), 0, 0, d, n, c);
    hipLaunchKernelGGL(k_probe<4>, dim3(1), dim3(threads), 0, 0, d, n, c);
    hipLaunchKernelGGL(k_probe<5>, dim3(1), dim3(threads), 0, 0, d, n, c);
    hipDeviceSynchronize();
    hipMemcpy(h, c, 96, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    for (int m = 0; m < 6; ++m)
      printf("waves=%2d %-18s slowest wave %.2f cycles/instr, fastest %.2f | CU throughput: one wave-instr per %.2f cycles\n", waves, names[m],
             h[2 * m] / (8.0 * n), h[2 * m + 1] / (8.0 * n), h[2 * m] / (8.0 * n * waves));
  }
  return 0;
}
