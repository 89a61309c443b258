// overlap_probe: do fp64 MFMA and fp64 vector FMA of two wavefronts on the SAME SIMD overlap?
// wave 0 runs MFMAs, wave 4 (same SIMD under round-robin placement) runs v_fma_f64; each alone, then together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k_probe(double* out, int n, long long* cyc, int mode) {   // mode bit0: wave 0 does MFMA, bit1: wave 4 does FMA
  __shared__ long long tend[8];
  const int wave = threadIdx.x >> 6;
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = out[threadIdx.x + i];
  double b = out[threadIdx.x + 9], c = out[threadIdx.x + 10];
  double4_t acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = double4_t{a[0], a[1], a[2], a[3]};
  __syncthreads();
  long long t0 = clock64();
  if (wave == 0 && (mode & 1)) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc[u & 3], 0, 0, 0);
    }
  }
  if (wave == 4 && (mode & 2)) {
    for (int i = 0; i < 8 * n; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = fma(b, c, a[u]);
    }
  }
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) tend[wave] = t1 - t0;
  __syncthreads();
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[0] = tend[0]; cyc[1] = tend[4]; }
}
int main() {
  double* d; long long* c; hipMalloc(&d, 1 << 20); hipMalloc(&c, 128); hipMemset(d, 0, 1 << 20);
  const int n = 5000;
  long long h[2];
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(512), 0, 0, d, n, c, mode);
    hipDeviceSynchronize();
    hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    printf("mode %d (1 = MFMA on wave 0, 2 = v_fma_f64 on wave 4, 3 = both): wave 0 %lld cycles (%.1f per MFMA), wave 4 %lld cycles (%.2f per FMA)\n", mode, h[0],
           h[0] / (8.0 * n), h[1], h[1] / (64.0 * n));
  }
  return 0;
}
