// lds_probe: LDS read throughput of one CU (one workgroup): bytes per clock64 cycle until the LAST wavefront finishes.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k_probe(double* out, int n, long long* cyc) {
  __shared__ long long tend[16];
  extern __shared__ double sm[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) sm[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int idx = MODE == 0 ? lane : MODE == 1 ? 2 * lane : MODE == 2 ? (lane & 15) : MODE == 3 ? lane * 55 % 4096 : lane;
  idx += wave * 64;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    const unsigned addr = (unsigned)(size_t)(sm + idx + (i & 7) * 128);   // LDS byte address
    if (MODE == 1) {
      double2 v0, v1, v2, v3;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:8192\n ds_read_b128 %2, %4 offset:16384\n ds_read_b128 %3, %4 offset:24576\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr));
      a0 += v0.x + v0.y; a1 += v1.x + v1.y; a2 += v2.x + v2.y; a3 += v3.x + v3.y;
    } else {
      double v0, v1, v2, v3;
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8192\n ds_read_b64 %2, %4 offset:16384\n ds_read_b64 %3, %4 offset:24576\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr));
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
  }
  long long t1 = clock64();
  if (lane == 0) tend[wave] = t1 - t0;
  __syncthreads();
  out[threadIdx.x] = a0 + a1 + a2 + a3;
  if (threadIdx.x == 0) {
    long long m = 0;
    for (int w = 0; w < (int)blockDim.x / 64; ++w) m = tend[w] > m ? tend[w] : m;
    cyc[MODE] = m;
  }
}
int main() {
  double* d; long long* c; hipMalloc(&d, 1 << 20); hipMalloc(&c, 128);
  const int n = 20000;
  long long h[8];
  const char* names[4] = {"ds_read_b64 consecutive", "ds_read_b128 consecutive", "ds_read_b64 16-periodic (broadcast)", "ds_read_b64 stride 55"};
  const int bytes[4] = {8, 16, 8, 8};
  hipFuncSetAttribute((const void*)k_probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k_probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k_probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k_probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int threads : {64, 256, 1024}) {
    hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(threads), 65536, 0, d, n, c);
    hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(threads), 65536, 0, d, n, c);
    hipLaunchKernelGGL(k_probe<2>, dim3(1), dim3(threads), 65536, 0, d, n, c);
    hipLaunchKernelGGL(k_probe<3>, dim3(1), dim3(threads), 65536, 0, d, n, c);
    hipDeviceSynchronize();
    hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
    for (int m = 0; m < 4; ++m)
      printf("waves=%2d %-36s %.2f cycles per wave-instr (CU), %.1f B/cycle delivered to lanes\n", threads / 64, names[m],
             h[m] / (4.0 * n * (threads / 64)), 4.0 * n * threads * bytes[m] / h[m]);
  }
  return 0;
}
