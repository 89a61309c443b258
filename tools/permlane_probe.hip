// permlane_probe.hip - what v_permlane32_swap / v_permlane16_swap do on gfx950 (printed: every 8th lane of new vdst, new src0)
#include <hip/hip_runtime.h>
__global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 1024); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) { for (int i = 0; i < 64; i += 8) printf("%4u", h[64 * r + i]); printf("\n"); }
  return 0;
}
