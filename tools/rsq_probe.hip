// rsq_probe.hip - how accurate is v_rsq_f64 / v_rcp_f64 on gfx950, and after one / two Newton steps?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double y0 = __builtin_amdgcn_rsq(v);
  double hx = 0.5 * v;
  double y1 = y0 * fma(-hx * y0, y0, 1.5);
  double y2 = y1 * fma(-hx * y1, y1, 1.5);
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = fma(fma(-v, r0, 1.0), r0, r0);
  { const double e = fma(-(v * y0), y0, 1.0); y2 = (i & 1) ? y2 : fma(y0 * e, fma(e, 0.375, 0.5), y0); }   // even entries: one cubic step instead
  o[5 * i] = y0; o[5 * i + 1] = y1; o[5 * i + 2] = y2; o[5 * i + 3] = r0; o[5 * i + 4] = r1;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(5 * n);
  for (int i = 0; i < n; ++i) x[i] = ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 200) - 100);
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 5 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(o.data(), dout, 5 * n * 8, hipMemcpyDeviceToHost);
  double e[5] = {0, 0, 0, 0, 0}, ecub = 0;
  for (int i = 0; i < n; ++i) {
    long double t = 1.0L / sqrtl((long double)x[i]), r = 1.0L / (long double)x[i];
    for (int q = 0; q < 2; ++q) e[q] = fmax(e[q], (double)fabsl((o[5 * i + q] - t) / t));
    if (i & 1) e[2] = fmax(e[2], (double)fabsl((o[5 * i + 2] - t) / t));
    else ecub = fmax(ecub, (double)fabsl((o[5 * i + 2] - t) / t));
    for (int q = 3; q < 5; ++q) e[q] = fmax(e[q], (double)fabsl((o[5 * i + q] - r) / r));
  }
  printf("max relative error: rsq seed %.3e (2^%.1f), 1 Newton %.3e, 2 Newton %.3e | rcp seed %.3e (2^%.1f), 1 Newton %.3e\n", e[0], log2(e[0]), e[1], e[2], e[3],
         log2(e[3]), e[4]);
  printf("one cubic step y (1 + e/2 + 3 e^2 / 8), e = 1 - x y^2: max relative error %.3e\n", ecub);
  return 0;
}
