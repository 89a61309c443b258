// mfma4_probe.hip - operand / result layout of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction),
// found by feeding one-hot operands: which (block, i, k) does lane l's A value stand for, which (block, k, j) its B value,
// which (block, i, j) its result?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
  double *da, *db, *dd, ha[64], hb[64], hd[64];
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  // for every pair (la, lb): A one-hot at lane la, B one-hot at lane lb -> which result lanes light up?
  int hit[64][64];
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      for (int i = 0; i < 64; ++i) { ha[i] = i == la; hb[i] = i == lb; }
      hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
      k<<<1, 64>>>(da, db, dd);
      hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
      hit[la][lb] = -1;
      for (int i = 0; i < 64; ++i) if (hd[i] != 0.0) hit[la][lb] = hit[la][lb] == -1 ? i : -2;
    }
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d meets B lanes:", la);
    for (int lb = 0; lb < 64; ++lb) if (hit[la][lb] != -1) printf(" %d->D%d", lb, hit[la][lb]);
    printf("\n");
  }
  return 0;
}
