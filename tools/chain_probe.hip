// chain_probe.hip - what does one 12 x 12 diagonal block of the node kernels cost on ONE wavefront, and what are the
// dependent-issue latencies of the instructions on its pivot chain (v_rsq_f64, v_mul_f64 / v_fma_f64, DPP forms)?
#include "../pysfm_amd/csrc/ba_bcr.h"
#include <cstdio>
#include <vector>
using namespace ba;

__global__ __launch_bounds__(1024) void k_probe(const double* A, long long* out, int reps) {
  __shared__ double G[16 * 17], dinv[16], Li[192], Idt[160];
  __shared__ int bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long tb = 0, tp = 0;
  for (int r = 0; r < reps; ++r) {
    if (tid < 192) Li[tid] = 0.0;
    bcr_identity_table(Idt, tid);
    for (int e = tid; e < 144; e += blockDim.x) G[(e / 12) * 17 + e % 12] = A[e];
    __syncthreads();
    if (wave == 0) {
      const long long t0 = clock64();
      bcr_diag_block<12, false>(G, 17, dinv, &bad, 0, lane, Li, Idt);
      lds_wave_sync();
      const long long dt = clock64() - t0;
      tb += dt;
      if (r == 0 && lane == 0) out[8] = dt;
      // the pivots alone, on registers
      double cl[12], di = 0.0;
      for (int p = 0; p < 12; ++p) cl[p] = A[(lane % 12) * 12 + p] + (p == lane % 12 ? 1.0 : 0.0);
      const long long t1 = clock64();
      bcr_diag_pivots<12, false>(std::make_integer_sequence<int, 12>{}, cl, lane & 15, di);
      asm volatile("" ::"v"(cl[11]), "v"(di));
      tp += clock64() - t1;
      if (cl[3] == 1234.5) out[9] = 1;
    }
    __syncthreads();
  }
  if (tid == 0) { out[0] = tb / reps; out[1] = tp / reps; }
  // dependent chains, 256 long
  if (wave == 0) {
    double x = 1.0 + 1e-9 * lane, y = 0.999999, z;
    long long t0 = clock64();
#pragma unroll
    for (int k = 0; k < 256; ++k) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    long long t1 = clock64();
    if (lane == 0) out[2] = (t1 - t0);
    x = 1.0 + 1e-9 * lane;
    t0 = clock64();
#pragma unroll
    for (int k = 0; k < 256; ++k) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
    t1 = clock64();
    if (lane == 0) out[3] = (t1 - t0);
    x = 1.0 + 1e-9 * lane;
    t0 = clock64();
#pragma unroll
    for (int k = 0; k < 256; ++k) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_mov_b64_dpp %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(x), "+v"(y));
    t1 = clock64();
    if (lane == 0) out[4] = (t1 - t0) / 2;
    x = 1.0 + 1e-9 * lane; z = 0.0;
    t0 = clock64();
#pragma unroll
    for (int k = 0; k < 256; ++k) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
    t1 = clock64();
    if (lane == 0) out[5] = (t1 - t0);
    // independent v_fma_f64 (issue rate of one wavefront)
    double a0 = x, a1 = y, a2 = 1.5, a3 = 2.5;
    t0 = clock64();
#pragma unroll
    for (int k = 0; k < 64; ++k)
      asm volatile("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));
    t1 = clock64();
    if (lane == 0) out[6] = (t1 - t0);
    // rsq followed by a dependent mul (trans -> valu forwarding)
    x = 1.0 + 1e-9 * lane;
    t0 = clock64();
#pragma unroll
    for (int k = 0; k < 128; ++k) asm volatile("v_rsq_f64 %0, %0\n\ts_nop 0\n\tv_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));
    t1 = clock64();
    if (lane == 0) out[7] = (t1 - t0);
    // v_mfma_f64_16x16x4_f64: one dependent chain, two and four interleaved chains (cycles per MFMA)
    {
      typedef double acc4 __attribute__((ext_vector_type(4)));
      acc4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      t0 = clock64();
#pragma unroll
      for (int k = 0; k < 64; ++k) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
      asm volatile("" ::"v"(c0));
      t1 = clock64();
      if (lane == 0) out[10] = t1 - t0;
      t0 = clock64();
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, c1, 0, 0, 0);
      }
      asm volatile("" ::"v"(c0), "v"(c1));
      t1 = clock64();
      if (lane == 0) out[11] = t1 - t0;
      t0 = clock64();
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c3, 0, 0, 0);
      }
      asm volatile("" ::"v"(c0), "v"(c1), "v"(c2), "v"(c3));
      t1 = clock64();
      if (lane == 0) out[12] = t1 - t0;
      if (c0[0] + c1[1] + c2[2] + c3[3] == 1234.5) out[9] = 3;
    }
    if (a0 + a1 + a2 + a3 + x + z == 1234.5) out[9] = 2;
  }
}

int main(int argc, char** argv) {
  std::vector<double> A(144);
  for (int i = 0; i < 12; ++i)
    for (int j = 0; j < 12; ++j) A[i * 12 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
  double* dA; long long* dout;
  hipMalloc(&dA, 144 * 8); hipMalloc(&dout, 16 * 8);
  hipMemcpy(dA, A.data(), 144 * 8, hipMemcpyHostToDevice);
  for (int threads : {64, 1024}) {
    long long o[16];
    k_probe<<<1, threads>>>(dA, dout, 50);
    k_probe<<<1, threads>>>(dA, dout, 50);
    hipMemcpy(o, dout, 16 * 8, hipMemcpyDeviceToHost);
    printf("workgroup of %4d: bcr_diag_block<12> %lld cycles (LDS loads + 12 pivots + stores of L, 1/diag, L^-1); the 12 pivots alone %lld\n", threads, o[0], o[1]);
    printf("   first call in the kernel (cold instruction cache): %lld cycles\n", o[8]);
    printf("   dependent chains, cycles per instruction: v_fma_f64 %.1f, v_rsq_f64 %.1f, v_mov_b64_dpp (+ s_nop 1) %.1f, v_fmac_f64_dpp (+ s_nop 1) %.1f; "
           "independent v_fma_f64 %.1f; v_rsq_f64 -> v_mul_f64 pair %.1f\n",
           o[2] / 256.0, o[3] / 256.0, o[4] / 256.0, o[5] / 256.0, o[6] / 256.0, o[7] / 128.0);
    printf("   v_mfma_f64_16x16x4_f64 from one wavefront, cycles per MFMA: one dependent chain %.1f, two interleaved %.1f, four interleaved %.1f\n",
           o[10] / 64.0, o[11] / 64.0, o[12] / 64.0);
  }
  return 0;
}
