// handover_probe: what does it cost to hand a value from one workgroup to another through global memory - on the same XCD and
// across XCDs?  Two workgroups of a 16-workgroup launch (workgroup w runs on XCD w % 8) play ping-pong on two words:
// a stores k, b waits for k and stores k into the second word, a waits for that, and so on.  One-way latency = time / (2 n).
//   agent : relaxed agent-scope store / load  (global_store sc1, global_load sc1)       - what the one-launch solvers use
//   system: relaxed system-scope store / load (sc0 sc1)
//   l2    : plain store, load with sc0 only   (workgroup scope: may be served by the CU's own L1 - checked for progress)
// A second test hands over a BLOCK of data the way k_bcr_eliminate_fused does: 23 KB of agent-scope stores by 1024 threads,
// s_waitcnt vmcnt(0), barrier, one word; the consumer polls the word and loads the block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE> __device__ __forceinline__ void st(long long* p, long long v) {
  if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (MODE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
}
template <int MODE> __device__ __forceinline__ long long ld(long long* p) {
  if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  long long v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int MODE>
__global__ void k_pingpong(int a, int b, int n, long long* words, long long* out) {
  const int w = blockIdx.x;
  if (threadIdx.x != 0 || (w != a && w != b)) return;
  const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
  long long* ping = words, * pong = words + 64;       // different cache lines
  const long long t0 = wall_clock64();
  long long spins = 0;
  for (long long k = 1; k <= n; ++k) {
    if (w == a) {
      st<MODE>(ping, k);
      while (ld<MODE>(pong) != k) { if (++spins > (1ll << 24)) break; }
    } else {
      while (ld<MODE>(ping) != k) { if (++spins > (1ll << 24)) break; }
      st<MODE>(pong, k);
    }
    if (spins > (1ll << 24)) break;
  }
  const long long t1 = wall_clock64();
  out[4 * (w == a ? 0 : 1)] = t1 - t0;
  out[4 * (w == a ? 0 : 1) + 1] = xcc;
  out[4 * (w == a ? 0 : 1) + 2] = spins;
}

// block hand-over: producer = workgroup a, consumer = workgroup b, `n` rounds; returns per round: stores issued -> acknowledged,
// acknowledged -> word seen by the consumer, word seen -> block loaded (100 MHz ticks, summed)
__global__ __launch_bounds__(1024) void k_block(int a, int b, int n, int doubles, double* buf, long long* words, long long* out) {
  const int w = blockIdx.x;
  if (w != a && w != b) return;
  __shared__ long long tsum[4];
  if (threadIdx.x < 4) tsum[threadIdx.x] = 0;
  __syncthreads();
  long long* flag = words, * back = words + 64;
  double acc = 0.0;
  for (long long k = 1; k <= n; ++k) {
    if (w == a) {
      const long long t0 = wall_clock64();
      for (int e = threadIdx.x; e < doubles; e += 1024) __hip_atomic_store(buf + e, (double)(k + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const long long t1 = wall_clock64();
      if (threadIdx.x == 0) {
        __hip_atomic_store(flag, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tsum[0] += t1 - t0;
        while (__hip_atomic_load(back, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != k) {}
        tsum[3] += wall_clock64() - t0;              // the whole round trip
      }
      __syncthreads();
    } else {
      if (threadIdx.x == 0) while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != k) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      const long long t2 = wall_clock64();
      double v[3] = {0, 0, 0};
      int q = 0;
      for (int e = threadIdx.x; e < doubles; e += 1024) v[q++ % 3] += __hip_atomic_load(buf + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += v[0] + v[1] + v[2];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const long long t3 = wall_clock64();
      if (threadIdx.x == 0) { tsum[2] += t3 - t2; __hip_atomic_store(back, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int o = 8 + 4 * (w == a ? 0 : 1);
    out[o] = tsum[0]; out[o + 1] = tsum[2]; out[o + 2] = tsum[3]; out[o + 3] = (long long)acc;
  }
}

int main() {
  long long* words; long long* out; double* buf;
  hipMalloc(&words, 4096); hipMalloc(&out, 4096); hipMalloc(&buf, 1 << 20);
  std::vector<long long> h(64);
  const int n = 2000;
  const char* names[3] = {"agent (sc1)", "system (sc0 sc1)", "plain store + sc0 load"};
  for (int mode = 0; mode < 3; ++mode)
    for (int b : {8, 1, 4, 9}) {
      hipMemset(words, 0, 4096); hipMemset(out, 0, 4096);
      if (mode == 0) hipLaunchKernelGGL(k_pingpong<0>, dim3(16), dim3(64), 0, 0, 0, b, n, words, out);
      if (mode == 1) hipLaunchKernelGGL(k_pingpong<1>, dim3(16), dim3(64), 0, 0, 0, b, n, words, out);
      if (mode == 2) hipLaunchKernelGGL(k_pingpong<2>, dim3(16), dim3(64), 0, 0, 0, b, n, words, out);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
      printf("ping-pong %-24s workgroups 0 (XCD %lld) <-> %d (XCD %lld): one way %.0f ns%s\n", names[mode], h[1], b, h[5],
             h[0] * 10.0 / (2.0 * n), (h[2] > (1ll << 24) || h[6] > (1ll << 24)) ? "  [NO PROGRESS: stale cache]" : "");
    }
  for (int doubles : {54, 2916, 3 * 2916})
    for (int b : {8, 1, 4}) {
      hipMemset(words, 0, 4096); hipMemset(out, 0, 4096);
      hipLaunchKernelGGL(k_block, dim3(16), dim3(1024), 0, 0, 0, b, n, doubles, buf, words, out);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
      printf("block of %5d doubles, workgroup 0 -> %d (%s XCD): stores -> acknowledged %.0f ns, block load after the word %.0f ns, whole round trip (incl. word back) %.0f ns\n",
             doubles, b, b % 8 == 0 ? "same" : "other", h[8] * 10.0 / n, h[13] * 10.0 / n, h[10] * 10.0 / n);
    }
  return 0;
}
