// ba_bcr_blocks.h - the building blocks the cyclic-reduction node kernels of ba_bcr.h, ba_bcr_wide.h and ba_dense.h share:
// the register-resident Cholesky of a 12 x 12 diagonal block with its inverse (DPP row broadcasts along the pivot chain),
// the panel tile below it on the matrix cores, the forward substitution of a block row.  Constants of the solvers.
#pragma once

#include "ba_device.h"

// Build with -DBA_BCR_PROFILE (make PROFILE=1) to have node 2 of the first level write
// its per-phase shader-cycle counts to info[8..12] (printed under BA_SOLVE_TRACE=1).
#ifdef BA_BCR_PROFILE
#ifndef BA_BCR_TRACE_KB
#define BA_BCR_TRACE_KB 0      // per-wavefront stamps of the split kernel: 0 = the prologue, k > 0 = phase 1 of block step k
#endif
#define BA_STAMP(var) const long long var = clock64()
#else
#define BA_STAMP(var)
#endif

#include <utility>

namespace ba {

constexpr int kBcrThreads = 1024;                // assemble (111 nodes at config 3: few workgroups, so make them wide)
constexpr int kBcrElimThreads = 1024;           // eliminate / backsolve: 16 wavefronts per node
constexpr int kBcrTicketWord = 61;             // info[61]: tickets of k_bcr_backsolve_fused (info = flags + 1, 64 flag words)
constexpr int kBcrRefineTicketWord = 58;       // info[58]: tickets of k_bcr_refine
constexpr long long kBcrNotYet = 0x7FFA5A5A5A5A5A5All;    // a NaN no computation produces: the mark of a solution entry that is not there yet
constexpr int kBcrMaxSpins = 1 << 20;           // bounded waits of the one-launch kernels: polls of ~1 us each
constexpr int kBcrTimedOut = 0x7f000001;        // status word of a workgroup that gave up waiting for another (BA_SOLVE_TIMED_OUT)

__host__ __device__ inline size_t bcr_lds_bytes(int B) { return ((size_t)4 * B * (B + 1) + 4 * B + 8 + 560) * sizeof(double); }   // + inverses of the current and the previous diagonal block, identity table

// value of lane K of my 16-lane row (v_mov_b64 with a DPP row_newbcast source)
template <int K>
__device__ __forceinline__ double mov_rowbcast(double v) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
  return r;
}

// acc += (lane K of my row of `row`) * y, with the two wait states a DPP read of a just-written VGPR needs
template <int K>
__device__ __forceinline__ void fmac_rowbcast_safe(double& acc, double row, double y) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(row), "v"(y), "n"(K));
}

// acc -= (lane K of my row of `row`) * y   (negation as a source modifier: nothing extra on the chain)
template <int K>
__device__ __forceinline__ void fnmac_rowbcast_safe(double& acc, double row, double y) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(row), "v"(y), "n"(K));
}

template <int NB, int Q, int... Ps>
__device__ __forceinline__ void bcr_diag_update(std::integer_sequence<int, Ps...>, double (&col)[NB], double uqc) {
  (fnmac_rowbcast_safe<Q + 1 + Ps>(col[Q + 1 + Ps], uqc, uqc), ...);     // col[p] -= U[q][p] U[q][c]   (entries p > c are never used)
}

// pivot Q of the NB x NB Cholesky on lanes 0..NB-1 of one 16-lane row (lane c owns column c of
// U = L^T): all cross-lane traffic is DPP row_newbcast - one instruction per broadcast on the chain
// of dependent pivots
template <int NB, int Q, bool KEEP_L>
__device__ __forceinline__ void bcr_diag_pivot(double (&col)[NB], int c, double& di) {
  // The dependent chain is: broadcast pivot -> 1/sqrt (seed, then ONE cubic step y (1 + e/2 + 3 e^2/8), e = 1 - x y^2, folded
  // into the scaling: u = w + (w e)(1/2 + 3 e / 8) with w = col y) -> rank-1 update of the next column; measured
  // (tools/chain_probe) 14.6 + 20 + 4 x 8.4 + 16.9 cycles.  The one wavefront that runs it also pays 5 - 8 cycles of issue for
  // every instruction off the chain: no selects and no failure test here (a pivot <= 0 turns its own 1/sqrt and everything after
  // it into NaN or inf - bcr_diag_block looks afterwards), and 1/sqrt itself is only formed where somebody needs it.
  // u is right for every lane that matters: lane Q holds the pivot itself (-> its square root), lanes below Q hold entries that
  // are never used again.
  const double piv = mov_rowbcast<Q>(col[Q]);
  const double y = __builtin_amdgcn_rsq(piv);
  const double e = fma(-(piv * y), y, 1.0);
  const double pe = fma(e, 0.375, 0.5);
  const double w = col[Q] * y;
  const double uqc = fma(w * e, pe, w);
  bcr_diag_update<NB, Q>(std::make_integer_sequence<int, NB - 1 - Q>{}, col, uqc);
  col[Q] = uqc;
  if constexpr (KEEP_L) {
    const double inv = fma(y * e, pe, y);
    if (c == Q) di = inv;
  } else if constexpr (Q == NB - 1) {
    di = fma(y * e, pe, y);                                            // (wave-uniform: the health of the whole block, see below)
  }
}

template <int NB, bool KEEP_L, int... Qs>
__device__ __forceinline__ void bcr_diag_pivots(std::integer_sequence<int, Qs...>, double (&col)[NB], int c, double& di) {
  (bcr_diag_pivot<NB, Qs, KEEP_L>(col, c, di), ...);
}

// diagonal block of NB unknowns at k0 (wavefront 0): the INVERSE of its Cholesky factor to Li ([.][12], lower triangular) and,
// with KEEP_L, the factor itself in place and 1/diag to dinv (the kernels that hand the factor on; the node kernels of the
// narrow reduction only ever use the inverse).  Lanes 0..NB-1 of every 16-lane row own the columns of
// the block (each row a replica: the DPP broadcasts are row-local), the lanes above them own columns of the identity.
// The rank-1 updates of the elimination turn those into L^-1 (right-looking Cholesky of [A | I] gives [L^T | L^-1]):
// the same instructions, nothing added to the chain of dependent pivots, and no triangular solve afterwards.
constexpr int kBcrIdtDoubles = 160;                                     // identity table [13][12] (+ padding)
__device__ __forceinline__ void bcr_identity_table(double* Idt, int tid) {
  if (tid < 156) Idt[tid] = (tid / 12 == tid % 12) ? 1.0 : 0.0;
}

template <int NB, bool KEEP_L = true>
__device__ __forceinline__ void bcr_diag_block(double* __restrict__ G, int ld, double* __restrict__ dinv, int* __restrict__ bad,
                                               int k0, int lane, double* __restrict__ Li, const double* __restrict__ Idt,
                                               long long* trace = nullptr) {
  constexpr int NA = 16 - NB;                                           // identity columns per 16-lane row
#ifdef BA_BCR_PROFILE
  const long long tr0 = clock64();
#endif
  const int i = lane & 15;
  const bool own = i < NB;
  const int c = own ? i : NB - 1;
  const int j = (lane >> 4) * NA + (i - NB);                            // identity column of a lane that owns one (j < NB)
  // A[p][c] from the lower triangle, or column j of the identity out of a table in LDS (Idt = [13][12], row 12 zero): one
  // address select and NB unconditional loads (selects on the loaded values compile to a branch around every load)
  const double* src = own ? G + (k0 + c) * ld + k0 : Idt + (j < NB ? j : 12) * 12;
  double cl[NB];
#pragma unroll
  for (int p = 0; p < NB; ++p) cl[p] = src[p];
  double di = 0.0;
#ifdef BA_BCR_PROFILE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long tr1 = clock64();
#endif
  bcr_diag_pivots<NB, KEEP_L>(std::make_integer_sequence<int, NB>{}, cl, own ? i : -1, di);
#ifdef BA_BCR_PROFILE
  asm volatile("" ::"v"(cl[NB - 1]), "v"(di));
  const long long tr2 = clock64();
#endif
  if constexpr (KEEP_L) {
    // not positive definite: the first lane whose 1 / sqrt(pivot) is not a positive finite number is the first bad pivot
    const unsigned long long notpd = __ballot(lane < NB && !(di > 0.0 && di < __builtin_huge_val()));
    if (notpd && lane == 0) *bad = k0 + __ffsll((long long)notpd);
  } else {
    // a bad pivot poisons every later one: 1 / sqrt of the LAST pivot tells whether all of them were positive
    if (lane == 0 && !(di > 0.0 && di < __builtin_huge_val())) *bad = k0 + 1;
  }
  if (KEEP_L && lane < NB) {
    dinv[k0 + c] = di;
#pragma unroll
    for (int p = 0; p < NB; ++p)
      if (p <= c) G[(k0 + c) * ld + k0 + p] = cl[p];                    // L[c][p] = U[p][c]
  } else if (!own && j < NB) {
#pragma unroll
    for (int q = 0; q < NB; ++q) Li[q * 12 + j] = cl[q];                // L^-1[q][j] (exact zeros above the diagonal)
  }
#ifdef BA_BCR_PROFILE
  if (trace) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long tr3 = clock64();
    if (lane == 0) { trace[0] = tr1 - tr0; trace[1] = tr2 - tr1; trace[2] = tr3 - tr2; }
  }
#endif
}

// rows i0 .. i0+15 of the panel below the diagonal block of nbw (12 or 6) unknowns at k0: X = A L_kk^-T on the matrix
// cores, in place, computed as its transpose X^T = L_kk^-1 A^T (K = 12, three v_mfma_f64_16x16x4_f64): the result leaves the
// matrix core as lane (row of the tile, k % 4) -> X[row][k], which is where the lane read A from (the stores reuse the load
// addresses) AND the layout of both operands of the update X X^T that comes next - pr[] hands the tile on in registers.
// Li = [16][12], rows 12..15 zero; for a 6-unknown block the entries of Li outside its 6 x 6 corner are whatever the block
// before left there: the columns of A they meet are fed as zeros.
__device__ __forceinline__ void bcr_panel_tile(double* __restrict__ G, int ld, int B, int k0, int i0, const double* __restrict__ Li,
                                               int lr, int lk, double (&pr)[3], int nbw = 12) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
  const int row = i0 + lr < B ? i0 + lr : B - 1;                        // rows past the end repeat the last one (never stored)
  const int ao = row * ld + k0 + lk, bo = lr * 12 + lk;
  double a0 = G[ao], a1 = G[ao + 4], a2 = G[ao + 8];
  if (nbw < 12) { a1 = lk < 2 ? a1 : 0.0; a2 = 0.0; }
  mfma_acc acc = {0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[bo], a0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[bo + 4], a1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[bo + 8], a2, acc, 0, 0, 0);
  pr[0] = acc[0]; pr[1] = acc[1]; pr[2] = acc[2];
  if (i0 + lr < B) {
    G[ao] = acc[0];
    if (nbw == 12 || lk < 2) G[ao + 4] = acc[1];
    if (nbw == 12) G[ao + 8] = acc[2];
  }
}

typedef double bcr_acc4 __attribute__((ext_vector_type(4)));

// Wavefront 0 between two diagonal blocks.  The tile of the next block column that holds the next diagonal block (rows
// kn .. kn+15) owes the panel one update, C -= X X^T over the first 16 rows X of the panel - the tile wavefront 0 computed
// itself and still holds in registers (pr, in the layout of BOTH operands).  The accumulator tile is fetched while the panel
// is being computed (it has been final since the barrier before), so after the barrier that ends the panel phase the three
// MFMAs start without an LDS round trip.
__device__ __forceinline__ bcr_acc4 bcr_prefetch_tile0(const double* __restrict__ sm, int ld, int kn, int lr, int lk) {
  const int cb = (kn + lk) * ld + kn + lr;
  return bcr_acc4{sm[cb], sm[cb + 4 * ld], sm[cb + 8 * ld], sm[cb + 12 * ld]};
}
__device__ __forceinline__ void bcr_urgent_tile0(double* __restrict__ sm, int ld, int B, int k0, int nb, int lr, int lk,
                                                 const double (&pr)[3], bcr_acc4 acc) {
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[0], -pr[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[1], -pr[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[2], -pr[2], acc, 0, 0, 0);
  const int cb = (k0 + lk) * ld + k0 + lr;
  if (k0 + 16 <= B) {                                                   // (wave-uniform) all 16 rows exist
    if (lr < nb) {
#pragma unroll
      for (int v = 0; v < 4; ++v) sm[cb + 4 * v * ld] = acc[v];
    }
  } else {
    const int rl = lr < nb ? B - k0 - lk : 0;
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (4 * v < rl) sm[cb + 4 * v * ld] = acc[v];
  }
}

// acc += (lane K of my 16-lane row of `row`) * y   (the DPP source comes from an LDS load: no hazard)
template <int K>
__device__ __forceinline__ void fmac_rowbcast(double& acc, double row, double y) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(row), "v"(y), "n"(K));
}

template <int NB, int... Ps>
__device__ __forceinline__ void bcr_fwd_dot(std::integer_sequence<int, Ps...>, double chunk, const double (&x)[NB], double& s0, double& s1) {
  (fmac_rowbcast<Ps>((Ps & 1) ? s1 : s0, chunk, x[Ps]), ...);
}

template <int NB, int Q>
__device__ __forceinline__ void bcr_fwd_row(const double* __restrict__ Lrow, int ld, const double* __restrict__ dk,
                                            double* __restrict__ X, int st, double (&x)[NB], double& chunk, double& dq) {
  double nchunk = 0.0, ndq = 0.0;
  if constexpr (Q + 1 < NB) { nchunk = Lrow[(Q + 1) * ld]; ndq = dk[Q + 1]; }      // in flight during row Q's chain
  double s0 = 0.0, s1 = 0.0;
  bcr_fwd_dot<NB>(std::make_integer_sequence<int, Q>{}, chunk, x, s0, s1);
  x[Q] = (x[Q] - (s0 + s1)) * dq;
  X[Q * st] = x[Q];
  chunk = nchunk; dq = ndq;
}

template <int NB, int... Qs>
__device__ __forceinline__ void bcr_fwd_rows(std::integer_sequence<int, Qs...>, const double* __restrict__ Lrow, int ld,
                                             const double* __restrict__ dk, double* __restrict__ X, int st, double (&x)[NB]) {
  double chunk = Lrow[0], dq = dk[0];
  (bcr_fwd_row<NB, Qs>(Lrow, ld, dk, X, st, x, chunk, dq), ...);
}

// x <- L_kk^-1 x for the NB entries x[0], x[st], ... (one right-hand side, or one row of the panel,
// per lane; EVERY lane of the wavefront must call this - DPP sources have to be live lanes).
// Row q of L_kk is fetched once per wavefront as a 16-lane-periodic register (lane l: entry l%16,
// Lrow = &L_kk[0][lane%16]) and its entries reach the FMAs as DPP row_newbcast operands.
template <int NB>
__device__ __forceinline__ void bcr_block_forward(const double* __restrict__ Lrow, int ld, const double* __restrict__ dk,
                                                  double* __restrict__ X, int st) {
  double x[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) x[q] = X[q * st];
  bcr_fwd_rows<NB>(std::make_integer_sequence<int, NB>{}, Lrow, ld, dk, X, st, x);
}

}  // namespace ba
