// ba_bcr.h - parallel solve of the block-banded reduced camera system by block cyclic
// reduction (BCR), the multi-CU path of solve_motion_normal_eqns (bundle_adjuster.py:281-312).
//
// k_band_solve walks the band with ONE workgroup: nco dependent 6x6 pivots, a few
// microseconds each.  Here the cameras are grouped into super-blocks of hb cameras
// (B = 6*hb unknowns); because the band half-width is hb, the system is block
// TRIDIAGONAL in super-blocks:  T[I,I] = D_I,  T[I,I+1] = U_I.  Cyclic reduction then
// eliminates every other super-block in parallel, one workgroup per eliminated node,
// log2(N) levels deep:
//
//   level with stride s: node i (neighbours l = i-s, r = i+s):
//     D_i = G G^T (Cholesky),  P = G^-1 T[i,l],  Q = G^-1 T[i,r],  g = G^-1 f_i
//     D_l -= P^T P,  D_r -= Q^T Q,  T[l,r] = -P^T Q,  f_l -= P^T g,  f_r -= Q^T g
//   back-substitution, levels in reverse:  x_i = G^-T (g - P x_l - Q x_r)
//
// Every Schur complement of an SPD matrix is SPD, so this is the same arithmetic as a
// Cholesky factorisation in nested-dissection order: results agree with k_band_solve and
// with the reference's LU to round-off.  A non-positive pivot is reported through *info.
// All fp64, all blocks dense B x B (B <= 60) in LDS.  Inside a node the chain of 6 hb dependent
// pivots is what cannot be parallelised; everything GEMM-shaped around it (K = 6 updates of the
// right-hand sides and of the trailing matrix, the three B x B x B neighbour products) runs on
// the fp64 matrix cores (v_mfma_f64_16x16x4_f64), because the vector forms are bound by LDS
// reads or by the one-FMA-per-8-cycles issue rate of a wavefront (tools/fma_probe, lds_probe).
#pragma once

#include "ba_bcr_blocks.h"

namespace ba {

#ifndef BA_BCR_TEMPLATES_ONLY      // (non-template kernels: compiled by ba_solve.hip alone)
// band (+ mask) -> D[N][B][B], U[N][B][B] = T[I,I+1], f[N][B]; cameras past nco and masked
// parameters become identity rows with zero right-hand side.  Super-blocks of cb >= hb cameras (B = 6 cb; cb = hb
// everywhere but in the solve that is spread over several GPUs, which picks cb so that the elimination tree splits evenly).
// NIT: entries per thread and round.  3 everywhere (B = 54: one round; the wide solver's B = 138: seven) but in the narrow solve of a
// trial, which launches ceil(B^2 / 1024) workgroups per node with ONE entry per thread: 333 workgroups instead of 111 at config 3,
// the trial 2.5 us shorter (alternating runs, three out of three) though the kernel's own time hardly moves (5.4 -> 5.3 us).
template <int NIT = 3>
__global__ __launch_bounds__(kBcrThreads) void k_bcr_assemble(int nco, int hb, int cb, const double* __restrict__ S,
                                                              const double* __restrict__ b,
                                                              const unsigned char* __restrict__ mask,
                                                              double* __restrict__ Dm, double* __restrict__ Um,
                                                              double* __restrict__ fm, int* __restrict__ info,
                                                              double* __restrict__ xsol = nullptr, int* __restrict__ done = nullptr,
                                                              const int* __restrict__ nodes = nullptr, double* __restrict__ mark = nullptr,
                                                              long long mark_n = 0) {
  const int B = 6 * cb, hb1 = hb + 1;
  const int I = nodes ? nodes[blockIdx.x] : blockIdx.x;      // (a rank of the distributed solve assembles its own nodes and the separators only)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *info = 0;                                    // status word of this solve (the eliminate levels only ever set it)
    info[kBcrTicketWord] = 0;                     // ticket counter of k_bcr_backsolve_fused
    info[kBcrTicketWord - 1] = 0;                 // ... and of k_bcr_eliminate_fused (kBcrElimTicketWord)
    info[kBcrRefineTicketWord] = 0;               // ... and of k_bcr_refine (ba_bcr_refine.h)
  }
  // everything the refinement step behind this solve will wait on: "not yet" (ba_bcr_refine.h; only when the step is going to run)
  if (blockIdx.y == 0)
    for (long long e = (long long)blockIdx.x * kBcrThreads + threadIdx.x; e < mark_n; e += (long long)gridDim.x * kBcrThreads)
      mark[e] = __longlong_as_double(kBcrNotYet);
  if (done && threadIdx.x < 4) done[4 * I + threadIdx.x] = 0;      // "this (node, role) has handed its results on": not yet
  // every load of a thread is issued before its first store (one round trip to memory instead of one per entry): the address
  // of an entry that is not in the band is that of S[0], its value is then multiplied away - a select on the loaded value
  // would compile to a branch around every load
  for (int base = blockIdx.y * NIT * kBcrThreads; base < B * B; base += gridDim.y * NIT * kBcrThreads) {      // (gridDim.y > 1: the big nodes of ba_bcr_big.h)
  double vd[NIT], vu[NIT], kd[NIT], ku[NIT], idv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = base + threadIdx.x + it * kBcrThreads;
    const int ec = e < B * B ? e : 0;
    const int r = ec / B, c = ec - r * B;
    const int i = I * cb + r / 6, j = I * cb + c / 6, a = r % 6, bb = c % 6;
    const bool inside = i < nco && j < nco;
    const bool masked = inside && mask && (!mask[6 * i + a] || !mask[6 * j + bb]);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const bool inband = inside && !masked && hi - lo <= hb;
    // diagonal block: its upper triangle; above the diagonal (i < j): entry (a, bb) of block (i, j); below: (bb, a) of block (j, i)
    const bool flip = i > j || (i == j && a > bb);
    const size_t off = inband ? band_block(lo, hi, hb1) + (flip ? bb * 6 + a : a * 6 + bb) : 0;
    vd[it] = S[off];
    kd[it] = inband ? 1.0 : 0.0;
    idv[it] = (!inside || masked) && r == c ? 1.0 : 0.0;           // cameras past the end, masked parameters: identity rows
    const int j2 = j + cb;
    const bool uin = i < nco && j2 < nco && j2 - i <= hb && !(mask && (!mask[6 * i + a] || !mask[6 * j2 + bb]));
    vu[it] = S[uin ? band_block(i, j2, hb1) + a * 6 + bb : 0];
    ku[it] = uin ? 1.0 : 0.0;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = base + threadIdx.x + it * kBcrThreads;
    if (e < B * B) {
      Dm[(size_t)I * B * B + e] = kd[it] != 0.0 ? vd[it] : idv[it];
      Um[(size_t)I * B * B + e] = ku[it] != 0.0 ? vu[it] : 0.0;
    }
  }
  }
  if (blockIdx.y != 0) return;
  for (int r = threadIdx.x; r < B; r += kBcrThreads) {
    const int i = I * cb + r / 6, a = r % 6;
    fm[(size_t)I * B + r] = (i < nco && (!mask || mask[6 * i + a])) ? b[6 * (size_t)i + a] : 0.0;
    if (xsol) xsol[(size_t)I * B + r] = __longlong_as_double(kBcrNotYet);      // "not solved yet" (k_bcr_backsolve_fused polls the data itself)
  }
}
#endif

// One elimination level.  blockIdx.x = k-th node of this level: i = s*(2k+1) - 1.
// Out: Gi[i] = G^-1 (lower triangular), Pm[i] = P, Qm[i] = Q, fm[i] = g; neighbours updated.
// HB is a template parameter so that the dense B x B (B = 6 HB) pieces unroll.  Phases:
// load -> fused Cholesky + forward substitution (see below) -> neighbour products -> store.
template <int HB>
__global__ __launch_bounds__(kBcrElimThreads) void k_bcr_eliminate(int N, int s, double* __restrict__ Dm,
                                                               double* __restrict__ Um, double* __restrict__ fm,
                                                               double* __restrict__ Pm, double* __restrict__ Qm,
                                                               double* __restrict__ Gi, int* __restrict__ info,
                                                               double* __restrict__ xout) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int B = 6 * HB, ld = B + 1;
  double* G = sm;                       // [B][ld]  D_i -> its Cholesky factor L (lower)
  double* Pl = G + (size_t)B * ld;      // [B][ld]  T[i,l] -> P
  double* Ql = Pl + (size_t)B * ld;     // [B][ld]  T[i,r] -> Q
  double* Xi = Ql + (size_t)B * ld;     // [B][ld]  identity -> G^-1
  double* g = Xi + (size_t)B * ld;      // [B]
  double* dinv = g + B;                 // [B]
  int* bad = reinterpret_cast<int*>(dinv + B + 2);
  double* Li = dinv + B + 4;            // [2][16][12]: rows 0..11 = inverse of diagonal block kb in half kb & 1 (lower triangular;
                                        // phase 1 reads block kb-1's while wavefront 0 writes block kb's), rows 12..15 zero
  const int tid = threadIdx.x;
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N) return;
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  constexpr size_t BB = (size_t)B * B;

  BA_STAMP(t0);
  if (tid == 0) *bad = 0;
  if (tid < 384) Li[tid] = 0.0;      // (a 6-unknown node never writes rows / columns 6..11 of it, and 0 x stale-LDS-NaN = NaN)
  double* Idt = Li + 384;
  bcr_identity_table(Idt, tid);
  {
    // all global loads of a thread are issued before the first LDS store (one round trip)
    constexpr int NIT = (B * B + kBcrElimThreads - 1) / kBcrElimThreads;
    double vd[NIT], vp[NIT], vq[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * kBcrElimThreads;
      const bool ok = e < B * B;
      vd[it] = ok ? Dm[(size_t)i * BB + e] : 0.0;
      vp[it] = (ok && haveL) ? Um[(size_t)l * BB + e] : 0.0;
      vq[it] = (ok && haveR) ? Um[(size_t)i * BB + e] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * kBcrElimThreads;
      if (e < B * B) {
        const int rr = e / B, cc = e - rr * B;
        G[rr * ld + cc] = vd[it];
        Pl[cc * ld + rr] = vp[it];                                // T[i,l] = T[l,i]^T
        Ql[rr * ld + cc] = vq[it];                                // T[i,r]
        Xi[rr * ld + cc] = rr == cc ? 1.0 : 0.0;
      }
    }
  }
  for (int e = tid; e < B; e += kBcrElimThreads) g[e] = fm[(size_t)i * B + e];
  __syncthreads();

  BA_STAMP(t1);
  // ---- blocked Cholesky D_i = L L^T (lower, in place, blocks of 12 unknowns) FUSED with the forward
  //      substitution L Y = R of the right-hand sides P | Q | I | g, as one right-looking elimination
  //      of the augmented matrix [D_i | R] with one block of look-ahead:
  //        phase 1  wavefront 0: 12x12 diagonal factor of block kb (DPP row broadcasts, registers only)
  //                 the others : what step kb-1 still owes, on the matrix cores - Y = L_pp^-1 R_p,
  //                              R_below -= L_panel Y, and the trailing update of D right of block
  //                              column kb (16-column tasks, three v_mfma_f64_16x16x4_f64 per tile)
  //        phase 2  panel of block column kb (one row per lane, L entries as DPP operands);
  //                 wavefront 0: inverse of the diagonal block from its registers (for the next phase 1)
  //        phase 3  the URGENT part of step kb: update of block column kb+1 only, so that the next
  //                 diagonal factor can start while the rest of the update is still owed
  //      The chain of 6 HB dependent pivots (about 300 cycles each) bounds this kernel; the rest rides along.
  // wave-uniform values live in SGPRs: every VALU instruction of the 16 wavefronts costs a CU issue slot
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int nP = haveL ? B : 0, nQ = haveR ? B : 0, ncol = nP + nQ + B + 1;      // absent neighbours: no columns
  auto rhs_column = [&](int c, int& st) -> int {                                 // offset into sm[] and row stride
    st = c < ncol - 1 ? ld : 1;
    return (int)((c < nP ? Pl + c : c < nP + nQ ? Ql + (c - nP) : c < ncol - 1 ? Xi + (c - nP - nQ) : g) - sm);
  };
  // Blocks of 12 unknowns (the last one may have 6).  Late-update tasks: one per 16 right-hand-side
  // columns, then one per 16-column tile of the trailing matrix, dealt to wavefronts 1..15.  (Keeping
  // SIMD 0 free for the pivot chain of wavefront 0 speeds that chain up by 25 %, but the MFMA work then
  // piles up on three SIMDs and phase 1 gets longer, not shorter: measured both ways.)
  constexpr int NBLK = (B + 11) / 12;
  const int nct = (ncol + 15) >> 4;
  const int myslot = wave - 1;            // 0..14 (wavefront 0 factors)
#ifdef BA_BCR_PROFILE
  long long ph[4] = {0, 0, 0, 0};
#endif
  double pr[3] = {0.0, 0.0, 0.0};                           // wavefront 0: the first tile of the panel, handed from phase 2 to the next phase 1
  bcr_acc4 cpre = {0.0, 0.0, 0.0, 0.0};                     // and the tile it updates there
#pragma unroll 1
  for (int kb = 0; kb < NBLK; ++kb) {
    const int k0 = 12 * kb;
    const bool last = kb == NBLK - 1;
    const int nb = last ? B - k0 : 12;                      // this block: 12, or 6 at the end
    const int kn = k0 + nb;                                 // first unknown after this block
#ifdef BA_BCR_PROFILE
    const long long p0 = clock64();
#endif
    // ---------------- phase 1
    // the update block column kb still owes to the panel of block kb - 1 (C -= panel panel^T, K = 12), one 16-row tile per task:
    // the tile that holds the diagonal block by wavefront 0 itself right before it factors it (no barrier in between), the
    // tiles below it as late-update tasks of this phase (they are due at the next barrier, when the panel needs them)
    auto urgent_tile = [&](int t) {
      typedef double mfma_acc __attribute__((ext_vector_type(4)));
      const int kp = k0 - 12, i0 = k0 + 16 * t;
      const int ao = (i0 + lr) * ld + kp + lk, bo = (k0 + lr) * ld + kp + lk, cb = (i0 + lk) * ld + k0 + lr;
      mfma_acc acc = {sm[cb], sm[cb + 4 * ld], sm[cb + 8 * ld], sm[cb + 12 * ld]};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao], -sm[bo], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao + 4], -sm[bo + 4], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao + 8], -sm[bo + 8], acc, 0, 0, 0);
      if (i0 + 16 <= B) {                                    // (wave-uniform) all 16 rows exist: one lane mask for the four stores
        if (lr < nb) {
#pragma unroll
          for (int v = 0; v < 4; ++v) sm[cb + 4 * v * ld] = acc[v];
        }
      } else {
        const int rl = lr < nb ? B - i0 - lk : 0;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (4 * v < rl) sm[cb + 4 * v * ld] = acc[v];
      }
    };
    if (wave == 0) {
      __builtin_amdgcn_s_setprio(3);                        // the pivot chain is the critical path of the node
      if (kb > 0) { bcr_urgent_tile0(sm, ld, B, k0, nb, lr, lk, pr, cpre); lds_wave_sync(); }
      if (nb == 12) bcr_diag_block<12, false>(G, ld, dinv, bad, k0, lane, Li + 192 * (kb & 1), Idt);
      else bcr_diag_block<6, false>(G, ld, dinv, bad, k0, lane, Li + 192 * (kb & 1), Idt);
      __builtin_amdgcn_s_setprio(0);
    } else if (kb > 0 && myslot >= 0) {
      const double* Lp = Li + 192 * ((kb - 1) & 1);         // inverse of the previous diagonal block
      // what step kb-1 still owes, on the matrix cores: C -= A B with K = 12 (three
      // v_mfma_f64_16x16x4_f64), A = panel of block kp (16 rows of the tile), and
      //   right-hand sides, rows >= k0:                       B = Y_kp (12 x 16 columns)
      //   lower tiles of D right of block column kb:          B = panel^T
      // Loads are unconditional: out-of-range rows / columns stay inside the LDS arrays and only
      // reach accumulator rows / columns that are never stored.  Index arithmetic is 32-bit with
      // 24-bit multiplies: every VALU instruction of the busy wavefronts costs a CU issue slot.
      typedef double mfma_acc __attribute__((ext_vector_type(4)));
      const int kp = k0 - 12;
      const int ngt = (B - kn + 15) >> 4;                   // column tiles of the trailing matrix right of this block
      const int nsu = ((B - k0 + 15) >> 4) - 1;             // tiles of block column kb below the one wavefront 0 takes
      for (int task = myslot; task < nct + ngt + nsu; task += 15) {
        if (task >= nct + ngt) { urgent_tile(task - nct - ngt + 1); continue; }
        int t0, t1, i00, co, cst;
        bool cok, full;
        double nb0, nb1, nb2;                                  // the NEGATED B operand of the three k-steps
        if (task < nct) {
          // columns of the identity right of block kp are still zero in rows kp..kp+11: nothing to subtract
          if (16 * task >= nP + nQ + kp + 12 && 16 * task + 15 < ncol - 1) continue;
          const int col = 16 * task + lr;
          cok = col < ncol; full = 16 * task + 15 < ncol;
          int xst;
          const int xoff = rhs_column(cok ? col : ncol - 1, xst);
          t0 = 0; t1 = (B - k0 + 15) >> 4; i00 = k0;
          co = xoff + __mul24(lk, xst); cst = xst;
          // Y_kp = L_pp^-1 R_kp for these 16 columns (the inverse comes from phase 2 of the previous step).
          // The result lands in exactly the B-operand layout of the updates below: no LDS round trip.
          const int ro = xoff + __mul24(kp + lk, xst), r4 = 4 * xst;
          mfma_acc y = {0.0, 0.0, 0.0, 0.0};
          y = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[lr * 12 + lk], sm[ro], y, 0, 0, 0);
          y = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[lr * 12 + 4 + lk], sm[ro + r4], y, 0, 0, 0);
          y = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[lr * 12 + 8 + lk], sm[ro + 2 * r4], y, 0, 0, 0);
          if (cok) { sm[ro] = y[0]; sm[ro + r4] = y[1]; sm[ro + 2 * r4] = y[2]; }
          nb0 = -y[0]; nb1 = -y[1]; nb2 = -y[2];
        } else {
          const int gtile = task - nct, c0 = kn + 16 * gtile;
          t0 = gtile; t1 = (B - kn + 15) >> 4; i00 = kn;
          cok = c0 + lr < B; full = c0 + 15 < B;
          const int bo = (c0 + lr) * ld + kp + lk;
          co = lk * ld + c0 + lr; cst = ld;
          nb0 = -sm[bo]; nb1 = -sm[bo + 4]; nb2 = -sm[bo + 8];
        }
        const int c4 = 4 * cst;
        int ao = (i00 + 16 * t0 + lr) * ld + kp + lk;          // A entry of this lane in the first tile
        int cb = co + __mul24(i00 + 16 * t0, cst);             // first accumulator entry of this lane
        int rows = B - (i00 + 16 * t0);                        // rows left from the top of the tile
        double a0 = sm[ao], a1 = sm[ao + 4], a2 = sm[ao + 8];
        mfma_acc acc = {sm[cb], sm[cb + c4], sm[cb + 2 * c4], sm[cb + 3 * c4]};
        for (int t = t0; t < t1; ++t) {
          const double a0c = a0, a1c = a1, a2c = a2;
          mfma_acc accc = acc;
          const int cbc = cb, rc = rows;
          if (t + 1 < t1) {                                    // the next tile's operands are in flight
            ao += 16 * ld; cb += 16 * cst; rows -= 16;
            a0 = sm[ao]; a1 = sm[ao + 4]; a2 = sm[ao + 8];
            acc = mfma_acc{sm[cb], sm[cb + c4], sm[cb + 2 * c4], sm[cb + 3 * c4]};
          }
          accc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0c, nb0, accc, 0, 0, 0);
          accc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1c, nb1, accc, 0, 0, 0);
          accc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2c, nb2, accc, 0, 0, 0);
          if (full && rc >= 16) {                              // wave-uniform: the whole tile exists
#pragma unroll
            for (int v = 0; v < 4; ++v) sm[cbc + v * c4] = accc[v];
          } else {
            const int rl = cok ? rc - lk : 0;                  // rows v with 4 v < rl exist
#pragma unroll
            for (int v = 0; v < 4; ++v)
              if (4 * v < rl) sm[cbc + v * c4] = accc[v];      // (the strict upper part of diagonal tiles is never read)
          }
        }
      }
    }
#ifdef BA_BCR_PROFILE
    const long long p1a = clock64();
    if (blockIdx.x == 1 && s == 1 && kb == 1 && lane == 0) info[44 + wave] = (int)(p1a - p0);
#endif
    __syncthreads();
#ifdef BA_BCR_PROFILE
    const long long p1 = clock64();
#endif
    // ---------------- phase 2: panel, rows below the diagonal block: X = A L_kk^-T, one 16-row tile per wavefront
    if (kn + 16 * wave < B) {                                // (nb == 12 here; at most 4 tiles)
      if (wave == 0) cpre = bcr_prefetch_tile0(sm, ld, kn, lr, lk);
      bcr_panel_tile(G, ld, B, k0, kn + 16 * wave, Li + 192 * (kb & 1), lr, lk, pr);
    }
    __syncthreads();
#ifdef BA_BCR_PROFILE
    const long long p2 = clock64();
#endif
#ifdef BA_BCR_PROFILE
    ph[0] += p1a - p0; ph[1] += p1 - p0; ph[2] += p2 - p1; ph[3] += 0;
#endif
  }
#ifdef BA_BCR_PROFILE
  if (tid == 0 && blockIdx.x == 1 && s == 1) { info[14] = (int)ph[0]; info[15] = (int)ph[1]; info[16] = (int)ph[2]; info[17] = (int)ph[3]; }
#endif
  if (myslot >= 0) {
    // the last block row of the right-hand sides: Y = L_pp^-1 R_p (nothing below it)
    typedef double mfma_acc __attribute__((ext_vector_type(4)));
    constexpr int KL = 12 * (NBLK - 1), NL = B - KL;          // last block: start and size (12 or 6)
    const double* Ll = Li + 192 * ((NBLK - 1) & 1);         // inverse of the last diagonal block
    for (int task = myslot; task < nct; task += 15) {
      const int col = 16 * task + lr;
      const bool cok = col < ncol;
      int xst;
      const int xoff = rhs_column(cok ? col : ncol - 1, xst);
      const int ro = xoff + __mul24(KL + lk, xst), r4 = 4 * xst;
      mfma_acc y = {0.0, 0.0, 0.0, 0.0};
      y = __builtin_amdgcn_mfma_f64_16x16x4f64(Ll[lr * 12 + lk], sm[ro], y, 0, 0, 0);
      if (NL == 12) {
        y = __builtin_amdgcn_mfma_f64_16x16x4f64(Ll[lr * 12 + 4 + lk], sm[ro + r4], y, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f64_16x16x4f64(Ll[lr * 12 + 8 + lk], sm[ro + 2 * r4], y, 0, 0, 0);
        if (cok) { sm[ro] = y[0]; sm[ro + r4] = y[1]; sm[ro + 2 * r4] = y[2]; }
      } else {                                               // 6 rows: k = 4, 5 of the second step only
        const double r1 = lk < 2 ? sm[ro + (lk < 2 ? r4 : 0)] : 0.0;
        y = __builtin_amdgcn_mfma_f64_16x16x4f64(Ll[lr * 12 + 4 + lk], r1, y, 0, 0, 0);
        if (cok) { sm[ro] = y[0]; if (lk < 2) sm[ro + r4] = y[1]; }
      }
    }
  }
  __syncthreads();
  if (*bad) {
    if (tid == 0) atomicMax(info, i * B + *bad);
    return;
  }
  BA_STAMP(t2);

  BA_STAMP(t3);
  // ---- neighbour updates P^T P, Q^T Q, P^T Q on the fp64 matrix cores: one wavefront per
  //      16 x 16 output tile (lower tiles only for the two symmetric products), K in steps of
  //      4 with v_mfma_f64_16x16x4_f64 - per step each lane reads ONE entry of A and of B
  //      from LDS (A^T tile: lane -> column 16 ti + lane%16, row 4 ks + lane/16).  The vector
  //      FMA forms of this are bound by LDS reads (3x3 register tiles) or by the issue rate of
  //      a wavefront; the matrix core runs a tile at 16 FMA/clk from a single wavefront.
  {
    typedef double mfma_acc __attribute__((ext_vector_type(4)));
    constexpr int NT = (B + 15) / 16, NSYM = NT * (NT + 1) / 2, NTASK = 2 * NSYM + NT * NT, KST = (B + 3) / 4;
    const int ln = lane & 15, lk = lane >> 4;
    for (int task = wave; task < NTASK; task += kBcrElimThreads / 64) {
      int which, ti, tj;
      if (task < 2 * NSYM) {
        which = task < NSYM ? 0 : 1;
        tri_decode(task - which * NSYM, NT, tj, ti);         // tj <= ti: D is only ever read in its lower triangle
      } else {
        which = 2;
        ti = (task - 2 * NSYM) / NT; tj = (task - 2 * NSYM) % NT;
      }
      if ((which == 0 && !haveL) || (which == 1 && !haveR) || (which == 2 && !(haveL && haveR))) continue;
      const double* A = (which == 1 ? Ql : Pl) + 16 * ti + ln;
      const double* Bm = (which == 0 ? Pl : Ql) + 16 * tj + ln;
      mfma_acc acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) {
        const int k = 4 * ks + lk;
        const bool in = 4 * ks + 3 < B || k < B;              // rows past B belong to the next matrix: feed zeros
        const double ar = A[k * ld], br = Bm[k * ld];
        const double a = in ? ar : 0.0, b = in ? br : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
      // C/D layout of the f64 form: lane -> column lane%16, register v -> row lane/16 + 4 v
      double* dst = which == 2 ? Um + (size_t)l * BB : Dm + (size_t)(which == 0 ? l : r) * BB;
      const int col = 16 * tj + ln;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * ti + lk + 4 * v;
        if (row < B && col < B) {
          if (which == 2) dst[(size_t)row * B + col] = -acc[v];          // new T[l,r]
          else if (col <= row) atomic_add_f64(dst + (size_t)row * B + col, -acc[v]);
        }
      }
    }
    for (int c = kBcrElimThreads - 1 - tid; c < 2 * B; c += kBcrElimThreads) {     // the last wavefronts have fewer tiles
      const bool left = c < B;
      if ((left && !haveL) || (!left && !haveR)) continue;
      const double* A = left ? Pl + c : Ql + (c - B);
      double acc = 0.0;
      for (int k = 0; k < B; ++k) acc += A[k * ld] * g[k];
      atomic_add_f64(fm + (size_t)(left ? l : r) * B + (left ? c : c - B), -acc);
    }
  }
#ifdef BA_BCR_PROFILE
  __syncthreads();      // profile builds only: separate the products from the store phase
#endif
  BA_STAMP(t4);
  // ---- keep what the back-substitution needs
  for (int e = tid; e < B * B; e += kBcrElimThreads) {
    const int rr = e / B, cc = e - rr * B;
    Pm[(size_t)i * BB + e] = Pl[rr * ld + cc];
    Qm[(size_t)i * BB + e] = Ql[rr * ld + cc];
    Gi[(size_t)i * BB + e] = cc <= rr ? Xi[rr * ld + cc] : 0.0;
  }
  __syncthreads();                       // the products above read g; only now overwrite fm[i]
  for (int e = tid; e < B; e += kBcrElimThreads) fm[(size_t)i * B + e] = g[e];
  if (!haveL && !haveR) {
    // the root of the elimination tree: nothing to wait for, x_i = G^-T g right here (its
    // back-substitution launch is skipped by the host)
    for (int task = tid; task < 4 * B; task += kBcrElimThreads) {
      const int m = task >> 2, q4 = task & 3;
      double acc = 0.0;
      for (int k = m + q4; k < B; k += 4) acc += Xi[k * ld + m] * g[k];
      acc += dpp_pair<0xB1>(acc);
      acc += dpp_pair<0x4E>(acc);
      if (q4 == 0) xout[(size_t)i * B + m] = acc;
    }
  }
#ifdef BA_BCR_PROFILE
  if (tid == 0 && blockIdx.x == 1 && s == 1) {
    const long long t5 = clock64();
    int* o = info + 8;
    o[0] = (int)(t1 - t0); o[1] = (int)(t2 - t1); o[2] = (int)(t3 - t2); o[3] = (int)(t4 - t3); o[4] = (int)(t5 - t4);
  }
#endif
}

// --------------------------------------------------------------------------
// The same elimination level with a node SPREAD OVER THREE COMPUTE UNITS (blockIdx.y = role).  What bounds
// k_bcr_eliminate is, in equal parts, the chain of 6 HB dependent pivots and the matrix-core throughput of ONE
// compute unit for the 3 B + 1 right-hand-side columns [T_il | T_ir | I | f] and the three B^3 neighbour products.
// The columns are independent, so each role factors D_i for itself (redundant: the chain cannot be shared anyway) and
// takes a third of the rest:
//     role 0 (left)     P = G^-1 T_il, g       ->  Pm[i];   D_l -= P^T P,  f_l -= P^T g
//     role 1 (right)    Q = G^-1 T_ir, g       ->  Qm[i];   D_r -= Q^T Q,  f_r -= Q^T g
//     role 2 (inverse)  G^-1 (lower), g        ->  Gi[i], gm[i];  the root node also solves x_i = G^-T g
// The cross product T[l,r] = -P^T Q would need both P and Q in one place; it is not formed at all at this level.
// The NEXT level forms the couplings it needs from the stored factors of the node eliminated between the two
// survivors (j = i -/+ s/2):  T[i,l] = -Q_j^T P_j,  T[i,r] = -P_j'^T Q_j'  (one B^3 product on the matrix cores in the
// prologue of roles 0 / 1; level s = 1 reads the assembled couplings Um).  Pm, Qm are kept for the back-substitution
// anyway, so nothing extra is stored, and U is no longer read and written inside one launch.
// With a third of the columns the late-update tasks of a step (<= 7) fit on the wavefronts of SIMDs 1..3, so SIMD 0
// is left to the pivot chain of wavefront 0 (25 % faster chain; with all columns on one CU that lost more than it won).
// --------------------------------------------------------------------------
#ifndef BA_BCR_PRODUCT_WAVES
#define BA_BCR_PRODUCT_WAVES 15     // wavefronts of the prologue's coupling product: all but the chain's (15), or only those of SIMDs 1..3 (12)
#endif
// (nodes of 12 and 13 cameras, B = 72 / 78: three matrices - the coupling product's result shares its LDS with one of its operands, see bcr_split_node)
__host__ __device__ inline size_t bcr_split_lds_bytes(int B) { return ((size_t)(B > 6 * kBcrMaxHB ? 3 : 4) * B * (B + 1) + 4 * B + 8 + 560) * sizeof(double); }

// Four independent 4 x 4 blocks of A^T Bm (all B rows of the two k-major operands) in one chain of v_mfma_f64_4x4x4_4b_f64
// (tools/mfma4_probe.hip: lane 16 k + 4 b + i feeds A_b[i][k], lane 16 k + 4 b + j feeds B_b[k][j], D_b[i][j] comes out in
// lane 16 i + 4 b + j; 17.5 cycles per instruction).  Block b = (lane >> 2) & 3 of this lane covers columns acol .. acol+3 of
// A and bcol .. bcol+3 of Bm; returns the lane's entry, row acol + (lane >> 4), column bcol + (lane & 3), of the product.
// For the edge strips of a B x B product when B is a few columns past a multiple of 16 (B = 54: 6): a 6-wide strip costs
// two of these chains (2 x 14 x 17.5 cycles) per 16 columns, where the 16 x 16 x 4 form pays a whole tile (14 x 64).
// Columns past the end of a row read the start of the next one: whatever is there only reaches entries nobody stores.
template <int B>
__device__ __forceinline__ double bcr_mfma4_blocks(const double* __restrict__ A, const double* __restrict__ Bm, int ld, int acol,
                                                   int bcol, int lane) {
  constexpr int KST = (B + 3) / 4;
  const int kq = lane >> 4, m = lane & 3;
  double ar[KST], br[KST];
#pragma unroll
  for (int ks = 0; ks < KST; ++ks) {
    const int k = 4 * ks + kq;
    const bool in = 4 * ks + 3 < B || k < B;                            // rows past B belong to the next matrix: feed zeros
    const int kc = in ? k : 0;
    const double a_ = A[kc * ld + acol + m], b_ = Bm[kc * ld + bcol + m];
    ar[ks] = in ? a_ : 0.0; br[ks] = in ? b_ : 0.0;
  }
  double acc = 0.0, acc2 = 0.0;
#pragma unroll
  for (int ks = 0; ks < KST; ks += 2) {
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(ar[ks], br[ks], acc, 0, 0, 0);
    if (ks + 1 < KST) acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(ar[ks + 1], br[ks + 1], acc2, 0, 0, 0);
  }
  return acc + acc2;
}
// the edge of such a product in tasks of one chain each: q < NG * RB: rows of the edge (block row q % RB) x 16 columns (group
// q / RB); then the same number for columns of the edge x 16 rows (with_right only); last the corner (2 x 2 blocks)
__device__ __forceinline__ void bcr_edge_task(int q, int NG, int RB, bool with_right, int EB, int lane, int& acol, int& bcol) {
  const int blk = (lane >> 2) & 3, nb = NG * RB;
  if (q < nb) { acol = EB + 4 * (q % RB); bcol = 16 * (q / RB) + 4 * blk; }
  else if (with_right && q < 2 * nb) { const int q2 = q - nb; acol = 16 * (q2 / RB) + 4 * blk; bcol = EB + 4 * (q2 % RB); }
  else { acol = EB + 4 * (blk >> 1); bcol = EB + 4 * (blk & 1); }
}

// What crosses workgroups INSIDE one launch (k_bcr_eliminate_fused: a level's outputs are the next level's inputs) goes
// through relaxed agent-scope accesses, which bypass / write through the caches that are not coherent across the chip's
// eight L2s; k_bcr_eliminate_split (one launch per level) uses plain ones.
template <bool COHERENT>
__device__ __forceinline__ double bcr_ld(const double* p) {
  if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COHERENT>
__device__ __forceinline__ void bcr_st(double* p, double v) {
  if constexpr (COHERENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// 16-byte agent-coherent loads, K of them in flight at once and ONE wait - all inside one asm statement, so that nothing can
// read a destination register before its data is there.  (tools/handover_probe.hip: what a workgroup pays for fetching the
// three B x B inputs of a node with relaxed agent-scope accesses is their NUMBER, not a latency - 8748 eight-byte loads take
// 2.7 us, 2916 take 1.1 us, 54 take 0.35 us: these loads bypass the L2, every wavefront instruction is its own trip to memory.)
typedef double bcr_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bcr_ld16(const double* p0, const double* p1, const double* p2, bcr_d2& v0, bcr_d2& v1, bcr_d2& v2) {
  asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}
__device__ __forceinline__ void bcr_ld16(const double* p0, const double* p1, const double* p2, const double* p3, const double* p4,
                                         const double* p5, bcr_d2& v0, bcr_d2& v1, bcr_d2& v2, bcr_d2& v3, bcr_d2& v4, bcr_d2& v5) {
  asm volatile("global_load_dwordx4 %0, %6, off sc1\n\tglobal_load_dwordx4 %1, %7, off sc1\n\tglobal_load_dwordx4 %2, %8, off sc1\n\t"
               "global_load_dwordx4 %3, %9, off sc1\n\tglobal_load_dwordx4 %4, %10, off sc1\n\tglobal_load_dwordx4 %5, %11, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5)
               : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5) : "memory");
}
__device__ __forceinline__ void bcr_ld16(const double* p0, const double* p1, const double* p2, const double* p3, const double* p4,
                                         const double* p5, const double* p6, const double* p7, const double* p8, bcr_d2& v0, bcr_d2& v1,
                                         bcr_d2& v2, bcr_d2& v3, bcr_d2& v4, bcr_d2& v5, bcr_d2& v6, bcr_d2& v7, bcr_d2& v8) {
  asm volatile("global_load_dwordx4 %0, %9, off sc1\n\tglobal_load_dwordx4 %1, %10, off sc1\n\tglobal_load_dwordx4 %2, %11, off sc1\n\t"
               "global_load_dwordx4 %3, %12, off sc1\n\tglobal_load_dwordx4 %4, %13, off sc1\n\tglobal_load_dwordx4 %5, %14, off sc1\n\t"
               "global_load_dwordx4 %6, %15, off sc1\n\tglobal_load_dwordx4 %7, %16, off sc1\n\tglobal_load_dwordx4 %8, %17, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7), "=&v"(v8)
               : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7), "v"(p8) : "memory");
}
__device__ __forceinline__ void bcr_ld16(const double* p0, const double* p1, bcr_d2& v0, bcr_d2& v1) {
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(v0), "=&v"(v1) : "v"(p0), "v"(p1) : "memory");
}
__device__ __forceinline__ void bcr_st16(double* p, bcr_d2 v) {       // (acknowledged like any store: s_waitcnt vmcnt(0) before the word that publishes it)
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

#ifndef BA_BCR_WIDE_HANDOVER
#define BA_BCR_WIDE_HANDOVER 1      // fused kernel: the inputs of a node and the factor it hands on move 16 bytes per lane
#endif
#ifndef BA_BCR_WIDE_MAX_ROUNDS
#define BA_BCR_WIDE_MAX_ROUNDS 3    // ... for nodes whose B x B / 2 pairs are at most that many per thread (3: up to 13 cameras)
#endif
#ifndef BA_BCR_RHS_ON_SIMD0
#define BA_BCR_RHS_ON_SIMD0 1       // nodes of 12, 13 cameras: right-hand-side wavefronts on the pivot chain's SIMD (0: none, as for the narrower nodes)
#endif
#ifndef BA_BCR_STORE_FIRST
#define BA_BCR_STORE_FIRST 1        // fused kernel: the factors P / Q leave for memory BEFORE the neighbour products (their latency rides under the MFMAs)
#endif

// What a (node, role) of the one-launch elimination waits for and publishes (words of done[], see k_bcr_eliminate_fused):
// pq[0..1]: the two factors P_j, Q_j its coupling is formed from are in memory; d[0..1]: every node that adds to D_i, f_i
// has added; my_pq / my_d: this role's own two words.  Null = nothing to wait for / to say.
struct BcrDeps {
  const int* pq[2];
  const int* d[2];
  int* my_pq;
  int* status;
};
__device__ __forceinline__ void bcr_wait_done(const int* flag, int* status) {
  int v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int spins = 0; v == 0; ++spins) {
    if (spins >= kBcrMaxSpins) { atomicMax(status, kBcrTimedOut); break; }
    __builtin_amdgcn_s_sleep(1);
    v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One (node, role) of a split elimination level - the body of k_bcr_eliminate_split and of k_bcr_eliminate_fused (FUSED: the
// inputs come from workgroups of the same launch, `dep` says which words to wait for).  Returns false when D_i turned out
// not to be positive definite (status word set, nothing handed on).
template <int HB, bool FUSED>
__device__ __forceinline__ bool bcr_split_node(double* __restrict__ sm, int N, int s, int i, int role, double* __restrict__ Dm,
                                               const double* __restrict__ Um, double* __restrict__ fm,
                                               double* Pm, double* Qm, double* __restrict__ Gi, double* __restrict__ gm,
                                               int* __restrict__ info, double* __restrict__ xout, const BcrDeps dep,
                                               long long* __restrict__ tline = nullptr) {
  typedef double mfma_acc __attribute__((ext_vector_type(4)));
#ifdef BA_BCR_PROFILE
#define BA_TLINE(k) do { if (tline && threadIdx.x == 0) tline[k] = wall_clock64(); } while (0)
#else
#define BA_TLINE(k)
#endif
  constexpr int B = 6 * HB, ld = B + 1;
  // Nodes of more than kBcrMaxHB cameras (B = 72, 78): four matrices do not fit the 160 KB, and Ta only lives until the coupling
  // product R = -A^T Bm has been formed - R takes its place: the product stays in the registers it left the matrix cores in
  // until every wavefront has read its operands (one barrier more per node; the narrower nodes keep the layout they had).
  constexpr bool kAlias = B > 6 * kBcrMaxHB;
  double* G = sm;                       // [B][ld]  D_i -> its Cholesky factor L (lower)
  double* R = G + (size_t)B * ld;       // [B][ld]  right-hand sides of this role: T_il | T_ir | I  ->  P | Q | G^-1
  double* Ta = kAlias ? R : R + (size_t)B * ld;      // [B][ld]  prologue: factors of the node eliminated one level down
  double* Tb = Ta + (size_t)B * ld;     // [B][ld]
  double* g = Tb + (size_t)B * ld;      // [B]
  double* dinv = g + B;                 // [B]
  int* bad = reinterpret_cast<int*>(dinv + B + 2);
  double* Li = dinv + B + 4;            // [2][16][12]: inverse of diagonal block kb in half kb & 1 (lower triangular), rows 12..15 zero
  const int tid = threadIdx.x;
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  constexpr size_t BB = (size_t)B * B;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lk = lane >> 4;

  if (tid == 0) { bad[0] = 0; bad[1] = 0; }
  if (tid < 384) Li[tid] = 0.0;      // (a 6-unknown node never writes rows / columns 6..11 of it, and 0 x stale-LDS-NaN = NaN)
  double* Idt = Li + 384;
  bcr_identity_table(Idt, tid);
  // ... and the words behind the table, up to the end of the node's LDS: a 6-unknown node's 16-row tile reads reach them with a
  // zero factor (tests/test_gpu_fuzz.py poisons LDS with NaNs between problems: the counters that once lived here hid it)
  if (tid >= 156 && tid < 176) Idt[tid] = 0.0;
  if constexpr (FUSED) {
    // (selects, not indexing: a runtime index would put the struct into scratch memory)
    const int* word = tid == 0 ? dep.d[0] : tid == 1 ? dep.d[1] : tid == 2 ? dep.pq[0] : dep.pq[1];
    if (tid < 4 && word) bcr_wait_done(word, dep.status);
    __syncthreads();
  }
  BA_TLINE(1);
#ifdef BA_BCR_PROFILE
  long long pst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long* dtrace = reinterpret_cast<long long*>(Li + 384 + kBcrIdtDoubles);
  const long long pt0 = clock64();
#endif
  {
    // all global loads of a thread are issued before the first LDS store (one round trip)
    constexpr int NIT = (B * B + kBcrElimThreads - 1) / kBcrElimThreads;
    const bool direct = s == 1 && role < 2;                           // level 1: the assembled couplings
    const bool prod = s > 1 && role < 2;                              // deeper: form the coupling from the factors of node j
    const int j = role == 0 ? i - (s >> 1) : i + (s >> 1);
    const double* srcU = Um + (size_t)(role == 0 ? l : i) * BB;
    // FUSED, above the first level: D_i and the two factors as 16-byte pairs (entries 2 q, 2 q + 1 share a row: B is even),
    // half as many trips to memory; pairs past the end read pair 0 again (never stored)
    constexpr int NPAIR = B * B / 2, NITW = (NPAIR + kBcrElimThreads - 1) / kBcrElimThreads;
    const bool wide = FUSED && BA_BCR_WIDE_HANDOVER && NITW <= BA_BCR_WIDE_MAX_ROUNDS && s > 1;
    if (wide) {
      const double* pd = Dm + (size_t)i * BB;
      const double* pa = prod ? Pm + (size_t)j * BB : pd;
      const double* pb = prod ? Qm + (size_t)j * BB : pd;
      const int q0 = tid, q1 = tid + kBcrElimThreads, q2 = tid + 2 * kBcrElimThreads;
      const int o0 = 2 * (q0 < NPAIR ? q0 : 0), o1 = 2 * (q1 < NPAIR ? q1 : 0), o2 = 2 * (q2 < NPAIR ? q2 : 0);
      bcr_d2 d0, a0, b0, d1 = {0.0, 0.0}, a1 = {0.0, 0.0}, b1 = {0.0, 0.0}, d2 = {0.0, 0.0}, a2 = {0.0, 0.0}, b2 = {0.0, 0.0};
      if (!prod && NITW == 3) { bcr_ld16(pd + o0, pd + o1, pd + o2, d0, d1, d2); a0 = b0 = d0; }      // (the inverse role: D_i only)
      else if (!prod) { bcr_ld16(pd + o0, pd + o1, d0, d1); a0 = b0 = d0; }
      else if (NITW == 1) bcr_ld16(pd + o0, pa + o0, pb + o0, d0, a0, b0);
      else if (NITW == 2) bcr_ld16(pd + o0, pa + o0, pb + o0, pd + o1, pa + o1, pb + o1, d0, a0, b0, d1, a1, b1);
      else bcr_ld16(pd + o0, pa + o0, pb + o0, pd + o1, pa + o1, pb + o1, pd + o2, pa + o2, pb + o2, d0, a0, b0, d1, a1, b1, d2, a2, b2);
      auto put = [&](int q, bcr_d2 dv, bcr_d2 av, bcr_d2 bv) {
        if (q < NPAIR) {
          const int e = 2 * q, rr = e / B, cc = e - rr * B, o = rr * ld + cc;
          G[o] = dv[0]; G[o + 1] = dv[1];
          if (prod) {
            Ta[o] = av[0]; Ta[o + 1] = av[1];                        // P_j
            Tb[o] = bv[0]; Tb[o + 1] = bv[1];                        // Q_j
          } else {
            R[o] = rr == cc ? 1.0 : 0.0; R[o + 1] = rr == cc + 1 ? 1.0 : 0.0;
          }
        }
      };
      put(q0, d0, a0, b0);
      if (NITW > 1) put(q1, d1, a1, b1);
      if (NITW > 2) put(q2, d2, a2, b2);
    }
    double vd[NIT], va[NIT], vb[NIT];
    if (!wide) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * kBcrElimThreads;
      const bool ok = e < B * B;
      vd[it] = ok ? bcr_ld<FUSED>(Dm + (size_t)i * BB + e) : 0.0;
      va[it] = (ok && direct) ? srcU[e] : (ok && prod) ? bcr_ld<FUSED>(Pm + (size_t)j * BB + e) : 0.0;
      vb[it] = (ok && prod) ? bcr_ld<FUSED>(Qm + (size_t)j * BB + e) : 0.0;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * kBcrElimThreads;
      if (e < B * B) {
        const int rr = e / B, cc = e - rr * B;
        G[rr * ld + cc] = vd[it];
        if (direct) {
          if (role == 0) R[cc * ld + rr] = va[it];                  // T[i,l] = T[l,i]^T
          else R[rr * ld + cc] = va[it];                            // T[i,r]
        } else if (prod) {
          Ta[rr * ld + cc] = va[it];                                // P_j
          Tb[rr * ld + cc] = vb[it];                                // Q_j
        } else {
          R[rr * ld + cc] = rr == cc ? 1.0 : 0.0;
        }
      }
    }
    }
    for (int e = tid; e < B; e += kBcrElimThreads) g[e] = bcr_ld<FUSED>(fm + (size_t)i * B + e);
    __syncthreads();
    BA_TLINE(2);
#ifdef BA_BCR_PROFILE
    pst[0] = clock64() - pt0;
#endif
  }
  // ---- block 0's diagonal factor (wavefront 0: it needs D_i only) WHILE the other wavefronts form this role's coupling
  //      from the factors of the node eliminated one level down
  constexpr int NBLK = (B + 11) / 12;
  // (kAlias: the tasks of a wavefront of the coupling product, held in registers over a barrier)
  constexpr int PNT = (B + 15) / 16, PEB = 16 * (PNT - 1), PEE = B - PEB, PRB = (PEE + 3) / 4;
  constexpr bool kPEdge4 = PNT > 1 && PEE <= 8;
  constexpr int PNTF = kPEdge4 ? PNT - 1 : PNT, PNF = PNTF * PNTF, PNS = kPEdge4 ? 2 * PNTF * PRB + 1 : 0;
  constexpr int pbase = PNF < 15 ? PNF : 0, pnsw = 15 - pbase;
  constexpr int MAXT = kAlias ? (PNF + 14) / 15 : 1, MAXE = kAlias && kPEdge4 ? (PNS + pnsw - 1) / pnsw : 1;
  static_assert(!kAlias || BA_BCR_PRODUCT_WAVES == 15, "nodes of more than 11 cameras: the coupling product on wavefronts 1..15");
  mfma_acc ptile[MAXT];
  double pedge[MAXE];
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    if (B >= 12) bcr_diag_block<12, false>(G, ld, dinv, bad, 0, lane, Li, Idt);
    else bcr_diag_block<6, false>(G, ld, dinv, bad, 0, lane, Li, Idt);
    __builtin_amdgcn_s_setprio(0);
#ifdef BA_BCR_PROFILE
    pst[2] += clock64() - pt0 - pst[0];
#endif
  } else if (kAlias && s > 1 && role < 2) {
    constexpr int KST = (B + 3) / 4;
    const double* A = role == 0 ? Tb : Ta;
    const double* Bm = role == 0 ? Ta : Tb;
    const int p = wave - 1;
    if constexpr (kPEdge4) {
#pragma unroll
      for (int u = 0; u < MAXE; ++u) {
        const int q = p - pbase + u * pnsw;
        pedge[u] = 0.0;
        if (p >= pbase && q < PNS) {
          int acol, bcol;
          bcr_edge_task(q, PNTF, PRB, true, PEB, lane, acol, bcol);
          pedge[u] = bcr_mfma4_blocks<B>(A, Bm, ld, acol, bcol, lane);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
      const int task = p + 15 * u;
      ptile[u] = mfma_acc{0.0, 0.0, 0.0, 0.0};
      if (task < PNF) {
        const int ti = task / PNTF, tj = task - ti * PNTF;
        // (B = 78: the operands in two halves of K - all of them at once, next to the tiles held over the barrier, do not fit 128 registers)
        constexpr int NH = KST > 18 ? 2 : 1, KH = (KST + NH - 1) / NH;
        mfma_acc acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int half = 0; half < NH; ++half) {
          double ar[KH], br[KH];
#pragma unroll
          for (int q = 0; q < KH; ++q) {
            const int ks = half * KH + q, k = 4 * ks + lk;
            const bool in = ks < KST && (4 * ks + 3 < B || k < B);
            const int kc = in ? k : 0;
            const double a_ = A[kc * ld + 16 * ti + lr], b_ = Bm[kc * ld + 16 * tj + lr];
            ar[q] = in ? a_ : 0.0; br[q] = in ? b_ : 0.0;
          }
#pragma unroll
          for (int q = 0; q < KH; q += 2) {
            if (half * KH + q < KST) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[q], br[q], acc, 0, 0, 0);
            if (q + 1 < KH && half * KH + q + 1 < KST) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[q + 1], br[q + 1], acc2, 0, 0, 0);
          }
        }
        ptile[u] = acc + acc2;
      }
    }
  } else if (s > 1 && role < 2) {
    // R = -A^T Bm with (A, Bm) = (Q_j, P_j) for the left role, (P_j', Q_j') for the right one: 16 x 16 output tiles over
    // wavefronts 1..15, K in steps of 4 (lane -> column 16 t + lane % 16 of the k-major operand, k = 4 ks + lane / 16)
    constexpr int NT = (B + 15) / 16, KST = (B + 3) / 4;
    const double* A = role == 0 ? Tb : Ta;
    const double* Bm = role == 0 ? Ta : Tb;
    // B a few columns past a multiple of 16 (54 = 48 + 6): the last tile row and column on v_mfma_f64_4x4x4 (bcr_mfma4_blocks),
    // 13 chains of 245 cycles for what would be 7 tiles of 896; the 9 full tiles one per wavefront, the chains on the other six
    constexpr int EB = 16 * (NT - 1), EE = B - EB, RB = (EE + 3) / 4;
    constexpr bool kEdge4 = NT > 1 && EE <= 8 && BA_BCR_PRODUCT_WAVES == 15;
    constexpr int NTF = kEdge4 ? NT - 1 : NT;                 // tile rows / columns done as 16 x 16 tiles
    if (kEdge4) {
      constexpr int NF = NTF * NTF, NS = 2 * NTF * RB + 1, base = NF < 15 ? NF : 0, nsw = 15 - base;
      const int p = wave - 1;
      if (p >= base)
        for (int q = p - base; q < NS; q += nsw) {
          int acol, bcol;
          bcr_edge_task(q, NTF, RB, true, EB, lane, acol, bcol);
          const double v = bcr_mfma4_blocks<B>(A, Bm, ld, acol, bcol, lane);
          const int row = acol + (lane >> 4), col = bcol + (lane & 3);
          if (row < B && col < B) R[row * ld + col] = -v;
        }
    }
    const int pslot = BA_BCR_PRODUCT_WAVES == 15 ? wave - 1 : ((wave & 3) ? wave - 1 - (wave >> 2) : -1);
    for (int task = pslot; task >= 0 && task < NTF * NTF; task += BA_BCR_PRODUCT_WAVES) {
      const int ti = task / NTF, tj = task - ti * NTF;
      // all operands of the tile first (one LDS round trip), then the chain of MFMAs; two accumulators halve the chain
      double ar[KST], br[KST];
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) {
        const int k = 4 * ks + lk;
        const bool in = 4 * ks + 3 < B || k < B;              // rows past B belong to the next matrix: feed zeros
        const int kc = in ? k : 0;
        const double a_ = A[kc * ld + 16 * ti + lr], b_ = Bm[kc * ld + 16 * tj + lr];
        ar[ks] = in ? a_ : 0.0; br[ks] = in ? b_ : 0.0;
      }
      mfma_acc acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KST; ks += 2) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[ks], br[ks], acc, 0, 0, 0);
        if (ks + 1 < KST) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[ks + 1], br[ks + 1], acc2, 0, 0, 0);
      }
      const int col = 16 * tj + lr;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * ti + lk + 4 * v;
        if (row < B && col < B) R[row * ld + col] = -(acc[v] + acc2[v]);
      }
    }
  }
#ifdef BA_BCR_PROFILE
  if (BA_BCR_TRACE_KB == 0 && i == 3 * s - 1 && s == 2 && role == 0 && lane == 0) info[44 + wave] = (int)(clock64() - pt0 - pst[0]);      // prologue, per wavefront
#endif
  if constexpr (kAlias) {
    if (s > 1 && role < 2) {              // (the same for every wavefront of the workgroup)
      __syncthreads();                    // every operand of the product has been read: R may take Ta's place
      if (wave != 0) {
        const int p = wave - 1;
        if constexpr (kPEdge4) {
#pragma unroll
          for (int u = 0; u < MAXE; ++u) {
            const int q = p - pbase + u * pnsw;
            if (p >= pbase && q < PNS) {
              int acol, bcol;
              bcr_edge_task(q, PNTF, PRB, true, PEB, lane, acol, bcol);
              const int row = acol + (lane >> 4), col = bcol + (lane & 3);
              if (row < B && col < B) R[row * ld + col] = -pedge[u];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
          const int task = p + 15 * u;
          if (task < PNF) {
            const int ti = task / PNTF, tj = task - ti * PNTF, col = 16 * tj + lr;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int row = 16 * ti + lk + 4 * v;
              if (row < B && col < B) R[row * ld + col] = -ptile[u][v];
            }
          }
        }
      }
    }
  }
  __syncthreads();
  BA_TLINE(3);

#ifdef BA_BCR_PROFILE
  const long long pt1 = clock64();
  pst[1] = pt1 - pt0 - pst[0];
#endif
  // ---- blocked Cholesky D_i = L L^T fused with the forward substitution L Y = [R | g], one block of look-ahead:
  //      the structure of k_bcr_eliminate (see there) with B + 1 right-hand-side columns instead of 3 B + 1
  constexpr int ncol = B + 1;
  auto rhs_column = [&](int c, int& st) -> int {                                 // offset into sm[] and row stride
    st = c < B ? ld : 1;
    return (int)((c < B ? R + c : g) - sm);
  };
  constexpr int nct = (ncol + 15) >> 4;
  // late-update workers: the wavefronts of SIMDs 1..3 (wave % 4 != 0): 12 slots; SIMD 0 belongs to the pivot chain
  int myslot_ = (wave & 3) ? wave - 1 - (wave >> 2) : -1;
  // the last nct of them own 16 columns of the right-hand sides each (none of the wavefronts 0..3, which compute the panel in phase 2)
  int rhs_ct_ = (myslot_ >= 12 - nct) ? 11 - myslot_ : -1;
  int nworkers_ = 12 - nct;
  if constexpr (kAlias && BA_BCR_RHS_ON_SIMD0 > 0) {
    // Nodes of 12 and 13 cameras: five 16-column tiles of right-hand sides at 6 - 7 block rows each are 90 - 105 MFMAs a step,
    // the late updates 30 - 42 more: on the matrix cores of three SIMDs that is MORE than the pivot chain takes (phase 1 of a
    // step: 4.9 - 6.1 k cycles on the SIMDs with two right-hand-side wavefronts against 3.0 k of wavefront 0).  One tile goes
    // to a wavefront of SIMD 0 (the chain's: it pays a little), the others one per SIMD + one; the late updates in the order
    // heaviest first to the SIMDs with one right-hand-side wavefront.
    static_assert(nct == 5, "five right-hand-side tiles");
    if (BA_BCR_RHS_ON_SIMD0 == 1) {
      rhs_ct_ = wave == 13 ? 0 : wave == 14 ? 1 : wave == 15 ? 2 : wave == 10 ? 3 : wave == 4 ? 4 : -1;
      myslot_ = wave == 1 ? 0 : wave == 3 ? 1 : wave == 5 ? 2 : wave == 7 ? 3 : wave == 9 ? 4 : wave == 11 ? 5 : wave == 2 ? 6 : wave == 6 ? 7 : -1;
      nworkers_ = 8;
    } else {
      rhs_ct_ = wave == 13 ? 0 : wave == 14 ? 1 : wave == 15 ? 2 : wave == 8 ? 3 : wave == 4 ? 4 : -1;
      myslot_ = wave == 1 ? 0 : wave == 3 ? 1 : wave == 2 ? 2 : wave == 5 ? 3 : wave == 7 ? 4 : wave == 6 ? 5 : wave == 9 ? 6 : wave == 11 ? 7 : wave == 10 ? 8 : -1;
      nworkers_ = 9;
    }
  }
  const int myslot = myslot_, rhs_ct = rhs_ct_, nworkers = nworkers_;
  const int rhs_col = 16 * (rhs_ct >= 0 ? rhs_ct : 0) + lr;
  const bool rhs_cok = rhs_col < ncol;
  int rhs_xst;
  const int rhs_xoff = rhs_column(rhs_cok ? rhs_col : ncol - 1, rhs_xst);
  mfma_acc racc[NBLK];                                      // all their block rows, in the accumulator layout of the matrix core
#pragma unroll
  for (int r = 0; r < NBLK; ++r) {
    racc[r] = mfma_acc{0.0, 0.0, 0.0, 0.0};
    if (rhs_ct >= 0) {
      const int ro = rhs_xoff + __mul24(12 * r + lk, rhs_xst), r4 = 4 * rhs_xst;
      const bool six = 12 * r + 12 > B;                       // the 6-unknown block at the end: rows 6.. do not exist
      racc[r][0] = sm[ro];
      racc[r][1] = (!six || lk < 2) ? sm[(!six || lk < 2) ? ro + r4 : ro] : 0.0;
      racc[r][2] = !six ? sm[!six ? ro + 2 * r4 : ro] : 0.0;
    }
  }
  double pr[3] = {0.0, 0.0, 0.0};                           // wavefront 0: the first tile of the panel, handed from phase 2 to the next phase 1
  bcr_acc4 cpre = {0.0, 0.0, 0.0, 0.0};                     // and the tile it updates there
#pragma unroll 1
  for (int kb = 0; kb < NBLK; ++kb) {
    // Seven block rows of right-hand sides in registers (B = 78): what a lane derives from its number - tile offsets, the
    // addresses of its right-hand-side column - is formed again at every step.  Hoisted out of this loop those values did not fit
    // the 128 registers next to seven accumulator tiles: they were spilled and came back from scratch memory in the middle of
    // a step (phase 1 took 8.7 k cycles a step instead of 4.2 k).
    int lane_v = lane;
    if constexpr (NBLK >= 7) asm volatile("" : "+v"(lane_v));
    const int lane = lane_v, lr = lane_v & 15, lk = lane_v >> 4;
    const int rhs_col = 16 * (rhs_ct >= 0 ? rhs_ct : 0) + lr;
    const bool rhs_cok = rhs_col < ncol;
    int rhs_xst;
    const int rhs_xoff = rhs_column(rhs_cok ? rhs_col : ncol - 1, rhs_xst);
    const int k0 = 12 * kb;
    const bool last = kb == NBLK - 1;
    const int nb = last ? B - k0 : 12;                      // this block: 12, or 6 at the end
    const int kn = k0 + nb;                                 // first unknown after this block
    // ---------------- phase 1 (block 0 was factored above, next to the prologue; it has no late updates)
#ifdef BA_BCR_PROFILE
    const long long q0 = clock64();
#endif
    // the update block column kb still owes to the panel of block kb - 1 (C -= panel panel^T, K = 12), one 16-row tile per task
    auto urgent_tile = [&](int t) {
      const int kp = k0 - 12, i0 = k0 + 16 * t;
      const int ao = (i0 + lr) * ld + kp + lk, bo = (k0 + lr) * ld + kp + lk, cb = (i0 + lk) * ld + k0 + lr;
      mfma_acc acc = {sm[cb], sm[cb + 4 * ld], sm[cb + 8 * ld], sm[cb + 12 * ld]};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao], -sm[bo], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao + 4], -sm[bo + 4], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[ao + 8], -sm[bo + 8], acc, 0, 0, 0);
      if (i0 + 16 <= B) {                                    // (wave-uniform) all 16 rows exist: one lane mask for the four stores
        if (lr < nb) {
#pragma unroll
          for (int v = 0; v < 4; ++v) sm[cb + 4 * v * ld] = acc[v];
        }
      } else {
        const int rl = lr < nb ? B - i0 - lk : 0;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (4 * v < rl) sm[cb + 4 * v * ld] = acc[v];
      }
    };
    if (kb == 0) {
    } else if (wave == 0) {
      __builtin_amdgcn_s_setprio(3);                        // the pivot chain is the critical path of the node
      bcr_urgent_tile0(sm, ld, B, k0, nb, lr, lk, pr, cpre);       // the tile that holds this diagonal block: by the chain's own
      lds_wave_sync();                                      // wavefront, no barrier between the update and the factor
#ifdef BA_BCR_PROFILE
      pst[5] += clock64() - q0;
#endif
#ifdef BA_BCR_PROFILE
      if (nb == 12) bcr_diag_block<12, false>(G, ld, dinv, bad, k0, lane, Li + 192 * (kb & 1), Idt, kb == 2 ? dtrace : nullptr);
#else
      if (nb == 12) bcr_diag_block<12, false>(G, ld, dinv, bad, k0, lane, Li + 192 * (kb & 1), Idt);
#endif
      else bcr_diag_block<6, false>(G, ld, dinv, bad, k0, lane, Li + 192 * (kb & 1), Idt);
      __builtin_amdgcn_s_setprio(0);
#ifdef BA_BCR_PROFILE
      pst[2] += clock64() - q0;
#endif
    } else if (rhs_ct >= 0) {
      // Right-hand sides: this wavefront owns 16 columns of them and keeps ALL their block rows in registers (racc[r], as they
      // leave the matrix core).  Step kb: finish block row kb-1, Y = L_pp^-1 racc[kb-1] (three MFMAs, stored in place), and take
      // it out of every block row below, racc[r] -= L[r, kb-1] Y (three independent MFMAs per row; L[r, kb-1] = rows of panel
      // kb-1) - all while wavefront 0 factors diagonal block kb, and well inside that time.  The form this replaces carried every
      // remaining row of the right-hand sides through LDS at every step, which made LDS traffic, not the pivot chain, the
      // length of phase 1 (4.5 k cycles against 3.0 k at the first step).
      if (!(role == 2 && 16 * rhs_ct >= k0 && 16 * rhs_ct + 15 < ncol - 1)) {      // (identity columns right of block kb-1: Y is zero)
        const double* Lp = Li + 192 * ((kb - 1) & 1);       // inverse of the previous diagonal block
        const int kp = k0 - 12;
        mfma_acc cur = racc[0];
#pragma unroll
        for (int r = 1; r < NBLK; ++r)
          if (r == kb - 1) cur = racc[r];
        // every operand of this step in ONE LDS round trip, before the first MFMA: the rows of panel kb-1 for all block rows
        // (those above kb are fetched for nothing - cheaper than a round trip per row, which is what a branch per row compiles to)
        if constexpr (NBLK <= 6) {
          const double l0 = Lp[lr * 12 + lk], l1 = Lp[lr * 12 + 4 + lk], l2 = Lp[lr * 12 + 8 + lk];
          double a[NBLK > 1 ? NBLK - 1 : 1][3];
  #pragma unroll
          for (int r = 1; r < NBLK; ++r) {
            const int arow = 12 * r + lr < B ? 12 * r + lr : B - 1;        // rows past the end repeat the last one (results never used)
            const int ao = arow * ld + kp + lk;
            a[r - 1][0] = sm[ao]; a[r - 1][1] = sm[ao + 4]; a[r - 1][2] = sm[ao + 8];
          }
  #pragma unroll
          for (int r = 1; r < NBLK; ++r) asm volatile("" : "+v"(a[r - 1][0]), "+v"(a[r - 1][1]), "+v"(a[r - 1][2]));
          mfma_acc y = {0.0, 0.0, 0.0, 0.0};
          y = __builtin_amdgcn_mfma_f64_16x16x4f64(l0, cur[0], y, 0, 0, 0);
          y = __builtin_amdgcn_mfma_f64_16x16x4f64(l1, cur[1], y, 0, 0, 0);
          y = __builtin_amdgcn_mfma_f64_16x16x4f64(l2, cur[2], y, 0, 0, 0);
          const int ro = rhs_xoff + __mul24(kp + lk, rhs_xst), r4 = 4 * rhs_xst;
          if (rhs_cok) { sm[ro] = y[0]; sm[ro + r4] = y[1]; sm[ro + 2 * r4] = y[2]; }
          const double ny0 = -y[0], ny1 = -y[1], ny2 = -y[2];
  #pragma unroll
          for (int r = 1; r < NBLK; ++r) {
            if (r >= kb) {                                       // (wave-uniform)
              racc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r - 1][0], ny0, racc[r], 0, 0, 0);
              racc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r - 1][1], ny1, racc[r], 0, 0, 0);
              racc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r - 1][2], ny2, racc[r], 0, 0, 0);
            }
          }
        } else {
          // (seven block rows, B = 78: the rows' operands two rows at a time, the next two in flight under the MFMAs of these - all
          //  at once do not fit next to seven accumulator tiles in 128 registers)
          constexpr int NRB = NBLK <= 6 ? (NBLK > 1 ? NBLK - 1 : 1) : 2;      // block rows per round trip
          constexpr int NBAT = NBLK > 1 ? (NBLK - 1 + NRB - 1) / NRB : 0;
          const double l0 = Lp[lr * 12 + lk], l1 = Lp[lr * 12 + 4 + lk], l2 = Lp[lr * 12 + 8 + lk];
          double a[2][NRB][3];
          auto fetch = [&](int bt, double (&dst)[NRB][3]) {
  #pragma unroll
            for (int q = 0; q < NRB; ++q) {
              const int r = 1 + bt * NRB + q;
              if (r < NBLK) {
                const int arow = 12 * r + lr < B ? 12 * r + lr : B - 1;      // rows past the end repeat the last one (results never used)
                const int ao = arow * ld + kp + lk;
                dst[q][0] = sm[ao]; dst[q][1] = sm[ao + 4]; dst[q][2] = sm[ao + 8];
              }
            }
          };
          if (NBAT > 0) fetch(0, a[0]);
  #pragma unroll
          for (int q = 0; q < NRB; ++q) asm volatile("" : "+v"(a[0][q][0]), "+v"(a[0][q][1]), "+v"(a[0][q][2]));
          mfma_acc y = {0.0, 0.0, 0.0, 0.0};
          y = __builtin_amdgcn_mfma_f64_16x16x4f64(l0, cur[0], y, 0, 0, 0);
          y = __builtin_amdgcn_mfma_f64_16x16x4f64(l1, cur[1], y, 0, 0, 0);
          y = __builtin_amdgcn_mfma_f64_16x16x4f64(l2, cur[2], y, 0, 0, 0);
          const int ro = rhs_xoff + __mul24(kp + lk, rhs_xst), r4 = 4 * rhs_xst;
          if (rhs_cok) { sm[ro] = y[0]; sm[ro + r4] = y[1]; sm[ro + 2 * r4] = y[2]; }
          const double ny0 = -y[0], ny1 = -y[1], ny2 = -y[2];
  #pragma unroll
          for (int bt = 0; bt < NBAT; ++bt) {
            if (bt + 1 < NBAT) fetch(bt + 1, a[(bt + 1) & 1]);
  #pragma unroll
            for (int q = 0; q < NRB; ++q) {
              const int r = 1 + bt * NRB + q;
              if (r < NBLK && r >= kb) {                           // (wave-uniform)
                racc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bt & 1][q][0], ny0, racc[r], 0, 0, 0);
                racc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bt & 1][q][1], ny1, racc[r], 0, 0, 0);
                racc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bt & 1][q][2], ny2, racc[r], 0, 0, 0);
              }
            }
          }
        }
      }
    } else if (myslot >= 0 && myslot < nworkers) {
      const int kp = k0 - 12;
      const int ngt = (B - kn + 15) >> 4;                   // column tiles of the trailing matrix right of this block
      const int nsu = ((B - k0 + 15) >> 4) - 1;             // tiles of block column kb below the one wavefront 0 takes
      for (int task = myslot; task < ngt + nsu; task += nworkers) {
        if (task >= ngt) { urgent_tile(task - ngt + 1); continue; }
        // lower tiles of D right of block column kb: C -= A B with K = 12, A = panel of block kp (16 rows of the tile), B = panel^T
        const int gtile = task, c0 = kn + 16 * gtile;
        const int t0 = gtile, t1 = (B - kn + 15) >> 4;
        const bool cok = c0 + lr < B, full = c0 + 15 < B;
        const int bo = (c0 + lr) * ld + kp + lk;
        const int co = lk * ld + c0 + lr, c4 = 4 * ld;
        const double nb0 = -sm[bo], nb1 = -sm[bo + 4], nb2 = -sm[bo + 8];      // the NEGATED B operand of the three k-steps
        int ao = (kn + 16 * t0 + lr) * ld + kp + lk;           // A entry of this lane in the first tile
        int cb = co + (kn + 16 * t0) * ld;                     // first accumulator entry of this lane
        int rows = B - (kn + 16 * t0);                         // rows left from the top of the tile
        double a0 = sm[ao], a1 = sm[ao + 4], a2 = sm[ao + 8];
        mfma_acc acc = {sm[cb], sm[cb + c4], sm[cb + 2 * c4], sm[cb + 3 * c4]};
        for (int t = t0; t < t1; ++t) {
          const double a0c = a0, a1c = a1, a2c = a2;
          mfma_acc accc = acc;
          const int cbc = cb, rc = rows;
          if (t + 1 < t1) {                                    // the next tile's operands are in flight
            ao += 16 * ld; cb += 16 * ld; rows -= 16;
            a0 = sm[ao]; a1 = sm[ao + 4]; a2 = sm[ao + 8];
            acc = mfma_acc{sm[cb], sm[cb + c4], sm[cb + 2 * c4], sm[cb + 3 * c4]};
          }
          accc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0c, nb0, accc, 0, 0, 0);
          accc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1c, nb1, accc, 0, 0, 0);
          accc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2c, nb2, accc, 0, 0, 0);
          if (full && rc >= 16) {                              // wave-uniform: the whole tile exists
#pragma unroll
            for (int v = 0; v < 4; ++v) sm[cbc + v * c4] = accc[v];
          } else {
            const int rl = cok ? rc - lk : 0;                  // rows v with 4 v < rl exist
#pragma unroll
            for (int v = 0; v < 4; ++v)
              if (4 * v < rl) sm[cbc + v * c4] = accc[v];      // (the strict upper part of diagonal tiles is never read)
          }
        }
      }
    }
#ifdef BA_BCR_PROFILE
    if (i == 3 * s - 1 && s == 2 && role == 0 && kb == BA_BCR_TRACE_KB && lane == 0) info[44 + wave] = (int)(clock64() - q0);      // phase 1 of one block step, per wavefront
#endif
    if (kb > 0) __syncthreads();
#ifdef BA_BCR_PROFILE
    const long long q1 = clock64();
    pst[3] += q1 - q0;
#endif
    // ---------------- phase 2: panel, rows below the diagonal block: X = A L_kk^-T, one 16-row tile per wavefront
    constexpr int NPT = (B - 12 + 15) / 16 > 4 ? (B - 12 + 15) / 16 : 4;      // tiles of the widest panel (B = 78: five - wavefront 4, idle in phase 1, takes the fifth)
    if (wave < NPT) {
      if (kn + 16 * wave < B) {                              // (nb == 12 here)
        if (wave == 0) cpre = bcr_prefetch_tile0(sm, ld, kn, lr, lk);
        bcr_panel_tile(G, ld, B, k0, kn + 16 * wave, Li + 192 * (kb & 1), lr, lk, pr);
      }
    }
#if defined(BA_BCR_PROFILE) && defined(BA_BCR_TRACE_PH2)
    if (i == 3 * s - 1 && s == 2 && role == 0 && kb == BA_BCR_TRACE_KB && lane == 0) info[44 + wave] = (int)(clock64() - q1);      // phase 2 of one block step, per wavefront
#endif
    __syncthreads();
#ifdef BA_BCR_PROFILE
    const long long q2 = clock64();
    pst[4] += q2 - q1;
#endif
  }
#ifdef BA_BCR_PROFILE
  const long long pt2 = clock64();
#endif
  if (rhs_ct >= 0) {
    // the last block row of the right-hand sides: Y = L_pp^-1 racc[NBLK - 1] (nothing below it)
    constexpr int KL = 12 * (NBLK - 1), NL = B - KL;          // last block: start and size (12 or 6)
    const double* Ll = Li + 192 * ((NBLK - 1) & 1);         // inverse of the last diagonal block
    const mfma_acc cur = racc[NBLK - 1];
    const int ro = rhs_xoff + __mul24(KL + lk, rhs_xst), r4 = 4 * rhs_xst;
    mfma_acc y = {0.0, 0.0, 0.0, 0.0};
    y = __builtin_amdgcn_mfma_f64_16x16x4f64(Ll[lr * 12 + lk], cur[0], y, 0, 0, 0);
    if (NL == 12) {
      y = __builtin_amdgcn_mfma_f64_16x16x4f64(Ll[lr * 12 + 4 + lk], cur[1], y, 0, 0, 0);
      y = __builtin_amdgcn_mfma_f64_16x16x4f64(Ll[lr * 12 + 8 + lk], cur[2], y, 0, 0, 0);
      if (rhs_cok) { sm[ro] = y[0]; sm[ro + r4] = y[1]; sm[ro + 2 * r4] = y[2]; }
    } else {                                                 // 6 rows: k = 4, 5 of the second step only (what Ll holds beyond is finite)
      y = __builtin_amdgcn_mfma_f64_16x16x4f64(Ll[lr * 12 + 4 + lk], lk < 2 ? cur[1] : 0.0, y, 0, 0, 0);
      if (rhs_cok) { sm[ro] = y[0]; if (lk < 2) sm[ro + r4] = y[1]; }
    }
  }
  __syncthreads();
  BA_TLINE(4);
  if (*bad) {
    if (tid == 0 && role == 2) atomicMax(info, i * B + *bad);
    return false;
  }
#ifdef BA_BCR_PROFILE
  const long long pt3 = clock64();
  pst[6] = pt3 - pt2;
#endif

  if (role < 2) {
    if constexpr (FUSED && BA_BCR_STORE_FIRST) {
      // the factor leaves for memory first: the next level multiplies it (and the back-substitution reads it)
      double* out = (role == 0 ? Pm : Qm) + (size_t)i * BB;
      if constexpr (BA_BCR_WIDE_HANDOVER) {
        for (int q = tid; q < B * B / 2; q += kBcrElimThreads) {
          const int e = 2 * q, rr = e / B, cc = e - rr * B;
          bcr_st16(out + e, bcr_d2{R[rr * ld + cc], R[rr * ld + cc + 1]});
        }
      } else {
        for (int e = tid; e < B * B; e += kBcrElimThreads) {
          const int rr = e / B, cc = e - rr * B;
          bcr_st<true>(out + e, R[rr * ld + cc]);
        }
      }
    }
    // ---- this role's neighbour update: D_nb -= R^T R (lower tiles), f_nb -= R^T g; R is kept for the back-substitution
    const int nbr = role == 0 ? l : r;
    constexpr int NT = (B + 15) / 16, KST = (B + 3) / 4;
    constexpr int EB = 16 * (NT - 1), EE = B - EB, RB = (EE + 3) / 4;
    constexpr bool kEdge4 = NT > 1 && EE <= 8;               // the last tile row as chains of v_mfma_f64_4x4x4 (see the prologue)
    constexpr int NTF = kEdge4 ? NT - 1 : NT, NSYM = NTF * (NTF + 1) / 2;
    if (kEdge4) {
      constexpr int NS = NTF * RB + 1, base = NSYM < 16 ? NSYM : 0, nsw = 16 - base;
      double* dst = Dm + (size_t)nbr * BB;
      if (wave >= base)
        for (int q = wave - base; q < NS; q += nsw) {
          int acol, bcol;
          bcr_edge_task(q, NTF, RB, false, EB, lane, acol, bcol);
          const double v = bcr_mfma4_blocks<B>(R, R, ld, acol, bcol, lane);
          const int row = acol + (lane >> 4), col = bcol + (lane & 3);
          if (row < B && col <= row) atomic_add_f64(dst + (size_t)row * B + col, -v);
        }
    }
    for (int task = wave; task < NSYM; task += kBcrElimThreads / 64) {
      int ti, tj;
      tri_decode(task, NTF, tj, ti);                         // tj <= ti: D is only ever read in its lower triangle
      const double* A = R + 16 * ti + lr;
      const double* Bm = R + 16 * tj + lr;
      double ar[KST], br[KST];
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) {
        const int k = 4 * ks + lk;
        const bool in = 4 * ks + 3 < B || k < B;
        const int kc = in ? k : 0;
        const double a_ = A[kc * ld], b_ = Bm[kc * ld];
        ar[ks] = in ? a_ : 0.0; br[ks] = in ? b_ : 0.0;
      }
      mfma_acc acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KST; ks += 2) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[ks], br[ks], acc, 0, 0, 0);
        if (ks + 1 < KST) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[ks + 1], br[ks + 1], acc2, 0, 0, 0);
      }
      double* dst = Dm + (size_t)nbr * BB;
      const int col = 16 * tj + lr;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 16 * ti + lk + 4 * v;
        if (row < B && col <= row) atomic_add_f64(dst + (size_t)row * B + col, -(acc[v] + acc2[v]));
      }
    }
    for (int c = kBcrElimThreads - 1 - tid; c < B; c += kBcrElimThreads) {       // the last wavefronts have fewer tiles
      double acc = 0.0;
      for (int k = 0; k < B; ++k) acc += R[k * ld + c] * g[k];
      atomic_add_f64(fm + (size_t)nbr * B + c, -acc);
    }
#ifdef BA_BCR_PROFILE
    if (BA_BCR_TRACE_KB == 99 && i == 3 * s - 1 && s == 2 && role == 0 && lane == 0) info[44 + wave] = (int)(clock64() - pt3);      // products, per wavefront
#endif
    if constexpr (!(FUSED && BA_BCR_STORE_FIRST)) {
      double* out = (role == 0 ? Pm : Qm) + (size_t)i * BB;
      for (int e = tid; e < B * B; e += kBcrElimThreads) {
        const int rr = e / B, cc = e - rr * B;
        bcr_st<FUSED>(out + e, R[rr * ld + cc]);
      }
    }
  } else {
    for (int e = tid; e < B * B; e += kBcrElimThreads) {
      const int rr = e / B, cc = e - rr * B;
      bcr_st<FUSED>(Gi + (size_t)i * BB + e, cc <= rr ? R[rr * ld + cc] : 0.0);
    }
    for (int e = tid; e < B; e += kBcrElimThreads) bcr_st<FUSED>(gm + (size_t)i * B + e, g[e]);
    if (!haveL && !haveR) {
      // the root of the elimination tree: x_i = G^-T g right here (its back-substitution launch is skipped by the host)
      for (int task = tid; task < 4 * B; task += kBcrElimThreads) {
        const int m = task >> 2, q4 = task & 3;
        double acc = 0.0;
        for (int k = m + q4; k < B; k += 4) acc += R[k * ld + m] * g[k];
        acc += dpp_pair<0xB1>(acc);
        acc += dpp_pair<0x4E>(acc);
        if (q4 == 0) bcr_st<FUSED>(xout + (size_t)i * B + m, acc);
      }
    }
  }
#ifdef BA_BCR_PROFILE
  __syncthreads();
  if (tid == 0 && i == 3 * s - 1 && s == 2) {
    int* o = info + 8 + 10 * role;              // [load, prologue, diag factor (wave 0), phase 1, phase 2, phase 3, last rhs block, products + store]
    for (int q = 0; q < 7; ++q) o[q] = (int)pst[q];
    o[7] = (int)(clock64() - pt3);
    if (role == 0) { info[40] = (int)dtrace[0]; info[41] = (int)dtrace[1]; info[42] = (int)dtrace[2]; }
  }
#endif
  return true;
}

template <int HB>
__global__ __launch_bounds__(kBcrElimThreads) void k_bcr_eliminate_split(int N, int s, double* __restrict__ Dm,
                                                                     const double* __restrict__ Um, double* __restrict__ fm,
                                                                     double* __restrict__ Pm, double* __restrict__ Qm,
                                                                     double* __restrict__ Gi, double* __restrict__ gm,
                                                                     int* __restrict__ info, double* __restrict__ xout) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int role = blockIdx.y;
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N) return;
  if ((role == 0 && i - s < 0) || (role == 1 && i + s >= N)) return;       // no such neighbour: nothing to do for this role
  bcr_split_node<HB, false>(sm, N, s, i, role, Dm, Um, fm, Pm, Qm, Gi, gm, info, xout, BcrDeps{{nullptr, nullptr}, {nullptr, nullptr}, nullptr, info});
}

// --------------------------------------------------------------------------
// ALL back-substitution levels in one launch, for systems whose N nodes are resident at once (N <= compute units).
// What a level's launch spends most of its ~5 us on - staging P, Q, G^-1 of its nodes into LDS - does not depend on
// the levels above it, so here every node's workgroup stages its matrices at once, then waits for x_l and x_r to be
// PUBLISHED by the nodes above (k_bcr_assemble marks every entry of x "not yet"; a node polls the entries it needs),
// forms x_i = G^-T (g - P x_l - Q x_r) and stores it.  The root was solved by its elimination kernel.
// Six dependent launches of 4.7 us become one of ~2 us + 6 hand-overs.
// --------------------------------------------------------------------------
__device__ __forceinline__ double bcr_wait_value(const double* p, int* status) {
  // relaxed agent-scope polls of the DATA (write-through stores, cache-bypassing loads): one memory round trip per
  // hand-over.  Acquire loads / release fences here invalidate and write back whole L2s: 20 - 30 us per hand-over
  // with a hundred workgroups doing it at once (measured), and a separate ready flag costs a second round trip.
  // The spin is bounded (about a second): whatever goes wrong upstream that nobody has thought of must end in a failed solve
  // (status word > 0: the caller rejects the trial / falls back), not in a GPU that never comes back.  The "not yet" NaN this
  // returns then turns into ordinary NaNs downstream, which nobody waits on.
  // A FAILED elimination (a pivot that is not positive: status word set by some node of this or an earlier launch) leaves
  // solution entries unwritten for good - the root never solves: nobody may wait for them, so every poll looks at the status
  // word as well.
  double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int spins = 0; __double_as_longlong(v) == kBcrNotYet; ++spins) {
    if (spins >= kBcrMaxSpins) { atomicMax(status, kBcrTimedOut); break; }
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    __builtin_amdgcn_s_sleep(2);
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return v;
}

// node i of the back-substitution: stage P_i, Q_i, G_i^-1, g_i (COHERENT: they were written by workgroups of this same launch -
// relaxed agent-scope loads), wait for x_l, x_r, form and publish x_i
template <bool COHERENT, int U = (kBcrSplitMaxHB * 6 * kBcrSplitMaxHB * 6 + kBcrElimThreads - 1) / kBcrElimThreads>      // (U: entries of a B x B matrix per thread)
__device__ __forceinline__ void bcr_backsolve_node(double* __restrict__ sm, int N, int B, int i, const double* gm,
                                                   const double* Pm, const double* Qm, const double* Gi, double* x, int* status) {
  const int ld = B + 1;
  double* MP = sm;                       // [B][ld] P
  double* MQ = MP + (size_t)B * ld;      // [B][ld] Q
  double* MG = MQ + (size_t)B * ld;      // [B][ld] G^-1
  double* w = MG + (size_t)B * ld;       // [B]
  double* xl = w + B;                    // [B]
  double* xr = xl + B;                   // [B]
  const int tid = threadIdx.x;
  const int s = (i + 1) & -(i + 1);      // the level that eliminated node i: i = s (2 k + 1) - 1
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  if (!haveL && !haveR) return;          // the root: its elimination kernel wrote x_i
  const size_t BB = (size_t)B * B;
  {
    double vp[U], vq[U], vg[U];
    const double wv = tid < B ? bcr_ld<COHERENT>(gm + (size_t)i * B + tid) : 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + kBcrElimThreads * u;
      const bool in = e < B * B;
      vp[u] = (in && haveL) ? bcr_ld<COHERENT>(Pm + (size_t)i * BB + e) : 0.0;
      vq[u] = (in && haveR) ? bcr_ld<COHERENT>(Qm + (size_t)i * BB + e) : 0.0;
      vg[u] = in ? bcr_ld<COHERENT>(Gi + (size_t)i * BB + e) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + kBcrElimThreads * u;
      if (e < B * B) {
        const int rr = e / B, cc = e - rr * B;
        MP[rr * ld + cc] = vp[u]; MQ[rr * ld + cc] = vq[u]; MG[rr * ld + cc] = vg[u];
      }
    }
    if (tid < B) w[tid] = wv;
  }
  // x_l, x_r from the nodes above, as soon as they are there (k_bcr_assemble marked every entry "not yet")
  if (tid < B) {
    xl[tid] = haveL ? bcr_wait_value(x + (size_t)l * B + tid, status) : 0.0;
    xr[tid] = haveR ? bcr_wait_value(x + (size_t)r * B + tid, status) : 0.0;
  }
  __syncthreads();
  // the two matrix-vector products of a hand-over with 16 lanes per row (the whole workgroup busy for four terms each
  // instead of a fifth of it for fourteen: every cycle here is on the chain of six dependent hand-overs)
  for (int base = 0; base < 16 * B; base += kBcrElimThreads) {         // w -= P xl + Q xr
    const int task = base + tid, kraw = task >> 4, q = task & 15;
    const int k = kraw < B ? kraw : B - 1;                               // (rows past the end repeat the last: DPP sources must be live lanes)
    double acc = 0.0;
    for (int c = q; c < B; c += 16) acc += MP[k * ld + c] * xl[c] + MQ[k * ld + c] * xr[c];
    acc += dpp_pair<0xB1>(acc);
    acc += dpp_pair<0x4E>(acc);
    acc += dpp_pair<0x141>(acc);
    acc += dpp_pair<0x140>(acc);
    if (q == 0 && kraw < B) w[k] -= acc;
  }
  __syncthreads();
  for (int base = 0; base < 16 * B; base += kBcrElimThreads) {         // x = (G^-1)^T w
    const int task = base + tid, mraw = task >> 4, q = task & 15;
    const int m = mraw < B ? mraw : B - 1;
    double acc = 0.0;
    for (int k = m + q; k < B; k += 16) acc += MG[k * ld + m] * w[k];
    acc += dpp_pair<0xB1>(acc);
    acc += dpp_pair<0x4E>(acc);
    acc += dpp_pair<0x141>(acc);
    acc += dpp_pair<0x140>(acc);
    if (q == 0 && mraw < B) __hip_atomic_store(x + (size_t)i * B + m, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#ifndef BA_BCR_TEMPLATES_ONLY      // (non-template kernels: compiled by ba_solve.hip alone)
__global__ __launch_bounds__(kBcrElimThreads) void k_bcr_backsolve_fused(int N, int B, const double* __restrict__ gm_split,
                                                                         const double* __restrict__ gm_one, int split_stride,
                                                                         const double* __restrict__ Pm,
                                                                         const double* __restrict__ Qm,
                                                                         const double* __restrict__ Gi, double* x,
                                                                         const int* __restrict__ order, int* ticket) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x;
  // NO DEADLOCK, whatever else shares the GPU: a workgroup takes a ticket when it STARTS and works on order[ticket],
  // and `order` lists the nodes level by level from the root down - so a node only ever waits for nodes whose
  // workgroups have started before it (they are resident or done: started workgroups are never preempted).  Waiting
  // by blockIdx instead hung two processes on one GPU, each kernel holding compute units the other's parents needed.
  // a failed elimination (matrix not positive definite: status word set by an earlier launch) leaves solution entries
  // unwritten: nobody may wait for them
  // (ONE thread reads the status word and takes the ticket, the workgroup decides on what it broadcast: every thread reading
  // the word for itself could split the workgroup when another one sets it in between)
  int* my_ticket = reinterpret_cast<int*>(sm + (size_t)3 * B * (B + 1) + 3 * B);      // (in the dynamic area: the kernel may ask for all 160 KB of it)
  if (tid == 0) {
    const int st = __hip_atomic_load(ticket - kBcrTicketWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    my_ticket[1] = st;
    my_ticket[0] = st != 0 ? 0 : atomicAdd(ticket, 1);
  }
  __syncthreads();
  if (my_ticket[1] != 0) return;
  const int i = order[my_ticket[0]];
  const int s = (i + 1) & -(i + 1);
  __syncthreads();
  bcr_backsolve_node<false>(sm, N, B, i, s >= split_stride ? gm_split : gm_one, Pm, Qm, Gi, x, ticket - kBcrTicketWord);
}
#endif

// --------------------------------------------------------------------------
// ALL split elimination levels in ONE launch.  One workgroup per (node, role) of every level from stride s_first up; a
// workgroup takes a ticket when it STARTS and works on work[ticket], `work` listing the (node, role) pairs level by level
// from the leaves up - so a workgroup only ever waits for workgroups that started before it (resident or done; the same
// argument as in k_bcr_backsolve_fused: no deadlock whatever else shares the GPU, and levels wider than the chip simply run
// in rounds).  What a (node i, stride s) needs from the level below arrives through global memory:
//     D_i, f_i  final once every node that adds to them is done:  (i - s/2, right role) and (i + s', left role), s' the
//               largest stride <= s/2 with i + s' < N  (those two waited, each in its turn, for the smaller strides on its side)
//     P_j, Q_j  of the node eliminated between i and its neighbour (j = i -/+ s/2, both roles): the coupling of the left /
//               right role
// Every workgroup ends by publishing done[node][role] AFTER its stores and atomics have been acknowledged (s_waitcnt
// vmcnt(0), barrier, one relaxed agent-scope store); a consumer polls the (at most three) words it needs with one lane each,
// then loads.  All of that data moves with relaxed agent-scope accesses: no fences (an acquire / release at agent scope
// invalidates / writes back a whole L2: DESIGN.md).  The wait is bounded (~1 s): status word, not a hung GPU.
// What the launch boundaries cost before: the set-up of a node (LDS tables, identity right-hand sides) now happens while it
// waits, the factors leave for memory under the neighbour products, a level's early finishers hand over at once instead
// of at the end of the launch, and workgroups of the upper levels never queue behind a kernel boundary.
// --------------------------------------------------------------------------
constexpr int kBcrElimTicketWord = kBcrTicketWord - 1;         // info[60]: tickets of k_bcr_eliminate_fused (k_bcr_assemble clears it)

template <int HB>
__global__ __launch_bounds__(kBcrElimThreads) void k_bcr_eliminate_fused(int N, int s_first, double* __restrict__ Dm,
                                                                     const double* __restrict__ Um, double* __restrict__ fm,
                                                                     double* Pm, double* Qm, double* __restrict__ Gi,
                                                                     double* __restrict__ gm, int* __restrict__ info,
                                                                     double* __restrict__ xout, const int* __restrict__ work,
                                                                     int* done, long long* __restrict__ trace) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  int* my_work = reinterpret_cast<int*>(sm + (bcr_split_lds_bytes(6 * HB) / sizeof(double)) - 2);      // (the last two doubles of the area: nothing else lives there)
  if (threadIdx.x == 0) {
    const int ticket = atomicAdd(info + kBcrElimTicketWord, 1);
    my_work[0] = work[ticket];
    my_work[1] = ticket;
  }
  __syncthreads();
  const int item = __builtin_amdgcn_readfirstlane(*my_work);
  if ((item & 3) == 3) {
    // a BACK-SUBSTITUTION item (they follow the elimination items in the work list, root down: their tickets are later than
    // everything they wait for): node i's factors are in memory once its three roles have said so - words 2, 3 (left / right
    // role) and 0 (inverse role) of done[4 i ...]; nodes eliminated by an earlier launch (stride < s_first) wait for nothing
    const int i = item >> 2, s = (i + 1) & -(i + 1);
    __syncthreads();
    if (s >= s_first) {
      const int* word = threadIdx.x == 0 ? done + 4 * i : threadIdx.x == 1 ? (i - s >= 0 ? done + 4 * i + 2 : nullptr)
                                                                            : (i + s < N ? done + 4 * i + 3 : nullptr);
      if (threadIdx.x < 3 && word) bcr_wait_done(word, info);
    }
    __syncthreads();
    constexpr int UB = HB <= kBcrMaxHB ? (kBcrMaxHB * 6 * kBcrMaxHB * 6 + kBcrElimThreads - 1) / kBcrElimThreads : (36 * HB * HB + kBcrElimThreads - 1) / kBcrElimThreads;
    if (s >= s_first) bcr_backsolve_node<true, UB>(sm, N, 6 * HB, i, gm, Pm, Qm, Gi, xout, info);
    else bcr_backsolve_node<false, UB>(sm, N, 6 * HB, i, fm, Pm, Qm, Gi, xout, info);
    return;
  }
#ifdef BA_BCR_PROFILE
  // time line of this workgroup (100 MHz wall clock, the same on every compute unit): [start, producers done, loaded, coupling
  // formed, factored, handed on, 4 node + role, XCD] at trace[8 ticket]
  long long* tline = trace ? trace + 8 * (size_t)my_work[1] : nullptr;
  if (tline && threadIdx.x == 0) {
    tline[0] = wall_clock64();
    tline[6] = item;
    tline[7] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID
  }
#else
  long long* tline = nullptr;
#endif
  const int i = item >> 2, role = item & 3;
  const int s = (i + 1) & -(i + 1);             // the level that eliminates node i: i = s (2 k + 1) - 1
  __syncthreads();                              // (my_work is read; the body may use the whole area)
  // the words of done[4 node + k]: k = 0 / 1: the left / right role's factor (P / Q) is in memory; k = 2 / 3: the left / right
  // role has added to its neighbour's D and f
  BcrDeps dep{{nullptr, nullptr}, {nullptr, nullptr}, role < 2 ? done + 4 * i + role : nullptr, info};
  if (s > s_first) {
    const int h = s >> 1;
    int sr = h;
    while (sr >= s_first && i + sr >= N) sr >>= 1;
    const int right = sr >= s_first ? i + sr : -1;        // the largest-stride node that adds to D_i from the right (its left role)
    dep.d[0] = done + 4 * (i - h) + 3;                     // ... and from the left: the right role of i - s/2
    dep.d[1] = right >= 0 ? done + 4 * right + 2 : nullptr;
    if (role < 2) {
      const int j = role == 0 ? i - h : i + h;             // the node eliminated between i and this role's neighbour
      dep.pq[0] = done + 4 * j + 2;
      dep.pq[1] = done + 4 * j + 3;
    }
  }
  const bool ok = bcr_split_node<HB, true>(sm, N, s, i, role, Dm, Um, fm, Pm, Qm, Gi, gm, info, xout, dep, tline);
  // publish: everything this workgroup stored or added is acknowledged by memory before the word says so
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && role < 2) {
    if (!ok) __hip_atomic_store(done + 4 * i + role, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (not positive definite: nobody may wait)
    __hip_atomic_store(done + 4 * i + 2 + role, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0 && role == 2)      // the inverse role: G^-1 and g are in memory (the back-substitution items wait for this)
    __hip_atomic_store(done + 4 * i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef BA_BCR_PROFILE
  if (tline && threadIdx.x == 0) tline[5] = wall_clock64();
#endif
}

// --------------------------------------------------------------------------
// The elimination level for reduced systems that are NOT positive definite - the reference solves its reduced system by LU
// (numpy.linalg.solve = gesv, bundle_adjuster.py:302-305), which also "succeeds" on a symmetric matrix that the round-off of
// a nearly singular scene (the damping decayed to 1e-10 at the noise floor, the free scale of a monocular reconstruction) has
// pushed indefinite; a Cholesky factorisation reports such a matrix, it cannot solve it.  Here a node is eliminated by
// Gauss-Jordan with PARTIAL PIVOTING inside its B x B block:  [Zl | Zr | z] = D_i^-1 [T_il | T_ir | f_i]  with row
// interchanges among the B rows of the node; the neighbours take the same Schur-complement updates as in the Cholesky kernels,
//     D_l -= T_il^T Zl,  D_r -= T_ir^T Zr,  T[l,r] = -T_il^T Zr,  f_l -= T_il^T z,  f_r -= T_ir^T z,
// and the back-substitution kernels run unchanged on  P = Zl, Q = Zr, g = z, G^-1 = I  (x_i = z - Zl x_l - Zr x_r).
// This is block LU in the cyclic-reduction order with pivoting restricted to a node: it needs every D_i it meets to be
// non-singular (status word = first zero pivot, like LAPACK's info), not positive.  Plain vector code, one workgroup per node,
// a launch per level: it only runs after the Cholesky solve has failed (~0.2 ms for 1000 cameras against 70 ms for
// rocSOLVER's LU of the flattened 5994 x 5994 system).
// --------------------------------------------------------------------------
__host__ __device__ inline size_t bcr_lu_lds_bytes(int B) { return ((size_t)B * (3 * B + 1) + 2 * B + 16) * sizeof(double); }

template <int HB>
__global__ __launch_bounds__(kBcrElimThreads) void k_bcr_eliminate_lu(int N, int s, double* __restrict__ Dm, double* __restrict__ Um,
                                                                      double* __restrict__ fm, double* __restrict__ Pm,
                                                                      double* __restrict__ Qm, double* __restrict__ Gi,
                                                                      int* __restrict__ info, double* __restrict__ xout) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int B = 6 * HB, W = 3 * B + 1;                  // row of the augmented matrix: D | T_il | T_ir | f
  double* A = sm;                                           // [B][W]
  double* colk = A + (size_t)B * W;                         // [B]  column k of the current step
  int* rowof = reinterpret_cast<int*>(colk + B);            // [B]  row that was pivot of column k; then [B] used flags
  int* used = rowof + B;
  int* piv = used + B;                                      // [2]  pivot row of this step, singular flag
  const int tid = threadIdx.x;
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N) return;
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  constexpr size_t BB = (size_t)B * B;
  // T[i,l] = T[l,i]^T and T[i,r] come from U of the level below (k_bcr_eliminate_lu writes T[l,r] into U[l] like k_bcr_eliminate)
  for (int e = tid; e < B * B; e += kBcrElimThreads) {
    const int rr = e / B, cc = e - rr * B;
    A[rr * W + cc] = rr >= cc ? Dm[(size_t)i * BB + e] : Dm[(size_t)i * BB + cc * B + rr];       // D is kept in its lower triangle
    A[rr * W + B + cc] = haveL ? Um[(size_t)l * BB + cc * B + rr] : 0.0;
    A[rr * W + 2 * B + cc] = haveR ? Um[(size_t)i * BB + e] : 0.0;
  }
  for (int e = tid; e < B; e += kBcrElimThreads) { A[e * W + 3 * B] = fm[(size_t)i * B + e]; used[e] = 0; }
  if (tid == 0) piv[1] = 0;
  __syncthreads();
  for (int k = 0; k < B; ++k) {
    // pivot: the largest |A[r][k]| among the rows that have not been a pivot row yet (first wavefront; B <= 66: two rows per lane)
    if (tid < 64) {
      double best = -1.0; int brow = -1;
      for (int rr = tid; rr < B; rr += 64) {
        const double v = used[rr] ? -1.0 : fabs(A[rr * W + k]);
        if (v > best) { best = v; brow = rr; }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int orow = __shfl_xor(brow, off);
        if (ob > best || (ob == best && orow >= 0 && (brow < 0 || orow < brow))) { best = ob; brow = orow; }
      }
      if (tid == 0) {
        piv[0] = brow;
        if (!(best > 0.0) || !(best < __builtin_huge_val())) piv[1] = k + 1;       // exactly singular (or NaN): LAPACK's info
        else { used[brow] = 1; rowof[k] = brow; }
      }
    }
    __syncthreads();
    if (piv[1]) break;
    const int p = piv[0];
    const double inv = 1.0 / A[p * W + k];
    for (int rr = tid; rr < B; rr += kBcrElimThreads) colk[rr] = rr == p ? 0.0 : A[rr * W + k];
    __syncthreads();
    // scale the pivot row (columns > k only: the rest is never read again), then eliminate column k from every other row
    for (int c = k + 1 + tid; c < W; c += kBcrElimThreads) A[p * W + c] *= inv;
    __syncthreads();
    const int ncol = W - (k + 1);
    for (int e = tid; e < B * ncol; e += kBcrElimThreads) {
      const int rr = e / ncol, c = k + 1 + (e - rr * ncol);
      A[rr * W + c] -= colk[rr] * A[p * W + c];             // (colk[p] = 0: the pivot row stays)
    }
    __syncthreads();
  }
  if (piv[1]) {
    if (tid == 0) atomicMax(info, i * B + piv[1]);
    return;
  }
  // unknown k sits in row rowof[k]:  Zl = rows of columns B..2B-1, Zr = 2B..3B-1, z = column 3B
  // ---- neighbour updates with the ORIGINAL couplings (re-read from U: the copies in A are overwritten)
  for (int e = tid; e < B * B; e += kBcrElimThreads) {
    const int a = e / B, b = e - a * B;                     // entry (a, b) of the B x B results
    double sll = 0.0, srr = 0.0, slr = 0.0;
    for (int k = 0; k < B; ++k) {
      const int row = rowof[k];
      const double til = haveL ? Um[(size_t)l * BB + (size_t)a * B + k] : 0.0;      // T_il[k][a] = U_l[a][k]
      const double tir = haveR ? Um[(size_t)i * BB + (size_t)k * B + a] : 0.0;      // T_ir[k][a]
      sll += til * A[row * W + B + b];
      srr += tir * A[row * W + 2 * B + b];
      slr += til * A[row * W + 2 * B + b];
    }
    if (haveL && b <= a) atomic_add_f64(Dm + (size_t)l * BB + e, -sll);
    if (haveR && b <= a) atomic_add_f64(Dm + (size_t)r * BB + e, -srr);
    A[a * W + b] = slr;             // T[l,r] = -slr goes to U[l] below, once everybody has read U[l]; until then in the D part of A (free now)
  }
  for (int c = tid; c < 2 * B; c += kBcrElimThreads) {
    const bool left = c < B;
    const int a = left ? c : c - B;
    if ((left && !haveL) || (!left && !haveR)) continue;
    double acc = 0.0;
    for (int k = 0; k < B; ++k) {
      const double t = left ? Um[(size_t)l * BB + (size_t)a * B + k] : Um[(size_t)i * BB + (size_t)k * B + a];
      acc += t * A[rowof[k] * W + 3 * B];
    }
    atomic_add_f64(fm + (size_t)(left ? l : r) * B + a, -acc);
  }
  __syncthreads();                                          // every read of U[l] is done
  // ---- what the back-substitution needs: P = Zl, Q = Zr, G^-1 = I, g = z; the new coupling T[l,r] = -T_il^T Zr into U[l]
  for (int e = tid; e < B * B; e += kBcrElimThreads) {
    const int k = e / B, c = e - k * B;
    const int row = rowof[k];
    Pm[(size_t)i * BB + e] = A[row * W + B + c];
    Qm[(size_t)i * BB + e] = A[row * W + 2 * B + c];
    Gi[(size_t)i * BB + e] = k == c ? 1.0 : 0.0;
    if (haveL && haveR) Um[(size_t)l * BB + e] = -A[k * W + c];       // (the stash: entry (k, c) of T_il^T Zr)
  }
  __syncthreads();
  for (int k = tid; k < B; k += kBcrElimThreads) {
    const double z = A[rowof[k] * W + 3 * B];
    fm[(size_t)i * B + k] = z;
    if (!haveL && !haveR) xout[(size_t)i * B + k] = z;      // the root: x_i = z
  }
}

#ifndef BA_BCR_TEMPLATES_ONLY      // (non-template kernels: compiled by ba_solve.hip alone)
// One back-substitution level: x_i = G^-T (g - P x_l - Q x_r) for the nodes of that level.
// P, Q and G^-1 are staged into LDS in one round trip; the two matrix-vector products use
// four lanes per row, the last one four lanes per column.
__global__ __launch_bounds__(kBcrElimThreads) void k_bcr_backsolve(int N, int B, int s, const double* __restrict__ fm,
                                                                   const double* __restrict__ Pm,
                                                                   const double* __restrict__ Qm,
                                                                   const double* __restrict__ Gi,
                                                                   double* __restrict__ x) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int ld = B + 1;
  double* MP = sm;                       // [B][ld] P
  double* MQ = MP + (size_t)B * ld;      // [B][ld] Q
  double* MG = MQ + (size_t)B * ld;      // [B][ld] G^-1
  double* w = MG + (size_t)B * ld;       // [B]
  double* xl = w + B;                    // [B]
  double* xr = xl + B;                   // [B]
  const int tid = threadIdx.x;
  const int i = s * (2 * blockIdx.x + 1) - 1;
  if (i >= N) return;
  const int l = i - s, r = i + s;
  const bool haveL = l >= 0, haveR = r < N;
  const size_t BB = (size_t)B * B;
  // all loads first (up to 6 entries of each matrix per thread: B <= 78), then the LDS stores: one memory
  // round trip for the whole staging instead of one per loop iteration
  {
    constexpr int U = (kBcrSplitMaxHB * 6 * kBcrSplitMaxHB * 6 + kBcrElimThreads - 1) / kBcrElimThreads;
    double vp[U], vq[U], vg[U];
    const double wv = tid < B ? fm[(size_t)i * B + tid] : 0.0;
    const double xlv = (tid < B && haveL) ? x[(size_t)l * B + tid] : 0.0;
    const double xrv = (tid < B && haveR) ? x[(size_t)r * B + tid] : 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + kBcrElimThreads * u;
      const bool in = e < B * B;
      vp[u] = (in && haveL) ? Pm[(size_t)i * BB + e] : 0.0;
      vq[u] = (in && haveR) ? Qm[(size_t)i * BB + e] : 0.0;
      vg[u] = in ? Gi[(size_t)i * BB + e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + kBcrElimThreads * u;
      if (e < B * B) {
        const int rr = e / B, cc = e - rr * B;
        MP[rr * ld + cc] = vp[u]; MQ[rr * ld + cc] = vq[u]; MG[rr * ld + cc] = vg[u];
      }
    }
    if (tid < B) { w[tid] = wv; xl[tid] = xlv; xr[tid] = xrv; }
  }
  __syncthreads();
  for (int task = tid; task < 4 * B; task += kBcrElimThreads) {      // w -= P xl + Q xr
    const int k = task >> 2, q4 = task & 3;
    double acc = 0.0;
    for (int c = q4; c < B; c += 4) acc += MP[k * ld + c] * xl[c] + MQ[k * ld + c] * xr[c];
    acc += dpp_pair<0xB1>(acc);
    acc += dpp_pair<0x4E>(acc);
    if (q4 == 0) w[k] -= acc;
  }
  __syncthreads();
  for (int task = tid; task < 4 * B; task += kBcrElimThreads) {      // x = (G^-1)^T w
    const int m = task >> 2, q4 = task & 3;
    double acc = 0.0;
    for (int k = m + q4; k < B; k += 4) acc += MG[k * ld + m] * w[k];
    acc += dpp_pair<0xB1>(acc);
    acc += dpp_pair<0x4E>(acc);
    if (q4 == 0) x[(size_t)i * B + m] = acc;
  }
}
#endif

}  // namespace ba
