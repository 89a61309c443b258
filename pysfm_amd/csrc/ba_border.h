// ba_border.h - the reduced camera system as a BAND PLUS A BORDER (kernels; compiled by ba_border.hip alone).
//
// The reference solves a dense S whatever cameras share tracks (bundle_adjuster.py:259-312).  A block band covers a camera
// sequence; a few long-range tracks - a loop closure: camera 3 and camera 503 see the same point - would widen the band to the
// whole matrix.  Instead the cameras at the far end of such tracks go to a BORDER: they are ordered last, the band is formed
// over the others (for the band's kernels a border camera is a camera that is not optimised), and
//
//        S = [ B   C ]     B  block band of the n1 = nco - k band cameras (half-width hb)
//            [ C^T D ]     C  6 n1 x 6 k, D 6 k x 6 k: every block that involves one of the k border cameras
//
// is solved by block elimination:  B [Y | y] = [C | b1]  (the cyclic reduction of ba_bcr.h factors B and solves for b1; its kept
// factors G^-1, P, Q then take the 6 k columns of C through the same elimination tree, k_bcr_apply),
// (D - C^T Y) x2 = b2 - C^T y  (one workgroup, Cholesky in LDS),  x1 = y - Y x2.
#pragma once

#include "ba_device.h"

namespace ba {

constexpr int kBordMaxCams = 21;          // border cameras at most: 126 unknowns fit the LDS of k_border_solve with their right-hand side
constexpr int kBordThreads = 256;

typedef double bord_acc4 __attribute__((ext_vector_type(4)));

// ---- the blocks of the border cameras (bundle_adjuster.py:230-234, 263-276 for every pair of cameras that involves one of them).
// The set-up lists, for every non-zero 6 x 6 block (camera at position pc, border camera ja), the pairs of observations
// (a in the border camera, c in the other camera, both of one track) that add to it - BorderBlock + pairs[].  One wavefront per
// block walks its pairs, every lane a share of them, and writes the block once: no atomics (a block of a border camera takes
// a term from every track the two cameras share - a thousand atomics on the same 36 words), nothing to clear, the same bits
// every time.
//   pc == n1 + ja (the diagonal block, pairs (a, a)):  D[ja, ja] = sum damped Jc^T Jc - T_a W_a^T,   b2[ja] = sum Jc^T r - T_a bP
//   pc <  n1 (a band camera):   C[pc, ja] = - sum W_c HPPinv W_a^T           (T_a = W_a HPPinv)
//   pc >= n1 (a border camera): D[pc - n1, ja] = - sum W_c HPPinv W_a^T      (both orders of a pair of border cameras are listed)
// C row-major [rows][ld] with row 6 pos + u, column 6 ja + v; D row-major [ld][ld]; b2 = the border part of b.
// A block's pairs are walked in CHUNKS of kBordChunkPairs (one wavefront each, two pairs a lane) that leave partial sums (two pairs a lane; 64-pair chunks measured slower: 24.6 against 19.2 us at config 3 with ten border cameras);
// k_border_sum adds a block's chunks up in their order and writes the block.
struct BorderBlock { int begin, end, pc, ja; };              // pairs [begin, end); its chunks start at chunk index `first` = see BorderChunk
struct BorderChunk { int block, begin, end, pad; };
constexpr int kBordChunkPairs = 128;
constexpr int kBordPartial = 42;                             // 36 entries + 6 of the right-hand side

template <bool TABLE>
__global__ __launch_bounds__(kBlock) void k_schur_border(DevProblem P, const double* __restrict__ cams, const double* __restrict__ X,
                                                         const BorderBlock* __restrict__ blocks, const BorderChunk* __restrict__ chunks, int nchunks,
                                                         const int2* __restrict__ pairs, int n1, double damping, const double* __restrict__ HPPinv,
                                                         const double* __restrict__ bP, double* __restrict__ partial) {
  const int wv = (int)((blockIdx.x * (unsigned)kBlock + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (wv >= nchunks) return;
  const BorderChunk ck = chunks[wv];
  BorderBlock bk = blocks[ck.block];
  bk.begin = ck.begin; bk.end = ck.end;
  const bool diag = bk.pc == n1 + bk.ja;
  double acc[36], rhs[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) rhs[i] = 0.0;
  for (int e = bk.begin + lane; e < bk.end; e += 64) {
    const int2 pr = pairs[e];
    const int na = pr.x, nc_ = pr.y, pt = P.obs_pt[na];
    const double x[3] = {X[(size_t)pt * 3], X[(size_t)pt * 3 + 1], X[(size_t)pt * 3 + 2]};
    double cm[12], er[2], r[2], Jc[12], Jp[6], Wa[18], Ta[18], Hi[6];
    load_cam(cams, P.obs_cam[na], cm);
    const double2 za = P.obs_z[na];
    obs_linearize<TABLE>(P.K, cm, x, za.x, za.y, P.sensor, er, r, Jc, Jp);
    block_W(Jc, Jp, Wa);
#pragma unroll
    for (int i = 0; i < 6; ++i) Hi[i] = HPPinv[(size_t)pt * 6 + i];
    block_T(Wa, Hi, Ta);
    if (diag) {
      const double g[3] = {bP[(size_t)pt * 3], bP[(size_t)pt * 3 + 1], bP[(size_t)pt * 3 + 2]};
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        rhs[u] += Jc[u] * r[0] + Jc[6 + u] * r[1] - (Ta[u * 3] * g[0] + Ta[u * 3 + 1] * g[1] + Ta[u * 3 + 2] * g[2]);
#pragma unroll
        for (int v = 0; v < 6; ++v) {
          double hv = Jc[u] * Jc[v] + Jc[6 + u] * Jc[6 + v];
          if (u == v) hv *= 1.0 + damping;
          acc[u * 6 + v] += hv - (Ta[u * 3] * Wa[v * 3] + Ta[u * 3 + 1] * Wa[v * 3 + 1] + Ta[u * 3 + 2] * Wa[v * 3 + 2]);
        }
      }
    } else {
      double Wc[18];
      load_cam(cams, P.obs_cam[nc_], cm);
      const double2 zc = P.obs_z[nc_];
      obs_linearize<TABLE>(P.K, cm, x, zc.x, zc.y, P.sensor, er, r, Jc, Jp);
      block_W(Jc, Jp, Wc);
#pragma unroll
      for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int v = 0; v < 6; ++v) acc[u * 6 + v] -= Wc[u * 3] * Ta[v * 3] + Wc[u * 3 + 1] * Ta[v * 3 + 1] + Wc[u * 3 + 2] * Ta[v * 3 + 2];
    }
  }
  // the lanes' shares added up in a fixed order; lane i < 36 keeps entry i
  double mine = 0.0, myr = 0.0;
#pragma unroll
  for (int i = 0; i < 36; ++i) { const double t = wave_sum(acc[i]); if (lane == i) mine = t; }
  if (diag) {
#pragma unroll
    for (int i = 0; i < 6; ++i) { const double t = wave_sum(rhs[i]); if (lane == i) myr = t; }
  }
  if (lane < 36) partial[(size_t)wv * kBordPartial + lane] = mine;
  if (lane < 6) partial[(size_t)wv * kBordPartial + 36 + lane] = myr;
}

// ... the chunks of a block added up in their order: 42 threads a block (chunk_first[b] .. chunk_first[b + 1] are its chunks)
__global__ __launch_bounds__(kBlock) void k_border_sum(const BorderBlock* __restrict__ blocks, int nblocks, const int* __restrict__ chunk_first,
                                                       const double* __restrict__ partial, int n1, double* __restrict__ C, double* __restrict__ D, int ld,
                                                       double* __restrict__ b2) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  const int b = t / kBordPartial, i = t % kBordPartial;
  if (b >= nblocks) return;
  const BorderBlock bk = blocks[b];
  const bool diag = bk.pc == n1 + bk.ja;
  if (i >= 36 && !diag) return;
  double sacc = 0.0;
  const int c0 = chunk_first[b], c1 = chunk_first[b + 1];
  for (int cb = c0; cb < c1; cb += 8) {                   // eight records in flight, added in their order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = cb + u < c1 ? partial[(size_t)(cb + u) * kBordPartial + i] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) sacc += v[u];
  }
  if (i >= 36) { b2[6 * bk.ja + i - 36] = sacc; return; }
  double* dst = bk.pc < n1 ? C + (size_t)(6 * bk.pc) * ld + 6 * bk.ja : D + (size_t)(6 * (bk.pc - n1)) * ld + 6 * bk.ja;
  dst[(size_t)(i / 6) * ld + i % 6] = sacc;
}

// ---- F = C with the rows / columns of masked camera parameters cleared; M [nb][nb] = D with masked border parameters turned into
// identity rows / columns; rv = b2 (0 at masked parameters); the status word of the border solve cleared
__global__ __launch_bounds__(kBlock) void k_border_prepare(long long rowsF, int rows1, int ld, int nb, const double* __restrict__ C,
                                                           const double* __restrict__ D, const double* __restrict__ b2,
                                                           const unsigned char* __restrict__ mask, double* __restrict__ F,
                                                           double* __restrict__ M, double* __restrict__ rv, int* __restrict__ binfo) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long nF = rowsF * ld;
  if (i == 0) *binfo = 0;
  if (i < nF) {
    const long long r = i / ld;
    const int c = (int)(i % ld);
    const bool keep = r < rows1 && c < nb && !(mask && (!mask[r] || !mask[rows1 + c]));
    F[i] = keep ? C[i] : 0.0;
    return;
  }
  const long long j = i - nF;
  if (j < (long long)nb * nb) {
    const int u = (int)(j / nb), v = (int)(j % nb);
    const bool mu = mask && !mask[rows1 + u], mv = mask && !mask[rows1 + v];
    M[j] = (mu || mv) ? (u == v ? 1.0 : 0.0) : D[(size_t)u * ld + v];
    if (v == 0) rv[u] = mu ? 0.0 : b2[u];
  }
}

// ---- 16 x 16 tile on the matrix cores: acc[i][j] += sign * sum_k A(i0 + i, k) X[k][j], k in [k0, k1) (multiples of 4).
// A(i, k) = Am[i * ra + k * ca] (LDS; rows / columns past the matrix are zero there), X[k * ldx + j] (LDS, 16 columns).
// Layout of v_mfma_f64_16x16x4_f64: lane (lr = lane & 15, lk = lane >> 4) feeds A[lr][lk] and X[lk][lr]; it receives
// acc[v] = entry (lk + 4 v, lr).
// k0, k1 are multiples of 16: four k-steps at a time, their eight operands in registers before the first of their MFMAs and the
// next four's loads issued before it too (an LDS round trip per MFMA would be three times the MFMA).
__device__ __forceinline__ void bord_tile_mac(bord_acc4& acc, const double* __restrict__ Am, int ra, int ca, int i0, const double* __restrict__ Xs,
                                              int ldx, int k0, int k1, int lr, int lk, bool negate) {
  if (k0 >= k1) return;
  const double* ap = Am + (i0 + lr) * ra + lk * ca;
  const double* xp = Xs + lk * ldx + lr;
  const int sa = 4 * ca, sx = 4 * ldx;
  double a[4], x[4], an[4], xn[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { a[q] = ap[k0 * ca + q * sa]; x[q] = xp[k0 * ldx + q * sx]; }
  for (int k = k0; k < k1; k += 16) {
    const bool more = k + 16 < k1;
    const int kn = more ? k + 16 : k;
#pragma unroll
    for (int q = 0; q < 4; ++q) { an[q] = ap[kn * ca + q * sa]; xn[q] = xp[kn * ldx + q * sx]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(negate ? -a[q] : a[q], x[q], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = an[q]; x[q] = xn[q]; }
  }
}

// A B x B row-major matrix from memory into registers (every load of a thread is issued before its first LDS store: one round
// trip), then into LDS as [Bq][lda] (Bq = 16 NRT >= B, lda = Bq + 1): entries outside the matrix - and, with lower_only, above
// its diagonal - are zero.  Wavefront w takes rows w, w + 4, ...; a lane a column (and column + 64 where Bq > 64).
template <int NRT>
struct BordMatrixRegs { double v[4 * NRT][NRT > 4 ? 2 : 1]; };

template <int NRT>
__device__ __forceinline__ void bord_load_matrix(BordMatrixRegs<NRT>& R, const double* __restrict__ src, int B, bool lower_only, bool present, int wave,
                                                 int lane) {
  constexpr int H = NRT > 4 ? 2 : 1;
#pragma unroll
  for (int u = 0; u < 4 * NRT; ++u) {
    const int r = wave + 4 * u;
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      const int c = lane + 64 * hh;
      const bool in = present && r < B && c < B && !(lower_only && c > r);
      R.v[u][hh] = in ? src[(size_t)r * B + c] : 0.0;
    }
  }
}
template <int NRT>
__device__ __forceinline__ void bord_store_matrix(const BordMatrixRegs<NRT>& R, double* __restrict__ dst, int wave, int lane) {
  constexpr int H = NRT > 4 ? 2 : 1, Bq = 16 * NRT, lda = Bq + 1;
#pragma unroll
  for (int u = 0; u < 4 * NRT; ++u) {
    const int r = wave + 4 * u;
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      const int c = lane + 64 * hh;
      if (c < Bq) dst[r * lda + c] = R.v[u][hh];
    }
  }
}
// ... and 16 columns [col0, col0 + 16) of the B rows of node `node` of F ([rows][ld]) as [Bq][17]
template <int NRT>
struct BordRhsRegs { double v[NRT]; };
template <int NRT>
__device__ __forceinline__ void bord_load_rhs(BordRhsRegs<NRT>& R, const double* __restrict__ F, int ld, int node, int B, int col0, bool present, int tid) {
#pragma unroll
  for (int u = 0; u < NRT; ++u) {
    const int r = (tid >> 4) + 16 * u;
    R.v[u] = (present && r < B) ? F[((size_t)node * B + r) * ld + col0 + (tid & 15)] : 0.0;
  }
}
template <int NRT>
__device__ __forceinline__ void bord_store_rhs(const BordRhsRegs<NRT>& R, double* __restrict__ dst, int tid) {
#pragma unroll
  for (int u = 0; u < NRT; ++u) dst[((tid >> 4) + 16 * u) * 17 + (tid & 15)] = R.v[u];
}

__host__ __device__ inline size_t bord_apply_lds_bytes(int B) {
  const int Bq = (B + 15) / 16 * 16, lda = Bq + 1;
  return ((size_t)2 * Bq * lda + (size_t)3 * Bq * 17) * sizeof(double);
}

// ---- the right-hand sides through the kept factors of the cyclic reduction (ba_bcr.h: node i eliminated at stride s = lowest
// set bit of i + 1 left G_i^-1 (lower triangular), P_i = G_i^-1 T[i, i - s], Q_i = G_i^-1 T[i, i + s]).
// FORWARD, level s: one workgroup per (surviving node j = 2 s (k + 1) - 1, 16 columns):
//     F_j -= Q_l^T (G_l^-1 F_l) + P_r^T (G_r^-1 F_r),   l = j - s, r = j + s (if < N)
// F_l, F_r are final (their nodes are eliminated at this level and only ever received updates from lower levels).
// BACKWARD, level s: one workgroup per (eliminated node i = s (2 k + 1) - 1, 16 columns):
//     F_i <- G_i^-T (G_i^-1 F_i - P_i X_l - Q_i X_r),   X_l = F_{i-s}, X_r = F_{i+s} already solved.
// NRT = 16-row tiles of a node (B <= 16 NRT): the staging loops unroll.
template <bool BACK, int NRT>
__global__ __launch_bounds__(kBordThreads) void k_bcr_apply(int N, int B, int s, const double* __restrict__ Pm, const double* __restrict__ Qm,
                                                            const double* __restrict__ Gi, double* __restrict__ F, int ld) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int Bq = 16 * NRT, lda = Bq + 1;
  double* A0 = sm;                           // G^-1
  double* A1 = A0 + (size_t)Bq * lda;        // P or Q
  double* X0 = A1 + (size_t)Bq * lda;        // [Bq][17]
  double* X1 = X0 + (size_t)Bq * 17;
  double* X2 = X1 + (size_t)Bq * 17;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int col0 = blockIdx.y * 16;
  const size_t BB = (size_t)B * B;
  constexpr int SLOTS = (NRT + 3) / 4;
  if (!BACK) {
    const int j = 2 * s * ((int)blockIdx.x + 1) - 1;
    if (j >= N) return;
    bord_acc4 acc[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) acc[q] = bord_acc4{0, 0, 0, 0};
    BordMatrixRegs<NRT> g, m;
    BordRhsRegs<NRT> f;
    bord_load_matrix<NRT>(g, Gi + (size_t)(j - s) * BB, B, true, true, wave, lane);
    bord_load_matrix<NRT>(m, Qm + (size_t)(j - s) * BB, B, false, true, wave, lane);
    bord_load_rhs<NRT>(f, F, ld, j - s, B, col0, true, tid);
    double fj[SLOTS][4];                                     // my entries of F_j (the update is added to them at the end: no read-modify-write round trip there)
#pragma unroll
    for (int q = 0; q < SLOTS; ++q)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 16 * (wave + 4 * q) + lk + 4 * v;
        fj[q][v] = (wave + 4 * q < NRT && r < B) ? F[((size_t)j * B + r) * ld + col0 + lr] : 0.0;
      }
    for (int side = 0; side < 2; ++side) {
      const int i = side == 0 ? j - s : j + s;
      if (i >= N) break;                                     // (uniform over the workgroup)
      bord_store_matrix<NRT>(g, A0, wave, lane);
      bord_store_matrix<NRT>(m, A1, wave, lane);
      bord_store_rhs<NRT>(f, X0, tid);
      if (side == 0 && j + s < N) {                          // the other neighbour's operands on their way meanwhile
        bord_load_matrix<NRT>(g, Gi + (size_t)(j + s) * BB, B, true, true, wave, lane);
        bord_load_matrix<NRT>(m, Pm + (size_t)(j + s) * BB, B, false, true, wave, lane);
        bord_load_rhs<NRT>(f, F, ld, j + s, B, col0, true, tid);
      }
      __syncthreads();
      // T = G_i^-1 F_i   (lower triangular: k <= row)
      for (int rt = wave; rt < NRT; rt += 4) {
        bord_acc4 t = {0, 0, 0, 0};
        bord_tile_mac(t, A0, lda, 1, 16 * rt, X0, 17, 0, 16 * rt + 16, lr, lk, false);
#pragma unroll
        for (int v = 0; v < 4; ++v) X1[(16 * rt + lk + 4 * v) * 17 + lr] = t[v];
      }
      __syncthreads();
      // update -= M^T T   (M = Q_l or P_r: A(i, k) = M[k][i])
#pragma unroll
      for (int q = 0; q < SLOTS; ++q) {
        const int rt = wave + 4 * q;
        if (rt < NRT) bord_tile_mac(acc[q], A1, 1, lda, 16 * rt, X1, 17, 0, Bq, lr, lk, true);
      }
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
      const int rt = wave + 4 * q;
      if (rt < NRT)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 16 * rt + lk + 4 * v;
          if (r < B) F[((size_t)j * B + r) * ld + col0 + lr] = fj[q][v] + acc[q][v];
        }
    }
  } else {
    const int i = s * (2 * (int)blockIdx.x + 1) - 1;
    if (i >= N) return;
    const int l = i - s, r_ = i + s;
    const bool haveL = l >= 0, haveR = r_ < N;
    {
      BordMatrixRegs<NRT> g, m;
      BordRhsRegs<NRT> f, xl;
      bord_load_matrix<NRT>(g, Gi + (size_t)i * BB, B, true, true, wave, lane);
      bord_load_matrix<NRT>(m, Pm + (size_t)i * BB, B, false, haveL, wave, lane);
      bord_load_rhs<NRT>(f, F, ld, i, B, col0, true, tid);
      bord_load_rhs<NRT>(xl, F, ld, haveL ? l : 0, B, col0, haveL, tid);
      bord_store_matrix<NRT>(g, A0, wave, lane);
      bord_store_matrix<NRT>(m, A1, wave, lane);
      bord_store_rhs<NRT>(f, X0, tid);
      bord_store_rhs<NRT>(xl, X1, tid);
    }
    BordMatrixRegs<NRT> qreg;                               // Q_i and X_r on their way while P_i X_l is formed
    BordRhsRegs<NRT> xr;
    bord_load_matrix<NRT>(qreg, Qm + (size_t)i * BB, B, false, haveR, wave, lane);
    bord_load_rhs<NRT>(xr, F, ld, haveR ? r_ : 0, B, col0, haveR, tid);
    __syncthreads();
    bord_acc4 w[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
      w[q] = bord_acc4{0, 0, 0, 0};
      const int rt = wave + 4 * q;
      if (rt < NRT) {
        bord_tile_mac(w[q], A0, lda, 1, 16 * rt, X0, 17, 0, 16 * rt + 16, lr, lk, false);
        if (haveL) bord_tile_mac(w[q], A1, lda, 1, 16 * rt, X1, 17, 0, Bq, lr, lk, true);
      }
    }
    __syncthreads();
    bord_store_matrix<NRT>(qreg, A1, wave, lane);
    bord_store_rhs<NRT>(xr, X1, tid);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
      const int rt = wave + 4 * q;
      if (rt < NRT) {
        if (haveR) bord_tile_mac(w[q], A1, lda, 1, 16 * rt, X1, 17, 0, Bq, lr, lk, true);
#pragma unroll
        for (int v = 0; v < 4; ++v) X2[(16 * rt + lk + 4 * v) * 17 + lr] = w[q][v];
      }
    }
    __syncthreads();
    // x = G^-T w   (A(i, k) = G^-1[k][i], zero for k < i)
    for (int rt = wave; rt < NRT; rt += 4) {
      bord_acc4 x = {0, 0, 0, 0};
      bord_tile_mac(x, A0, 1, lda, 16 * rt, X2, 17, 16 * rt, Bq, lr, lk, false);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 16 * rt + lk + 4 * v;
        if (r < B) F[((size_t)i * B + r) * ld + col0 + lr] = x[v];
      }
    }
  }
}

// ---- M -= C^T Y, rv -= C^T y over the rows of the band cameras (M [nb][nb] full and symmetric: the node kernel of the cyclic
// reduction factors it).  One workgroup per chunk of kBordRedRows rows; the tiles (tu <= tv) of 16 x 16 over its wavefronts,
// operands straight from memory, all of a tile's loads before its first MFMA; an off-diagonal tile is added to both triangles.
// Only the band cameras that share a track with a border camera have rows in C: `rcams` lists them (set-up).  A workgroup owns
// one 16 x 16 tile (tu <= tv; blockIdx.x = tile, the last one = the right-hand side) and four chunks of kBordRedCams cameras, one
// per wavefront (blockIdx.y * 4 + wave); the four partial tiles meet in LDS and leave as ONE set of atomics.
// (Global fp64 atomics run at ~9 per nanosecond on the whole device whatever their addresses - measured: the first version of the
// border reduction, 3.6 M atomics in 0.42 ms - and only the LOWER triangle of M is kept: all the node kernel reads.)
constexpr int kBordRedCams = 8, kBordRedRows = 6 * kBordRedCams;
__global__ __launch_bounds__(kBordThreads) void k_border_reduce(int nrcams, const int* __restrict__ rcams, int ld, int nb, const double* __restrict__ C,
                                                                const double* __restrict__ Y, const double* __restrict__ y,
                                                                const unsigned char* __restrict__ mask2, double* __restrict__ M, double* __restrict__ rv) {
  __shared__ double part[kBordThreads / 64][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int c0 = (blockIdx.y * (kBordThreads / 64) + wave) * kBordRedCams;
  const int nt = ld / 16, ntiles = nt * (nt + 1) / 2;
  const int tile = blockIdx.x;
  int rows[kBordRedCams];
#pragma unroll
  for (int q = 0; q < kBordRedCams; ++q) rows[q] = c0 + q < nrcams ? 6 * rcams[c0 + q] : -1;
  if (tile == ntiles) {                                   // rv -= C^T y: lane = column (and + 64)
    // (a camera at a time: its six rows of C and y in flight together - all 48 rows of the chunk at once spilled, 15 of the kernel's 19 us)
    double sacc[2] = {0.0, 0.0};
#pragma unroll
    for (int m = 0; m < kBordRedCams; ++m) {
      if (rows[m] < 0) continue;                          // (wave-uniform)
      double yv[6], cv[2][6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        yv[u] = y[rows[m] + u];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) cv[hh][u] = lane + 64 * hh < nb ? C[(size_t)(rows[m] + u) * ld + lane + 64 * hh] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) { sacc[0] += cv[0][u] * yv[u]; sacc[1] += cv[1][u] * yv[u]; }
    }
    part[wave][lane] = sacc[0];
    part[wave][lane + 64] = sacc[1];
    __syncthreads();
    if (tid < nb && !(mask2 && !mask2[tid])) atomic_add_f64(rv + tid, -(part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]));
    return;
  }
  int tu, tv;
  tri_decode(tile, nt, tu, tv);
  // (C as the reduction left it: the column of a masked border parameter - mask2[.] == 0 - must not reach its identity row of M;
  //  Y's masked columns and rows are zero already)
  const bool cu = 16 * tu + lr < nb && !(mask2 && !mask2[16 * tu + lr]);
  double a[kBordRedRows / 4], x[kBordRedRows / 4];
#pragma unroll
  for (int q = 0; q < kBordRedRows / 4; ++q) {            // this lane's rows: k = 4 q + lk
    const int k = 4 * q + lk;
    int base = rows[0];
#pragma unroll
    for (int m = 1; m < kBordRedCams; ++m) base = (k / 6 == m) ? rows[m] : base;
    const int kk = base + k % 6;
    a[q] = (base >= 0 && cu) ? C[(size_t)kk * ld + 16 * tu + lr] : 0.0;
    x[q] = base >= 0 ? Y[(size_t)kk * ld + 16 * tv + lr] : 0.0;
  }
  bord_acc4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < kBordRedRows / 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[q], x[q], acc, 0, 0, 0);
#pragma unroll
  for (int v = 0; v < 4; ++v) part[wave][(lk + 4 * v) * 16 + lr] = acc[v];
  __syncthreads();
  {
    const int i = tid >> 4, j = tid & 15;                 // entry (16 tu + i, 16 tv + j) of C^T Y: the lower triangle takes it as (w, u)
    const int u = 16 * tu + i, w = 16 * tv + j;
    if (u < nb && w < nb && (tu != tv || w >= u)) atomic_add_f64(M + (size_t)w * nb + u, part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]);
  }
}

__host__ __device__ inline size_t bord_solve_lds_bytes(int nb) { return ((size_t)(nb + 1) * (nb + 2) + 2 * (nb + 1) + 8) * sizeof(double); }

// ---- (D - C^T Y) x2 = b2 - C^T y for more border cameras than a node of the cyclic reduction holds (nb > 66): Cholesky in LDS,
// one workgroup, a thread per row (left-looking: row r's entry of column j is A[r][j] - sum_{k<j} L[r][k] L[j][k], one barrier
// per column; the right-hand side rides along as row nb).  A pivot that is not positive: *binfo = its index + 1.
__global__ __launch_bounds__(kBordThreads) void k_border_solve(int nb, const double* __restrict__ M, const double* __restrict__ rv, int* __restrict__ binfo,
                                                               double* __restrict__ x2) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n1 = nb + 1, lda = nb + 2 + ((nb & 1) ? 1 : 0), tid = threadIdx.x;      // (odd row stride: a column read by the threads of a wavefront spreads over the banks)
  double* A = sm;                              // [nb + 1][lda]: lower triangle, the right-hand side as row nb
  double* xs = A + (size_t)n1 * lda;
  int& bad = *reinterpret_cast<int*>(xs + n1);      // (in the dynamic block: a static word on top of a raised dynamic limit does not launch)
  if (tid == 0) bad = 0;
  for (int r0 = 0; r0 <= nb; r0 += 8) {            // a thread a column, eight rows in flight
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = r0 + u;
      v[u] = (r <= nb && tid < nb) ? (r == nb ? rv[tid] : (tid <= r ? M[(size_t)r * nb + tid] : 0.0)) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (r0 + u <= nb && tid < nb) A[(r0 + u) * lda + tid] = v[u];
  }
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    // the pivot: every thread forms it for itself from row j (final up to column j - 1): no second barrier
    double piv = A[j * lda + j];
    for (int k = 0; k < j; ++k) piv -= A[j * lda + k] * A[j * lda + k];
    if (!(piv > 0.0) || !(piv < __builtin_huge_val())) { if (tid == 0) bad = j + 1; break; }
    const double d = rsqrt_nr(piv);
    if (tid > j && tid <= nb) {
      const int r = tid;
      double v = A[r * lda + j];
      for (int k = 0; k < j; ++k) v -= A[r * lda + k] * A[j * lda + k];
      A[r * lda + j] = v * d;
    }
    __syncthreads();                           // (row j is not read again before its own entry (j, j) is: stored after the barrier)
    if (tid == j) A[j * lda + j] = piv * d;
    __syncthreads();
  }
  __syncthreads();
  if (bad) {
    if (tid == 0) *binfo = bad;
    for (int c = tid; c < nb; c += kBordThreads) x2[c] = 0.0;
    return;
  }
  // row nb now holds y = L^-1 rv; x = L^-T y, last unknown first
  for (int c = tid; c < nb; c += kBordThreads) xs[c] = A[nb * lda + c];
  __syncthreads();
  for (int j = nb - 1; j >= 0; --j) {
    const double xj = xs[j] / A[j * lda + j];
    __syncthreads();
    for (int c = tid; c < j; c += kBordThreads) xs[c] -= A[j * lda + c] * xj;
    if (tid == 0) xs[j] = xj;
    __syncthreads();
  }
  for (int c = tid; c < nb; c += kBordThreads) x2[c] = xs[c];
}

// ---- the band + border system that is NOT positive definite (border_solve_lu, ba_border.hip): a column of F out to a vector and a
// solution back into it; the status of a column's solve folded into one word
__global__ __launch_bounds__(kBlock) void k_border_column(int rows1, int ld, int c, double* __restrict__ F, double* __restrict__ v, int to_F,
                                                          const int* __restrict__ info, int* __restrict__ acc) {
  const int r = blockIdx.x * kBlock + threadIdx.x;
  if (r == 0 && acc && *acc == 0 && *info != 0) *acc = *info;      // (the first column solve that failed: the next one's assembly clears the word)
  if (r >= rows1) return;
  if (to_F) F[(size_t)r * ld + c] = v[r];
  else v[r] = F[(size_t)r * ld + c];
}

// (D - C^T Y) x2 = b2 - C^T y by Gaussian elimination with partial pivoting in LDS, one workgroup, a thread per row: what the reference's
// numpy.linalg.solve would do to the border's Schur complement.  M holds its LOWER triangle (k_border_reduce).  A pivot that is exactly
// zero: *binfo = its column + 1 (gesv's info).
__global__ __launch_bounds__(kBordThreads) void k_border_solve_lu(int nb, const double* __restrict__ M, const double* __restrict__ rv, int* __restrict__ binfo,
                                                                  double* __restrict__ x2) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lda = nb + 2 + ((nb & 1) ? 0 : 1), tid = threadIdx.x;      // (odd row stride; column nb = the right-hand side)
  double* A = sm;                              // [nb][lda]
  int* piv_row = reinterpret_cast<int*>(A + (size_t)nb * lda);      // [2]: the pivot's row, "zero pivot at"
  for (int e = tid; e < nb * nb; e += kBordThreads) {
    const int r = e / nb, c = e - r * nb;
    A[r * lda + c] = c <= r ? M[(size_t)r * nb + c] : M[(size_t)c * nb + r];
  }
  for (int r = tid; r < nb; r += kBordThreads) A[r * lda + nb] = rv[r];
  if (tid == 0) piv_row[1] = 0;
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    if (tid == 0) {                            // the entry of largest magnitude in column j from row j down (the first of equals: gesv)
      int p = j;
      double best = fabs(A[j * lda + j]);
      for (int r = j + 1; r < nb; ++r) { const double v = fabs(A[r * lda + j]); if (v > best) { best = v; p = r; } }
      piv_row[0] = p;
      if (!(best > 0.0)) piv_row[1] = j + 1;
    }
    __syncthreads();
    if (piv_row[1]) break;
    const int p = piv_row[0];
    if (p != j)
      for (int c = j + tid; c <= nb; c += kBordThreads) { const double t = A[j * lda + c]; A[j * lda + c] = A[p * lda + c]; A[p * lda + c] = t; }
    __syncthreads();
    const double inv = 1.0 / A[j * lda + j];
    if (tid > j && tid < nb) {
      const double m = A[tid * lda + j] * inv;
      for (int c = j + 1; c <= nb; ++c) A[tid * lda + c] -= m * A[j * lda + c];
    }
    __syncthreads();
  }
  __syncthreads();
  if (piv_row[1]) {
    if (tid == 0) *binfo = piv_row[1];
    for (int c = tid; c < nb; c += kBordThreads) x2[c] = 0.0;
    return;
  }
  for (int j = nb - 1; j >= 0; --j) {          // U x = y (column nb), last unknown first
    if (tid == 0) A[j * lda + nb] /= A[j * lda + j];
    __syncthreads();
    const double xj = A[j * lda + nb];
    for (int r = tid; r < j; r += kBordThreads) A[r * lda + nb] -= A[r * lda + j] * xj;
    __syncthreads();
  }
  for (int c = tid; c < nb; c += kBordThreads) x2[c] = A[c * lda + nb];
}

// ---- x1 = y - Y x2 for the band cameras, x2 behind them; the status of the border solve joins the solver's status word
__global__ __launch_bounds__(kBlock) void k_border_correct(int rows1, int nb, int ld, const double* __restrict__ Y, const double* __restrict__ x2,
                                                           double* __restrict__ dC, const int* __restrict__ binfo, int* __restrict__ info) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t == 0 && *binfo != 0) atomicCAS(info, 0, rows1 + *binfo);
  const int r = t >> 2, q4 = t & 3;                   // four lanes per row
  if (r < rows1) {
    double sacc = 0.0;
    for (int c0 = q4; c0 < nb; c0 += 32) {                // eight loads of a lane in flight
      double yv[8], xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int c = c0 + 4 * u; yv[u] = c < nb ? Y[(size_t)r * ld + c] : 0.0; xv[u] = c < nb ? x2[c] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) sacc += yv[u] * xv[u];
    }
    sacc += dpp_pair<0xB1>(sacc);
    sacc += dpp_pair<0x4E>(sacc);
    if (q4 == 0) dC[r] -= sacc;
  } else if (r < rows1 + nb && q4 == 0) {
    dC[r] = x2[r - rows1];
  }
}

// ---- dense (nco, nco, 6, 6)-style reads of the bordered system for ba_flatten_reduced: entry (p, q) of the flat matrix by
// internal parameter indices
__device__ __forceinline__ double bordered_entry(int p, int q, int n1cams, int hb, const double* __restrict__ S, const double* __restrict__ C,
                                                 const double* __restrict__ D, int ld) {
  const int i = p / 6, a = p % 6, j = q / 6, d = q % 6;
  if (i >= n1cams && j >= n1cams) return D[(size_t)(p - 6 * n1cams) * ld + (q - 6 * n1cams)];
  if (j >= n1cams) return C[(size_t)p * ld + (q - 6 * n1cams)];
  if (i >= n1cams) return C[(size_t)q * ld + (p - 6 * n1cams)];
  if (i <= j) return j - i <= hb ? S[band_block(i, j, hb + 1) + a * 6 + d] : 0.0;
  return i - j <= hb ? S[band_block(j, i, hb + 1) + d * 6 + a] : 0.0;
}

__global__ __launch_bounds__(kBlock) void k_flatten_bordered(int n1cams, int hb, int nkeep, const int* __restrict__ keep, const double* __restrict__ S,
                                                             const double* __restrict__ b, const double* __restrict__ C, const double* __restrict__ D,
                                                             int ld, double* __restrict__ Aout, double* __restrict__ rhs) {
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (tid >= (long long)nkeep * nkeep) return;
  const int r = (int)(tid / nkeep), c = (int)(tid % nkeep);
  Aout[tid] = bordered_entry(keep[r], keep[c], n1cams, hb, S, C, D, ld);
  if (c == 0) rhs[r] = b[keep[r]];
}

}  // namespace ba
