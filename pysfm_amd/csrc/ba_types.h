// ba_types.h - what the host side and the kernels of libpysfm_ba.so share: launch shapes, the work-list records
// ba_set_problem builds for the kernels, LDS budgets.  No kernel lives here (ba_obs_kernels.h, ba_schur_kernels.h,
// ba_band.h, ba_bcr*.h, ba_dense.h, ba_dist.h hold them, each compiled in exactly one translation unit).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ba_math.h"

namespace ba {

// ---- launch shapes and the problem as the kernels see it
constexpr int kBlock = 256;       // 4 wavefronts
constexpr int kWave = 64;
constexpr int kTile = 16;         // observations of one track staged per Schur work unit

struct DevProblem {
  int nc, nt, nco;
  int hb;                   // block half-bandwidth of the reduced system (see reduced-system layout below)
  long long nobs;
  const int* obs_cam;
  const int* obs_pt;
  const double2* obs_z;
  const int* pt_off;        // [nt+1]
  const int* cam_opt_pos;   // [nc]
  const unsigned char* pt_opt;  // [nt]
  double K[9];
  Sensor sensor;
};

// ---- the record of a trial (k_cost, k_backsub_groups)
constexpr int kCostBlocks = 2048;
struct HostResult { int singular_points; int solve_info; double partial[kCostBlocks]; };
// The solver status as it travels in the shards' trial record, which is SUMMED over the ranks: a time-out (a fault, never a
// property of the matrix) must still be recognisable after the sum, so it weighs more than any sum of pivot indices can.
constexpr double kTrialTimedOutWord = 1099511627776.0;      // 2^40 > ranks * 2^31
__host__ __device__ inline double trial_status_word(int solve_info) { return solve_info == 0x7f000001 ? kTrialTimedOutWord : (double)solve_info; }
__host__ __device__ inline int trial_status_of_sum(double sum, int ranks, bool own_parts) {
  if (sum >= kTrialTimedOutWord) return 0x7f000001;
  if (sum == 0.0) return 0;
  // every rank solved the same system (status x ranks) - or, with the solve spread over the ranks, its own part of it: any non-zero = failed
  const double v = own_parts ? fabs(sum) : sum / ranks;
  const double c = v < 1.0 ? (own_parts ? 1.0 : v) : (v > 2.0e9 ? 2.0e9 : v);
  return (int)(c + (c >= 0 ? 0.5 : -0.5));
}

// ---- k_camera_blocks
constexpr int kCamChunk = 2048;    // at most; the host shrinks it for scenes with few cameras (ba_set_problem)
struct CamUnit { int cam; int begin; int end; };

// ---- k_schur_pairs
struct SchurUnit { int pt; int row0; int col0; };
struct SchurChunk { int begin; int end; int p0; };     // units [begin, end), window start p0
constexpr int kSchurChunkUnits = 64;                  // at most this many units per workgroup
constexpr int kSchurTileBytes = 48 * 1024;             // LDS budget of the accumulation tile

constexpr int kSchurBlock = 1024;                      // 16 wavefronts share one accumulation tile

// ---- k_schur_groups, k_linearize_groups, k_backsub_groups
struct SchurGroup { int pt_begin; int pt_end; int L; int pad; };
constexpr int kGroupBlock = 512;                       // 8 wavefronts per workgroup
constexpr int kGroupMaxPts = 24;
constexpr int kGroupMaxL = 15;
constexpr int kGroupChunk = 8;                         // groups per workgroup

// ---- k_schur_groups_mfma
constexpr int kGmBlock = 256;                          // 4 wavefronts per workgroup (20 KB of staging each)
constexpr int kGmMaxL = 10;
constexpr int kGmChunk = 4;                            // groups per workgroup: one per wavefront
constexpr int kGmPts = 6;                              // points per batch
constexpr int kGmK = 20;                               // staged k rows: 3 per point, two zero rows
constexpr int kGmLd = 64;                              // staged row length (60 used)
constexpr int kGmMultiRoundCap = 60;                   // points per group when the groups need several rounds of workgroups (ba_set_problem)

// ---- k_schur_groups_mfma2
constexpr int kGm2Block = 512;
constexpr int kGm2Pairs = 4;
constexpr int kGm2DRows = 24;                          // D values per buffer (20 used)

__host__ __device__ inline size_t schur_mfma2_lds_bytes(int wn, int hb1) {
  return (size_t)kGm2Pairs * 2 * kGmK * kGmLd * sizeof(double) + (size_t)kGm2Pairs * 2 * kGm2DRows * sizeof(double) +
         (size_t)kGm2Pairs * (16 + 4) * sizeof(int) + 64 * sizeof(double) + (size_t)wn * ((size_t)hb1 * 36 + 6) * sizeof(double);
}

// ---- k_schur_groups_mfma3
constexpr int kGm3MaxL = 24;                            // track length up to which the launches re-linearise at most four times
constexpr int kGm3MaxSpan = 40;                         // widest window: 15 tiles per side, the last tile COLUMN alone fills the 15 accumulator tiles of a launch
constexpr int kGm3PosLen = 64;                          // optimised positions of a group's cameras (40 used)
constexpr int kGm3MaxTiles = 28;                        // accumulator tiles of one launch (15 fit without register spills; 21 and 28: launch_mfma3_set, ba_schur_window.hip)

struct Gm3Params { int nts; int Ld; int Kbuf; int np_cap; int wn; int do_rhs; int wb1; };      // wb1: blocks per row of the LDS window (the widest group)
// A group of k_schur_groups_mfma3: consecutive points (internal order) whose optimised cameras all lie in the window of
// W <= 40 (kGm3MaxSpan) consecutive optimised positions starting at `lo`.  tab[(k - pt_begin) * W + w] = the observation of point k
// in the camera at position lo + w, or -1: the camera lists need NOT be identical, only close (tracks of different
// lengths, missing observations) - a run of points with one camera list is the special case of a full table.
struct WinGroup { int pt_begin; int pt_end; int W; int lo; int tab; int pad0; int pad1; int pad2; };

__host__ __device__ constexpr int gm3_ntiles(int tj0, int tj1) { return (tj1 * (tj1 + 1) - tj0 * (tj0 + 1)) / 2; }
__host__ __device__ inline int gm3_np(int L, int np_cap) { int np = 64 / L; if (np > kGmPts) np = kGmPts; if (np > np_cap) np = np_cap; return np; }
__host__ __device__ inline size_t schur_mfma3_lds_bytes(int Kbuf, int Ld, int wn, int hb1) {      // hb1: blocks per window row (Gm3Params::wb1)
  return (size_t)kGm2Pairs * 2 * Kbuf * Ld * sizeof(double) + (size_t)kGm2Pairs * 2 * kGm2DRows * sizeof(double) +
         (size_t)kGm2Pairs * (kGm3PosLen + 4) * sizeof(int) + 64 * sizeof(double) + (size_t)wn * ((size_t)hb1 * 36 + 6) * sizeof(double);
}

// ---- k_schur_wide_mfma
constexpr int kGw7 = 7;                                 // consumers
constexpr int kGwBlock = 64 * (1 + kGw7);
constexpr int kGwMinTiles = 11;                         // tiles per side from which a group comes here (10: five launches of k_schur_groups_mfma3 are faster, 0.68 against 0.90 ms at L = 25)
constexpr int kGwMaxTiles = 15;
constexpr int kGwLd = 16 * kGwMaxTiles;                 // staged row: 240 doubles
constexpr int kGwK = 8;                                 // k rows per buffer (two points: 6 + 2 zero rows)
__host__ __device__ constexpr int gw_own(int nts) { return (nts * (nts + 1) / 2 + kGw7 - 1) / kGw7; }      // tiles per consumer: 8 .. 18

__host__ __device__ inline size_t schur_wide_lds_bytes() {
  return (size_t)2 * kGwK * kGwLd * sizeof(double) + (size_t)2 * kGwK * sizeof(double) + (size_t)(64 + 8 + 128) * sizeof(int);
}

// ---- k_schur_rect_mfma
constexpr int kRectSeg = 32;                            // positions per segment
constexpr int kRectTiles = 6 * kRectSeg / 16;           // 12 tiles per side of a segment
constexpr int kRectLd = 2 * 6 * kRectSeg;               // staged row: [A | B], 384 doubles
constexpr int kRectGroupPts = 96;                       // listed points per group at most (the host halves it until the groups fill the chip)
constexpr int kRectConsumers = kRectTiles / 2;          // consumer wavefronts of a workgroup: two tile columns each
constexpr int kRectBlock = 64 * (2 + kRectConsumers);      // two producers (even / odd points of the group) + the consumers
struct RectGroup { int n; int pts; int tab; int loA; int loB; int WB; int pad0; int pad1; };      // points rtab[pts ...], table rtab[tab + q * 64 + column]

__host__ __device__ inline size_t schur_rect_lds_bytes() {
  return (size_t)2 * 4 * kRectLd * sizeof(double) + (size_t)2 * 4 * sizeof(double) + (size_t)(64 + 8) * sizeof(int);
}

// ---- the dense-visibility reduction (k_dense_syrk, k_dense_rhs)
constexpr int kSyrkTile = 64;                          // output tile edge
constexpr int kSyrkKc = 32;                            // rows of Ud per LDS panel
constexpr int kDenseRhsRows = 32;

// ---- the group-packed point kernels
constexpr int kPtGroupMaxL = 24;    // (= kGm3MaxL: every scene the matrix-core reduction takes also takes the group-packed point kernels)

// ---- k_band_solve
constexpr int kMaxBandSolve = 21;
constexpr int kSolveThreads = 256;
// LDS budget of k_band_solve: the U window, two row buffers and a staging area of `ch`
// band rows (S on the way down, U on the way back) so that global latency is paid once
// per chunk instead of once per row.
// LDS row stride (doubles) of the U window: padded so that the HB rows a wavefront reads
// together (one per lane of a group, `36` doubles further along in each older row) fall on
// distinct LDS banks for ds_read2_b64 (32 banks of 4 B): (stride - 36) % 16 == 2.
__host__ __device__ constexpr int band_ring_stride(int hb) {
  return (hb + 1) * 36 + ((2 - 36 * hb) % 16 + 16) % 16;
}
__host__ __device__ inline size_t band_solve_fixed_doubles(int hb) {
  const size_t hbm = hb > 0 ? hb : 1;
  return hbm * band_ring_stride(hb) + hbm * 6 * 2 + 2 * ((size_t)(hb + 1) * 36 + 6) + 16 * 6 + 8;
}
__host__ __device__ inline size_t band_solve_row_doubles(int hb) { return (size_t)(hb + 1) * 36 + 12; }
__host__ __device__ inline int band_solve_chunk(int hb, size_t lds_bytes) {
  const size_t fixed = band_solve_fixed_doubles(hb) * 8 + (size_t)(hb + 64) * 6 + 64;
  if (lds_bytes <= fixed) return 0;
  size_t ch = (lds_bytes - fixed) / (band_solve_row_doubles(hb) * 8 + 6);
  return (int)(ch > 32 ? 32 : ch);
}
__host__ __device__ inline size_t band_solve_lds_bytes(int hb, int ch) {
  return band_solve_fixed_doubles(hb) * 8 + (size_t)ch * band_solve_row_doubles(hb) * 8 + (size_t)(ch + hb) * 6 + 64;
}

// ---- limits of the reduced-system solvers (ba_solve_reduced picks by half-bandwidth and size)
constexpr int kBcrMaxHB = 11;                  // ba_bcr.h: 4 matrices of B x (B+1) doubles must fit in LDS (B = 66: 145 KB)
constexpr int kBcrSplitMaxHB = 13;             // ba_bcr.h: the node over three workgroups (k_bcr_eliminate_split / _fused) needs three matrices of its own (B = 78: 148 KB)
constexpr int kBcrwMinHB = kBcrMaxHB + 1;      // ba_bcr_wide.h
constexpr int kBcrwMaxHB = 23;                 // B = 138: one B x (B+1) fp64 matrix = 150 KB of the 160 KB LDS (track length 24)
constexpr int kDcMaxN = 16000;                 // ba_dense.h: k_dense_backsolve keeps w[n] in LDS
__host__ __device__ inline size_t big_matrix_doubles(int B) { return (size_t)(3 * B + 1) * (3 * B); }      // ba_bcr_big.h: the work matrix K of a node

}  // namespace ba
