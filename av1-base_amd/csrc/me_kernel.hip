// me_kernel.hip - motion estimation for inter frames: integer-pel full search (SAD) and, with
// subpel = 1, the half-/quarter-sample refinement with an SATD cost (second half of this file).
//
// Replaces the motion search SVT-AV1 runs inside the av1an worker the reference forks
// (/root/reference/crates/daemon/src/encode/av1an.rs:126-139).  Encoder-side, non-normative; the algorithm is the
// one DESIGN.md §3.9 defines and oracle/av1o_enc.c (motion_search) restates: for every leaf block of the partition,
// cost(dx, dy) = SAD(source block, previous SOURCE frame displaced by (dx, dy)) + n * (|dx| + |dy|) over
// |dx|, |dy| <= R with the displaced block kept within 16 samples of the frame; minimum cost, ties to the first
// candidate in (dy, dx) raster order.
//
// MI355X mapping: the search is open loop (source against previous source), so it does not sit on the frame-by-frame
// reconstruction chain of a chunk: the inter frames are searched up front, a launch and an event per frame, on a second
// stream while the chain runs.  Grid x/y = (32x32 cells of the frame) x (2R+1 values of dy) = 34 680 waves per 1080p
// frame at R = 8.  A wave owns one cell and one
// dy: lane = 16 consecutive samples of one cell row (source in registers, the 16 + 2R reference samples the
// 2R+1 dx candidates need in registers, each loaded once); per dx the lane forms two 8-sample partial SADs,
// DPP row rotations turn them into the 16 8x8 sub-block SADs of the cell, and the leaf blocks of the
// cell (one 32x32 - its dx candidates then reduced one per lane - or 16x16 / 8x8 blocks at frame edges or smaller block
// sizes) sum their sub-blocks from LDS.
// Each leaf's best candidate of this dy goes into a 64-bit atomicMin on (cost << 16 | candidate index).
// Algorithmic HBM bytes per frame: source luma read once + reference luma read once = 2*L*b (L luma samples);
// the (2R+1)-fold re-reads of the reference by the dy waves of a cell are L2 hits.
#include <hip/hip_runtime.h>
#include "av1mi_dev.h"
#include "av1_tables.h"

namespace {

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// the split mask of frame f's superblock that holds luma position (x, y) (content-driven partition; av1mi_dev.h has the rule)
__device__ __forceinline__ uint32_t sb_mask(const Av1miDevParams &P, int f, int x, int y) {
  return P.part_map ? P.part_map[((size_t)f * P.sb_rows + (y >> 6)) * P.sb_cols + (x >> 6)] : 0u;
}
// Leaf block size (log2) at cell-local 8x8 unit (ux, uy) of the 32x32 cell at (cx, cy) of frame f, or 0 if the unit is not the
// origin of a leaf: the recon kernel's rule; a unit inside a 64x64 leaf reports 0 except at the leaf's origin (6).
__device__ __forceinline__ int leaf_bsl_cell(const Av1miDevParams &P, int f, int cx, int cy, int ux, int uy) {
  const int x = cx + ux * 8, y = cy + uy * 8;
  if (x >= P.width || y >= P.height) return 0;
  return av1mi_leaf_bsl_at(P.width, P.height, P.min_bs_log2, P.max_bs_log2, P.part_map != nullptr, sb_mask(P, f, x, y), x & ~63, y & ~63, x & 63, y & 63);
}

// The 32x32 cell at (cx, cy) lies in a 64x64 LEAF: the node of its superblock does not split (block_log2 = 6, its half point inside
// the frame both ways - it may overhang the edge by less than 32 samples - and, under a content-driven partition, its mask bit clear).
__device__ __forceinline__ bool cell_in_leaf64(const Av1miDevParams &P, int f, int cx, int cy) {
  return !av1mi_node_split(P.width, P.height, P.min_bs_log2, P.max_bs_log2, P.part_map != nullptr, sb_mask(P, f, cx, cy), cx & ~63, cy & ~63, 0, 0, 6);
}

template <typename PIX, int R>
__global__ void __launch_bounds__(64) motion_search_kernel(Av1miDevParams P, const PIX *__restrict__ frames,
                                                          unsigned long long *__restrict__ best_all /* [frame][8x8 unit] */, int frame0,
                                                          int vec_ok /* frames are 16-byte aligned */,
                                                          uint32_t *__restrict__ acc64 /* 64x64 leaves: [frame][superblock][candidate] SAD sums, zeroed */,
                                                          const uint32_t *__restrict__ centre /* pre-search: [frame][superblock] centre codes, or null */) {
  constexpr int NC = 2 * R + 1;
  __shared__ uint32_t sad8[NC][16];  // [dx][8x8 sub-block of the cell, raster]
  const int f = frame0 + blockIdx.z;
  if (!av1mi_frame_is_inter(P, f)) return;  // key frames have no reference
  const PIX *src = frames + (size_t)f * P.frame_samples, *ref = src - P.frame_samples;  // luma of this and of the previous source frame
  unsigned long long *best = best_all + (size_t)f * P.b8_rows * P.b8_cols;
  const int cells_x = (P.width + 31) >> 5;
  const int cell = blockIdx.x, dyi = blockIdx.y;
  const int cx = (cell % cells_x) * 32, cy = (cell / cells_x) * 32;
  // centre of the superblock's search (av1mi_dev.h: motion search keys): whole multiples of 8 luma samples, zero without a pre-search
  const uint32_t ccode = centre ? centre[((size_t)f * P.sb_rows + (cy >> 6)) * P.sb_cols + (cx >> 6)] : 0u;
  const int ccx = 8 * av1mi_sext12(ccode), ccy = 8 * av1mi_sext12(ccode >> 12);
  const unsigned long long ctop = (unsigned long long)ccode << 40;
  const int dy = dyi - R + ccy;
  const int lane = threadIdx.x, r = lane >> 1, half = lane & 1;
  const int W = P.width, H = P.height;
  // ---- source: 16 samples of row cy + r (edge samples replicated outside the frame), reference: samples
  // x0 - R .. x0 + 15 + R of row cy + r + dy (coordinates clamped to the frame); both packed two samples per register:
  // s2[k] = source (2k, 2k+1); rE[k] = reference (2k, 2k+1), rO[k] = (2k+1, 2k+2): the source pair k meets rE[k + dx/2]
  // for even dx and rO[k + (dx-1)/2] for odd dx, compared with v_sad_u16 (4 instructions per 8 samples).
  uint32_t s2[8], rE[8 + R], rO[8 + R];
  {
    const int y = cy + r, xs = cx + half * 16;
    int yr = cy + r + dy;
    yr = yr < 0 ? 0 : (yr > H - 1 ? H - 1 : yr);
    const int xr = xs - R + ccx;
    const PIX *srow = src + (size_t)(y < H ? y : H - 1) * P.stride_y, *rrow = ref + (size_t)yr * P.stride_y;
    if (sizeof(PIX) == 2 && vec_ok && y < H && xs + 16 <= W && xr >= 0 && xr + 16 + 2 * R <= W) {
      // interior, 16-bit samples: 16-byte loads (xs and xr are multiples of 8 samples)
      const uint4 *sv = reinterpret_cast<const uint4 *>(srow + xs), *rv = reinterpret_cast<const uint4 *>(rrow + xr);
#pragma unroll
      for (int q = 0; q < 2; q++) { const uint4 v = sv[q]; s2[4 * q] = v.x; s2[4 * q + 1] = v.y; s2[4 * q + 2] = v.z; s2[4 * q + 3] = v.w; }
#pragma unroll
      for (int q = 0; q < (8 + R) / 4; q++) { const uint4 v = rv[q]; rE[4 * q] = v.x; rE[4 * q + 1] = v.y; rE[4 * q + 2] = v.z; rE[4 * q + 3] = v.w; }
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        // source outside the frame: its last column / row replicated (srow is clamped), what an overhanging leaf is compared with
        const int xa = xs + 2 * k;
        const uint32_t p0 = (uint32_t)srow[xa < W ? xa : W - 1], p1 = (uint32_t)srow[xa + 1 < W ? xa + 1 : W - 1];
        s2[k] = p0 | (p1 << 16);
      }
#pragma unroll
      for (int k = 0; k < 8 + R; k++) {
        int xa = xr + 2 * k, xb = xa + 1;
        xa = xa < 0 ? 0 : (xa > W - 1 ? W - 1 : xa); xb = xb < 0 ? 0 : (xb > W - 1 ? W - 1 : xb);
        rE[k] = (uint32_t)rrow[xa] | ((uint32_t)rrow[xb] << 16);
      }
    }
#pragma unroll
    for (int k = 0; k < 8 + R; k++) rO[k] = k < 8 + R - 1 ? (rE[k] >> 16) | (rE[k + 1] << 16) : 0u;
  }
  // ---- per dx: two 8-sample partial SADs, reduced over the 8 rows of a sub-block (lane bits 1..3)
#pragma unroll
  for (int dxi = 0; dxi < NC; dxi++) {
    uint32_t a = 0, b = 0;
    const int h = dxi >> 1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      a = __builtin_amdgcn_sad_u16(s2[k], (dxi & 1) ? rO[k + h] : rE[k + h], a);
      b = __builtin_amdgcn_sad_u16(s2[k + 4], (dxi & 1) ? rO[k + 4 + h] : rE[k + 4 + h], b);
    }
    // sum over the 8 lanes of the same parity inside each row of 16 lanes (the 8 rows of a sub-block): rotations within the
    // row as DPP operands of the additions (row_ror 2, 4, 8) - a cross-lane shuffle each would go through the LDS pipe
    a += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x122, 0xF, 0xF, false); b += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x122, 0xF, 0xF, false);
    a += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x124, 0xF, 0xF, false); b += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x124, 0xF, 0xF, false);
    a += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x128, 0xF, 0xF, false); b += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x128, 0xF, 0xF, false);
    if ((r & 7) == 0) {  // lanes of the first row of each sub-block row: sub-blocks (r >> 3, 2 * half) and (.., 2 * half + 1)
      sad8[dxi][(r >> 3) * 4 + 2 * half] = a;
      sad8[dxi][(r >> 3) * 4 + 2 * half + 1] = b;
    }
  }
  __syncthreads();
  // ---- a cell of a 64x64 leaf: the leaf's SAD of a candidate is the sum over its four cells - every (cell, dy) wave adds its
  // cell sums to the leaf's candidate table; me64_reduce_kernel takes the minimum afterwards
  if (cell_in_leaf64(P, f, cx, cy)) {
    if (lane < NC) {
      uint32_t sad = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) sad += sad8[lane][i];
      const size_t leaf = (size_t)f * P.sb_rows * P.sb_cols + (size_t)(cy >> 6) * P.sb_cols + (cx >> 6);
      atomicAdd(&acc64[(leaf * NC + dyi) * NC + lane], sad);
    }
    return;
  }
  // ---- the usual cell: one 32x32 leaf.  Lane = dx candidate: each sums the 16 sub-block SADs of its dx, a wave minimum
  // over (cost, candidate) picks the best of this dy (as one lane looping over the candidates it was a third of the kernel)
  if (leaf_bsl_cell(P, f, cx, cy, 0, 0) == 5) {
    if (cy + dy >= -16 && cy + dy + 32 <= H + 16) {
      unsigned long long key = ~0ull;
      if (lane < NC) {
        const int dx = lane - R + ccx;
        if (!(cx + dx < -16 || cx + dx + 32 > W + 16)) {
          uint32_t sad = 0;
#pragma unroll
          for (int i = 0; i < 16; i++) sad += sad8[lane][i];
          const unsigned long long cost = (unsigned long long)sad + (unsigned long long)(32 * (iabs(dx) + iabs(dy)));
          key = ctop | (cost << 16) | (unsigned long long)(dyi * NC + lane);
        }
      }
      for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(key, o, 64); key = t < key ? t : key; }
      if (lane == 0 && key != ~0ull) atomicMin(&best[(size_t)(cy >> 3) * P.b8_cols + (cx >> 3)], key);
    }
    return;
  }
  // ---- other cells (smaller blocks, frame edges): lane u < 16 = 8x8 unit u; a leaf origin sums its sub-blocks for every dx
  if (lane < 16) {
    const int ux = lane & 3, uy = lane >> 2;
    const int bsl = leaf_bsl_cell(P, f, cx, cy, ux, uy);
    if (bsl) {
      const int n = 1 << bsl, n8 = n >> 3;
      const int x = cx + ux * 8, y = cy + uy * 8;
      if (y + dy >= -16 && y + dy + n <= H + 16) {
        unsigned long long bk = ~0ull;
        for (int dxi = 0; dxi < NC; dxi++) {
          const int dx = dxi - R + ccx;
          if (x + dx < -16 || x + dx + n > W + 16) continue;
          uint32_t sad = 0;
          for (int i = 0; i < n8; i++)
            for (int j = 0; j < n8; j++) sad += sad8[dxi][(uy + i) * 4 + ux + j];
          const unsigned long long cost = (unsigned long long)sad + (unsigned long long)(n * (iabs(dx) + iabs(dy)));
          const unsigned long long key = ctop | (cost << 16) | (unsigned long long)(dyi * NC + dxi);
          bk = key < bk ? key : bk;
        }
        if (bk != ~0ull) atomicMin(&best[(size_t)((cy >> 3) + uy) * P.b8_cols + (cx >> 3) + ux], bk);
      }
    }
  }
}

// One wave per superblock that is a 64x64 leaf: cost and minimum over the candidate table the search filled (same rule: cost =
// SAD + 64 * (|dx| + |dy|), the displaced block within 16 samples of the frame, ties to the first candidate in raster order).
__global__ void __launch_bounds__(64) me64_reduce_kernel(Av1miDevParams P, const uint32_t *__restrict__ acc64,
                                                        unsigned long long *__restrict__ best_all, int frame0, int R, const uint32_t *__restrict__ centre) {
  const int f = frame0 + blockIdx.y;
  if (!av1mi_frame_is_inter(P, f)) return;
  const int sb = blockIdx.x, x = (sb % P.sb_cols) * 64, y = (sb / P.sb_cols) * 64;
  if (!cell_in_leaf64(P, f, x, y)) return;
  const int NC = 2 * R + 1, lane = threadIdx.x;
  const uint32_t ccode = centre ? centre[(size_t)f * P.sb_rows * P.sb_cols + sb] : 0u;
  const int ccx = 8 * av1mi_sext12(ccode), ccy = 8 * av1mi_sext12(ccode >> 12);
  const unsigned long long ctop = (unsigned long long)ccode << 40;
  const uint32_t *tab = acc64 + ((size_t)f * P.sb_rows * P.sb_cols + sb) * NC * NC;
  unsigned long long key = ~0ull;
  for (int c = lane; c < NC * NC; c += 64) {
    const int dy = c / NC - R + ccy, dx = c % NC - R + ccx;
    if (x + dx < -16 || x + dx + 64 > P.width + 16 || y + dy < -16 || y + dy + 64 > P.height + 16) continue;
    const unsigned long long cost = (unsigned long long)tab[c] + (unsigned long long)(64 * (iabs(dx) + iabs(dy)));
    const unsigned long long k = ctop | (cost << 16) | (unsigned long long)c;
    key = k < key ? k : key;
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(key, o, 64); key = t < key ? t : key; }
  if (lane == 0) best_all[(size_t)f * P.b8_rows * P.b8_cols + (size_t)(y >> 3) * P.b8_cols + (x >> 3)] = key;
}

// ---------------------------------------------------------------------------------------------------------------
// Sub-sample refinement (subpel = 1; oracle/av1o_enc.c motion_search, second half).
// Around the full search's winner: the 8 half-sample neighbours, then the 8 quarter-sample neighbours of that stage's
// best.  A candidate is the block interpolated from the previous SOURCE frame with the prediction's own filter
// (EIGHTTAP, Round2 by 3 after the horizontal pass, by 11 and a clamp after the vertical one); cost = SATD (sum of the
// absolute 8x8 Hadamard coefficients of the difference, >> 3) + (n * (|mv.row| + |mv.col|) >> 3), the integer winner
// re-costed the same way first; a candidate replaces the best only when strictly cheaper, candidates in (row, col)
// raster order; the displaced block stays within 16 samples of the frame.
// One wave per cell of the largest block size (its leaf; the smaller leaves of a cell that straddles the frame edge in
// turn).  The (n + 8)^2 reference window
// around the integer winner - every candidate's taps lie inside it - and the source block are staged in LDS once, all
// loads in flight together.  Per candidate: horizontal pass, a lane = 8 consecutive outputs of a row from 16 window
// samples read as two 16-byte words (64 multiply-adds per 2 LDS reads); vertical pass, a lane = n^2/64 consecutive
// outputs of a column from its (n^2/64 + 7) intermediate values; SAD against the source block, wave sum.
// The Hadamard transform of 32x32 blocks (and of the quadrants of 64x64 ones) runs on the matrix cores straight from the vertical
// pass's registers (refine_eval, measured against the butterflies with tools/mfma_satd_ab.hip); the smaller blocks' runs as
// butterflies (24 additions per 8 values, rows through LDS, then columns).
// Output: (SAD << 36) | (u16 mv.row << 16) | u16 mv.col per leaf, which is what the recon kernel reads.
__constant__ int16_t c_subpel_me[2][16][8] = AV1_SUBPEL_FILTERS_INIT;

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

// 16 values of at most 15 bits + sign -> the byte fragments of their high bytes (v >> 8) and biased low bytes ((v & 255) - 128)
__device__ __forceinline__ void satd_split16(const int *v, v4i_t &lo, v4i_t &hi) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const unsigned a = (unsigned)v[4 * i], b = (unsigned)v[4 * i + 1], c = (unsigned)v[4 * i + 2], d = (unsigned)v[4 * i + 3];
    const unsigned ab = __builtin_amdgcn_perm(b, a, 0x05040100u);   // a.b0 a.b1 b.b0 b.b1
    const unsigned cd = __builtin_amdgcn_perm(d, c, 0x05040100u);
    lo[i] = (int)(__builtin_amdgcn_perm(cd, ab, 0x06040200u) ^ 0x80808080u);
    hi[i] = (int)__builtin_amdgcn_perm(cd, ab, 0x07050301u);
  }
}

__device__ __forceinline__ void hadamard8(int *v) {
  int t[8];
  t[0] = v[0] + v[4]; t[4] = v[0] - v[4]; t[1] = v[1] + v[5]; t[5] = v[1] - v[5]; t[2] = v[2] + v[6]; t[6] = v[2] - v[6]; t[3] = v[3] + v[7]; t[7] = v[3] - v[7];
  v[0] = t[0] + t[2]; v[2] = t[0] - t[2]; v[1] = t[1] + t[3]; v[3] = t[1] - t[3]; v[4] = t[4] + t[6]; v[6] = t[4] - t[6]; v[5] = t[5] + t[7]; v[7] = t[5] - t[7];
  t[0] = v[0] + v[1]; t[1] = v[0] - v[1]; t[2] = v[2] + v[3]; t[3] = v[2] - v[3]; t[4] = v[4] + v[5]; t[5] = v[4] - v[5]; t[6] = v[6] + v[7]; t[7] = v[6] - v[7];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = t[i];
}

// The (n + 8)^2 reference window at (wx, wy) and the n x n source block at (x, y) -> LDS, all loads in flight together
// (coordinates clamped to the frame: the reference as motion compensation clamps it, the source of an overhanging leaf with its
// last row / column replicated).
template <typename PIX, int LOG2N>
__device__ __forceinline__ void refine_stage(const Av1miDevParams &P, const PIX *__restrict__ src, const PIX *__restrict__ ref, int x, int y,
                                             int wx, int wy, uint16_t *win, uint16_t *srcb) {
  constexpr int n = 1 << LOG2N, WW = n + 8, WS = n + 16;   // window width, row stride (rows stay 16-byte aligned)
  constexpr int RPL = n * n / 64;
  const int lane = threadIdx.x, W = P.width, H = P.height;
  constexpr int TOT = WW * WW, K = (TOT + 63) / 64;
  uint16_t v[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int p = lane + 64 * k, i = p / WW, j = p - i * WW;
    int yy = wy + i, xx = wx + j;
    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
    xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
    v[k] = p < TOT ? (uint16_t)ref[(size_t)yy * P.stride_y + xx] : 0;
  }
  uint16_t sv[RPL];
#pragma unroll
  for (int k = 0; k < RPL; k++) {
    const int p = lane + 64 * k;
    int yy = y + (p >> LOG2N), xx = x + (p & (n - 1));
    yy = yy > H - 1 ? H - 1 : yy; xx = xx > W - 1 ? W - 1 : xx;
    sv[k] = (uint16_t)src[(size_t)yy * P.stride_y + xx];
  }
#pragma unroll
  for (int k = 0; k < K; k++) { const int p = lane + 64 * k, i = p / WW, j = p - i * WW; if (p < TOT) win[i * WS + j] = v[k]; }
#pragma unroll
  for (int k = 0; k < RPL; k++) srcb[lane + 64 * k] = sv[k];
  __syncthreads();
}

// One candidate (mr, mc) of the n x n block at (x, y) from the staged window: interpolation (EIGHTTAP, both roundings), then the
// wave sums of the absolute differences and of the absolute 8x8 Hadamard coefficients (not yet >> 3) against the source block.
template <typename PIX, int LOG2N>
__device__ __forceinline__ void refine_eval(const Av1miDevParams &P, int x, int y, int wx, int wy, int mr, int mc,
                                            const uint16_t *win, int16_t *mid, const uint16_t *srcb, int16_t *dif, int &sad_out, int &satd_out) {
  constexpr int n = 1 << LOG2N, WS = n + 16;
  constexpr int RPL = n * n / 64;                            // vertical pass: output rows per lane
  const int lane = threadIdx.x;
  const int maxv = (1 << P.bit_depth) - 1;
  const int px = (x << 4) + 2 * mc, py = (y << 4) + 2 * mr;
  const int ix = (px >> 4) - 3 - wx, iy = (py >> 4) - 3 - wy, fx = px & 15, fy = py & 15;   // 0 <= ix, iy <= 1
  int fh[8], fv[8];
#pragma unroll
  for (int t = 0; t < 8; t++) { fh[t] = c_subpel_me[0][fx][t]; fv[t] = c_subpel_me[0][fy][t]; }
  // horizontal pass: task = (row r of mid, segment of 8 outputs)
  constexpr int SEGS = n / 8, TASKS = (n + 7) * SEGS;
  for (int task = lane; task < TASKS; task += 64) {
    const int r = task / SEGS, sg = task - r * SEGS;
    const uint4 *wp = reinterpret_cast<const uint4 *>(win + (iy + r) * WS + 8 * sg);
    const uint4 q0 = wp[0], q1 = wp[1];
    const uint32_t d[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
    int a[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[2 * i] = (int)(d[i] & 0xFFFF); a[2 * i + 1] = (int)(d[i] >> 16); }
    int o[8];
    if (ix) {
#pragma unroll
      for (int j = 0; j < 8; j++) { int sum = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) sum += fh[t] * a[1 + j + t];
        o[j] = (sum + 4) >> 3; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) { int sum = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) sum += fh[t] * a[j + t];
        o[j] = (sum + 4) >> 3; }
    }
    uint4 w;
    w.x = (uint32_t)(uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16); w.y = (uint32_t)(uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16);
    w.z = (uint32_t)(uint16_t)o[4] | ((uint32_t)(uint16_t)o[5] << 16); w.w = (uint32_t)(uint16_t)o[6] | ((uint32_t)(uint16_t)o[7] << 16);
    *reinterpret_cast<uint4 *>(mid + r * n + 8 * sg) = w;
  }
  __syncthreads();
  // vertical pass: lane = column c, rows r0 .. r0 + RPL - 1
  int sad = 0;
  [[maybe_unused]] int dd[RPL];   // n = 32: the lane's 16 differences D[16h + j][c] - the first matrix product's A fragment
  {
    const int c = lane & (n - 1), r0 = (lane >> LOG2N) * RPL;
    int m[RPL + 7];
#pragma unroll
    for (int i = 0; i < RPL + 7; i++) m[i] = mid[(r0 + i) * n + c];
#pragma unroll
    for (int j = 0; j < RPL; j++) {
      int sum = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) sum += fv[t] * m[j + t];
      int v = (sum + 1024) >> 11;
      v = v < 0 ? 0 : (v > maxv ? maxv : v);
      const int d = (int)srcb[(r0 + j) * n + c] - v;
      sad += iabs(d);
      if constexpr (LOG2N == 5) dd[j] = d;
      else dif[(r0 + j) * n + c] = (int16_t)d;
    }
  }
  if constexpr (LOG2N == 5) {
    // SATD on the matrix cores (tools/mfma_satd_ab.hip: 0.36 against 0.47 ns per candidate for the butterflies below, and no LDS
    // round trip or barrier on a kernel that runs two waves per SIMD): Y = H' D^T H' with H' = I4 (x) H8, +-1 / 0 entries, the
    // operands split into a signed high byte and a biased low byte - v = 256 (v >> 8) + ((v & 255) - 128) + 128, the + 128 being
    // a constant matrix whose transform (1024 on the Hadamard index 0 of every sub-block) goes into the accumulators' start value.
    // v_mfma_i32_32x32x32_i8: lane (r, h) holds A[r][16h + j], B[16h + j][r]; accumulator reg of lane (c, h) = row
    // (reg & 3) + 8 (reg >> 2) + 4h, column c - so the first product's accumulators are the second's B fragment with k in that order.
    const int c = lane & 31, h = lane >> 5;
    v4i_t hb = {0, 0, 0, 0}, ha = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int k1 = 16 * h + j, k2 = (j & 3) + 8 * (j >> 2) + 4 * h;
      const int e1 = (k1 >> 3) == (c >> 3) ? ((__builtin_popcount(k1 & c & 7) & 1) ? 0xFF : 0x01) : 0;
      const int e2 = (k2 >> 3) == (c >> 3) ? ((__builtin_popcount(k2 & c & 7) & 1) ? 0xFF : 0x01) : 0;
      hb[j >> 2] |= e1 << (8 * (j & 3));
      ha[j >> 2] |= e2 << (8 * (j & 3));
    }
    const int bias1 = (c & 7) == 0 ? 1024 : 0, bias2 = h == 0 ? 1024 : 0;
    v4i_t lo, hi;
    satd_split16(dd, lo, hi);
    v16i_t acc = {0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi, hb, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = (acc[r] << 8) + bias1;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo, hb, acc, 0, 0, 0);
    int t[16];
#pragma unroll
    for (int r = 0; r < 16; r++) t[r] = acc[r];
    satd_split16(t, lo, hi);
    v16i_t acc2 = {0};
    acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ha, hi, acc2, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) acc2[r] = (acc2[r] << 8) + ((r & 3) == 0 ? bias2 : 0);
    acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ha, lo, acc2, 0, 0, 0);
    int satd = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) satd += iabs(acc2[r]);
    for (int o = 32; o > 0; o >>= 1) { sad += __shfl_xor(sad, o, 64); satd += __shfl_xor(satd, o, 64); }
    __syncthreads();   // `mid` is rewritten by the next candidate's horizontal pass
    sad_out = sad; satd_out = satd;
    return;
  }
  __syncthreads();
  // SATD: 8x8 Hadamard of the difference, rows (8 values of a row of a sub-block per task, written back), then
  // columns (absolute sum); n^2/64 sub-blocks x 8 tasks each
  constexpr int NB = n / 8, HT = NB * NB * 8;
  for (int task = lane; task < HT; task += 64) {
    const int b = task >> 3, i = task & 7;
    int16_t *rowp = dif + ((b / NB) * 8 + i) * n + (b % NB) * 8;
    const uint4 q = *reinterpret_cast<const uint4 *>(rowp);
    int v[8] = { (int16_t)(q.x & 0xFFFF), (int16_t)(q.x >> 16), (int16_t)(q.y & 0xFFFF), (int16_t)(q.y >> 16),
                 (int16_t)(q.z & 0xFFFF), (int16_t)(q.z >> 16), (int16_t)(q.w & 0xFFFF), (int16_t)(q.w >> 16) };
    hadamard8(v);
    uint4 w;
    w.x = (uint32_t)(uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16); w.y = (uint32_t)(uint16_t)v[2] | ((uint32_t)(uint16_t)v[3] << 16);
    w.z = (uint32_t)(uint16_t)v[4] | ((uint32_t)(uint16_t)v[5] << 16); w.w = (uint32_t)(uint16_t)v[6] | ((uint32_t)(uint16_t)v[7] << 16);
    *reinterpret_cast<uint4 *>(rowp) = w;
  }
  __syncthreads();
  int satd = 0;
  for (int task = lane; task < HT; task += 64) {
    const int b = task >> 3, j = task & 7;
    const int16_t *colp = dif + (b / NB) * 8 * n + (b % NB) * 8 + j;
    int v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = colp[i * n];
    hadamard8(v);
#pragma unroll
    for (int i = 0; i < 8; i++) satd += iabs(v[i]);
  }
  for (int o = 32; o > 0; o >>= 1) { sad += __shfl_xor(sad, o, 64); satd += __shfl_xor(satd, o, 64); }
  __syncthreads();
  sad_out = sad; satd_out = satd;
}

template <typename PIX, int LOG2N>
__device__ __forceinline__ void refine_leaf(const Av1miDevParams &P, const PIX *__restrict__ src, const PIX *__restrict__ ref, int x, int y,
                                            int &best_row, int &best_col, int &best_sad, uint16_t *win, int16_t *mid, uint16_t *srcb, int16_t *dif) {
  constexpr int n = 1 << LOG2N;
  const int W = P.width, H = P.height;
  const int wx = x + (best_col >> 3) - 4, wy = y + (best_row >> 3) - 4;   // window origin (the integer winner is a multiple of 8)
  refine_stage<PIX, LOG2N>(P, src, ref, x, y, wx, wy, win, srcb);
  long best_cost = 0;
  // step 8 = the integer winner itself, re-costed in SATD (phase 0: the filters copy); then half and quarter samples
  for (int step = 8; step >= 2; step >>= 1) {
    const int base_row = best_row, base_col = best_col;
    for (int k = 0; k < 9; k++) {
      if ((k == 4) != (step == 8)) continue;
      const int mr = base_row + (step == 8 ? 0 : (k / 3 - 1) * step), mc = base_col + (step == 8 ? 0 : (k % 3 - 1) * step);
      if (x * 8 + mc < -128 || (x + n) * 8 + mc > (W + 16) * 8 || y * 8 + mr < -128 || (y + n) * 8 + mr > (H + 16) * 8) continue;
      int sad, satd;
      refine_eval<PIX, LOG2N>(P, x, y, wx, wy, mr, mc, win, mid, srcb, dif, sad, satd);
      const long cost = (long)(satd >> 3) + (((long)n * (iabs(mr) + iabs(mc))) >> 3);
      if (step == 8 || cost < best_cost) { best_cost = cost; best_sad = sad; best_row = mr; best_col = mc; }
    }
  }
}

// A 64x64 leaf (block_log2 = 6): SAD and SATD are sums over 8x8 sub-blocks, so the leaf is evaluated as its four 32x32 quadrants with
// the same LDS tiles - per stage every quadrant is staged once (window around the leaf's integer winner) and gives its share
// of every candidate of the stage; the decision is taken on the sums.  Same candidates, order and costs as the smaller leaves.
template <typename PIX>
__device__ __forceinline__ void refine_leaf64(const Av1miDevParams &P, const PIX *__restrict__ src, const PIX *__restrict__ ref, int x, int y,
                                              int &best_row, int &best_col, int &best_sad, uint16_t *win, int16_t *mid, uint16_t *srcb, int16_t *dif) {
  constexpr int n = 64;
  const int W = P.width, H = P.height;
  const int irow = best_row, icol = best_col;   // the integer winner: every quadrant's window sits around it
  long best_cost = 0;
  for (int step = 8; step >= 2; step >>= 1) {
    const int base_row = best_row, base_col = best_col;
    int sad_acc[9], satd_acc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { sad_acc[k] = 0; satd_acc[k] = 0; }
    for (int q = 0; q < 4; q++) {
      const int qx = x + (q & 1) * 32, qy = y + (q >> 1) * 32;
      const int wx = qx + (icol >> 3) - 4, wy = qy + (irow >> 3) - 4;
      refine_stage<PIX, 5>(P, src, ref, qx, qy, wx, wy, win, srcb);
#pragma unroll
      for (int k = 0; k < 9; k++) {
        if ((k == 4) != (step == 8)) continue;
        const int mr = base_row + (step == 8 ? 0 : (k / 3 - 1) * step), mc = base_col + (step == 8 ? 0 : (k % 3 - 1) * step);
        if (x * 8 + mc < -128 || (x + n) * 8 + mc > (W + 16) * 8 || y * 8 + mr < -128 || (y + n) * 8 + mr > (H + 16) * 8) continue;
        int sad, satd;
        refine_eval<PIX, 5>(P, qx, qy, wx, wy, mr, mc, win, mid, srcb, dif, sad, satd);
        sad_acc[k] += sad; satd_acc[k] += satd;
      }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
      if ((k == 4) != (step == 8)) continue;
      const int mr = base_row + (step == 8 ? 0 : (k / 3 - 1) * step), mc = base_col + (step == 8 ? 0 : (k % 3 - 1) * step);
      if (x * 8 + mc < -128 || (x + n) * 8 + mc > (W + 16) * 8 || y * 8 + mr < -128 || (y + n) * 8 + mr > (H + 16) * 8) continue;
      const long cost = (long)(satd_acc[k] >> 3) + (((long)n * (iabs(mr) + iabs(mc))) >> 3);
      if (step == 8 || cost < best_cost) { best_cost = cost; best_sad = sad_acc[k]; best_row = mr; best_col = mc; }
    }
  }
}

template <typename PIX>
__global__ void __launch_bounds__(64) subpel_refine_kernel(Av1miDevParams P, const PIX *__restrict__ frames,
                                                          const unsigned long long *__restrict__ in_all, unsigned long long *__restrict__ out_all,
                                                          int frame0, int R, int cell_log2) {
  __shared__ __attribute__((aligned(16))) uint16_t win[40 * 48];
  __shared__ __attribute__((aligned(16))) int16_t mid[39 * 32];
  __shared__ uint16_t srcb[32 * 32];
  __shared__ __attribute__((aligned(16))) int16_t dif[32 * 32];
  const int f = frame0 + blockIdx.z;
  if (!av1mi_frame_is_inter(P, f)) return;
  const PIX *src = frames + (size_t)f * P.frame_samples, *ref = src - P.frame_samples;
  const int NC = 2 * R + 1;
  // this wave's cell of the largest block size: one leaf inside the frame, a few smaller ones where it straddles the edge
  const int u = 1 << (cell_log2 - 3);
  if (cell_log2 == 6 && cell_in_leaf64(P, f, blockIdx.x * 64, blockIdx.y * 64)) {   // this wave's cell is one 64x64 leaf
    const size_t slot = (size_t)f * P.b8_rows * P.b8_cols + (size_t)(blockIdx.y * 8) * P.b8_cols + blockIdx.x * 8;
    int idy, idx_, icost;
    av1mi_me_key_decode(in_all[slot], R, &idy, &idx_, &icost);
    int best_row = idy * 8, best_col = idx_ * 8, best_sad = 0;
    refine_leaf64<PIX>(P, src, ref, blockIdx.x * 64, blockIdx.y * 64, best_row, best_col, best_sad, win, mid, srcb, dif);
    if (threadIdx.x == 0)
      out_all[slot] = ((unsigned long long)(uint32_t)best_sad << 36) | ((unsigned long long)(uint16_t)(int16_t)best_row << 16) | (uint16_t)(int16_t)best_col;
    return;
  }
  for (int uy = blockIdx.y * u; uy < (int)(blockIdx.y + 1) * u; uy++)
    for (int ux = blockIdx.x * u; ux < (int)(blockIdx.x + 1) * u; ux++) {
      if (ux * 8 >= P.width || uy * 8 >= P.height) continue;
      const int bsl = leaf_bsl_cell(P, f, (ux >> 2) * 32, (uy >> 2) * 32, ux & 3, uy & 3);
      if (!bsl) continue;
      const size_t slot = (size_t)f * P.b8_rows * P.b8_cols + (size_t)uy * P.b8_cols + ux;
      const unsigned long long key = in_all[slot];   // the full search's key: (cost << 16) | candidate index
      int idy, idx_, icost;
      av1mi_me_key_decode(key, R, &idy, &idx_, &icost);
      int best_row = idy * 8, best_col = idx_ * 8;
      int best_sad = 0;
      switch (bsl) {
        case 5: refine_leaf<PIX, 5>(P, src, ref, ux * 8, uy * 8, best_row, best_col, best_sad, win, mid, srcb, dif); break;
        case 4: refine_leaf<PIX, 4>(P, src, ref, ux * 8, uy * 8, best_row, best_col, best_sad, win, mid, srcb, dif); break;
        default: refine_leaf<PIX, 3>(P, src, ref, ux * 8, uy * 8, best_row, best_col, best_sad, win, mid, srcb, dif); break;
      }
      if (threadIdx.x == 0)
        out_all[slot] = ((unsigned long long)(uint32_t)best_sad << 36) | ((unsigned long long)(uint16_t)(int16_t)best_row << 16) | (uint16_t)(int16_t)best_col;
    }
}

// ---- hierarchical search, first level (av1mi_params.me_presearch; SURVEY.md §8a row a13 "hierarchical: 1/4-res ... full").
// quarter_luma_kernel: q(Y, X) = (sum of the 4x4 luma samples + 8) >> 4 of every frame of the chunk (16-bit whatever the bit depth);
// one thread per quarter-resolution sample, four 4-sample loads each.  Algorithmic bytes L b + L / 8 per frame.
template <typename PIX>
__global__ void __launch_bounds__(256) quarter_luma_kernel(Av1miDevParams P, const PIX *__restrict__ frames, uint16_t *__restrict__ quarter) {
  const int qw = P.width >> 2, qh = P.height >> 2, f = blockIdx.y;
  const PIX *luma = frames + (size_t)f * P.frame_samples;
  uint16_t *q = quarter + (size_t)f * qw * qh;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < qw * qh; i += gridDim.x * 256) {
    const int y = i / qw, x = i - y * qw;
    uint32_t s = 8;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const PIX *row = luma + (size_t)(4 * y + r) * P.stride_y + 4 * x;
      if (sizeof(PIX) == 2) { const uint2 v = *reinterpret_cast<const uint2 *>(row); s += (v.x & 0xFFFF) + (v.x >> 16) + (v.y & 0xFFFF) + (v.y >> 16); }
      else { const uint32_t v = *reinterpret_cast<const uint32_t *>(row); s += (v & 0xFF) + ((v >> 8) & 0xFF) + ((v >> 16) & 0xFF) + (v >> 24); }
    }
    q[i] = (uint16_t)(s >> 4);
  }
}

// presearch_kernel: one workgroup of two waves per superblock of an inter frame.  The superblock's 16x16 quarter-resolution block and the
// 48x48 window of the previous frame around it (coordinates clamped to the plane) are staged in LDS; lane = dqx + 16, each wave takes half
// of the dqy range; a candidate's SAD two samples per instruction (the block's pair is a broadcast read, the window's pair an unaligned
// 4-byte read), cost = SAD + 16 (|dqx| + |dqy|); a candidate whose centre would push the superblock more than 16 samples out of the frame is
// skipped; minimum over (cost, raster index).  Writes the centre code of the winner rounded to whole multiples of 8 luma samples
// (av1mi_dev.h: motion search keys).  (As one wave with one sample per instruction it took as long as a frame's chain step, and the chain
// waited for it: 1080p IPPP 5.36 -> 4.89 k frames/s.)
__global__ void __launch_bounds__(128) presearch_kernel(Av1miDevParams P, const uint16_t *__restrict__ quarter, uint32_t *__restrict__ centre, int frame0) {
  __shared__ __attribute__((aligned(16))) uint16_t cur[16][16];
  __shared__ __attribute__((aligned(16))) uint16_t win[48][50];
  __shared__ uint32_t wbest[2];
  typedef uint32_t u32_any __attribute__((aligned(2)));
  const int f = frame0 + blockIdx.y;
  if (!av1mi_frame_is_inter(P, f)) return;
  const int sb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sx = (sb % P.sb_cols) * 64, sy = (sb / P.sb_cols) * 64;
  const int W = P.width, H = P.height, qw = W >> 2, qh = H >> 2;
  const uint16_t *qc = quarter + (size_t)f * qw * qh, *qp = qc - (size_t)qw * qh;
  for (int i = tid; i < 256; i += 128) {
    int y = (sy >> 2) + (i >> 4), x = (sx >> 2) + (i & 15);
    y = y > qh - 1 ? qh - 1 : y; x = x > qw - 1 ? qw - 1 : x;
    cur[i >> 4][i & 15] = qc[(size_t)y * qw + x];
  }
  for (int i = tid; i < 48 * 48; i += 128) {
    const int r = i / 48, c = i - r * 48;
    // window sample (r, c) = previous plane at (block row - 16 + r, block column - 16 + c), coordinates clamped to the plane
    int y = (sy >> 2) - 16 + r, x = (sx >> 2) - 16 + c;
    y = y < 0 ? 0 : (y > qh - 1 ? qh - 1 : y); x = x < 0 ? 0 : (x > qw - 1 ? qw - 1 : x);
    win[r][c] = qp[(size_t)y * qw + x];
  }
  __syncthreads();
  const int sbw = W - sx < 64 ? W - sx : 64, sbh = H - sy < 64 ? H - sy : 64;
  uint32_t best = 0xFFFFFFFFu;
  const int dqx = lane - 16;
  // wave 0: dqy -16 .. 0, wave 1: 1 .. 16
  for (int dqy = wave ? 1 : -16; dqy <= (wave ? 16 : 0); dqy++) {
    if (lane > 32) continue;
    const int ccx = ((dqx + 1) >> 1) * 8, ccy = ((dqy + 1) >> 1) * 8;
    if (sx + ccx < -16 || sx + ccx + sbw > W + 16 || sy + ccy < -16 || sy + ccy + sbh > H + 16) continue;
    uint32_t sad = 0;
    for (int i = 0; i < 16; i++) {
      const uint32_t *cr = reinterpret_cast<const uint32_t *>(&cur[i][0]);
      const uint16_t *wr = &win[i + dqy + 16][dqx + 16];
#pragma unroll
      for (int k = 0; k < 8; k++) sad = __builtin_amdgcn_sad_u16(cr[k], *reinterpret_cast<const u32_any *>(wr + 2 * k), sad);
    }
    const uint32_t cost = sad + 16u * (uint32_t)(iabs(dqx) + iabs(dqy));
    const uint32_t key = (cost << 11) | (uint32_t)((dqy + 16) * 33 + lane);
    best = key < best ? key : best;
  }
  for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(best, o, 64); best = t < best ? t : best; }
  if (lane == 0) wbest[wave] = best;
  __syncthreads();
  if (tid == 0) {
    const uint32_t bb = wbest[0] < wbest[1] ? wbest[0] : wbest[1];
    const int idx = (int)(bb & 0x7FF), bqy = idx / 33 - 16, bqx = idx % 33 - 16;
    const int c8y = (bqy + 1) >> 1, c8x = (bqx + 1) >> 1;   // centre / 8
    centre[(size_t)f * P.sb_rows * P.sb_cols + sb] = ((uint32_t)(c8y & 0xFFF) << 12) | (uint32_t)(c8x & 0xFFF);
  }
}

}  // namespace

// quarter-resolution luma of all P->n_frames frames of the chunk
extern "C" hipError_t av1mi_launch_quarter_luma(const Av1miDevParams *P, const void *frames, uint16_t *quarter, hipStream_t stream) {
  dim3 grid(128, P->n_frames);
  if (P->bit_depth == 8) hipLaunchKernelGGL(quarter_luma_kernel<uint8_t>, grid, dim3(256), 0, stream, *P, (const uint8_t *)frames, quarter);
  else hipLaunchKernelGGL(quarter_luma_kernel<uint16_t>, grid, dim3(256), 0, stream, *P, (const uint16_t *)frames, quarter);
  return hipGetLastError();
}
// search centres of frames [frame0, frame0 + count) (key frames are skipped)
extern "C" hipError_t av1mi_launch_presearch(const Av1miDevParams *P, const uint16_t *quarter, uint32_t *centre, int frame0, int count, hipStream_t stream) {
  hipLaunchKernelGGL(presearch_kernel, dim3(P->sb_rows * P->sb_cols, count), dim3(128), 0, stream, *P, quarter, centre, frame0);
  return hipGetLastError();
}

// Refines the vectors of frames [frame0, frame0 + count): `best` = the full search's keys (av1mi_launch_motion_search on the
// same stream before), `refined` = same layout, what the recon kernel reads with subpel = 1.
extern "C" hipError_t av1mi_launch_subpel_refine(const Av1miDevParams *P, const void *frames, const unsigned long long *best,
                                                 unsigned long long *refined, int me_range, int frame0, int count, hipStream_t stream) {
  // one wave per cell of the largest block size (every wave has work, and consecutive workgroups - which the dispatcher deals
  // round-robin to the 8 XCDs - are all active: a wave per 8x8 unit left 6 of 8 XCDs without a single leaf)
  const int g = P->max_bs_log2 > 6 ? 6 : P->max_bs_log2, cell = 1 << g;
  dim3 grid((P->width + cell - 1) / cell, (P->height + cell - 1) / cell, count);
  if (P->bit_depth == 8) hipLaunchKernelGGL((subpel_refine_kernel<uint8_t>), grid, dim3(64), 0, stream, *P, (const uint8_t *)frames, best, refined, frame0, me_range, g);
  else hipLaunchKernelGGL((subpel_refine_kernel<uint16_t>), grid, dim3(64), 0, stream, *P, (const uint16_t *)frames, best, refined, frame0, me_range, g);
  return hipGetLastError();
}

// best[] (n_frames x 8x8 units) must be filled with 0xFF bytes before the launch.  `frames`: the chunk's source frames
// (P->n_frames of them); every inter frame is searched against the source frame before it.  R must be 8 or 16.
// Searches frames [frame0, frame0 + count) of the chunk.
extern "C" hipError_t av1mi_launch_motion_search(const Av1miDevParams *P, const void *frames, unsigned long long *best, int me_range,
                                                 int frame0, int count, uint32_t *acc64 /* block_log2 = 6: zeroed [frame][superblock][candidate] table, else null */,
                                                 const uint32_t *centre /* pre-search centre codes [frame][superblock], or null */, hipStream_t stream) {
  const int cells = ((P->width + 31) >> 5) * ((P->height + 31) >> 5);
  dim3 grid(cells, 2 * me_range + 1, count);
  const int vec_ok = ((uintptr_t)frames & 15) == 0;  // frame size in bytes is a multiple of 32
  if (P->bit_depth == 8) {
    if (me_range == 8) hipLaunchKernelGGL((motion_search_kernel<uint8_t, 8>), grid, dim3(64), 0, stream, *P, (const uint8_t *)frames, best, frame0, vec_ok, acc64, centre);
    else hipLaunchKernelGGL((motion_search_kernel<uint8_t, 16>), grid, dim3(64), 0, stream, *P, (const uint8_t *)frames, best, frame0, vec_ok, acc64, centre);
  } else {
    if (me_range == 8) hipLaunchKernelGGL((motion_search_kernel<uint16_t, 8>), grid, dim3(64), 0, stream, *P, (const uint16_t *)frames, best, frame0, vec_ok, acc64, centre);
    else hipLaunchKernelGGL((motion_search_kernel<uint16_t, 16>), grid, dim3(64), 0, stream, *P, (const uint16_t *)frames, best, frame0, vec_ok, acc64, centre);
  }
  if (P->max_bs_log2 >= 6)
    hipLaunchKernelGGL(me64_reduce_kernel, dim3(P->sb_rows * P->sb_cols, count), dim3(64), 0, stream, *P, acc64, best, frame0, me_range, centre);
  return hipGetLastError();
}
