// cdef_pack_kernels.hip - (1) CDEF of the reconstructed frames, (2) bitstream packing.
//
// (1) replaces the CDEF stage of the external SVT-AV1 worker behind `run_av1an`
//     (/root/reference/crates/daemon/src/encode/av1an.rs:126-139):
//     AV1 spec §7.15 - 8x8 direction search (§7.15.2) and the constrained primary/secondary
//     filter (§7.15.3), 4:2:0, one strength set per frame (cdef_bits = 0).
//     MI355X mapping: one wave per 64x64 superblock; luma (68x68) and chroma (36x36 x2) tiles incl.
//     the 2-pixel halo are staged once into LDS with coalesced row loads, unavailable (outside
//     frame) samples carry a sentinel; each lane owns one 8x8 block (direction + filter).
// (2) replaces av1an's chunk concatenation (`-o`, av1an.rs:87) at tile
//     granularity: prefix sums over tile sizes, then every tile copies itself into the final
//     Section-5 OBU stream (temporal delimiter, sequence header, OBU_FRAME with tile sizes).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "av1mi_dev.h"

namespace {

// Cdef_Directions (spec §7.15.3) live in cdef_dir_offsets() as packed constants
__constant__ int c_div_table[9] = { 0, 840, 420, 280, 210, 168, 140, 120, 105 };

// CDEF reads the reconstruction straight from HBM/L2 through the vector L1 (every sample is touched ~5
// times by neighbouring lanes/rows of the same wave) and keeps only the per-8x8 decisions in LDS (256 B):
// the kernel can share a CU with the range-coding kernel, which holds 129 KB of LDS per CU.
struct CdefLds {
  uint8_t dir[64];       // luma direction of each 8x8 block
  uint8_t on[64];        // block is filtered
  uint16_t pri_y[64];    // variance-adjusted luma primary strength
};
__shared__ CdefLds g_cdef;

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int constrain(int diff, int threshold, int shift) {
  // shift = max(0, damping - floor(log2(threshold))), precomputed by the caller; threshold != 0
  const int mag = iabs(diff);
  int lim = threshold - (mag >> shift);
  lim = lim < 0 ? 0 : (lim > mag ? mag : lim);
  return diff < 0 ? -lim : lim;
}
__device__ __forceinline__ int damp_shift(int threshold, int damping) {
  if (!threshold) return 0;
  const int a = damping - (31 - __builtin_clz((unsigned)threshold));
  return a < 0 ? 0 : a;
}

// Cdef_Directions[d][k] = (dy, dx), packed 3 bits per direction (value + 2) so that the tap offsets cost
// a shift and a mask instead of a dependent table load per tap.
__device__ __forceinline__ void cdef_dir_offsets(int dir, int &dy0, int &dx0, int &dy1, int &dx1) {
  //            d: 7  6  5  4  3  2  1  0
  // k=0 dy+2:     3  3  3  3  2  2  2  1      dx+2: 2 2 2 3 3 3 3 3
  // k=1 dy+2:     4  4  4  4  3  2  1  0      dx+2: 1 2 3 4 4 4 4 4
  const int sh = 3 * dir;
  dy0 = ((0x6DB491 >> sh) & 7) - 2;   // 011 011 011 011 010 010 010 001
  dx0 = ((0x4936DB >> sh) & 7) - 2;   // 010 010 010 011 011 011 011 011
  dy1 = ((0x924688 >> sh) & 7) - 2;   // 100 100 100 100 011 010 001 000
  dx1 = ((0x29C924 >> sh) & 7) - 2;   // 001 010 011 100 100 100 100 100
}

// The constrained filter (§7.15.3) of BR consecutive rows of one column (this lane's) whose 8x8 (4x4 chroma) block is the same
// for all of them: the direction - so the tap offsets - and the strengths are per-lane constants of the batch.  ALL loads of the
// batch (centre + 4 primary taps, + 8 secondary taps when SEC) are issued before anything is computed: the row-at-a-time form
// (load centre, branch on the block's filter flag, load taps, filter, store) paid two memory latencies per row - 192 per
// superblock, ~600 k cycles per wave with 73 % of them parked on loads.  No branches: a block that is not filtered runs with both
// strengths 0 (the filter then returns the sample itself), a tap outside the frame (EDGE superblocks only) is replaced by the
// centre sample, which contributes nothing to the sum nor to the clamp range - exactly "not available" (§7.15.3).
// sample `idx` of a plane: the byte offset is formed in 32 bits (a frame is far below 4 GB), so the load takes its address as
// scalar base + 32-bit vector offset
template <typename PIX>
__device__ __forceinline__ int ld_px(const PIX *pl, int idx) {
  return *reinterpret_cast<const PIX *>(reinterpret_cast<const char *>(pl) + (unsigned)idx * (unsigned)sizeof(PIX));
}
template <typename PIX>
__device__ __forceinline__ void st_px(PIX *pl, int idx, int v) {
  *reinterpret_cast<PIX *>(reinterpret_cast<char *>(pl) + (unsigned)idx * (unsigned)sizeof(PIX)) = (PIX)v;
}
template <typename PIX, bool EDGE, bool SEC, int BR>
__device__ __forceinline__ void cdef_rows(const PIX *pl, PIX *out, int stride, int w, int h, int gx, int gy0, int pri, int sec, int pri_shift,
                                          int sec_shift, int dir, int coeff_shift, const PIX *srcp = nullptr, unsigned *acc = nullptr) {
  int oy[2], ox[2], sy[2][2], sx[2][2];
  cdef_dir_offsets(dir, oy[0], ox[0], oy[1], ox[1]);
  if (SEC) {
    cdef_dir_offsets((dir + 2) & 7, sy[0][0], sx[0][0], sy[0][1], sx[0][1]);
    cdef_dir_offsets((dir + 6) & 7, sy[1][0], sx[1][0], sy[1][1], sx[1][1]);
  }
  constexpr int NT = SEC ? 12 : 4;
  int ty[NT], tx[NT];   // tap offsets: primary k = 0, 1 (each with both signs), then the two secondary directions
#pragma unroll
  for (int k = 0; k < 2; k++) { ty[2 * k] = oy[k]; tx[2 * k] = ox[k]; ty[2 * k + 1] = -oy[k]; tx[2 * k + 1] = -ox[k]; }
  if (SEC) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        ty[4 + 4 * q + 2 * k] = sy[q][k]; tx[4 + 4 * q + 2 * k] = sx[q][k];
        ty[4 + 4 * q + 2 * k + 1] = -sy[q][k]; tx[4 + 4 * q + 2 * k + 1] = -sx[q][k];
      }
    }
  }
  // 32-bit unsigned sample indices from the (wave-uniform) plane pointer: the loads take the scalar-base + 32-bit-offset form
  // instead of a 64-bit address pair per tap
  int toff[NT];
#pragma unroll
  for (int k = 0; k < NT; k++) toff[k] = ty[k] * stride + tx[k];
  int c[BR], t[BR][NT];
  unsigned okm[BR];   // EDGE: bit k = tap k of the row is inside the frame
#pragma unroll
  for (int r = 0; r < BR; r++) {
    const int gy = EDGE ? (gy0 + r < h ? gy0 + r : h - 1) : gy0 + r;   // (rows below a partial superblock: loaded from the last row, never stored)
    const int base = gy * stride + gx;
    c[r] = ld_px(pl, base);
    okm[r] = 0;
#pragma unroll
    for (int k = 0; k < NT; k++) {
      if (EDGE) {
        int yy = gy + ty[k], xx = gx + tx[k];
        okm[r] |= (unsigned)(yy >= 0 && xx >= 0 && yy < h && xx < w) << k;
        yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
        xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
        t[r][k] = ld_px(pl, yy * stride + xx);
      } else {
        t[r][k] = ld_px(pl, base + toff[k]);
      }
    }
  }
  const int odd = (pri >> coeff_shift) & 1;
#pragma unroll
  for (int r = 0; r < BR; r++) {
    const int x = c[r];
    int sum = 0, mx = x, mn = x;
#pragma unroll
    for (int k = 0; k < NT; k++) {
      const int p = (EDGE && !((okm[r] >> k) & 1)) ? x : t[r][k];
      // taps k = 0, 1: primary, distance 1 (weight 4 or 3); 2, 3: primary, distance 2 (2 or 3); 4 .. 11: secondary (2, 2, 1, 1 per direction)
      const int wgt = k < 2 ? (odd ? 3 : 4) : (k < 4 ? (odd ? 3 : 2) : (((k - 4) & 3) < 2 ? 2 : 1));
      sum += wgt * (k < 4 ? constrain(p - x, pri, pri_shift) : constrain(p - x, sec, sec_shift));
      mx = p > mx ? p : mx;
      mn = p < mn ? p : mn;
    }
    int v = x + ((8 + sum - (sum < 0)) >> 4);
    v = v < mn ? mn : (v > mx ? mx : v);
    if (!EDGE || gy0 + r < h) {
      st_px(out, (gy0 + r) * stride + gx, v);
      // (chunk-wide launches: the squared error against the source is summed here - sse_kernel's second pass over the frames would
      // end after the range coder that CDEF runs beside)
      if (acc) { const int d = v - ld_px(srcp, (gy0 + r) * stride + gx); *acc += (unsigned)(d * d); }
    }
  }
}

template <typename PIX, bool EDGE, bool SEC, bool SSEV>
__device__ __forceinline__ void cdef_filter_sb(const Av1miDevParams &P, const PIX *fr, PIX *fo, int x0, int y0, int w, int h, int lane,
                                               int row0, int row1 /* luma rows [row0, row1) of the superblock, multiples of 8 */,
                                               const PIX *sf /* SSEV: the source frame */, unsigned long long *sse_f /* SSEV: the frame's three sums */) {
  const int coeff_shift = P.bit_depth - 8;
  // ---- luma: lane = column, batches of the 8 rows of a block row (row-contiguous HBM loads and stores)
  {
    const int sec = (P.cdef_y_sec == 3 ? 4 : P.cdef_y_sec) << coeff_shift;
    const int damping = P.cdef_damping + coeff_shift;
    const int sec_shift = damp_shift(sec, damping);
    unsigned acc = 0, *const accp = SSEV ? &acc : nullptr;
    if (lane < w) {
      for (int r = row0; r < (row1 < h ? row1 : h); r += 8) {
        const int b = (r >> 3) * 8 + (lane >> 3);
        const bool on = g_cdef.on[b] != 0;
        const int pri = on ? (int)g_cdef.pri_y[b] : 0, sec_b = on ? sec : 0;
        const int dir = P.cdef_y_pri == 0 ? 0 : g_cdef.dir[b];
        // batches: 8 rows with primary taps only in interior superblocks; 4 rows with the secondary taps or the frame-edge tests
        if (SEC && sec) {
          cdef_rows<PIX, EDGE, true, 4>(fr, fo, P.stride_y, P.width, P.height, x0 + lane, y0 + r, pri, sec_b, damp_shift(pri, damping), sec_shift, dir, coeff_shift, sf, accp);
          cdef_rows<PIX, EDGE, true, 4>(fr, fo, P.stride_y, P.width, P.height, x0 + lane, y0 + r + 4, pri, sec_b, damp_shift(pri, damping), sec_shift, dir, coeff_shift, sf, accp);
        } else if (EDGE) {
          cdef_rows<PIX, true, false, 4>(fr, fo, P.stride_y, P.width, P.height, x0 + lane, y0 + r, pri, 0, damp_shift(pri, damping), 0, dir, coeff_shift, sf, accp);
          cdef_rows<PIX, true, false, 4>(fr, fo, P.stride_y, P.width, P.height, x0 + lane, y0 + r + 4, pri, 0, damp_shift(pri, damping), 0, dir, coeff_shift, sf, accp);
        } else {
          cdef_rows<PIX, false, false, 8>(fr, fo, P.stride_y, P.width, P.height, x0 + lane, y0 + r, pri, 0, damp_shift(pri, damping), 0, dir, coeff_shift, sf, accp);
        }
      }
    }
    if constexpr (SSEV) {
      unsigned long long t = acc;
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      if (lane == 0 && t) atomicAdd(&sse_f[0], t);
    }
  }
  // ---- chroma: lanes 0-31 = columns of U, lanes 32-63 = columns of V; batches of the 4 rows of a block row
  {
    const int pl = lane >> 5, col = lane & 31;
    const PIX *cp = fr + (pl ? P.plane_off_v : P.plane_off_u);
    PIX *op = fo + (pl ? P.plane_off_v : P.plane_off_u);
    const int pri0 = P.cdef_uv_pri << coeff_shift;
    const int sec0 = (P.cdef_uv_sec == 3 ? 4 : P.cdef_uv_sec) << coeff_shift;
    const int damping = P.cdef_damping + coeff_shift - 1;
    const int pri_shift = damp_shift(pri0, damping), sec_shift = damp_shift(sec0, damping);
    const int wc = w >> 1, hc = h >> 1;
    const PIX *sp = SSEV ? sf + (pl ? P.plane_off_v : P.plane_off_u) : nullptr;
    unsigned acc = 0, *const accp = SSEV ? &acc : nullptr;
    if (col < wc) {
      for (int r = row0 >> 1; r < ((row1 >> 1) < hc ? (row1 >> 1) : hc); r += 4) {
        const int b = (r >> 2) * 8 + (col >> 2);
        const bool on = g_cdef.on[b] != 0;
        const int pri = on ? pri0 : 0, sec = on ? sec0 : 0;
        const int dir = pri0 == 0 ? 0 : (int)g_cdef.dir[b];
        if (SEC && sec0) cdef_rows<PIX, EDGE, true, 4>(cp, op, P.stride_c, P.width >> 1, P.height >> 1, (x0 >> 1) + col, (y0 >> 1) + r, pri, sec, pri_shift, sec_shift, dir, coeff_shift, sp, accp);
        else cdef_rows<PIX, EDGE, false, 4>(cp, op, P.stride_c, P.width >> 1, P.height >> 1, (x0 >> 1) + col, (y0 >> 1) + r, pri, 0, pri_shift, 0, dir, coeff_shift, sp, accp);
      }
    }
    if constexpr (SSEV) {   // the two halves of the wave: U, V
      unsigned long long t = acc;
      for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      if ((lane & 31) == 0 && t) atomicAdd(&sse_f[1 + pl], t);
    }
  }
}

// Direction search §7.15.2 of one 8x8 block held in registers (px[i][j], already >> coeff_shift and - 128): direction and the
// variance that scales the luma primary strength.  The eight directions' partial sums would need 120 registers at once; they
// are built in two groups of four, which keeps the callers at 6 waves/SIMD without spilling.
__device__ __forceinline__ void cdef_direction_8x8(const int (&px)[8][8], int &ydir, int &var) {
  int cost[8];
#pragma unroll
  for (int a = 0; a < 8; a++) cost[a] = 0;
#pragma unroll
  for (int grp = 0; grp < 2; grp++) {
    int partial[4][15];
#pragma unroll
    for (int a = 0; a < 4; a++) {
#pragma unroll
      for (int b = 0; b < 15; b++) partial[a][b] = 0;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int x = px[i][j];
        if (grp == 0) {
          partial[0][i + j] += x;            // direction 0
          partial[1][i + j / 2] += x;        // 1
          partial[2][i] += x;                // 2
          partial[3][3 + i - j / 2] += x;    // 3
        } else {
          partial[0][7 + i - j] += x;        // 4
          partial[1][3 - i / 2 + j] += x;    // 5
          partial[2][j] += x;                // 6
          partial[3][i / 2 + j] += x;        // 7
        }
      }
    }
    const int d0 = grp * 4;
    {  // directions 2 and 6: 8 sums
      int c = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) c += partial[2][i] * partial[2][i];
      cost[d0 + 2] = c * 105;
    }
    {  // directions 0 and 4: 15 diagonals
      int c = 0;
#pragma unroll
      for (int i = 0; i < 7; i++) c += (partial[0][i] * partial[0][i] + partial[0][14 - i] * partial[0][14 - i]) * c_div_table[i + 1];
      cost[d0] = c + partial[0][7] * partial[0][7] * 105;
    }
#pragma unroll
    for (int q = 1; q < 4; q += 2) {  // odd directions: 11 sums
      int c = 0;
#pragma unroll
      for (int j = 0; j < 5; j++) c += partial[q][3 + j] * partial[q][3 + j];
      c *= 105;
#pragma unroll
      for (int j = 0; j < 3; j++) c += (partial[q][j] * partial[q][j] + partial[q][10 - j] * partial[q][10 - j]) * c_div_table[2 * j + 2];
      cost[d0 + q] = c;
    }
  }
  int best = 0;
  ydir = 0;
#pragma unroll
  for (int d = 0; d < 8; d++)
    if (cost[d] > best) { best = cost[d]; ydir = d; }
  int opp = 0;
#pragma unroll
  for (int d = 0; d < 8; d++)
    if (d == ((ydir + 4) & 7)) opp = cost[d];
  var = (best - opp) >> 10;
}

// variance-adjusted luma primary strength (§7.15.1)
__device__ __forceinline__ int cdef_adjusted_pri(int pri_y, int var, int coeff_shift) {
  int pri = pri_y << coeff_shift;
  const int v6 = var >> 6;
  const int var_str = v6 ? ((31 - __builtin_clz((unsigned)v6)) < 12 ? (31 - __builtin_clz((unsigned)v6)) : 12) : 0;
  return var ? (pri * (4 + var_str) + 8) >> 4 : 0;
}

// Chunk-wide launches: the direction search in a kernel of its own.  A wave per superblock stages the 64x64 luma samples in LDS
// with row-contiguous 16-byte loads (lane = 8 samples of a row: 1 KB per instruction) and every lane then reads its own 8x8
// block from there - as one kernel with the filter each lane loaded its block straight from HBM, 64 lanes touching 64 different
// 16-byte segments per instruction (8192 cache-line requests per superblock against 640 for the whole filter pass), and the
// filter kernel cannot hold an LDS tile beside the range coder (145 KB of LDS per CU).  This kernel runs beside symbolize; the
// filter kernel (beside the range coder) reads {adjusted primary strength << 3 | direction} per 8x8 block.
template <typename PIX>
__global__ void __launch_bounds__(64) cdef_dir_kernel(Av1miDevParams P, const PIX *__restrict__ rec, const Av1miBlkInfo *__restrict__ blk,
                                                     uint16_t *__restrict__ dirtab /* [frame][superblock][64] */) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[64][72];   // row pitch 144 B: 16-byte aligned, block rows land on different bank halves
  const int sbs_per_frame = P.sb_rows * P.sb_cols;
  const int f = blockIdx.x / sbs_per_frame, sb = blockIdx.x % sbs_per_frame;
  const int sbr = sb / P.sb_cols, sbc = sb % P.sb_cols, lane = threadIdx.x;
  const PIX *fr = rec + (size_t)f * P.frame_samples;
  const int x0 = sbc * 64, y0 = sbr * 64, coeff_shift = P.bit_depth - 8;
  const int b8r = lane >> 3, b8c = lane & 7;
  const bool inside = (sbr * 8 + b8r) < P.b8_rows && (sbc * 8 + b8c) < P.b8_cols;
  int skip = 1;
  if (inside) skip = blk[(size_t)f * P.b8_rows * P.b8_cols + (size_t)(sbr * 8 + b8r) * P.b8_cols + sbc * 8 + b8c].skip;
  const bool sb_on = __ballot(inside && !skip) != 0ull;
  if (!P.enable_cdef || !sb_on) return;   // nothing of this superblock is filtered: the filter kernel does not read its entries
  {
    const int c8 = (lane & 7) * 8;
    const bool col_in = x0 + c8 < P.width;   // (width is a multiple of 8: a group of 8 samples is inside or outside as a whole)
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int row = k * 8 + (lane >> 3);
      if (col_in && y0 + row < P.height) {
        const PIX *p = fr + (size_t)(y0 + row) * P.stride_y + x0 + c8;
        uint4 w;
        if (sizeof(PIX) == 2) {
          w = *reinterpret_cast<const uint4 *>(p);
        } else {
          const uint2 q = *reinterpret_cast<const uint2 *>(p);
          w.x = (q.x & 0xFF) | ((q.x & 0xFF00) << 8); w.y = ((q.x >> 16) & 0xFF) | ((q.x >> 24) << 16);
          w.z = (q.y & 0xFF) | ((q.y & 0xFF00) << 8); w.w = ((q.y >> 16) & 0xFF) | ((q.y >> 24) << 16);
        }
        *reinterpret_cast<uint4 *>(&tile[row][c8]) = w;
      }
    }
  }
  __syncthreads();
  int ydir = 0, var = 0;
  if (inside && !skip) {
    int px[8][8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint4 w = *reinterpret_cast<const uint4 *>(&tile[b8r * 8 + i][b8c * 8]);
      const uint32_t d[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
      for (int j = 0; j < 4; j++) {
        px[i][2 * j] = (int)((d[j] & 0xFFFF) >> coeff_shift) - 128;
        px[i][2 * j + 1] = (int)((d[j] >> 16) >> coeff_shift) - 128;
      }
    }
    cdef_direction_8x8(px, ydir, var);
  }
  dirtab[(size_t)blockIdx.x * 64 + lane] = (uint16_t)((cdef_adjusted_pri(P.cdef_y_pri, var, coeff_shift) << 3) | ydir);
}

// NS = 1: one wave per superblock (chunk-wide launches: throughput).  NS = 4 / 8: one wave per 16- / 8-row strip of a superblock
// (one-frame launches of inter chunks: NS x the waves and 1 / NS of the filter work per wave: latency).
// TAB: directions and adjusted strengths come from cdef_dir_kernel's table instead of being searched here.
// SEC: some secondary strength is non-zero (the default strengths have none: the instantiation without carries no secondary-tap code).
// SSEV: the squared error of the output against the source `src` is added to sse[frame][plane] (chunk-wide launches only).
template <typename PIX, int NS, bool TAB, bool SEC, bool SSEV = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) cdef_sb_kernel(Av1miDevParams P, const PIX *__restrict__ rec, PIX *__restrict__ fin,
                                                    const Av1miBlkInfo *__restrict__ blk, const uint16_t *__restrict__ dirtab,
                                                    const PIX *__restrict__ src = nullptr, unsigned long long *__restrict__ sse = nullptr) {
  const int sbs_per_frame = P.sb_rows * P.sb_cols;
  const int strip = NS == 1 ? 0 : (int)(blockIdx.x % NS);
  const int item = blockIdx.x / NS;
  const int f = item / sbs_per_frame, sb = item % sbs_per_frame;
  const int sbr = sb / P.sb_cols, sbc = sb % P.sb_cols;
  const int lane = threadIdx.x;
  const PIX *fr = rec + (size_t)f * P.frame_samples;
  PIX *fo = fin + (size_t)f * P.frame_samples;
  const int x0 = sbc * 64, y0 = sbr * 64;
  const int coeff_shift = P.bit_depth - 8;
  const int w = P.width - x0 < 64 ? P.width - x0 : 64, h = P.height - y0 < 64 ? P.height - y0 : 64;
  // ---- per-8x8 decisions: lane = 8x8 block
  const int b8r = lane >> 3, b8c = lane & 7;
  const bool inside = (sbr * 8 + b8r) < P.b8_rows && (sbc * 8 + b8c) < P.b8_cols;
  int skip = 1;
  if (inside) skip = blk[(size_t)f * P.b8_rows * P.b8_cols + (size_t)(sbr * 8 + b8r) * P.b8_cols + sbc * 8 + b8c].skip;
  // cdef_idx of the superblock is coded (== 0) iff some block in it is not skipped (§5.11.56)
  const bool sb_on = __ballot(inside && !skip) != 0ull;
  const bool do_filter = P.enable_cdef && sb_on && inside && !skip;
  const bool mine = NS == 1 || b8r / (8 / NS) == strip;   // this wave decides (and filters) only the blocks of its strip
  {
    int ydir = 0, pri = 0;
    if constexpr (TAB) {
      if (do_filter && mine) { const int e = dirtab[(size_t)item * 64 + lane]; ydir = e & 7; pri = e >> 3; }
    } else {
      int var = 0;
      if (do_filter && mine) {
        // direction search §7.15.2 on the block's 8 rows of 8 samples (one-frame launches of inter chunks: the lane reads its block
        // from L1/L2; the second group's reads hit L1)
        const PIX *ty = fr + (size_t)(y0 + b8r * 8) * P.stride_y + x0 + b8c * 8;
        int px[8][8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
          for (int j = 0; j < 8; j++) px[i][j] = ((int)ty[(size_t)i * P.stride_y + j] >> coeff_shift) - 128;
        }
        cdef_direction_8x8(px, ydir, var);
      }
      pri = cdef_adjusted_pri(P.cdef_y_pri, var, coeff_shift);
    }
    g_cdef.dir[lane] = (uint8_t)ydir;
    g_cdef.on[lane] = (uint8_t)do_filter;
    g_cdef.pri_y[lane] = (uint16_t)pri;
  }
  __syncthreads();
  // taps reach 2 samples beyond the superblock (1 in chroma): interior superblocks need no tests
  const bool edge = x0 < 2 || y0 < 2 || x0 + 66 > P.width || y0 + 66 > P.height;
  const int row0 = NS == 1 ? 0 : strip * (64 / NS), row1 = NS == 1 ? 64 : (strip + 1) * (64 / NS);
  const PIX *sf = SSEV ? src + (size_t)f * P.frame_samples : nullptr;
  unsigned long long *sse_f = SSEV ? sse + (size_t)f * 3 : nullptr;
  if (edge) cdef_filter_sb<PIX, true, SEC, SSEV>(P, fr, fo, x0, y0, w, h, lane, row0, row1, sf, sse_f);
  else cdef_filter_sb<PIX, false, SEC, SSEV>(P, fr, fo, x0, y0, w, h, lane, row0, row1, sf, sse_f);
}

// ------------------------------------------------------------------------------ SSE (PSNR)
// 16 bytes per lane per load (8 or 16 samples); plane sizes are multiples of 16 samples (width, height multiples of 8),
// so a vector never straddles two planes.  VEC = false: sample-wise fallback for frame pointers that are not 16-byte aligned.
template <typename PIX, bool VEC>
__global__ void __launch_bounds__(256) sse_kernel(Av1miDevParams P, const PIX *__restrict__ a, const PIX *__restrict__ b,
                                                 unsigned long long *__restrict__ sse /* [n_frames][3] */) {
  const int f = blockIdx.y;
  const PIX *pa = a + (size_t)f * P.frame_samples, *pb = b + (size_t)f * P.frame_samples;
  const long ny = (long)P.width * P.height, nc = ny >> 2;
  unsigned long long acc[3] = { 0, 0, 0 };
  if (VEC) {
    constexpr int SPV = 16 / (int)sizeof(PIX);  // samples per vector
    const uint4 *va = reinterpret_cast<const uint4 *>(pa), *vb = reinterpret_cast<const uint4 *>(pb);
    const long nvec = P.frame_samples / SPV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
      const uint4 x = va[i], y = vb[i];
      const uint32_t xs[4] = { x.x, x.y, x.z, x.w }, ys[4] = { y.x, y.y, y.z, y.w };
      unsigned s = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (sizeof(PIX) == 2) {
          const int d0 = (int)(xs[k] & 0xFFFF) - (int)(ys[k] & 0xFFFF), d1 = (int)(xs[k] >> 16) - (int)(ys[k] >> 16);
          s += (unsigned)(d0 * d0) + (unsigned)(d1 * d1);
        } else {
#pragma unroll
          for (int q = 0; q < 4; q++) { const int d = (int)((xs[k] >> (8 * q)) & 0xFF) - (int)((ys[k] >> (8 * q)) & 0xFF); s += (unsigned)(d * d); }
        }
      }
      const long i0 = i * SPV;
      const int pl = i0 < ny ? 0 : (i0 < ny + nc ? 1 : 2);
      acc[pl] += s;
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < P.frame_samples; i += (long)gridDim.x * 256) {
      const int d = (int)pa[i] - (int)pb[i];
      const int pl = i < ny ? 0 : (i < ny + nc ? 1 : 2);
      acc[pl] += (unsigned long long)(d * d);
    }
  }
  for (int pl = 0; pl < 3; pl++) {
    unsigned long long v = acc[pl];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sse[f * 3 + pl], v);
  }
}

// ------------------------------------------------------------------------------ packing
__device__ __forceinline__ int leb128_len(uint32_t v) {
  int n = 1;
  while (v >>= 7) n++;
  return n;
}

// one workgroup per frame: tile offsets inside the OBU_FRAME payload, frame (temporal unit) size
__global__ void __launch_bounds__(256) frame_layout_kernel(Av1miDevParams P, const uint32_t *__restrict__ tile_bytes,
                                                          uint32_t *__restrict__ tile_off, uint32_t *__restrict__ frame_size,
                                                          uint32_t *__restrict__ payload_size, int *__restrict__ overflow) {
  __shared__ uint32_t part[256];
  const int f = blockIdx.x, nt = P.tile_rows * P.tile_cols, t = threadIdx.x;
  const uint32_t *tb = tile_bytes + (size_t)f * nt;
  const int per = (nt + 255) / 256;
  uint32_t s = 0;
  for (int i = t * per; i < (t + 1) * per && i < nt; i++) {
    // a tile that outgrew its slot or its symbol stream (0xFFFFFFFF): flag it; the host re-runs the
    // chunk with larger capacities and pack_tiles_kernel does nothing in this pass
    if (tb[i] > (uint32_t)P.tile_slot_bytes) atomicExch(overflow, 1);
    s += tb[i] + (i < nt - 1 ? (uint32_t)P.tile_size_bytes : 0u);
  }
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    // temporal unit = TD + [sequence header, key frames only] + OBU_FRAME header + size + payload
    const int inter = av1mi_frame_is_inter(P, f);
    uint32_t run = (uint32_t)(inter ? P.inter_hdr_bytes : P.frame_hdr_bytes);
    for (int i = 0; i < 256; i++) { uint32_t v = part[i]; part[i] = run; run += v; }
    payload_size[f] = run;
    frame_size[f] = 2u + (inter ? 0u : (uint32_t)P.seq_hdr_bytes) + 1u + (uint32_t)leb128_len(run) + run;
  }
  __syncthreads();
  uint32_t run = part[t];
  for (int i = t * per; i < (t + 1) * per && i < nt; i++) {
    tile_off[(size_t)f * nt + i] = run;
    run += tb[i] + (i < nt - 1 ? (uint32_t)P.tile_size_bytes : 0u);
  }
}

__global__ void chunk_layout_kernel(int n_frames, const uint32_t *__restrict__ frame_size, unsigned long long *__restrict__ frame_off) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    unsigned long long run = 0;
    for (int f = 0; f < n_frames; f++) { frame_off[f] = run; run += frame_size[f]; }
    frame_off[n_frames] = run;
  }
}

// one wave per tile: copy the tile (and, for tile 0, the frame's headers) to its final position
__global__ void __launch_bounds__(64) pack_tiles_kernel(Av1miDevParams P, const uint8_t *__restrict__ slots,
                                                       const uint32_t *__restrict__ tile_bytes, const uint32_t *__restrict__ tile_off,
                                                       const uint32_t *__restrict__ payload_size,
                                                       const unsigned long long *__restrict__ frame_off,
                                                       const uint8_t *__restrict__ hdr_blob, uint8_t *__restrict__ out,
                                                       const int *__restrict__ overflow) {
  if (*overflow) return;  // sizes and offsets are meaningless: nothing may be written
  const int nt = P.tile_rows * P.tile_cols;
  const int f = blockIdx.x / nt, t = blockIdx.x % nt, lane = threadIdx.x;
  const uint32_t pay = payload_size[f];
  const int ll = leb128_len(pay);
  uint8_t *fo = out + frame_off[f];
  const int inter = av1mi_frame_is_inter(P, f);
  const int seq_bytes = inter ? 0 : P.seq_hdr_bytes, hdr_bytes = inter ? P.inter_hdr_bytes : P.frame_hdr_bytes;
  const int prefix = 2 + seq_bytes + 1 + ll;  // TD + [sequence header] + OBU_FRAME header + size
  if (t == 0) {
    if (lane == 0) {
      fo[0] = 0x12; fo[1] = 0x00;
      uint8_t *q = fo + 2 + seq_bytes;
      q[0] = 0x32;
      uint32_t v = pay;
      for (int i = 0; i < ll; i++) { uint8_t b = v & 0x7F; v >>= 7; if (v) b |= 0x80; q[1 + i] = b; }
    }
    for (int i = lane; i < seq_bytes; i += 64) fo[2 + i] = hdr_blob[i];
    for (int i = lane; i < hdr_bytes; i += 64) fo[prefix + i] = hdr_blob[P.seq_hdr_bytes + (size_t)f * P.hdr_slot_bytes + i];
  }
  const uint32_t n = tile_bytes[blockIdx.x];
  uint8_t *dst = fo + prefix + tile_off[blockIdx.x];
  if (t < nt - 1) {
    if (lane < P.tile_size_bytes) dst[lane] = (uint8_t)((n - 1) >> (8 * lane));
    dst += P.tile_size_bytes;
  }
  // The range-coding kernel left 16-bit pre-carry entries d[i] = byte | carry << 8 (carry: +1 to the number formed by the
  // bytes before i).  Final byte i = (lo(d[i]) + hi(d[i+1]) + carry-in from the right) & 0xFF: a long addition whose
  // carries are resolved 64 positions at a time, from the end of the tile, with generate/propagate masks and one 64-bit
  // add per chunk (carry-lookahead: carries = (X + Y + cin) ^ P with X = G | P, Y = G).
  const uint16_t *pre = reinterpret_cast<const uint16_t *>(slots) + (size_t)blockIdx.x * P.tile_slot_bytes;
  unsigned long long carry = 0;
  for (long base = n ? (long)((n - 1) & ~63u) : -1; base >= 0; base -= 64) {
    const uint32_t i = (uint32_t)base + (uint32_t)lane;
    const bool valid = i < n;
    const unsigned d = valid ? pre[i] : 0u, dn = (i + 1 < n) ? pre[i + 1] : 0u;
    const unsigned sv = (d & 0xFFu) + (dn >> 8);                 // 0 .. 256
    // bit (63 - lane): carries run from larger i (low bits) to smaller i (high bits)
    const unsigned long long G = __brevll(__ballot(valid && sv == 256u)), Pm = __brevll(__ballot(valid && sv == 255u));
    const unsigned long long X = G | Pm;
    const unsigned long long S1 = X + G, S = S1 + carry;
    const unsigned long long cout = (S1 < X) | (S < S1);         // carry out of bit 63 = into the previous chunk
    const unsigned long long C = S ^ Pm;                         // S ^ X ^ Y: carry into every position
    const unsigned cin = (unsigned)((C >> (63 - lane)) & 1ull);
    if (valid) dst[i] = (uint8_t)(sv + cin);
    carry = cout;
  }
}

}  // namespace

// dirtab == nullptr: one kernel (direction search + filter); else the filter reads cdef_dir_kernel's table (av1mi_launch_cdef_dir before)
extern "C" hipError_t av1mi_launch_cdef(const Av1miDevParams *P, const void *rec, void *fin, const Av1miBlkInfo *blk, const uint16_t *dirtab,
                                        const void *src, unsigned long long *sse /* both or neither: the chunk-wide launch also sums the squared error */,
                                        hipStream_t stream) {
  const int grid = P->n_frames * P->sb_rows * P->sb_cols;
  const bool strips = P->n_frames == 1;  // a one-frame launch sits on an inter chunk's serial chain
  static const bool exp_strips = getenv("AV1MI_CDEF_STRIPS") != nullptr;   // experiment: 16-row strips for chunk-wide launches too
  const bool sec = P->cdef_y_sec != 0 || P->cdef_uv_sec != 0;
  const bool sse_here = src && sse && !strips && !exp_strips;
  // one-frame launches: 8-row strips (4 080 waves at 1080p, still one round of the chip) - 16-row strips were 2.5 us per frame slower
  // (AV1MI_CDEF_NS4 brings them back: same-box A/B of the chain)
  static const bool ns8 = getenv("AV1MI_CDEF_NS4") == nullptr;
#define CDEF_LAUNCH2(PIXT, SECV)                                                                                                           \
  do {                                                                                                                                     \
    if (exp_strips && dirtab) hipLaunchKernelGGL((cdef_sb_kernel<PIXT, 4, true, SECV>), dim3(grid * 4), dim3(64), 0, stream, *P, (const PIXT *)rec, (PIXT *)fin, blk, dirtab); \
    else if (strips && ns8) hipLaunchKernelGGL((cdef_sb_kernel<PIXT, 8, false, SECV>), dim3(grid * 8), dim3(64), 0, stream, *P, (const PIXT *)rec, (PIXT *)fin, blk, dirtab); \
    else if (strips || exp_strips) hipLaunchKernelGGL((cdef_sb_kernel<PIXT, 4, false, SECV>), dim3(grid * 4), dim3(64), 0, stream, *P, (const PIXT *)rec, (PIXT *)fin, blk, dirtab); \
    else if (dirtab && sse_here) hipLaunchKernelGGL((cdef_sb_kernel<PIXT, 1, true, SECV, true>), dim3(grid), dim3(64), 0, stream, *P, (const PIXT *)rec, (PIXT *)fin, blk, dirtab, (const PIXT *)src, sse); \
    else if (dirtab) hipLaunchKernelGGL((cdef_sb_kernel<PIXT, 1, true, SECV>), dim3(grid), dim3(64), 0, stream, *P, (const PIXT *)rec, (PIXT *)fin, blk, dirtab); \
    else if (sse_here) hipLaunchKernelGGL((cdef_sb_kernel<PIXT, 1, false, SECV, true>), dim3(grid), dim3(64), 0, stream, *P, (const PIXT *)rec, (PIXT *)fin, blk, dirtab, (const PIXT *)src, sse); \
    else hipLaunchKernelGGL((cdef_sb_kernel<PIXT, 1, false, SECV>), dim3(grid), dim3(64), 0, stream, *P, (const PIXT *)rec, (PIXT *)fin, blk, dirtab); \
  } while (0)
#define CDEF_LAUNCH(PIXT) do { if (sec) CDEF_LAUNCH2(PIXT, true); else CDEF_LAUNCH2(PIXT, false); } while (0)
  if (P->bit_depth == 8) CDEF_LAUNCH(uint8_t); else CDEF_LAUNCH(uint16_t);
#undef CDEF_LAUNCH2
#undef CDEF_LAUNCH
  return hipGetLastError();
}

// the direction search of a chunk-wide CDEF as a kernel of its own (see cdef_dir_kernel); dirtab: n_frames x superblocks x 64 entries
extern "C" hipError_t av1mi_launch_cdef_dir(const Av1miDevParams *P, const void *rec, const Av1miBlkInfo *blk, uint16_t *dirtab, hipStream_t stream) {
  const int grid = P->n_frames * P->sb_rows * P->sb_cols;
  if (P->bit_depth == 8) hipLaunchKernelGGL((cdef_dir_kernel<uint8_t>), dim3(grid), dim3(64), 0, stream, *P, (const uint8_t *)rec, blk, dirtab);
  else hipLaunchKernelGGL((cdef_dir_kernel<uint16_t>), dim3(grid), dim3(64), 0, stream, *P, (const uint16_t *)rec, blk, dirtab);
  return hipGetLastError();
}

extern "C" hipError_t av1mi_launch_sse(const Av1miDevParams *P, const void *a, const void *b, unsigned long long *sse, hipStream_t stream) {
  dim3 grid(64, P->n_frames);
  const bool vec = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;  // frame size in bytes is a multiple of 16 (1.5 * w * h, w and h multiples of 8)
  if (P->bit_depth == 8) {
    if (vec) hipLaunchKernelGGL((sse_kernel<uint8_t, true>), grid, dim3(256), 0, stream, *P, (const uint8_t *)a, (const uint8_t *)b, sse);
    else hipLaunchKernelGGL((sse_kernel<uint8_t, false>), grid, dim3(256), 0, stream, *P, (const uint8_t *)a, (const uint8_t *)b, sse);
  } else {
    if (vec) hipLaunchKernelGGL((sse_kernel<uint16_t, true>), grid, dim3(256), 0, stream, *P, (const uint16_t *)a, (const uint16_t *)b, sse);
    else hipLaunchKernelGGL((sse_kernel<uint16_t, false>), grid, dim3(256), 0, stream, *P, (const uint16_t *)a, (const uint16_t *)b, sse);
  }
  return hipGetLastError();
}

extern "C" hipError_t av1mi_launch_pack(const Av1miDevParams *P, const uint8_t *slots, const uint32_t *tile_bytes, uint32_t *tile_off,
                                        uint32_t *frame_size, uint32_t *payload_size, unsigned long long *frame_off,
                                        const uint8_t *hdr_blob, uint8_t *out, int *overflow, int stage, hipStream_t stream) {
  const int nt = P->tile_rows * P->tile_cols;
  if (stage == 0) {
    hipLaunchKernelGGL(frame_layout_kernel, dim3(P->n_frames), dim3(256), 0, stream, *P, tile_bytes, tile_off, frame_size, payload_size, overflow);
    hipLaunchKernelGGL(chunk_layout_kernel, dim3(1), dim3(64), 0, stream, P->n_frames, frame_size, frame_off);
  } else {
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(P->n_frames * nt), dim3(64), 0, stream, *P, slots, tile_bytes, tile_off, payload_size, frame_off, hdr_blob, out, overflow);
  }
  return hipGetLastError();
}
