// lr_kernel.hip - loop restoration: per-unit Wiener decision and filter on luma.
//
// Replaces the restoration-filter search and application inside the SVT-AV1 worker behind `run_av1an`
// (/root/reference/crates/daemon/src/encode/av1an.rs:126-139).  Normative part: AV1 spec §7.17.3 (units offset by 8 luma
// rows), §7.17.4 (separable 7-tap Wiener filter, Round2 by 3 then 11, intermediate clamp) and §7.17.6 (rows outside the
// 64-row stripe come from the pre-CDEF frame, at most 2 rows away).  Encoder part: a unit takes the
// candidate filter with the smallest SSE against the source, or none; restated in oracle/av1o_lr.c.
//
// MI355X mapping: waves of 16 rows of a 64x64 unit (lane = column), two phases (see lr_unit_kernel).  Per stripe of the unit the horizontal pass of 70 rows goes to
// LDS as int16 (the spec's clamp keeps it in 16 bits for 8/10 bit), the vertical pass reads 7 LDS rows per sample.  The
// three candidates are evaluated for their SSE only (with enable_lr = 2 also three self-guided candidates: the A/B grids of
// both box-filter passes go to LDS per stripe section, see sgr_grid); the winner is applied in a last pass that writes the final
// reconstruction (chroma is copied: FrameRestorationType = NONE).  Algorithmic HBM bytes: CDEF frame read + pre-CDEF rows
// at stripe edges + source read + final write = ~3*L*b + N*b per frame.
#include <hip/hip_runtime.h>
#include "av1mi_dev.h"

namespace {

__constant__ int8_t c_wiener_cand[3][3] = { { 0, 0, -4 }, { 1, -3, -6 }, { 3, -7, 15 } };

// a wave works on at most 16 rows (LR_SLICES): the source window of those rows (3 more above, 2..3 below, 3 columns either
// side; get_source_sample's stripe rule applied per row) is staged once and both filters read it from LDS
__shared__ uint16_t g_win[22][72];
__shared__ int16_t g_mid[22][72];   // Wiener: horizontal-pass output [row][lane]

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// horizontal pass of rows [ya - 3, yb + 3) for column xs + lane, from the staged window, into g_mid[row - (ya - 3)][lane]
__device__ __forceinline__ void wiener_h(int bd, int ya, int yb, const int *f, int lane) {
  const int offset = 1 << (bd + 7 - 3 - 1), limit = (1 << (bd + 1 + 7 - 3)) - 1;
  for (int r = 0; r < yb - ya + 6; r++) {
    int s = 0;
#pragma unroll
    for (int t = 0; t < 7; t++) s += f[t] * (int)g_win[r][lane + t];
    g_mid[r][lane] = (int16_t)clampi((s + 4) >> 3, -offset, limit - offset);
  }
}
__device__ __forceinline__ int wiener_v(int row_in_mid, int lane, const int *f, int maxv) {
  int s = 0;
#pragma unroll
  for (int t = 0; t < 7; t++) s += f[t] * (int)g_mid[row_in_mid + t][lane];
  return clampi((s + 1024) >> 11, 0, maxv);
}
__device__ __forceinline__ void taps_of(int k, int *f) {
  const int c0 = c_wiener_cand[k][0], c1 = c_wiener_cand[k][1], c2 = c_wiener_cand[k][2];
  f[0] = f[6] = c0; f[1] = f[5] = c1; f[2] = f[4] = c2; f[3] = 128 - 2 * (c0 + c1 + c2);
}
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- self-guided restoration (enable_lr = 2; §7.17.3 self guided filter process / box filter process) ---------------
// Candidates (oracle/av1o_lr.c av1o_sgr_candidates): parameter set 9 (pass 0: r = 2, eps 68; pass 1: r = 1, eps 15) with the
// weights (xqd0, xqd1) = (31, 31), (0, 31), (31, 95): both box-filter passes are computed once per sample, the candidates
// differ only in the final blend.
__constant__ int8_t c_sgr_cand[3][3] = { { 9, 31, 31 }, { 9, 0, 31 }, { 9, 31, 95 } };
// A (<= 256) and B of the box filter at the positions a slice of SGR_ROWS rows of a unit section needs: rows ya - 1 .. yb
// (<= SGR_ROWS + 2), columns xs - 1 .. xs + 64.  Slices keep the LDS footprint small (whole 64-row sections
// needed 62 KB: 2 waves per CU, half the SIMDs idle, 3x slower).
#define SGR_ROWS 16   /* == the rows of a wave's slice */
// Pass 0 (r = 2) is only evaluated at odd rows: its grids hold every second row (row index >> 1; a slice starts at an even row, so
// the odd rows ya - 1, ya + 1 .. have even indices).  That is 3.5 KB less: 17 KB per wave, 9 waves per CU instead of 7, and the
// 2040 working waves of a 1080p frame are resident at once (the same step the inter pass took).
__shared__ uint16_t g_sgrA0[(SGR_ROWS + 2) / 2][66], g_sgrA1[SGR_ROWS + 2][66];
__shared__ int32_t g_sgrB0[(SGR_ROWS + 2) / 2][66], g_sgrB1[SGR_ROWS + 2][66];

// get_source_sample (§7.17.6): the row of the frame that supplies restoration input row y of the stripe [s0, s1]
template <typename PIX>
__device__ __forceinline__ const PIX *lr_row(const Av1miDevParams &P, const PIX *cdef, const PIX *pre, int y, int s0, int s1) {
  int yy = clampi(y, 0, P.true_h - 1);
  const PIX *fr = cdef;
  if (yy < s0) { yy = yy > s0 - 2 ? yy : s0 - 2; fr = pre; }
  else if (yy > s1) { yy = yy < s1 + 2 ? yy : s1 + 2; fr = pre; }
  return fr + (size_t)yy * P.stride_y;
}

// A and B from the box sums (sum of samples b, of squares a) of a (2r+1)^2 window
template <int R>
__device__ __forceinline__ void sgr_ab(uint32_t a, uint32_t b, int bd, uint32_t &A, int32_t &B) {
  constexpr uint32_t n = (2 * R + 1) * (2 * R + 1), eps = R == 2 ? 68 : 15, n2e = n * n * eps;
  constexpr uint32_t s = ((1u << 20) + n2e / 2) / n2e, one_by_n = ((1u << 12) + n / 2) / n;
  const uint32_t a8 = (a + ((1u << (2 * (bd - 8))) >> 1)) >> (2 * (bd - 8));
  const uint32_t d = (b + ((1u << (bd - 8)) >> 1)) >> (bd - 8);
  const uint32_t p = a8 * n > d * d ? a8 * n - d * d : 0;
  const uint32_t z = (uint32_t)(((unsigned long long)p * s + (1u << 19)) >> 20);
  const uint32_t a2 = z >= 255 ? 256 : (z == 0 ? 1 : ((z << 8) + z / 2) / (z + 1));
  A = a2;
  B = (int32_t)(((unsigned long long)(256 - a2) * b * one_by_n + (1u << 11)) >> 12);
}

// Source window of a unit section for the box sums: restoration input rows ya - 3 .. yb + 2 (get_source_sample's stripe rule
// per row), columns xs - 3 .. xs + 66 (clamped to the frame) -> win[row - (ya - 3)][col - (xs - 3)], all loads independent.
template <typename PIX>
__device__ __forceinline__ void lr_stage(const Av1miDevParams &P, const PIX *cdef, const PIX *pre, int xs, int ya, int yb, int s0, int s1, int lane) {
  uint16_t (*win)[72] = g_win;
  const int rows = yb - ya + 6, W = P.true_w;
#pragma unroll 8
  for (int p = lane; p < rows * 70; p += 64) {
    const int i = p / 70, j = p - i * 70;
    win[i][j] = (uint16_t)lr_row<PIX>(P, cdef, pre, ya - 3 + i, s0, s1)[clampi(xs - 3 + j, 0, W - 1)];
  }
}

// A/B of pass PASS (radius R) for the section rows ya - 1 .. yb and columns xs - 1 .. xs + 64 -> g_sgrA/B[PASS][row - (ya - 1)][col - (xs - 1)],
// from the staged window.  Pass 0 is only read on odd rows.  Lane = column xs + lane with a sliding window of row sums; the
// two edge columns are shared out over the lanes afterwards (direct sums).
template <int PASS>
__device__ __forceinline__ void sgr_grid(int bd, int ya, int yb, int lane) {
  constexpr int R = PASS == 0 ? 2 : 1, WN = 2 * R + 1;
  const uint16_t (*win)[72] = g_win;
  uint32_t h1[WN], h2[WN];   // ring of the last WN row sums
#pragma unroll
  for (int t = 0; t < WN; t++) { h1[t] = 0; h2[t] = 0; }
  for (int yy = ya - 1 - R; yy <= yb + R; yy++) {
    const uint16_t *row = win[yy - (ya - 3)] + lane + 3 - R;
    uint32_t r1 = 0, r2 = 0;
#pragma unroll
    for (int t = 0; t < WN; t++) { const uint32_t c = row[t]; r1 += c; r2 += c * c; }
#pragma unroll
    for (int t = 0; t < WN - 1; t++) { h1[t] = h1[t + 1]; h2[t] = h2[t + 1]; }
    h1[WN - 1] = r1; h2[WN - 1] = r2;
    const int yc = yy - R;   // centre row of the window that just became complete
    if (yc >= ya - 1 && (PASS == 1 || (yc & 1))) {
      uint32_t b = 0, a = 0;
#pragma unroll
      for (int t = 0; t < WN; t++) { b += h1[t]; a += h2[t]; }
      uint32_t A; int32_t B;
      sgr_ab<R>(a, b, bd, A, B);
      if (PASS == 0) { g_sgrA0[(yc - (ya - 1)) >> 1][lane + 1] = (uint16_t)A; g_sgrB0[(yc - (ya - 1)) >> 1][lane + 1] = B; }
      else { g_sgrA1[yc - (ya - 1)][lane + 1] = (uint16_t)A; g_sgrB1[yc - (ya - 1)][lane + 1] = B; }
    }
  }
  const int rows = yb - ya + 2;
  for (int task = lane; task < rows * 2; task += 64) {
    const int ri = task >> 1, side = task & 1;
    const int yc = ya - 1 + ri, j = side ? 67 : 2;   // window column of xs + 64 / xs - 1
    if (PASS == 0 && !(yc & 1)) continue;
    uint32_t a = 0, b = 0;
    for (int dy = -R; dy <= R; dy++) {
      const uint16_t *row = win[yc + dy - (ya - 3)] + j - R;
#pragma unroll
      for (int t = 0; t < WN; t++) { const uint32_t c = row[t]; b += c; a += c * c; }
    }
    uint32_t A; int32_t B;
    sgr_ab<R>(a, b, bd, A, B);
    if (PASS == 0) { g_sgrA0[ri >> 1][side ? 65 : 0] = (uint16_t)A; g_sgrB0[ri >> 1][side ? 65 : 0] = B; }
    else { g_sgrA1[ri][side ? 65 : 0] = (uint16_t)A; g_sgrB1[ri][side ? 65 : 0] = B; }
  }
}

// the two box-filter outputs (flt0: r = 2, flt1: r = 1) of sample (x = xs + lane, y) from the grids; cur = the CDEF sample
__device__ __forceinline__ void sgr_flt(int lane, int y, int ya, int cur, int &flt0, int &flt1) {
  const int ri = y - (ya - 1), c = lane + 1;
  {
    int a = 0, b = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
      for (int dx = -1; dx <= 1; dx++) {
        const int w = (dx == 0 || dy == 0) ? 4 : 3;
        a += w * (int)g_sgrA1[ri + dy][c + dx]; b += w * g_sgrB1[ri + dy][c + dx];
      }
    flt1 = (a * cur + b + (1 << 8)) >> 9;
  }
  {
    int a = 0, b = 0;
    if (y & 1) {  // odd row: the row itself, weights 5 6 5, shift 4
      const int r0 = ri >> 1;
      a = 5 * (int)g_sgrA0[r0][c - 1] + 6 * (int)g_sgrA0[r0][c] + 5 * (int)g_sgrA0[r0][c + 1];
      b = 5 * g_sgrB0[r0][c - 1] + 6 * g_sgrB0[r0][c] + 5 * g_sgrB0[r0][c + 1];
      flt0 = (a * cur + b + (1 << 7)) >> 8;
    } else {      // even row: the rows above and below, shift 5
#pragma unroll
      for (int dy = -1; dy <= 1; dy += 2) {
        const int r0 = (ri + dy) >> 1;
        a += 5 * (int)g_sgrA0[r0][c - 1] + 6 * (int)g_sgrA0[r0][c] + 5 * (int)g_sgrA0[r0][c + 1];
        b += 5 * g_sgrB0[r0][c - 1] + 6 * g_sgrB0[r0][c] + 5 * g_sgrB0[r0][c + 1];
      }
      flt0 = (a * cur + b + (1 << 8)) >> 9;
    }
  }
}
__device__ __forceinline__ int sgr_blend(int cur, int flt0, int flt1, int w0, int w1, int maxv) {
  const int u = cur << 4, w2 = 128 - w0 - w1;
  const int v = w1 * u + w0 * flt0 + w2 * flt1;   // set 9: both radii non-zero
  return clampi((v + (1 << 10)) >> 11, 0, maxv);
}

// A unit is LR_SLICES waves, one per 16 of its rows (the last unit of a column has up to 103), in two launches: PHASE 0 adds
// the slice's SSE of every candidate to the unit's sums (atomics), PHASE 1 reads the sums, takes the same decision in every
// slice and applies it to its rows.  (As one wave per unit the kernel took 221 us of an inter frame's serial chain.)
#define LR_SLICES 7
template <typename PIX, bool SGR, int PHASE>
__global__ void __launch_bounds__(64) lr_unit_kernel(Av1miDevParams P, const PIX *__restrict__ pre, const PIX *__restrict__ cdef,
                                                    const PIX *__restrict__ src, PIX *__restrict__ out, uint8_t *__restrict__ choice,
                                                    unsigned long long *__restrict__ unit_sse /* [frame][unit][8] */) {
  // units and stripes follow the signalled size; the last unit of a row/column also carries the padding up to the coded
  // size (copied, never filtered), so the whole frame buffer is defined
  const int urows = (P.true_h + 32) / 64 > 0 ? (P.true_h + 32) / 64 : 1, ucols = (P.true_w + 32) / 64 > 0 ? (P.true_w + 32) / 64 : 1;
  const int per_frame = urows * ucols;
  const int item = blockIdx.x / LR_SLICES, slice = blockIdx.x % LR_SLICES;
  const int f = item / per_frame, u = item % per_frame, ur = u / ucols, uc = u % ucols;
  const int lane = threadIdx.x;
  const size_t fo = (size_t)f * P.frame_samples;
  pre += fo; cdef += fo; src += fo; out += fo;
  const int uy0 = ur ? ur * 64 - 8 : 0, uy1 = ur == urows - 1 ? P.true_h : ur * 64 + 56;   // the unit's rows
  const int y0 = uy0 + slice * 16, y1 = y0 + 16 < uy1 ? y0 + 16 : uy1;                      // this wave's rows
  if (y0 >= uy1) return;
  const int x0 = uc * 64, x1 = uc == ucols - 1 ? P.true_w : x0 + 64;
  const int maxv = (1 << P.bit_depth) - 1;
  unsigned long long *usse = unit_sse + (size_t)item * 8;
  if constexpr (PHASE == 0) {
  // ---- SSE without restoration and with each candidate
  unsigned long long sse[7] = { 0, 0, 0, 0, 0, 0, 0 };
  for (int xs = x0; xs < x1; xs += 64) {
    const int x = xs + lane;
    const bool active = x < x1;
    for (int st = (y0 + 8) / 64; st * 64 - 8 < y1; st++) {
      const int s0 = st * 64 - 8, s1 = s0 + 63;
      const int ya = y0 > s0 ? y0 : s0, yb = y1 < s1 + 1 ? y1 : s1 + 1;
      __syncthreads();
      lr_stage<PIX>(P, cdef, pre, xs, ya, yb, s0, s1, lane);   // the section's source window, once, for every candidate
      __syncthreads();
      int cur[16], sv[16];   // this lane's CDEF and source samples of the section (<= 16 rows)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int y = ya + i < yb ? ya + i : yb - 1;
        cur[i] = (int)g_win[y - ya + 3][lane + 3];
        sv[i] = active ? (int)src[(size_t)y * P.stride_y + x] : cur[i];
      }
#pragma unroll
      for (int i = 0; i < 16; i++)
        if (ya + i < yb) { const int d = cur[i] - sv[i]; sse[0] += (unsigned long long)(d * d); }
      for (int k = 0; k < 3; k++) {
        int tf[7];
        taps_of(k, tf);
        wiener_h(P.bit_depth, ya, yb, tf, lane);   // g_mid is private to the lane's column: no barrier needed
#pragma unroll
        for (int i = 0; i < 16; i++)
          if (active && ya + i < yb) { const int d = wiener_v(i, lane, tf, maxv) - sv[i]; sse[k + 1] += (unsigned long long)(d * d); }
      }
      if constexpr (SGR) {
        sgr_grid<0>(P.bit_depth, ya, yb, lane);
        sgr_grid<1>(P.bit_depth, ya, yb, lane);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; i++)
          if (active && ya + i < yb) {
            int f0, f1;
            sgr_flt(lane, ya + i, ya, cur[i], f0, f1);
#pragma unroll
            for (int k = 0; k < 3; k++) {
              const int d = sgr_blend(cur[i], f0, f1, c_sgr_cand[k][1], c_sgr_cand[k][2], maxv) - sv[i];
              sse[4 + k] += (unsigned long long)(d * d);
            }
          }
      }
    }
  }
  for (int k = 0; k < (SGR ? 7 : 4); k++) {
    const unsigned long long v = wave_sum64(sse[k]);
    if (lane == 0 && v) atomicAdd(&usse[k], v);
  }
  } else {
  int best = 0;
  {
    unsigned long long bs = usse[0];
    const int ncand = SGR ? 6 : 3;
    for (int k = 0; k < ncand; k++) {
      const unsigned long long s = usse[k + 1];
      if (s < bs) { bs = s; best = k + 1; }
    }
  }
  if (lane == 0 && slice == 0) choice[item] = (uint8_t)best;
  // ---- apply: luma of the slice, and the co-located chroma (copied)
  for (int xs = x0; xs < x1; xs += 64) {
    const int x = xs + lane;
    const bool active = x < x1;
    for (int st = (y0 + 8) / 64; st * 64 - 8 < y1; st++) {
      const int s0 = st * 64 - 8, s1 = s0 + 63;
      const int ya = y0 > s0 ? y0 : s0, yb = y1 < s1 + 1 ? y1 : s1 + 1;
      if (best) {
        __syncthreads();
        lr_stage<PIX>(P, cdef, pre, xs, ya, yb, s0, s1, lane);
        __syncthreads();
      }
      if (best > 3) {
        if constexpr (SGR) {
          sgr_grid<0>(P.bit_depth, ya, yb, lane);
          sgr_grid<1>(P.bit_depth, ya, yb, lane);
          __syncthreads();
          if (active)
            for (int y = ya; y < yb; y++) {
              const int cur = (int)g_win[y - ya + 3][lane + 3];
              int f0, f1;
              sgr_flt(lane, y, ya, cur, f0, f1);
              out[(size_t)y * P.stride_y + x] = (PIX)sgr_blend(cur, f0, f1, c_sgr_cand[best - 4][1], c_sgr_cand[best - 4][2], maxv);
            }
        }
      } else if (best) {
        int tf[7];
        taps_of(best - 1, tf);
        wiener_h(P.bit_depth, ya, yb, tf, lane);
        if (active)
          for (int y = ya; y < yb; y++) out[(size_t)y * P.stride_y + x] = (PIX)wiener_v(y - ya, lane, tf, maxv);
      } else if (active) {
        for (int y = ya; y < yb; y++) out[(size_t)y * P.stride_y + x] = cdef[(size_t)y * P.stride_y + x];
      }
    }
  }
  {
    // padding between the signalled and the coded size (< 8 samples): copied with the last unit of the row / column (its last slice)
    const int py1 = (ur == urows - 1 && y1 == uy1) ? P.height : y1, px1 = uc == ucols - 1 ? P.width : x1;
    for (int y = y0; y < py1; y++)
      for (int x = x0 + lane; x < px1; x += 64)
        if (y >= y1 || x >= x1) out[(size_t)y * P.stride_y + x] = cdef[(size_t)y * P.stride_y + x];
    const int cy0 = y0 >> 1, cy1 = py1 >> 1, cx0 = x0 >> 1, cx1 = px1 >> 1;
    for (int pl = 0; pl < 2; pl++) {
      const size_t po = pl ? P.plane_off_v : P.plane_off_u;
      for (int y = cy0; y < cy1; y++)
        for (int x = cx0 + lane; x < cx1; x += 64) out[po + (size_t)y * P.stride_c + x] = cdef[po + (size_t)y * P.stride_c + x];
    }
  }
  }  // PHASE 1
}

template <typename PIX, bool SGR>
void launch_lr_phases(const Av1miDevParams *P, int grid, const PIX *pre, const PIX *cdef, const PIX *src, PIX *out, uint8_t *choice,
                      unsigned long long *unit_sse, hipStream_t stream) {
  hipLaunchKernelGGL((lr_unit_kernel<PIX, SGR, 0>), dim3(grid), dim3(64), 0, stream, *P, pre, cdef, src, out, choice, unit_sse);
  hipLaunchKernelGGL((lr_unit_kernel<PIX, SGR, 1>), dim3(grid), dim3(64), 0, stream, *P, pre, cdef, src, out, choice, unit_sse);
}

}  // namespace

// unit_sse: P->n_frames x units x 8 sums (scratch of the two phases; cleared here unless the caller did - the frame loop of a P chunk
// clears the whole chunk's once instead of putting a fill between every frame's kernels on the chain).
extern "C" hipError_t av1mi_launch_lr(const Av1miDevParams *P, const void *pre, const void *cdef, const void *src, void *out, uint8_t *choice,
                                      unsigned long long *unit_sse, int clear, hipStream_t stream) {
  const int urows = (P->true_h + 32) / 64 > 0 ? (P->true_h + 32) / 64 : 1, ucols = (P->true_w + 32) / 64 > 0 ? (P->true_w + 32) / 64 : 1;
  const int units = P->n_frames * urows * ucols, grid = units * LR_SLICES;
  if (clear) {
    hipError_t e = hipMemsetAsync(unit_sse, 0, (size_t)units * 8 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
  }
  // enable_lr = 2 (RESTORE_SWITCHABLE): the instantiation with the self-guided candidates (24 KB of LDS per wave)
  if (P->bit_depth == 8) {
    if (P->enable_lr == 2) launch_lr_phases<uint8_t, true>(P, grid, (const uint8_t *)pre, (const uint8_t *)cdef, (const uint8_t *)src, (uint8_t *)out, choice, unit_sse, stream);
    else launch_lr_phases<uint8_t, false>(P, grid, (const uint8_t *)pre, (const uint8_t *)cdef, (const uint8_t *)src, (uint8_t *)out, choice, unit_sse, stream);
  } else {
    if (P->enable_lr == 2) launch_lr_phases<uint16_t, true>(P, grid, (const uint16_t *)pre, (const uint16_t *)cdef, (const uint16_t *)src, (uint16_t *)out, choice, unit_sse, stream);
    else launch_lr_phases<uint16_t, false>(P, grid, (const uint16_t *)pre, (const uint16_t *)cdef, (const uint16_t *)src, (uint16_t *)out, choice, unit_sse, stream);
  }
  return hipGetLastError();
}
