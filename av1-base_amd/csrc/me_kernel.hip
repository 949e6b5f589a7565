// me_kernel.hip - integer-pel full-search motion estimation for inter frames.
//
// Replaces the motion search SVT-AV1 runs inside the av1an worker the reference forks
// (/root/reference/crates/daemon/src/encode/av1an.rs:126-139).  Encoder-side, non-normative; the algorithm is the
// one DESIGN.md §3.9 defines and oracle/av1o_enc.c (motion_search) restates: for every leaf block of the partition,
// cost(dx, dy) = SAD(source block, previous reconstruction displaced by (dx, dy)) + n * (|dx| + |dy|) over
// |dx|, |dy| <= R with the displaced block kept within 16 samples of the frame; minimum cost, ties to the first
// candidate in (dy, dx) raster order.
//
// MI355X mapping: frames of a chunk are coded one after the other (each P frame needs the previous
// reconstruction), so one launch sees only one frame: the grid is (32x32 cells of the frame) x (2R+1 values of
// dy) = 34 680 waves at 1080p, R = 8 - enough to fill the chip from a single frame.  A wave owns one cell and one
// dy: lane = 16 consecutive samples of one cell row (source in registers, the 16 + 2R reference samples the
// 2R+1 dx candidates need in registers, each loaded once); per dx the lane forms two 8-sample partial SADs,
// xor-shuffles over the row bits turn them into the 16 8x8 sub-block SADs of the cell, and the leaf blocks of the
// cell (one 32x32, or 16x16 / 8x8 blocks at frame edges or smaller block sizes) sum their sub-blocks from LDS.
// Each leaf's best candidate of this dy goes into a 64-bit atomicMin on (cost << 16 | candidate index).
// Algorithmic HBM bytes per frame: source luma read once + reference luma read once = 2*L*b (L luma samples);
// the (2R+1)-fold re-reads of the reference by the dy waves of a cell are L2 hits.
#include <hip/hip_runtime.h>
#include "av1mi_dev.h"

namespace {

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// Leaf block size (log2) at cell-local 8x8 unit (ux, uy) of the 32x32 cell at (cx, cy), or 0 if the unit is not the
// origin of a leaf: same rule as the recon kernel's leaf_bsl_at restricted to sizes <= 32.
__device__ __forceinline__ int leaf_bsl_cell(const Av1miDevParams &P, int cx, int cy, int ux, int uy) {
  const int x = cx + ux * 8, y = cy + uy * 8;
  if (x >= P.width || y >= P.height) return 0;
  for (int bsl = 5; bsl >= 3; bsl--) {
    const int n = 1 << bsl;
    const int ox = x & ~(n - 1), oy = y & ~(n - 1);
    const bool split = (bsl > P.max_bs_log2 && bsl > 3) || ((oy + n > P.height || ox + n > P.width) && bsl > 3);
    if (!split) return (ox == x && oy == y) ? bsl : 0;
  }
  return 0;
}

template <typename PIX, int R>
__global__ void __launch_bounds__(64) motion_search_kernel(Av1miDevParams P, const PIX *__restrict__ src, const PIX *__restrict__ ref,
                                                          unsigned long long *__restrict__ best /* per 8x8 unit of the frame */) {
  constexpr int NC = 2 * R + 1;
  __shared__ uint32_t sad8[NC][16];  // [dx][8x8 sub-block of the cell, raster]
  const int cells_x = (P.width + 31) >> 5;
  const int cell = blockIdx.x, dyi = blockIdx.y, dy = dyi - R;
  const int cx = (cell % cells_x) * 32, cy = (cell / cells_x) * 32;
  const int lane = threadIdx.x, r = lane >> 1, half = lane & 1;
  const int W = P.width, H = P.height;
  // ---- source: 16 samples of row cy + r (zeros outside the frame: those sub-blocks are never used)
  int s[16];
  {
    const int y = cy + r, x0 = cx + half * 16;
#pragma unroll
    for (int j = 0; j < 16; j++) s[j] = (y < H && x0 + j < W) ? (int)src[(size_t)y * P.stride_y + x0 + j] : 0;
  }
  // ---- reference: samples x0 - R .. x0 + 15 + R of row cy + r + dy, coordinates clamped to the frame
  int rf[16 + 2 * R];
  {
    int y = cy + r + dy;
    y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
    const int x0 = cx + half * 16 - R;
    const PIX *row = ref + (size_t)y * P.stride_y;
    if (x0 >= 0 && x0 + 16 + 2 * R <= W) {
#pragma unroll
      for (int j = 0; j < 16 + 2 * R; j++) rf[j] = (int)row[x0 + j];
    } else {
#pragma unroll
      for (int j = 0; j < 16 + 2 * R; j++) { int x = x0 + j; x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x); rf[j] = (int)row[x]; }
    }
  }
  // ---- per dx: two 8-sample partial SADs, reduced over the 8 rows of a sub-block (lane bits 1..3)
#pragma unroll
  for (int dxi = 0; dxi < NC; dxi++) {
    int a = 0, b = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { a += iabs(s[j] - rf[j + dxi]); b += iabs(s[j + 8] - rf[j + 8 + dxi]); }
    a += __shfl_xor(a, 2, 64); b += __shfl_xor(b, 2, 64);
    a += __shfl_xor(a, 4, 64); b += __shfl_xor(b, 4, 64);
    a += __shfl_xor(a, 8, 64); b += __shfl_xor(b, 8, 64);
    if ((r & 7) == 0) {  // lanes of the first row of each sub-block row: sub-blocks (r >> 3, 2 * half) and (.., 2 * half + 1)
      sad8[dxi][(r >> 3) * 4 + 2 * half] = (uint32_t)a;
      sad8[dxi][(r >> 3) * 4 + 2 * half + 1] = (uint32_t)b;
    }
  }
  __syncthreads();
  // ---- leaves of the cell: lane u < 16 = 8x8 unit u; a leaf origin sums its sub-blocks for every dx
  if (lane < 16) {
    const int ux = lane & 3, uy = lane >> 2;
    const int bsl = leaf_bsl_cell(P, cx, cy, ux, uy);
    if (bsl) {
      const int n = 1 << bsl, n8 = n >> 3;
      const int x = cx + ux * 8, y = cy + uy * 8;
      if (y + dy >= -16 && y + dy + n <= H + 16) {
        unsigned long long bk = ~0ull;
        for (int dxi = 0; dxi < NC; dxi++) {
          const int dx = dxi - R;
          if (x + dx < -16 || x + dx + n > W + 16) continue;
          uint32_t sad = 0;
          for (int i = 0; i < n8; i++)
            for (int j = 0; j < n8; j++) sad += sad8[dxi][(uy + i) * 4 + ux + j];
          const unsigned long long cost = (unsigned long long)sad + (unsigned long long)(n * (iabs(dx) + iabs(dy)));
          const unsigned long long key = (cost << 16) | (unsigned long long)(dyi * NC + dxi);
          bk = key < bk ? key : bk;
        }
        if (bk != ~0ull) atomicMin(&best[(size_t)((cy >> 3) + uy) * P.b8_cols + (cx >> 3) + ux], bk);
      }
    }
  }
}

}  // namespace

// best[] must be filled with 0xFF bytes before the launch.  `src`, `ref`: luma planes of the frame and of the
// previous frame's final reconstruction.  R must be 8 or 16.
extern "C" hipError_t av1mi_launch_motion_search(const Av1miDevParams *P, const void *src, const void *ref, unsigned long long *best,
                                                 int me_range, hipStream_t stream) {
  const int cells = ((P->width + 31) >> 5) * ((P->height + 31) >> 5);
  dim3 grid(cells, 2 * me_range + 1);
  if (P->bit_depth == 8) {
    if (me_range == 8) hipLaunchKernelGGL((motion_search_kernel<uint8_t, 8>), grid, dim3(64), 0, stream, *P, (const uint8_t *)src, (const uint8_t *)ref, best);
    else hipLaunchKernelGGL((motion_search_kernel<uint8_t, 16>), grid, dim3(64), 0, stream, *P, (const uint8_t *)src, (const uint8_t *)ref, best);
  } else {
    if (me_range == 8) hipLaunchKernelGGL((motion_search_kernel<uint16_t, 8>), grid, dim3(64), 0, stream, *P, (const uint16_t *)src, (const uint16_t *)ref, best);
    else hipLaunchKernelGGL((motion_search_kernel<uint16_t, 16>), grid, dim3(64), 0, stream, *P, (const uint16_t *)src, (const uint16_t *)ref, best);
  }
  return hipGetLastError();
}
