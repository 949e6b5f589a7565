// deblock_kernel.hip - AV1 deblocking filter (spec §7.14), in place on the reconstruction before
// CDEF.  Replaces the loop-filter stage of the SVT-AV1 worker behind `run_av1an`
// (/root/reference/crates/daemon/src/encode/av1an.rs:126-139).  Restated in oracle/av1o_deblock.c (pinned by dav1d).
//
// Structure of this build: square blocks, transform == block, no segmentation or loop-filter deltas - every transform
// edge is a block edge and the level is frame-wide.  Per plane all vertical edges are filtered, then all horizontal
// edges; within one pass the edges are independent (a filter modifies at most 6 samples and reads at most 7 on each
// side, and reaches that far only when both neighbouring transforms are >= 16 wide), so one pass is one launch:
// one thread per 4-sample edge segment of any plane.  HBM bound: each pass reads and writes the samples next to the
// edges; algorithmic bytes <= 2 * 2 * N * b per frame (both passes, read + write).
#include <hip/hip_runtime.h>
#include "av1mi_dev.h"

namespace {

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// one sample position across an edge: px[-k*step] = p(k-1), px[k*step] = q(k).  len: 4, 6 (chroma), 8 or 16 = filterLen
template <typename PIX>
__device__ __forceinline__ void filter_sample(PIX *px, long step, int plane, int lim, int blim, int thr, int len, int bd) {
  const int one = 1 << (bd - 8);
  int t[16];  // t[8 + i] = sample i (i < 0: p(-i-1), i >= 0: q(i)) for i in -8..7
  const int reach = len == 16 ? 7 : (len == 8 ? 4 : (len == 6 ? 3 : 2));
#pragma unroll
  for (int i = -7; i < 7; i++) t[8 + i] = (i >= -reach && i < reach) ? (int)px[i * step] : 0;
#define P(k) t[7 - (k)]
#define Q(k) t[8 + (k)]
  const int p0 = P(0), p1 = P(1), q0 = Q(0), q1 = Q(1);
  const bool hev = iabs(p1 - p0) > thr || iabs(q1 - q0) > thr;
  bool mask = iabs(p1 - p0) > lim || iabs(q1 - q0) > lim || iabs(p0 - q0) * 2 + iabs(p1 - q1) / 2 > blim;
  if (len >= 6) mask = mask || iabs(P(2) - p1) > lim || iabs(Q(2) - q1) > lim;
  if (len >= 8) mask = mask || iabs(P(3) - P(2)) > lim || iabs(Q(3) - Q(2)) > lim;
  if (mask) return;
  bool flat = false, flat2 = false;
  if (len >= 6) {
    flat = iabs(p1 - p0) <= one && iabs(q1 - q0) <= one && iabs(P(2) - p0) <= one && iabs(Q(2) - q0) <= one;
    if (len >= 8) flat = flat && iabs(P(3) - p0) <= one && iabs(Q(3) - q0) <= one;
  }
  if (len >= 16) flat2 = iabs(P(4) - p0) <= one && iabs(Q(4) - q0) <= one && iabs(P(5) - p0) <= one && iabs(Q(5) - q0) <= one &&
                         iabs(P(6) - p0) <= one && iabs(Q(6) - q0) <= one;
  if (len == 4 || !flat) {
    // narrow filter §7.14.6.3
    const int lo = -(1 << (bd - 1)), hi = (1 << (bd - 1)) - 1, half = 0x80 << (bd - 8);
    const int ps1 = p1 - half, ps0 = p0 - half, qs0 = q0 - half, qs1 = q1 - half;
    int f = hev ? clampi(ps1 - qs1, lo, hi) : 0;
    f = clampi(f + 3 * (qs0 - ps0), lo, hi);
    const int f1 = clampi(f + 4, lo, hi) >> 3, f2 = clampi(f + 3, lo, hi) >> 3;
    px[0] = (PIX)(clampi(qs0 - f1, lo, hi) + half);
    px[-step] = (PIX)(clampi(ps0 + f2, lo, hi) + half);
    if (!hev) {
      const int g = (f1 + 1) >> 1;
      px[step] = (PIX)(clampi(qs1 - g, lo, hi) + half);
      px[-2 * step] = (PIX)(clampi(ps1 + g, lo, hi) + half);
    }
  } else {
    // wide filter §7.14.6.4: 2n + 1 taps (n = 6 / 3 / 2) whose weights sum to 1 << log2size
    const int log2size = (len == 16 && flat2) ? 4 : 3;
    const int n = log2size == 4 ? 6 : (plane == 0 ? 3 : 2), n2 = (log2size == 3 && plane == 0) ? 0 : 1;
    int out[12];
#pragma unroll
    for (int i = -6; i < 6; i++) {
      int s = 0;
      if (i >= -n && i < n) {
#pragma unroll
        for (int j = -6; j <= 6; j++) {
          if (j < -n || j > n) continue;
          const int p = clampi(i + j, -(n + 1), n);
          s += t[8 + p] * (iabs(j) <= n2 ? 2 : 1);
        }
        s = (s + (1 << (log2size - 1))) >> log2size;
      }
      out[i + 6] = s;
    }
#pragma unroll
    for (int i = -6; i < 6; i++)
      if (i >= -n && i < n) px[i * step] = (PIX)out[i + 6];
  }
#undef P
#undef Q
}

// PASS 0: vertical edges, 1: horizontal edges.  grid.x covers the 4x4 positions of all three planes, grid.y = frame.
template <typename PIX, int PASS>
__global__ void __launch_bounds__(256) deblock_kernel(Av1miDevParams P, PIX *__restrict__ rec, const Av1miBlkInfo *__restrict__ blk) {
  const int f = blockIdx.y;
  const long n_luma = (long)P.mi_rows * P.mi_cols, n_chroma = n_luma >> 2;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= n_luma + 2 * n_chroma) return;
  const int plane = id < n_luma ? 0 : (id < n_luma + n_chroma ? 1 : 2);
  const long local = id - (plane == 0 ? 0 : (plane == 1 ? n_luma : n_luma + n_chroma));
  const int ss = plane > 0;
  const int pcols = P.mi_cols >> ss;
  const int r4 = (int)(local / pcols), c4 = (int)(local % pcols);   // 4x4 position in the plane
  const int row = r4 << ss, col = c4 << ss;                         // the same in luma 4x4 units
  const int lvl = plane == 0 ? P.lf_level[PASS] : P.lf_level[plane + 1];
  if (!lvl) return;
  if (col * 4 >= P.true_w || row * 4 >= P.true_h) return;           // onScreen (§7.14.2)
  if (PASS == 0 ? c4 == 0 : r4 == 0) return;
  const Av1miBlkInfo *info = blk + (size_t)f * P.b8_rows * P.b8_cols;
  const int prow = row - (PASS ? 1 << ss : 0), pcol = col - (PASS ? 0 : 1 << ss);
  const int bsl = info[(size_t)(row >> 1) * P.b8_cols + (col >> 1)].bsl, pbsl = info[(size_t)(prow >> 1) * P.b8_cols + (pcol >> 1)].bsl;
  int txw = (1 << bsl) >> ss, ptxw = (1 << pbsl) >> ss;
  txw = txw < 4 ? 4 : txw; ptxw = ptxw < 4 ? 4 : ptxw;
  const int x = c4 * 4, y = r4 * 4;
  if (((PASS == 0 ? x : y) & (txw - 1)) != 0) return;                // not a transform (= block) edge
  const int base = txw < ptxw ? txw : ptxw;
  const int len = plane == 0 ? (base >= 16 ? 16 : base) : (base >= 8 ? 6 : 4);
  const int sharp = P.lf_sharpness;
  const int shift = sharp > 4 ? 2 : (sharp > 0 ? 1 : 0);
  const int limit = sharp > 0 ? clampi(lvl >> shift, 1, 9 - sharp) : ((lvl >> shift) > 1 ? (lvl >> shift) : 1);
  const int sh = P.bit_depth - 8;
  const int lim = limit << sh, blim = (2 * (lvl + 2) + limit) << sh, thr = (lvl >> 4) << sh;
  const long stride = plane ? P.stride_c : P.stride_y;
  PIX *pl = rec + (size_t)f * P.frame_samples + (plane == 0 ? 0 : (plane == 1 ? P.plane_off_u : P.plane_off_v));
#pragma unroll
  for (int i = 0; i < 4; i++) {
    PIX *px = pl + (size_t)(y + (PASS ? 0 : i)) * stride + x + (PASS ? i : 0);
    filter_sample<PIX>(px, PASS ? stride : 1, plane, lim, blim, thr, len, P.bit_depth);
  }
}

}  // namespace

// both passes over P->n_frames frames of `rec`, in place
extern "C" hipError_t av1mi_launch_deblock(const Av1miDevParams *P, void *rec, const Av1miBlkInfo *blk, hipStream_t stream) {
  const long n = (long)P->mi_rows * P->mi_cols * 3 / 2;
  dim3 grid((unsigned)((n + 255) / 256), P->n_frames);
  if (P->bit_depth == 8) {
    hipLaunchKernelGGL((deblock_kernel<uint8_t, 0>), grid, dim3(256), 0, stream, *P, (uint8_t *)rec, blk);
    hipLaunchKernelGGL((deblock_kernel<uint8_t, 1>), grid, dim3(256), 0, stream, *P, (uint8_t *)rec, blk);
  } else {
    hipLaunchKernelGGL((deblock_kernel<uint16_t, 0>), grid, dim3(256), 0, stream, *P, (uint16_t *)rec, blk);
    hipLaunchKernelGGL((deblock_kernel<uint16_t, 1>), grid, dim3(256), 0, stream, *P, (uint16_t *)rec, blk);
  }
  return hipGetLastError();
}
