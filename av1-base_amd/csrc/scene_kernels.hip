// scene_kernels.hip - luma frame-difference statistics for the scene-cut chunker.
//
// Replaces the scene detection av1an runs before it splits a clip into chunks for its workers
// (`--workers N --temp DIR`, /root/reference/crates/daemon/src/encode/av1an.rs:100-104; SURVEY.md §8a
// row a9): SAD of the luma plane of every frame against its predecessor.  The cut rule itself is a few
// integer comparisons per frame and runs on the host (av1mi_host.cpp: decide_scene_cuts).
//
// HBM-bound streaming kernel: every luma sample of the chunk is read twice (as frame t and as frame
// t-1) with 16-byte loads per lane; algorithmic bytes 2*L*b per frame pair (L luma samples).
#include <hip/hip_runtime.h>
#include "av1mi_dev.h"

namespace {

__device__ __forceinline__ unsigned sad_u8x4(unsigned a, unsigned b, unsigned acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }

template <int BPS>
__device__ __forceinline__ unsigned sad16(const uint4 &a, const uint4 &b) {
  unsigned s = 0;
  if (BPS == 1) {
    s = sad_u8x4(a.x, b.x, s); s = sad_u8x4(a.y, b.y, s); s = sad_u8x4(a.z, b.z, s); s = sad_u8x4(a.w, b.w, s);
  } else {
    s = __builtin_amdgcn_sad_u16(a.x, b.x, s); s = __builtin_amdgcn_sad_u16(a.y, b.y, s);
    s = __builtin_amdgcn_sad_u16(a.z, b.z, s); s = __builtin_amdgcn_sad_u16(a.w, b.w, s);
  }
  return s;
}

// grid (x = slices of a frame, y = frame index t).  prev0 = predecessor of frame 0 (may be null: sad[0] = 0).
template <int BPS>
__global__ void __launch_bounds__(256) luma_sad_kernel(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ prev0,
                                                      size_t frame_bytes, size_t luma_bytes,
                                                      unsigned long long *__restrict__ sad /* [n_frames] */) {
  const int t = blockIdx.y;
  const uint8_t *cur = frames + (size_t)t * frame_bytes;
  const uint8_t *prv = t ? cur - frame_bytes : prev0;
  if (!prv) return;
  unsigned long long acc = 0;
  if (((frame_bytes | luma_bytes) & 15) == 0) {  // sizes that are multiples of 8 both ways: 16-byte loads
    const size_t nvec = luma_bytes >> 4;
    const uint4 *c4 = reinterpret_cast<const uint4 *>(cur), *p4 = reinterpret_cast<const uint4 *>(prv);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) acc += sad16<BPS>(c4[i], p4[i]);
  } else {                                       // any other (even) size: sample by sample
    const size_t ns = luma_bytes / BPS;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < ns; i += (size_t)gridDim.x * 256) {
      const int a = BPS == 1 ? (int)cur[i] : (int)reinterpret_cast<const uint16_t *>(cur)[i];
      const int b = BPS == 1 ? (int)prv[i] : (int)reinterpret_cast<const uint16_t *>(prv)[i];
      acc += (unsigned long long)(a < b ? b - a : a - b);
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&sad[t], acc);
}

// ---- frame sizes that are not multiples of 8: the encoder works on frames edge-extended to the next multiple of 8.
// pad: tight planar I420 frames of w x h -> frames of cw x ch, last column / row replicated; crop: the reverse.
template <typename PIX>
__global__ void __launch_bounds__(256) pad_frames_kernel(const PIX *__restrict__ in, PIX *__restrict__ out, int w, int h, int cw, int ch) {
  const size_t in_frame = (size_t)w * h * 3 / 2, out_frame = (size_t)cw * ch * 3 / 2;
  const PIX *fi = in + (size_t)blockIdx.y * in_frame;
  PIX *fo = out + (size_t)blockIdx.y * out_frame;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < out_frame; i += (size_t)gridDim.x * 256) {
    const size_t ny = (size_t)cw * ch, nc = ny >> 2;
    const int pl = i < ny ? 0 : (i < ny + nc ? 1 : 2);
    const size_t o = i - (pl == 0 ? 0 : (pl == 1 ? ny : ny + nc));
    const int pcw = pl ? cw >> 1 : cw, pw = pl ? w >> 1 : w, ph = pl ? h >> 1 : h;
    int y = (int)(o / pcw), x = (int)(o % pcw);
    y = y < ph ? y : ph - 1; x = x < pw ? x : pw - 1;
    const size_t ioff = pl == 0 ? 0 : (pl == 1 ? (size_t)w * h : (size_t)w * h + (size_t)(w >> 1) * (h >> 1));
    fo[i] = fi[ioff + (size_t)y * pw + x];
  }
}
template <typename PIX>
__global__ void __launch_bounds__(256) crop_frames_kernel(const PIX *__restrict__ in, PIX *__restrict__ out, int w, int h, int cw, int ch) {
  const size_t in_frame = (size_t)cw * ch * 3 / 2, out_frame = (size_t)w * h * 3 / 2;
  const PIX *fi = in + (size_t)blockIdx.y * in_frame;
  PIX *fo = out + (size_t)blockIdx.y * out_frame;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < out_frame; i += (size_t)gridDim.x * 256) {
    const size_t ny = (size_t)w * h, nc = ny >> 2;
    const int pl = i < ny ? 0 : (i < ny + nc ? 1 : 2);
    const size_t o = i - (pl == 0 ? 0 : (pl == 1 ? ny : ny + nc));
    const int pw = pl ? w >> 1 : w, pcw = pl ? cw >> 1 : cw;
    const int y = (int)(o / pw), x = (int)(o % pw);
    const size_t ioff = pl == 0 ? 0 : (pl == 1 ? (size_t)cw * ch : (size_t)cw * ch + (size_t)(cw >> 1) * (ch >> 1));
    fo[i] = fi[ioff + (size_t)y * pcw + x];
  }
}

// ---- content-driven partition (av1mi_params.partition_search; SURVEY.md §8a row a10: the block-size decision SVT-AV1
// spends most of `--preset 3` on, /root/reference/crates/daemon/src/encode/av1an.rs:14).  Open loop, from the SOURCE luma: a node of the
// partition tree splits when its four quadrants differ in activity - the largest quadrant variance exceeds four times the smallest plus
// (ac_q / 16)^2 - so every frame of a chunk is decided in one streaming pass before any tile walk.  One wave per superblock: lane = 8x8
// unit in Z order (sum and sum of squares of its 64 samples, coordinates beyond the frame repeating the last column / row: what the
// block's transform will see), then one lane per node - sixteen 16x16, four 32x32, one 64x64 - compares its quadrants.  Writes the
// superblock's split mask (av1mi_dev.h: av1mi_node_split reads it).  Algorithmic bytes L b per frame (L luma samples).
template <typename PIX>
__global__ void __launch_bounds__(64) partition_kernel(Av1miDevParams P, const PIX *__restrict__ frames, uint32_t *__restrict__ part) {
  __shared__ uint32_t uS[64], uQ[64];
  __shared__ uint32_t bits[21];
  const int f = blockIdx.y, sb = blockIdx.x, lane = threadIdx.x;
  const int sb_x = (sb % P.sb_cols) * 64, sb_y = (sb / P.sb_cols) * 64;
  const PIX *luma = frames + (size_t)f * P.frame_samples;
  {
    const int ux = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), uy = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int x0 = sb_x + 8 * ux, y0 = sb_y + 8 * uy;
    uint32_t S = 0, Q = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int yy = y0 + i < P.height ? y0 + i : P.height - 1;
      const PIX *row = luma + (size_t)yy * P.stride_y;
      if (x0 + 8 <= P.width) {   // eight samples of the row in one load (the coded width is a multiple of 8)
        if (sizeof(PIX) == 2) {
          const uint4 v = *reinterpret_cast<const uint4 *>(row + x0);
          const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
          for (int k = 0; k < 4; k++) { const uint32_t a = w[k] & 0xFFFF, b = w[k] >> 16; S += a + b; Q += a * a + b * b; }
        } else {
          const uint2 v = *reinterpret_cast<const uint2 *>(row + x0);
          const uint32_t w[2] = { v.x, v.y };
#pragma unroll
          for (int k = 0; k < 2; k++)
#pragma unroll
            for (int b8 = 0; b8 < 4; b8++) { const uint32_t a = (w[k] >> (8 * b8)) & 0xFF; S += a; Q += a * a; }
        }
      } else {
        for (int j = 0; j < 8; j++) { const int xx = x0 + j < P.width ? x0 + j : P.width - 1; const uint32_t a = row[xx]; S += a; Q += a * a; }
      }
    }
    uS[lane] = S; uQ[lane] = Q;
  }
  __syncthreads();
  if (lane < 21) {
    // node of this lane: 0 .. 15 the 16x16 nodes (Z order; quadrants = 1 unit each), 16 .. 19 the 32x32 nodes (4 units per quadrant), 20 the
    // 64x64 node (16 units per quadrant); the units of a node are consecutive in Z order
    const int per_q = lane < 16 ? 1 : (lane < 20 ? 4 : 16);
    const int first = lane < 16 ? 4 * lane : (lane < 20 ? 16 * (lane - 16) : 0);
    const unsigned long long m = 64ull * per_q, t = (unsigned long long)(P.ac_q >> 4);
    unsigned long long vmin = ~0ull, vmax = 0;
    for (int q = 0; q < 4; q++) {
      unsigned long long S = 0, Q = 0;
      for (int u = 0; u < per_q; u++) { S += uS[first + q * per_q + u]; Q += uQ[first + q * per_q + u]; }
      const unsigned long long V = m * Q - S * S;
      vmin = V < vmin ? V : vmin;
      vmax = V > vmax ? V : vmax;
    }
    // bit 0: the 64x64 node, bits 1 .. 4: the 32x32 nodes, bits 5 .. 20: the 16x16 nodes
    bits[lane < 16 ? 5 + lane : (lane < 20 ? 1 + (lane - 16) : 0)] = vmax > 4 * vmin + t * t * m * m;
  }
  __syncthreads();
  if (lane == 0) {
    uint32_t mask = 0;
    for (int b = 0; b < 21; b++) mask |= (bits[b] & 1u) << b;
    part[(size_t)f * P.sb_rows * P.sb_cols + sb] = mask;
  }
}

}  // namespace

// split masks of every superblock of P->n_frames frames (partition_kernel)
extern "C" hipError_t av1mi_launch_partition(const Av1miDevParams *P, const void *frames, uint32_t *part, hipStream_t stream) {
  dim3 grid(P->sb_rows * P->sb_cols, P->n_frames);
  if (P->bit_depth == 8) hipLaunchKernelGGL(partition_kernel<uint8_t>, grid, dim3(64), 0, stream, *P, (const uint8_t *)frames, part);
  else hipLaunchKernelGGL(partition_kernel<uint16_t>, grid, dim3(64), 0, stream, *P, (const uint16_t *)frames, part);
  return hipGetLastError();
}

extern "C" hipError_t av1mi_launch_pad(const void *in, void *out, int w, int h, int cw, int ch, int bit_depth, int n_frames, int crop,
                                       hipStream_t stream) {
  dim3 grid(256, n_frames);
  if (bit_depth == 8) {
    if (crop) hipLaunchKernelGGL(crop_frames_kernel<uint8_t>, grid, dim3(256), 0, stream, (const uint8_t *)in, (uint8_t *)out, w, h, cw, ch);
    else hipLaunchKernelGGL(pad_frames_kernel<uint8_t>, grid, dim3(256), 0, stream, (const uint8_t *)in, (uint8_t *)out, w, h, cw, ch);
  } else {
    if (crop) hipLaunchKernelGGL(crop_frames_kernel<uint16_t>, grid, dim3(256), 0, stream, (const uint16_t *)in, (uint16_t *)out, w, h, cw, ch);
    else hipLaunchKernelGGL(pad_frames_kernel<uint16_t>, grid, dim3(256), 0, stream, (const uint16_t *)in, (uint16_t *)out, w, h, cw, ch);
  }
  return hipGetLastError();
}

// `frames`/`prev0` must be 16-byte aligned device pointers (checked by the caller); sad[] zeroed by the caller.
extern "C" hipError_t av1mi_launch_luma_sad(const Av1miDevParams *P, const void *frames, const void *prev0, unsigned long long *sad,
                                            hipStream_t stream) {
  const int bps = P->bit_depth > 8 ? 2 : 1;
  const size_t frame_bytes = (size_t)P->frame_samples * bps, luma_bytes = (size_t)P->width * P->height * bps;
  dim3 grid(128, P->n_frames);
  if (bps == 1)
    hipLaunchKernelGGL(luma_sad_kernel<1>, grid, dim3(256), 0, stream, (const uint8_t *)frames, (const uint8_t *)prev0, frame_bytes, luma_bytes, sad);
  else
    hipLaunchKernelGGL(luma_sad_kernel<2>, grid, dim3(256), 0, stream, (const uint8_t *)frames, (const uint8_t *)prev0, frame_bytes, luma_bytes, sad);
  return hipGetLastError();
}
