// recon_kernel.hip - closed-loop reconstruction: one tile (1 or 2x2 64x64 superblocks) per wavefront, every tile of every
// key frame of a chunk in one launch (recon_sb_kernel); inter frames one at a time, in two launches - every block as an
// inter block in parallel (recon_inter_pre_kernel), then the tile walk that decides and codes the intra winners - a workgroup of
// two waves per tile there: the luma wave takes the decisions and posts them through LDS, the chroma wave follows one behind.
//
// Replaces the per-superblock inner loop that the reference runs inside an external SVT-AV1
// worker (av1an -> SVT-AV1, reached through `run_av1an`,
// /root/reference/crates/daemon/src/encode/av1an.rs:126-139):
// directional/smooth/Paeth/DC intra prediction (AV1 spec §7.11.2) with angle deltas and, in the EXT instantiations, the intra
// edge filter / upsampling (§7.11.2.7-12) and chroma from luma (§7.11.5); forward DCT/ADST, dead-zone
// quantiser, normative dequantiser (§7.12.3) and inverse DCT/ADST (§7.13.3).
//
// MI355X mapping: one 64-lane wave owns one tile; the edges its blocks predict from
// live in LDS line buffers for the whole walk, so neighbour edges never touch
// HBM; source pixels are read once (coalesced rows); transforms run one row or
// column per lane on VGPR-resident straight-line butterflies (txfm_gen.h) with the 2-D
// transposition staged through a padded LDS tile (stride n+1: conflict-free for both row and
// column access) - except the forward transform of luma 32x32 blocks, an exact-integer matrix product on the matrix cores
// (eight v_mfma_i32_32x32x32_i8 on signed-byte halves, fdct32_matrix.h; measured against the butterflies by
// tools/mfma_fwd32_ab.hip); quantised levels are staged in LDS and leave as 16-byte-per-lane stores.
// The transform items are `noinline` functions: the library is built with -fno-optimize-sibling-calls so that they save no
// callee-saved registers (DESIGN.md §4.2 iv).
// Algorithmic HBM traffic per superblock: source read once, reconstruction written once, levels
// written once (SURVEY.md §8d "stage A").
#include <hip/hip_runtime.h>
#include <type_traits>
#include "av1mi_dev.h"
#include "av1_tables.h"
// This file is compiled twice: as is (leaf blocks up to 32x32: the kernels keep their 10 KB of LDS and 4 waves per SIMD), and
// through recon64_kernel.hip with AV1MI_RECON_BIG = 1 (block_log2 = 6: 64x64 luma blocks with the 64-point transform, 32x32
// chroma; 64x64 tiles of LDS per wave and the 64-point networks' registers, so 2 waves per SIMD).
#ifndef AV1MI_RECON_BIG
#define AV1MI_RECON_BIG 0
#endif
// ... and once per sample type: as is for 16-bit samples (10-bit video), through recon8_kernel.hip / recon64_8_kernel.hip with
// AV1MI_RECON_PIX8 = 1 for 8-bit samples - four translation units that compile side by side (one unit took 13 minutes).
#ifndef AV1MI_RECON_PIX8
#define AV1MI_RECON_PIX8 0
#endif
#define MAXN (AV1MI_RECON_BIG ? 64 : 32)   /* largest transform block side */
#define AV1_TXFM_FN static __device__ __forceinline__
// Round2(w0 * a + w1 * b, 12) of the butterfly rotations with 24-bit multiplies (v_mul_i32_i24 / v_mad_i32_i24: full rate; the
// generic form's two 64-bit multiply-adds run at a quarter of it and were 474 of a 32x32 block's 7.6 k instructions).  Exact
// whenever the operands have at most 24 bits and the sum fits 32: the weights have 13 bits (|w| <= 4096, w0^2 + w1^2 = 4096^2),
// the data stays below 2^19 - inverse transforms: the spec's conformance bound of 8 + BitDepth bits on every intermediate
// (§7.13.2.1), which the dequantiser's clamp enforces at the input; forward transforms: residual << 2 through a 32-point network
// (tools/gen_txfm.py's bound) - so |w0 * a + w1 * b| <= 4096 * sqrt(2) * 2^18.5 < 2^31.
#define AV1_HALF_BTF_DEFINED
static __device__ __forceinline__ int32_t av1_half_btf(int32_t w0, int32_t a, int32_t w1, int32_t b) {
  return (__mul24(w0, a) + __mul24(w1, b) + 2048) >> 12;
}
#define AV1_HALF_BTF1_DEFINED
static __device__ __forceinline__ int32_t av1_half_btf1(int32_t w, int32_t a) { return (__mul24(w, a) + 2048) >> 12; }
#include "txfm_gen.h"
#include "fdct32_matrix.h"
#include "intra_pieces.h"

namespace {

enum { DC_PRED, V_PRED, H_PRED, D45_PRED, D135_PRED, D113_PRED, D157_PRED, D203_PRED, D67_PRED,
       SMOOTH_PRED, SMOOTH_V_PRED, SMOOTH_H_PRED, PAETH_PRED };

__constant__ uint8_t c_sm_weights[4 + 8 + 16 + 32 + 64] = {
  255, 149, 85, 64,
  255, 197, 146, 105, 73, 50, 37, 32,
  255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16,
  255, 240, 225, 210, 196, 182, 169, 157, 145, 133, 122, 111, 101, 92, 83, 74, 66, 59, 52, 45, 39, 34, 29, 25, 21, 17, 14, 12, 10, 9, 8, 8,
  255, 248, 240, 233, 225, 218, 210, 203, 196, 189, 182, 176, 169, 163, 156, 150, 144, 138, 133, 127, 121, 116, 111, 106, 101, 96, 91, 86, 82, 77, 73, 69,
  65, 61, 57, 54, 50, 47, 44, 41, 38, 35, 32, 29, 27, 25, 22, 20, 18, 16, 15, 13, 12, 10, 9, 8, 7, 6, 6, 5, 5, 4, 4, 4 };
// Dr_Intra_Derivative indexed by angle/3 rounded down is not injective, so index by angle (values: intra_pieces.h)
__constant__ int c_dr_deriv[91] = AV1MI_DR_DERIV_INIT;   // (32-bit: the scalar unit has no 16-bit loads, and the angle is the same in every lane)
// floor(64 k / Dr_Intra_Derivative[angle]) for k = 1 .. 32 as (64 k * magic) >> 22 (the row where a 90 < angle < 180 prediction switches edges)
__constant__ uint32_t c_dr_magic[91] = AV1MI_DR_MAGIC_INIT;
// Mode_To_Txfm (spec §6.10.x): 0 DCT_DCT 1 ADST_DCT 2 DCT_ADST 3 ADST_ADST
// luma 32x32 forward transform on the matrix cores: per lane the operand fragments of fdct32_matrix.h (stage-1 B low / high bytes,
// stage-2 A low / high bytes)
__device__ const uint32_t c_fdct32_frag[64][16] = AV1_FDCT32_FRAG_INIT;

// LDS per superblock-wave (~9 KB, so ~4 waves fit a SIMD): decoder-style line buffers instead of the
// whole reconstructed superblock.  above[p][x] = bottom row of the last block reconstructed over column
// x, left[p][y] = right column of the last block reconstructed over row y, corner[p][yk][xk] = pixel
// (4*y4-1, 4*x4-1).  Z-order + quadtree alignment guarantee that whenever the decoded-block map says an
// edge is available these hold exactly the pixels spec §7.11.2 asks for.
// line buffers of a tile of TSB x TSB superblocks (coordinates tile-local); only the instantiations for two-superblock
// tiles reference - and therefore allocate - the larger set
template <int TSB>
struct LineLds {
  // (one row of 64 TSB luma + two of 32 TSB chroma samples each: with three rows of 64 the 32x32 build's wave held 8 348 B of LDS,
  // 156 more than lets 20 waves share a CU)
  uint16_t above_[128 * TSB], left_[128 * TSB];
  __device__ __forceinline__ uint16_t *above(int plane) { return above_ + (plane ? 32 * TSB * (plane + 1) : 0); }
  __device__ __forceinline__ uint16_t *left(int plane) { return left_ + (plane ? 32 * TSB * (plane + 1) : 0); }
  uint16_t corner[3][8 * TSB + 1][8 * TSB + 1];   // luma: every 8 samples (the smallest luma block), chroma: every 4
  // intra edge filter (get_filter_type, spec 7.11.2.8): [luma | chroma][8x8-luma unit of the tile] = the last block reconstructed over
  // that column / beside that row was predicted with a smooth mode
  uint8_t sm_above[2][8 * TSB], sm_left[2][8 * TSB];
};
__shared__ LineLds<1> g_lines1;
__shared__ LineLds<2> g_lines2;
template <int TSB> struct LinesSel;
template <> struct LinesSel<1> { static __device__ __forceinline__ LineLds<1> &get() { return g_lines1; } };
template <> struct LinesSel<2> { static __device__ __forceinline__ LineLds<2> &get() { return g_lines2; } };
#define LN (LinesSel<TSB>::get())
struct alignas(16) SbLds {
  uint16_t blkpix[MAXN * MAXN];     // prediction, then reconstruction, of the current transform block
  uint16_t srcblk[MAXN * MAXN];     // source pixels of the block; reused for the quantised levels
  // 2-D transform staging (every intermediate fits 16 bits).  A 64x64 block keeps 32 rows of 64 after its first pass,
  // and its residual is never stored (the column pass subtracts source and prediction itself): 32 x 65 for it, 2 x 32 x 33 for the
  // chroma pair of the 64x64 build - half of 64 x 65, which was what held that build at five waves per CU
  int16_t scratch[MAXN > 32 ? 2 * 32 * 33 : MAXN * (MAXN + 1)];
  // element i of a lane group's edge at [group * EDGS + 8 + i] (i >= -1; padded up to 3 N + 8 for the piece-wise predictors): element 0 is
  // 16-byte aligned.  Two groups of blocks up to MAXN / 2, or one block of MAXN.
#define AV1MI_EDGS (MAXN > 32 ? 120 : 72)
  uint16_t edge_a[2 * AV1MI_EDGS + 8];
  uint16_t edge_l[2 * AV1MI_EDGS + 8];
  uint32_t blkdec[2][19];       // luma, chroma (U and V decode together): row y + 1 of the 4x4-unit map, bit x + 1
  uint8_t smw[64];              // smooth weights of the current block size
  int eobs[4];                  // eob of the current block's Y, U, V transform blocks
};
__shared__ SbLds g_sb;
__shared__ SbLds g_sb_c;   // the chroma wave's block state in a split tile walk (referenced - and allocated - only there)
// WV: 0 = the wave does luma and chroma of a block in turn; a SPLIT tile walk is a workgroup of two waves - 1 = its luma wave
// (takes the decisions and posts them), 2 = its chroma wave (follows one decision behind, on LDS of its own)
template <int WV> struct SbSel { static __device__ __forceinline__ SbLds *get() { return &g_sb; } };
template <> struct SbSel<2> { static __device__ __forceinline__ SbLds *get() { return &g_sb_c; } };
#define S (SbSel<WV>::get())
// luma wave -> chroma wave, per leaf block (index = superblock-in-tile * 64 + Z index of the block's first 8x8 unit):
// dec = Q_VALID | the decision as tx_item returns it, posted the moment it is taken; e0 = Q_VALID | the luma eob, when the luma
// block is finished (the chroma wave writes the block-info entry, which holds both)
struct SplitQueue { int dec[4 * 64]; int e0[4 * 64]; };
__shared__ SplitQueue g_q;
// P-frame tile walk: everything a block of the superblock would fetch from HBM on the walk's serial chain, staged once per superblock
// with all loads in flight together (referenced - and allocated - by the inter instantiations only).  A decision used to pay four
// dependent HBM round trips (search key, provisional block info, source block, the inter version's bottom row / right column):
// ~ 4 of its ~ 7 us.  Samples as 16-bit whatever the bit depth; coordinates beyond the frame repeat the last row / column.
struct WalkStage {
  uint16_t src_y[64 * 64];        // source luma of the superblock
  uint16_t row_y[8][64];          // provisional reconstruction (every block as an inter block): luma rows 8 k + 7 ...
  uint16_t col_y[8][64];          // ... and columns 8 k + 7 - the bottom row / right column of any leaf is one of them
  uint16_t row_c[2][8][32];       // U, V: rows 4 k + 3
  uint16_t col_c[2][8][32];       //       columns 4 k + 3
  unsigned long long key[64];     // motion search result per 8x8 unit
  uint16_t pre_eob[64][4];        // the inter version's eobs per 8x8 unit (Y, U, V)
};
__shared__ __attribute__((aligned(16))) WalkStage g_ws;
#define Q_VALID 0x40000000
__device__ __forceinline__ int q_wait(const int *slot) {
  int v;
  while (!((v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) & Q_VALID)) __builtin_amdgcn_s_sleep(2);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return __builtin_amdgcn_readfirstlane(v) & (Q_VALID - 1);
}
__device__ __forceinline__ void q_post(int *slot, int v) {
  __hip_atomic_store(slot, v | Q_VALID, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Ordering between the lanes of ONE wave that exchange data through LDS: the wave's LDS operations complete in order, so
// waiting for them is all a "barrier" has to do.  (__syncthreads() would be a barrier of the whole workgroup, and the two waves
// of a split tile walk do not run the same number of them.)
// The fences name the LDS address space only: a generic workgroup fence also waits for every global store the wave has in flight
// (the levels and the reconstruction of the block just finished - about a microsecond each), which nothing here depends on.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- diagnostic build only (-DAV1MI_STAMPS, tools/stamp_recon.py): where a transform item spends its cycles.  Lane 0 adds the
// s_memtime difference of every phase to LDS sums per item class; the wave adds them to g_stamp_sum when it ends.  The
// product build compiles none of this.
#ifdef AV1MI_STAMPS
#define STAMP_CLASSES 6
#define STAMP_PHASES 8
__device__ unsigned long long g_stamp_sum[STAMP_CLASSES * STAMP_PHASES + 3];   // + wave cycles, waves, wave time in 100 MHz ticks
struct StampLds { unsigned long long last, acc[STAMP_CLASSES * STAMP_PHASES]; };
__shared__ StampLds g_stamp;
__device__ __forceinline__ void stamp_phase(int cls, int phase) {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0);   // everything the phase issued has landed
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
  if (threadIdx.x == 0) {
    if (phase >= 0) g_stamp.acc[cls * STAMP_PHASES + phase] += t - g_stamp.last;
    g_stamp.last = t;
  }
}
#define STAMP(phase) stamp_phase(NPL == 1 ? 5 - LOG2N : 7 - LOG2N, (phase))
#else
#define STAMP(phase) do { } while (0)
#endif
__constant__ int16_t c_subpel[2][16][8] = AV1_SUBPEL_FILTERS_INIT;  // EIGHTTAP, and its 4-tap form for 4-sample blocks

// Wave reductions by DPP row shifts and broadcasts (an addition or a maximum each) instead of butterfly shuffles: a shuffle is an LDS
// round trip, six of them in a row were ~400 cycles of pure latency per reduction, and a block pass makes five.
// Inclusive "scan" with OP along the wave: afterwards lane 31 holds the reduction of lanes 0-31 and lane 63 that of all 64.
#define AV1MI_DPP_SCAN(x_, OP)                                                                 \
  do {                                                                                         \
    x_ = OP(x_, __builtin_amdgcn_update_dpp(x_, x_, 0x111, 0xF, 0xF, false)); /* row_shr:1 */   \
    x_ = OP(x_, __builtin_amdgcn_update_dpp(x_, x_, 0x112, 0xF, 0xF, false)); /* row_shr:2 */   \
    x_ = OP(x_, __builtin_amdgcn_update_dpp(x_, x_, 0x114, 0xF, 0xF, false)); /* row_shr:4 */   \
    x_ = OP(x_, __builtin_amdgcn_update_dpp(x_, x_, 0x118, 0xF, 0xF, false)); /* row_shr:8 */   \
    x_ = OP(x_, __builtin_amdgcn_update_dpp(x_, x_, 0x142, 0xA, 0xF, false)); /* row_bcast:15 */\
  } while (0)
__device__ __forceinline__ int dpp_add(int a, int b) { return a + b; }
__device__ __forceinline__ int dpp_max(int a, int b) { return a > b ? a : b; }
// (the shifted-in value of lanes without a source is the lane's own value, `old` = x: harmless for a maximum; for a sum the lanes
// that matter - the last lane of every row - always have a source)
__device__ __forceinline__ int wave_sum(int v) {
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
  return __builtin_amdgcn_readlane(x, 63);
}
// sum inside each half of the wave (lanes 0-31 | 32-63), returned to every lane of the half
__device__ __forceinline__ int half_sum(int v, int lane) {
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
  const int lo = __builtin_amdgcn_readlane(x, 31), hi = __builtin_amdgcn_readlane(x, 63);
  return lane < 32 ? lo : hi;
}
__device__ __forceinline__ int wave_max(int v) {
  int x = v;
  AV1MI_DPP_SCAN(x, dpp_max);
  x = dpp_max(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xC, 0xF, false));
  return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ int half_max(int v, int lane) {
  int x = v;
  AV1MI_DPP_SCAN(x, dpp_max);
  const int lo = __builtin_amdgcn_readlane(x, 31), hi = __builtin_amdgcn_readlane(x, 63);
  return lane < 32 ? lo : hi;
}
// componentwise maximum of two packed pairs of unsigned 16-bit values (v_pk_max_u16)
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
  typedef unsigned short pm2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pm2, a), __builtin_bit_cast(pm2, b)));
}
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int rshift_round(int v, int s) { return s ? (v + (1 << (s - 1))) >> s : v; }
__device__ __forceinline__ int clamp_bits(int v, int bits) {
  int lo = -(1 << (bits - 1)), hi = (1 << (bits - 1)) - 1;
  return v < lo ? lo : (v > hi ? hi : v);
}

// ---- 1-D transforms (type: 0 DCT, 1 ADST, 3 identity) -----------------------------------------
// identity transform (spec 7.13.2.15; the forward one is the same scaling): x * sqrt(2), * 2, * 2 sqrt(2) for 4, 8, 16 points
template <int LOG2N>
__device__ __forceinline__ void ident1d(int32_t *x) {
#pragma unroll
  for (int i = 0; i < (1 << LOG2N); i++) x[i] = LOG2N == 2 ? (x[i] * 5793 + 2048) >> 12 : (LOG2N == 3 ? x[i] * 2 : (x[i] * 11586 + 2048) >> 12);
}
template <int LOG2N> struct Tx1d;
template <int V> struct NzTag { static constexpr int value = V; };
template <> struct Tx1d<2> {
  template <int NZ> static __device__ __forceinline__ void inv_nz(int32_t *x, int t) { inv(x, t); }
  static __device__ __forceinline__ void iadst(int32_t *x) {
    // spec §7.13.2.6 inverse ADST4 (sinpi 1321 2482 3344 3803)
    int x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    int s0 = 1321 * x0, s1 = 2482 * x0, s2 = 3344 * x1, s3 = 3803 * x2, s4 = 1321 * x2, s5 = 2482 * x3, s6 = 3803 * x3;
    int s7 = x0 - x2 + x3;
    s0 = s0 + s3; s1 = s1 - s4; s3 = s2; s2 = 3344 * s7;
    s0 = s0 + s5; s1 = s1 - s6;
    x[0] = (s0 + s3 + 2048) >> 12; x[1] = (s1 + s3 + 2048) >> 12; x[2] = (s2 + 2048) >> 12; x[3] = (s0 + s1 - s3 + 2048) >> 12;
  }
  static __device__ __forceinline__ void fadst(int32_t *x) {
    int x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    int s0 = 1321 * x0, s1 = 3803 * x0, s2 = 2482 * x1, s3 = 1321 * x1, s4 = 3344 * x2, s5 = 3803 * x3, s6 = 2482 * x3;
    int s7 = x0 + x1 - x3;
    int y0 = s0 + s2 + s5, y1 = 3344 * s7, y2 = s1 - s3 + s6, y3 = s4;
    x[0] = (y0 + y3 + 2048) >> 12; x[1] = (y1 + 2048) >> 12; x[2] = (y2 - y3 + 2048) >> 12; x[3] = (y2 - y0 + y3 + 2048) >> 12;
  }
  static __device__ __forceinline__ void fwd(int32_t *x, int t) { if (t == 3) ident1d<2>(x); else if (t) fadst(x); else av1_fdct4(x); }
  static __device__ __forceinline__ void inv(int32_t *x, int t) { if (t == 3) ident1d<2>(x); else if (t) iadst(x); else av1_idct4(x); }
};
// inv_nz<NZ>: the inverse transform of a vector whose elements NZ .. are zero (the caller has made them so) - the pruned networks of
// txfm_gen.h where one exists for the size, else the full one.  Most blocks' coefficients lie in a small low-frequency corner.
template <> struct Tx1d<3> {
  static __device__ __forceinline__ void fwd(int32_t *x, int t) { if (t == 3) ident1d<3>(x); else if (t) av1_fadst8(x); else av1_fdct8(x); }
  static __device__ __forceinline__ void inv(int32_t *x, int t) { if (t == 3) ident1d<3>(x); else if (t) av1_iadst8(x); else av1_idct8(x); }
  template <int NZ> static __device__ __forceinline__ void inv_nz(int32_t *x, int t) {
    if constexpr (NZ == 4) { if (t == 3) ident1d<3>(x); else if (t) av1_iadst8_nz4(x); else av1_idct8_nz4(x); } else inv(x, t);
  }
};
template <> struct Tx1d<4> {
  static __device__ __forceinline__ void fwd(int32_t *x, int t) { if (t == 3) ident1d<4>(x); else if (t) av1_fadst16(x); else av1_fdct16(x); }
  static __device__ __forceinline__ void inv(int32_t *x, int t) { if (t == 3) ident1d<4>(x); else if (t) av1_iadst16(x); else av1_idct16(x); }
  template <int NZ> static __device__ __forceinline__ void inv_nz(int32_t *x, int t) {
    if constexpr (NZ == 4) { if (t == 3) ident1d<4>(x); else if (t) av1_iadst16_nz4(x); else av1_idct16_nz4(x); }
    else if constexpr (NZ == 8) { if (t == 3) ident1d<4>(x); else if (t) av1_iadst16_nz8(x); else av1_idct16_nz8(x); }
    else inv(x, t);
  }
};
template <> struct Tx1d<5> {
  static __device__ __forceinline__ void fwd(int32_t *x, int) { av1_fdct32(x); }
  static __device__ __forceinline__ void inv(int32_t *x, int) { av1_idct32(x); }
  template <int NZ> static __device__ __forceinline__ void inv_nz(int32_t *x, int) {
    if constexpr (NZ == 4) av1_idct32_nz4(x); else if constexpr (NZ == 8) av1_idct32_nz8(x); else if constexpr (NZ == 16) av1_idct32_nz16(x); else av1_idct32(x);
  }
};
#if AV1MI_RECON_BIG
template <> struct Tx1d<6> {
  static __device__ __forceinline__ void fwd(int32_t *x, int) { av1_fdct64(x); }
  static __device__ __forceinline__ void inv(int32_t *x, int) { av1_idct64(x); }
  template <int NZ> static __device__ __forceinline__ void inv_nz(int32_t *x, int) {
    if constexpr (NZ == 8) av1_idct64_nz8(x); else if constexpr (NZ == 16) av1_idct64_nz16(x); else av1_idct64(x);
  }
};
#endif

// position of (row, col) in the default zig-zag scan of an n x n block:
// odd anti-diagonals run with increasing row, even ones with increasing column.
__device__ __forceinline__ int scan_index(int row, int col, int n) {
  int d = row + col;
  int before = d < n ? (d * (d + 1)) >> 1 : n * n - (((2 * n - 1 - d) * (2 * n - d)) >> 1);
  int lo = d - (n - 1) > 0 ? d - (n - 1) : 0;
  return before + ((d & 1) ? row - lo : col - lo);
}

// ---- intra edge filter helpers (spec 7.11.2.9 strength selection, 7.11.2.10 upsample selection; w = h = n)
__device__ __forceinline__ int ef_strength(int wh, int type, int delta) {
  const int d = delta < 0 ? -delta : delta;
  int s = 0;
  if (type == 0) {
    if (wh <= 8) { if (d >= 56) s = 1; }
    else if (wh <= 16) { if (d >= 40) s = 1; }
    else if (wh <= 24) { if (d >= 8) s = 1; if (d >= 16) s = 2; if (d >= 32) s = 3; }
    else if (wh <= 32) { if (d >= 1) s = 1; if (d >= 4) s = 2; if (d >= 32) s = 3; }
    else { if (d >= 1) s = 3; }
  } else {
    if (wh <= 8) { if (d >= 40) s = 1; if (d >= 64) s = 2; }
    else if (wh <= 16) { if (d >= 20) s = 1; if (d >= 48) s = 2; }
    else if (wh <= 24) { if (d >= 4) s = 3; }
    else { if (d >= 1) s = 3; }
  }
  return s;
}
__device__ __forceinline__ int ef_use_upsample(int wh, int type, int delta) {
  const int d = delta < 0 ? -delta : delta;
  if (d <= 0 || d >= 40) return 0;
  return type ? wh <= 8 : wh <= 16;
}
// one filtered edge element (7.11.2.12): R[k] = raw element k (k >= 0), `corner` = element -1 (after the corner filter); i = the element
// (-1 .. ), sz = number of elements the filter covers counted from element -1
__device__ __forceinline__ int ef_element(const uint16_t *R, int corner, int i, int sz, int strength) {
  const int ii = i + 1;
  if (i < 0) return corner;
  if (strength == 0 || ii >= sz) return R[i];
  const int k0 = strength == 3 ? 2 : 0, k1 = strength == 2 ? 5 : 4, k2 = strength == 1 ? 8 : (strength == 2 ? 6 : 4);
  auto E = [&](int k) { k = k < 0 ? 0 : (k > sz - 1 ? sz - 1 : k); return k == 0 ? corner : (int)R[k - 1]; };
  return (k0 * (E(ii - 2) + E(ii + 2)) + k1 * (E(ii - 1) + E(ii + 1)) + k2 * E(ii) + 8) >> 4;
}

// ---- intra prediction of one pixel (spec §7.11.2) -------------------
// A / L = the above / left edge with index -1 valid (-2 when upsampled): the block's raw edges, or for a directional mode under
// enable_intra_edge_filter the filtered (and, up_a / up_l, upsampled) copies; dx/dy = Dr_Intra_Derivative values.
template <int LOG2N, int WV>
__device__ __forceinline__ int pred_pixel(int mode, int r, int c, int dcval, int ang, int dx, int dy, const uint16_t *A, const uint16_t *L,
                                          int up_a, int up_l) {
  constexpr int N = 1 << LOG2N;
  switch (mode) {
    case DC_PRED: return dcval;
    case PAETH_PRED: {
      int tl = A[-1], t = A[c], l = L[r];
      int base = t + l - tl;
      int pl = iabs(base - l), pt = iabs(base - t), ptl = iabs(base - tl);
      return (pl <= pt && pl <= ptl) ? l : (pt <= ptl ? t : tl);
    }
    case SMOOTH_PRED: {
      int wr = S->smw[r], wc = S->smw[c];
      return (wr * A[c] + (256 - wr) * L[N - 1] + wc * L[r] + (256 - wc) * A[N - 1] + 256) >> 9;
    }
    case SMOOTH_V_PRED: {
      int wr = S->smw[r];
      return (wr * A[c] + (256 - wr) * L[N - 1] + 128) >> 8;
    }
    case SMOOTH_H_PRED: {
      int wc = S->smw[c];
      return (wc * L[r] + (256 - wc) * A[N - 1] + 128) >> 8;
    }
    default: {   // V_PRED .. D67_PRED at the angle `ang` = the mode's base angle + 3 * angle delta
      if (ang == 90) return A[c];
      if (ang == 180) return L[r];
      if (ang < 90) {
        int idx = (r + 1) * dx;
        int base = (idx >> (6 - up_a)) + (c << up_a), sh = ((idx << up_a) >> 1) & 31;
        const int max_base = (2 * N - 1) << up_a;
        if (base < max_base) return (A[base] * (32 - sh) + A[base + 1] * sh + 16) >> 5;
        return A[max_base];
      } else if (ang < 180) {
        int idx = (c << 6) - (r + 1) * dx;
        int base = idx >> (6 - up_a);
        if (base >= -(1 << up_a)) {
          int sh = ((idx << up_a) >> 1) & 31;
          return (A[base] * (32 - sh) + A[base + 1] * sh + 16) >> 5;
        }
        idx = (r << 6) - (c + 1) * dy;
        base = idx >> (6 - up_l);
        int sh = ((idx << up_l) >> 1) & 31;
        return (L[base] * (32 - sh) + L[base + 1] * sh + 16) >> 5;
      } else {
        int idx = (c + 1) * dy;
        int base = (idx >> (6 - up_l)) + (r << up_l), sh = ((idx << up_l) >> 1) & 31;
        return (L[base] * (32 - sh) + L[base + 1] * sh + 16) >> 5;
      }
    }
  }
}

struct SbCtx {
  const Av1miDevParams *P;
  int lane;
  int sb_x, sb_y;          // superblock origin in luma pixels
  int tox, toy;            // the same relative to the tile origin (0 or 64): line buffers are tile-local
};

// Inter decision of the current block (inter frames): in = the motion search's result for the block, out (luma
// pass) = whether motion compensation won; the chroma pass follows it.
struct InterInfo {
  const void *ref;         // previous frame's final reconstruction (frame base), nullptr on key frames
  int mv_row, mv_col;      // 1/8 luma samples (multiples of 8)
  int sad_inter;           // luma SAD of that vector
  int is_inter;
  int pre_eob[3];          // PH == 2: the eobs of the block's inter version (recon_inter_pre_kernel computed it already)
};

// Motion-compensated sample (spec §7.11.3.4, unscaled reference, BILINEAR filter, not compound): position in 1/16
// plane samples = (coordinate << 4) + mv_q4; rounding InterRound0 = 3, InterRound1 = 11; reference coordinates are
// clamped to the plane.  Integer positions (every luma sample of this build) reduce to a copy.
template <typename PIX>
__device__ __forceinline__ int mc_sample(const PIX *plane, int stride, int last_x, int last_y, int px, int py, int maxv) {
  const int ix = px >> 4, fx = px & 15, iy = py >> 4, fy = py & 15;
  const int x0 = ix < 0 ? 0 : (ix > last_x ? last_x : ix), y0 = iy < 0 ? 0 : (iy > last_y ? last_y : iy);
  const int a = plane[(size_t)y0 * stride + x0];
  if ((fx | fy) == 0) return a;
  const int x1 = ix + 1 < 0 ? 0 : (ix + 1 > last_x ? last_x : ix + 1), y1 = iy + 1 < 0 ? 0 : (iy + 1 > last_y ? last_y : iy + 1);
  const int b = plane[(size_t)y0 * stride + x1], c = plane[(size_t)y1 * stride + x0], d = plane[(size_t)y1 * stride + x1];
  const int h0 = ((128 - 8 * fx) * a + 8 * fx * b + 4) >> 3, h1 = ((128 - 8 * fx) * c + 8 * fx * d + 4) >> 3;
  const int v = ((128 - 8 * fy) * h0 + 8 * fy * h1 + 1024) >> 11;
  return v < 0 ? 0 : (v > maxv ? maxv : v);
}

// Motion-compensated block at a sub-sample position with the EIGHTTAP filter (subpel = 1; spec §7.11.3.4, unscaled
// reference, not compound): the (N + 7)^2 reference window (coordinates clamped to the signalled frame) is staged in
// LDS once, horizontal pass -> Round2 by 3 -> 16-bit intermediate, vertical pass -> Round2 by 11 -> clamp.  Writes the
// prediction of lane group `grp` into dst[0 .. N*N).  px0 / py0: position of the block's first sample in 1/16 samples.
template <typename PIX, int LOG2N, int NPL>
__device__ __forceinline__ void mc_block_8tap(const PIX *rp, int stride, int last_x, int last_y, int px0, int py0, int maxv,
                                              int grp, int sl, uint16_t *buf, uint16_t *dst) {
  // In strips of 16 rows: the window of a strip ((16 + 7) x (N + 7) samples) and its horizontal-pass output fit the block's source
  // and staging tiles, which are free while the block is predicted BEFORE its source is loaded - the kernel needs no LDS of its own
  // for the window (5.6 KB: 12 instead of 16 waves per CU in the inter pass).
  constexpr int N = 1 << LOG2N, WN = N + 7, G = 64 / NPL, RS = N < 16 ? N : 16, WR = RS + 7;
  constexpr int WSZ = (WR * WN + 3) & ~3, MSZ = WR * N;
  static_assert(NPL * (WSZ + MSZ) <= MAXN * MAXN + (MAXN > 32 ? 2 * 32 * 33 : MAXN * (MAXN + 1)), "the strip's window and intermediate must fit srcblk + scratch");
  uint16_t *win = buf + grp * (WSZ + MSZ);
  int16_t *mid = reinterpret_cast<int16_t *>(win + WSZ);
  const int ix0 = (px0 >> 4) - 3, iy0 = (py0 >> 4) - 3;
  const int16_t *fh = c_subpel[N <= 4][px0 & 15], *fv = c_subpel[N <= 4][py0 & 15];
#pragma nounroll
  for (int s0 = 0; s0 < N; s0 += RS) {
    {  // all loads of the window in flight together (a loop of dependent load -> LDS store pairs paid the latency every time)
      constexpr int K = (WR * WN + G - 1) / G;
      PIX v[K];
#pragma unroll
      for (int k = 0; k < K; k++) {
        const int p = sl + k * G, i = p / WN, j = p - i * WN;
        int yy = iy0 + s0 + i, xx = ix0 + j;
        yy = yy < 0 ? 0 : (yy > last_y ? last_y : yy);
        xx = xx < 0 ? 0 : (xx > last_x ? last_x : xx);
        v[k] = rp[(size_t)yy * stride + xx];
      }
#pragma unroll
      for (int k = 0; k < K; k++) { const int p = sl + k * G; if (p < WR * WN) win[p] = (uint16_t)v[k]; }
    }
    wave_sync();
    for (int p = sl; p < WR * N; p += G) {
      const int r = p >> LOG2N, c = p & (N - 1);
      const uint16_t *wp = win + r * WN + c;
      int sum = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) sum += fh[t] * (int)wp[t];
      mid[p] = (int16_t)((sum + 4) >> 3);
    }
    wave_sync();
    for (int p = sl; p < RS * N; p += G) {
      const int r = p >> LOG2N, c = p & (N - 1);
      int sum = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) sum += fv[t] * (int)mid[(r + t) * N + c];
      const int v = (sum + 1024) >> 11;
      dst[s0 * N + p] = (uint16_t)(v < 0 ? 0 : (v > maxv ? maxv : v));
    }
    wave_sync();
  }
}

// One transform block per lane GROUP.  NPL = 1: the whole wave works on one block of `plane0` (luma,
// with the mode decision).  NPL = 2: lanes 0-31 work on the U block and lanes 32-63 on the V block of
// the same position at the same time (same mode, independent data) - chroma transforms are at most
// 16 wide, so this doubles the lanes that do useful work.  Steps: [luma: mode decision by closed-loop
// SAD, DESIGN.md §3.3] -> prediction -> forward transform -> dead-zone quantiser -> normative
// dequantiser + inverse transform -> reconstruction (HBM + line buffers).
// `mode_io`: in = mode to use (chroma), out = decided mode (luma).  eob_out[g] = eob of group g.
// PH = 0: the whole item.  Inter frames are reconstructed in two launches (recon_inter_pre_kernel, then recon_sb_kernel):
// PH = 1 codes the block as an inter block, no neighbours involved (no edges, no mode decision, no line buffers);
// PH = 2 takes the intra/inter decision with the neighbours in place and, when motion compensation wins, only moves the
// finished block's edges from HBM into the line buffers - the block is coded here only if intra prediction wins.
// Everything comes BY VALUE and the decision goes back in the return value ((is_inter << 8) | mode): a structure passed by
// reference to a `noinline` function lives in scratch memory, and every field access was a per-lane scratch load (the P-frame
// tile walk carried 348 B of scratch per lane for it).  The wave-uniform arguments are laundered through readfirstlane so that
// the compiler keeps them - and everything derived from them: addresses, edge availability, loop bounds - on the scalar unit.
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// (the pointers are rebuilt as pointers into GLOBAL memory: a pointer that arrives through a `noinline` function's arguments is a
// generic one, its accesses FLAT instructions - slower, and counted on the LDS counter too, so that a wait for an LDS read also waits
// for every global store in flight)
template <typename T>
__device__ __forceinline__ T *uniform_p(T *p) {
  typedef T __attribute__((address_space(1))) *G;
  const unsigned long long a = (unsigned long long)p;
  return (T *)(G)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a));
}
// (a pointer into LDS: the low 32 bits of its generic form are the LDS offset)
template <typename T>
__device__ __forceinline__ T *uniform_lds_p(T *p) {
  typedef T __attribute__((address_space(3))) *L;
  return (T *)(L)(unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)p);
}
// the quantiser-matrix table (its address is a field of the parameter block): a pointer into global memory
typedef const Av1miQmEntry __attribute__((address_space(1))) *QmTab;
__device__ __forceinline__ QmTab qm_table(const Av1miDevParams *P) { return (QmTab)(unsigned long long)P->qm_tab; }
// the parameter block: constant address space - its fields become scalar loads through the constant cache instead of per-lane FLAT
// loads the wave waits for on the spot (nothing writes the block while a kernel runs)
__device__ __forceinline__ const Av1miDevParams *uniform_params(const Av1miDevParams *p) {
  typedef const Av1miDevParams __attribute__((address_space(4))) *C;
  const unsigned long long a = (unsigned long long)p;
  return (const Av1miDevParams *)(C)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a));
}
// EXT: the instantiation contains the optional intra tools (edge filter / upsampling, chroma from luma); the default operating
// point runs the instantiations without them - their code alone cost 5 % (the items are larger than the instruction cache).
template <typename PIX, int LOG2N, int NPL, bool INTER, int TSB, bool QM, int PH, int WV, bool EXT>
__device__ __attribute__((noinline)) int tx_item(SbCtx cx, const PIX *frame, PIX *rec_frame, int plane0, int x0, int y0,
                                                 int mode_io, InterInfo ii, int16_t *lv_out0, int16_t *lv_out1, int *eob_out, int post_idx) {
  post_idx = uniform_i(post_idx);
  cx.P = uniform_params(cx.P); cx.sb_x = uniform_i(cx.sb_x); cx.sb_y = uniform_i(cx.sb_y); cx.tox = uniform_i(cx.tox); cx.toy = uniform_i(cx.toy);
  frame = uniform_p(frame); rec_frame = uniform_p(rec_frame); plane0 = uniform_i(plane0); x0 = uniform_i(x0); y0 = uniform_i(y0);
  mode_io = uniform_i(mode_io); lv_out0 = uniform_p(lv_out0); lv_out1 = uniform_p(lv_out1); eob_out = uniform_lds_p(eob_out);
  if (INTER) {
    ii.ref = uniform_p(ii.ref); ii.mv_row = uniform_i(ii.mv_row); ii.mv_col = uniform_i(ii.mv_col); ii.sad_inter = uniform_i(ii.sad_inter);
    ii.is_inter = uniform_i(ii.is_inter); ii.pre_eob[0] = uniform_i(ii.pre_eob[0]); ii.pre_eob[1] = uniform_i(ii.pre_eob[1]); ii.pre_eob[2] = uniform_i(ii.pre_eob[2]);
  }
  constexpr int N = 1 << LOG2N;
  constexpr int ST = N + 1;
  // MM: luma 32x32 blocks run the forward transform as a matrix product on the matrix cores: the residual rows are
  // then read with 128-bit loads, so they lie 32 apart (16-byte aligned) instead of 33
  constexpr bool MM = LOG2N == 5 && NPL == 1;
  // VEC: block classes whose DC / V / H prediction and residual are written eight samples per lane and step (128-bit LDS accesses):
  // luma 32x32 and the chroma 16x16 pair; their residual rows lie 32 / 24 apart (16-byte aligned)
  constexpr bool VEC = MM || (LOG2N == 4 && NPL == 2);
  constexpr int STR = MM ? 32 : (VEC ? 24 : ST);
  constexpr int G = 64 / NPL;              // lanes per group
  constexpr int PIXO = NPL == 1 ? 0 : (N > 16 ? 1024 : 512); // per-group offset inside srcblk / blkpix (NPL == 2: N <= 16, or 32 in the 64x64 build)
  constexpr int SCRO = NPL == 1 ? 0 : (N > 16 ? 32 * 33 : 16 * 24);   // (16 rows of stride 24: see STR)
  constexpr int EDGO = NPL == 1 ? 0 : AV1MI_EDGS;
  static_assert(NPL == 1 ? 8 + 3 * N + 9 <= 2 * AV1MI_EDGS + 8 : 8 + 3 * N + 9 <= AV1MI_EDGS, "padded edge does not fit its array");
  constexpr int EB = 8;   // index of element 0 inside a group's edge array
  // PVOK: block classes with the piece-wise intra predictors (eight samples of a row per lane and step; below)
  constexpr bool PVOK = N >= 8;
  const Av1miDevParams *P = cx.P;
  const int lane = cx.lane;
  const int grp = NPL == 1 ? 0 : lane >> 5, sl = NPL == 1 ? lane : lane & 31;
  const int plane = plane0 + grp;
  const int po = grp * PIXO, so = grp * SCRO, eo = grp * EDGO;
  const int bd = P->bit_depth;
  // decoded-block map lookups (spec §5.11.35): above-right / below-left availability
  const int pc = plane0 > 0;
  const int cgs = plane0 > 0 ? 2 : 3;   // log2 of the corner grid's pitch in this plane
  const int r4 = y0 >> 2, c4 = x0 >> 2;
  constexpr int step = (N >> 2) > 0 ? (N >> 2) : 1;
  const int have_ar = PH == 1 ? 0 : (int)((S->blkdec[pc][r4 - 1 + 1] >> (c4 + step + 1)) & 1u);
  const int have_bl = PH == 1 ? 0 : (int)((S->blkdec[pc][r4 + step + 1] >> (c4 - 1 + 1)) & 1u);
  const long poff = plane == 0 ? 0 : (plane == 1 ? P->plane_off_u : P->plane_off_v);
  const int gs = plane0 ? P->stride_c : P->stride_y;
  const int gx = (plane0 ? cx.sb_x >> 1 : cx.sb_x) + x0, gy = (plane0 ? cx.sb_y >> 1 : cx.sb_y) + y0;
  // tile-local plane coordinates of the block: what the line buffers and edge availability are indexed by
  const int lx = (plane0 ? cx.tox >> 1 : cx.tox) + x0, ly = (plane0 ? cx.toy >> 1 : cx.toy) + y0;
  const int have_above = ly > 0, have_left = lx > 0;
  // a block may overhang the right / bottom frame edge (by less than half its size): the source is read with its last column /
  // row replicated, the reconstruction is stored only inside the plane
  const int pw_lim = (plane0 ? P->width >> 1 : P->width) - gx, ph_lim = (plane0 ? P->height >> 1 : P->height) - gy;   // samples of the block inside
  const bool overhang = pw_lim < N || ph_lim < N;
  // PH == 2, motion compensation won: the block is already reconstructed in HBM - bottom row, right column and corners go to
  // the line buffers, the decoded-block map and the eob are set, nothing else happens
  auto inter_done = [&]() {
    if constexpr (PH == 2) {
      // the block's bottom row and right column come from the superblock's staged rows / columns (g_ws; the row index of a leaf's last
      // row inside the frame is 7 mod 8 - 3 mod 4 in chroma - whatever its size, block sizes and coded frame sizes being multiples of 8)
      const int rb = N - 1 < ph_lim ? N - 1 : ph_lim - 1, cb = N - 1 < pw_lim ? N - 1 : pw_lim - 1;   // last row / column inside
      const uint16_t *srow = plane == 0 ? g_ws.row_y[(y0 + rb) >> 3] + x0 : g_ws.row_c[plane - 1][(y0 + rb) >> 2] + x0;
      const uint16_t *scol = plane == 0 ? g_ws.col_y[(x0 + cb) >> 3] + y0 : g_ws.col_c[plane - 1][(x0 + cb) >> 2] + y0;
      if (sl < N) {
        LN.above(plane)[lx + sl] = srow[sl < pw_lim ? sl : pw_lim - 1];
        LN.left(plane)[ly + sl] = scol[sl < ph_lim ? sl : ph_lim - 1];
      }
      if (sl < (N >> cgs)) {
        const int j = sl + 1, q = (j << cgs) - 1;
        LN.corner[plane][(ly + N) >> cgs][(lx >> cgs) + j] = srow[q < pw_lim ? q : pw_lim - 1];
        LN.corner[plane][(ly >> cgs) + j][(lx + N) >> cgs] = scol[q < ph_lim ? q : ph_lim - 1];
      }
      if (lane < step) S->blkdec[pc][r4 + lane + 1] |= ((1u << step) - 1u) << (c4 + 1);   // (a lane per row of the block)
      if (EXT && lane < (plane0 ? N >> 2 : N >> 3)) { LN.sm_above[pc][(lx >> (plane0 ? 2 : 3)) + lane] = 0; LN.sm_left[pc][(ly >> (plane0 ? 2 : 3)) + lane] = 0; }
      if (sl == 0) eob_out[grp] = plane == 0 ? ii.pre_eob[0] : (plane == 1 ? ii.pre_eob[1] : ii.pre_eob[2]);   // (no dynamic index: the array stays in registers)
      wave_sync();
    }
  };
  if constexpr (PH == 2) {
    if (plane0 > 0 && ii.is_inter) { inter_done(); return (ii.is_inter << 8) | (mode_io & 0x7F); }
  }
  STAMP(-1);
  // ---- sub-sample motion compensation (PH == 1 items of the sub-sample kernels: EXT says so there) - first of all, while the source
  // and staging tiles are still free for its window
  bool mc_in_lds = false;  // the motion-compensated prediction already sits in blkpix
  if constexpr (INTER && PH == 1) {
    const int ss = plane0 > 0;
    const int px0 = (gx << 4) + ((2 * ii.mv_col) >> ss), py0 = (gy << 4) + ((2 * ii.mv_row) >> ss);
    const int last_x = ((P->true_w + ss) >> ss) - 1, last_y = ((P->true_h + ss) >> ss) - 1;
    if constexpr (EXT) {
      if (P->subpel && ((px0 | py0) & 15)) {   // sub-sample position: EIGHTTAP
        mc_block_8tap<PIX, LOG2N, NPL>(static_cast<const PIX *>(ii.ref) + poff, gs, last_x, last_y, px0, py0, (1 << bd) - 1, grp, sl, S->srcblk, S->blkpix + po);
        mc_in_lds = true;
      }
    }
    if constexpr (sizeof(PIX) == 2 && N >= 8) {
      // whole-sample position with the block inside the reference: the prediction is a copy - 16-byte pieces of the reference's rows
      // (any 2-byte alignment: the vector points anywhere) instead of one clamped 2-byte load per sample
      const int ix = px0 >> 4, iy = py0 >> 4;
      if (!mc_in_lds && uniform_i(((px0 | py0) & 15) == 0 && ix >= 0 && iy >= 0 && ix + N - 1 <= last_x && iy + N - 1 <= last_y)) {
        constexpr int CPR = N / 8;
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        typedef u4 u4_any __attribute__((aligned(2)));
        const PIX *rp = static_cast<const PIX *>(ii.ref) + poff;
        // (every load of the copy before its first LDS store - in one loop the compiler keeps source order, load -> wait -> store ->
        // load ...: a memory round trip per piece; unconditional, the index clamped: a load under a condition is waited for on the spot)
        constexpr int CNT = (N * CPR + G - 1) / G;
        u4 pv[CNT];
#pragma unroll
        for (int k = 0; k < CNT; k++) {
          const int q = sl + k * G < N * CPR ? sl + k * G : 0, r = q / CPR, c8 = q - r * CPR;
          pv[k] = *reinterpret_cast<const u4_any *>(rp + (size_t)(iy + r) * gs + ix + 8 * c8);
        }
#pragma unroll
        for (int k = 0; k < CNT; k++) {
          const int q = sl + k * G, r = q / CPR, c8 = q - r * CPR;
          if (q < N * CPR) *reinterpret_cast<u4 *>(&S->blkpix[po + r * N + 8 * c8]) = pv[k];
        }
        mc_in_lds = true;
      }
    }
  }
  // ---- source block -> LDS.  16-bit samples of a block inside the frame go as 16-byte pieces of a row (8 samples: two loads per
  // lane for a 32x32 block instead of sixteen 2-byte ones, full cache lines; the stamps build had the sixteen at 25 - 30 % of a
  // block pass); 8-bit samples, 4-wide blocks and blocks that overhang the frame edge (clamped coordinates) sample by sample.
  bool staged_src = false;
  if constexpr (PH == 2 && NPL == 1 && N >= 8) {
    if (plane0 == 0) {   // the walk's luma item: the superblock's source is staged (g_ws), edge samples already replicated
      typedef unsigned int u4 __attribute__((ext_vector_type(4)));
      constexpr int CPR = N / 8;
#pragma unroll
      for (int q = sl; q < N * CPR; q += G) {
        const int r = q / CPR, c8 = q - r * CPR;
        *reinterpret_cast<u4 *>(&S->srcblk[r * N + 8 * c8]) = *reinterpret_cast<const u4 *>(&g_ws.src_y[(y0 + r) * 64 + x0 + 8 * c8]);
      }
      staged_src = true;
    }
  }
  if (!staged_src) {
    const PIX *pl = frame + poff;
    bool wide = false;
    if constexpr (sizeof(PIX) == 2 && N >= 8) {
      wide = uniform_i(!overhang && ((gs | gx) & 7) == 0);
      if (wide) {
        constexpr int CPR = N / 8;   // 16-byte pieces per row
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        constexpr int CNT = (N * CPR + G - 1) / G;   // (all loads first: see the motion-compensated copy above)
        u4 sv[CNT];
#pragma unroll
        for (int k = 0; k < CNT; k++) {
          const int q = sl + k * G < N * CPR ? sl + k * G : 0, r = q / CPR, c8 = q - r * CPR;
          sv[k] = *reinterpret_cast<const u4 *>(pl + (size_t)(gy + r) * gs + gx + 8 * c8);
        }
#pragma unroll
        for (int k = 0; k < CNT; k++) {
          const int q = sl + k * G, r = q / CPR, c8 = q - r * CPR;
          if (q < N * CPR) *reinterpret_cast<u4 *>(&S->srcblk[po + r * N + 8 * c8]) = sv[k];
        }
      }
    }
    if constexpr (sizeof(PIX) == 1 && N >= 16) {   // 8-bit samples: 16 of them per 16-byte load, widened to the tile's 16-bit layout
      wide = uniform_i(!overhang && ((gs | gx) & 15) == 0);
      if (wide) {
        constexpr int CPR = N / 16;
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = sl; q < N * CPR; q += G) {
          const int r = q / CPR, c16 = q - r * CPR;
          const u4 v = *reinterpret_cast<const u4 *>(pl + (size_t)(gy + r) * gs + gx + 16 * c16);
          u4 lo, hi;
#pragma unroll
          for (int k = 0; k < 2; k++) {
            lo[2 * k] = __builtin_amdgcn_perm(0u, v[k], 0x0c010c00u); lo[2 * k + 1] = __builtin_amdgcn_perm(0u, v[k], 0x0c030c02u);
            hi[2 * k] = __builtin_amdgcn_perm(0u, v[2 + k], 0x0c010c00u); hi[2 * k + 1] = __builtin_amdgcn_perm(0u, v[2 + k], 0x0c030c02u);
          }
          *reinterpret_cast<u4 *>(&S->srcblk[po + r * N + 16 * c16]) = lo;
          *reinterpret_cast<u4 *>(&S->srcblk[po + r * N + 16 * c16 + 8]) = hi;
        }
      }
    }
    if (!wide) {
#pragma unroll   // all of the block's loads in flight together (N*N/G <= 16 per lane)
      for (int p = sl; p < N * N; p += G) {
        int r = p >> LOG2N, c = p & (N - 1);
        if (overhang) { r = r < ph_lim ? r : ph_lim - 1; c = c < pw_lim ? c : pw_lim - 1; }
        S->srcblk[po + p] = (uint16_t)pl[(size_t)(gy + r) * gs + gx + c];
      }
    }
  }
  STAMP(7);   // (diagnostic build: the source loads alone)
  // ---- edges from the line buffers (spec §7.11.2; tile == superblock: nothing outside it is available)
  int dcv = 0;
  if constexpr (PH != 1) {
  {
    const int max_x = ((plane0 ? P->width >> 1 : P->width) - 1) - (gx - lx);   // frame limit, tile-local
    const int max_y = ((plane0 ? P->height >> 1 : P->height) - 1) - (gy - ly);
    // (PVOK: the piece-wise predictors read up to element 3 N + 8 of an edge - the elements beyond 2 N - 1 repeat the last one, which is
    // what a prediction at an angle below 90 degrees takes beyond max_base)
    for (int i = sl; i < (PVOK ? 3 * N + 9 : 2 * N); i += G) {
      int a, l;
      if (!have_above && have_left) a = LN.left(plane)[ly];           // pixel (y0, x0-1)
      else if (!have_above) a = (1 << (bd - 1)) - 1;
      else {
        int lim = lx + (have_ar ? 2 * N : N) - 1;
        if (lim > max_x) lim = max_x;
        a = LN.above(plane)[lx + i < lim ? lx + i : lim];             // pixel (y0-1, .)
      }
      if (!have_left && have_above) l = LN.above(plane)[lx];          // pixel (y0-1, x0)
      else if (!have_left) l = (1 << (bd - 1)) + 1;
      else {
        int lim = ly + (have_bl ? 2 * N : N) - 1;
        if (lim > max_y) lim = max_y;
        l = LN.left(plane)[ly + i < lim ? ly + i : lim];              // pixel (., x0-1)
      }
      S->edge_a[eo + EB + i] = (uint16_t)a;
      S->edge_l[eo + EB + i] = (uint16_t)l;
    }
    if (sl == 0) {
      int tl;
      if (have_above && have_left) tl = LN.corner[plane][ly >> cgs][lx >> cgs];
      else if (have_above) tl = LN.above(plane)[lx];
      else if (have_left) tl = LN.left(plane)[ly];
      else tl = 1 << (bd - 1);
      S->edge_a[eo + EB - 1] = (uint16_t)tl;
      S->edge_l[eo + EB - 1] = (uint16_t)tl;
    }
    constexpr int WOFF = LOG2N == 2 ? 0 : (LOG2N == 3 ? 4 : (LOG2N == 4 ? 12 : (LOG2N == 5 ? 28 : 60)));
    // (only when a smooth mode can be asked for: a per-lane table load whose latency the following barrier would expose in every item;
    // chroma follows luma's mode, so the candidate mask covers both)
    if ((P->mode_mask & 0xE00u) && lane < N) S->smw[lane] = c_sm_weights[WOFF + lane];
  }
  wave_sync();
  STAMP(0);   // source -> LDS, edges from the line buffers
  // ---- DC value (sum within the lane group)
  {
    int s = 0;
    if (sl < N) s = (have_above ? S->edge_a[eo + EB + sl] : 0) + (have_left ? S->edge_l[eo + EB + sl] : 0);
    s = NPL == 1 ? wave_sum(s) : half_sum(s, lane);
    if (have_above && have_left) dcv = (s + N) >> (LOG2N + 1);
    else if (have_above || have_left) dcv = (s + (N >> 1)) >> LOG2N;
    else dcv = 1 << (bd - 1);
  }
  } else {
    wave_sync();
  }
  // ---- mode decision (luma) + final prediction: one loop, the last trip writes the prediction
  // mode_io: bits 0-3 the mode, bits 4-6 the angle delta + 3 (chroma passes follow the luma decision; luma passes decide)
  int best_mode = mode_io & 15, best_delta = ((mode_io >> 4) & 7) - 3, best_sad = 0x7FFFFFFF, sad_dc = -1;
  // direction parameters of a directional mode at an angle delta (§7.11.2.4)
  // (mode and delta are the same in every lane; saying so turns the table reads into scalar loads from the constant cache - as
  // per-lane loads each was a global round trip the wave waited for on the spot, two or three per candidate - and the mode's base
  // angle comes out of a 64-bit literal: 90, 180, 45, 135, 113, 157, 203, 67 for V_PRED .. D67_PRED)
  auto dir_params = [&](int mode, int delta, int &ang, int &dx, int &dy) {
    ang = 0; dx = 0; dy = 0;
    mode = uniform_i(mode); delta = uniform_i(delta);
    if (mode >= V_PRED && mode <= D67_PRED) {
      ang = (int)((0x43CB9D71872DB45Aull >> (8 * (mode - V_PRED))) & 255u) + 3 * delta;
      if (ang < 90) dx = c_dr_deriv[ang];
      else if (ang > 90 && ang < 180) { dx = c_dr_deriv[180 - ang]; dy = c_dr_deriv[ang - 90]; }
      else if (ang > 180) dy = c_dr_deriv[270 - ang];
    }
  };
  // enable_intra_edge_filter (spec 7.11.2.7 - 7.11.2.12): a directional candidate at an angle other than 90 / 180 predicts from a
  // filtered - and, for blocks up to 8x8, upsampled - copy of the edges.  The copy lives in the transform staging area (free
  // until the residual is written); element i of the above edge at FA[i], i = -2 .. , left edge at FL[i].
  int ef_type = 0;   // get_filter_type(): a neighbour predicted with a smooth mode
  if constexpr (PH != 1 && EXT) {
    if (P->edge_filter) {
      const int ux = lx >> (plane0 ? 2 : 3), uy = ly >> (plane0 ? 2 : 3);
      ef_type = uniform_i((have_above && LN.sm_above[pc][ux]) || (have_left && LN.sm_left[pc][uy]));
    }
  }
  constexpr int FEL = PVOK ? 3 * N + 16 : (N >= 64 ? 136 : 72);   // (PVOK: elements -2 .. 3 N + 8, see the padding of the raw edges)
  uint16_t *const FA = reinterpret_cast<uint16_t *>(S->scratch + so) + 2, *const FL = FA + FEL;
  auto dir_edges = [&](int ang, const uint16_t *&A, const uint16_t *&L, int &up_a, int &up_l) -> bool {
    A = S->edge_a + EB + eo; L = S->edge_l + EB + eo; up_a = 0; up_l = 0;
    if (PH == 1 || !EXT || !P->edge_filter || ang == 0 || ang == 90 || ang == 180) return false;
    const uint16_t *RA = A, *RL = L;
    const int n_top = N < pw_lim ? N : pw_lim, n_left = N < ph_lim ? N : ph_lim;
    int corner = RA[-1];
    if (ang > 90 && ang < 180 && 2 * N >= 24) corner = (RL[0] * 5 + corner * 6 + RA[0] * 5 + 8) >> 4;   // 7.11.2.7
    const int st_a = have_above ? ef_strength(2 * N, ef_type, ang - 90) : 0, st_l = have_left ? ef_strength(2 * N, ef_type, ang - 180) : 0;
    const int sz_a = n_top + (ang < 90 ? N : 0) + 1, sz_l = n_left + (ang > 180 ? N : 0) + 1;
    wave_sync();   // the previous candidate's reads of FA / FL are done
    for (int i = sl - 1; i < (PVOK ? 3 * N + 9 : 2 * N); i += G) {
      const int ic = i < 2 * N ? i : 2 * N - 1;   // (padding: the last element again, filtered as it is)
      FA[i] = (uint16_t)ef_element(RA, corner, ic, sz_a, st_a);
      FL[i] = (uint16_t)ef_element(RL, corner, ic, sz_l, st_l);
    }
    wave_sync();
    if constexpr (N <= 8) {   // 7.11.2.11: one lane per source element, at most 16 of them
      up_a = ef_use_upsample(2 * N, ef_type, ang - 90); up_l = ef_use_upsample(2 * N, ef_type, ang - 180);
      if (up_a | up_l) {
        const int maxv = (1 << bd) - 1;
        const int npa = N + (ang < 90 ? N : 0), npl = N + (ang > 180 ? N : 0);
        int va = 0, da = 0, vl = 0, dl = 0;
        auto up_one = [&](const uint16_t *buf, int i, int np, int &v, int &d2) {
          const int d0 = buf[i == 0 ? -1 : i - 2], d1 = buf[i - 1], d3 = buf[i + 1 >= np ? np - 1 : i + 1];
          d2 = buf[i];
          v = (-d0 + 9 * d1 + 9 * d2 - d3 + 8) >> 4;
          v = v < 0 ? 0 : (v > maxv ? maxv : v);
        };
        if (up_a && sl < npa) up_one(FA, sl, npa, va, da);
        if (up_l && sl < npl) up_one(FL, sl, npl, vl, dl);
        const int a_m1 = FA[-1], l_m1 = FL[-1];
        wave_sync();
        if (up_a && sl < npa) { FA[2 * sl - 1] = (uint16_t)va; FA[2 * sl] = (uint16_t)da; if (sl == 0) FA[-2] = (uint16_t)a_m1; }
        if (up_l && sl < npl) { FL[2 * sl - 1] = (uint16_t)vl; FL[2 * sl] = (uint16_t)dl; if (sl == 0) FL[-2] = (uint16_t)l_m1; }
        wave_sync();
      }
    }
    A = FA; L = FL;
    return true;
  };
  int first = (PH != 1 && NPL == 1 && plane0 == 0) ? 0 : 13;
  // ---- piece-wise intra predictors (PVOK; intra_pieces.h): eight samples of a row per lane and step, packed two to a register, instead
  // of one sample per lane and step through pred_pixel (whose switch, index arithmetic and dependent 2-byte LDS reads made a candidate
  // ~ 900 wave instructions on a 32x32 block: with all 13 candidates 89 % of a luma item).  The SADs of candidates that predict from the
  // left edge at an angle are taken in the transposed layout, against a transposed copy of the source kept where the prediction goes later.
  uint16_t *const TT = S->blkpix + po;
  bool have_tt = false;   // the transposed source tile exists (built when the first candidate needs it)
  // write == false: the lane's part of the candidate's SAD; write == true: the prediction goes to blkpix (the caller forms the residual)
  auto pv_run = [&](int mode, int ang, int dx, int dy, const uint16_t *EA, const uint16_t *EL, bool write) -> int {
    if constexpr (PVOK) {
      namespace pc = av1mi_pieces;
      mode = uniform_i(mode); ang = uniform_i(ang); dx = uniform_i(dx); dy = uniform_i(dy);
      const uint16_t *A0 = S->edge_a + EB + eo, *L0 = S->edge_l + EB + eo;
      const bool left_part = pc::is_dir(mode, ang) && ang > 90;
      if (left_part && !write && !have_tt) {
        pc::transpose_lane<N, G>(sl, S->srcblk + po, TT);
        wave_sync();
        have_tt = true;
      }
      const uint32_t magic = (left_part && ang < 180) ? c_dr_magic[180 - ang] : 0u;   // (dx = c_dr_deriv[180 - ang])
      uint32_t acc = pc::pass_t<N, G>(sl, mode, ang, dy, magic, EL, TT, S->blkpix + po, write);
      if (write && left_part && ang < 180) wave_sync();
      acc += pc::pass_n<N, G>(sl, mode, ang, dx, dcv, EA, A0, L0, S->smw, S->srcblk + po, S->blkpix + po, write);
      return (int)acc;
    } else {
      return 0;
    }
  };
  if (PH != 1 && NPL == 1 && plane0 == 0 && (PVOK ? (P->mode_mask & 0x7u) != 0 : P->mode_mask == 0x7u)) {
    // DC, V and H - the default candidate set - in one pass over the block instead of three
    int s_dc = 0, s_v = 0, s_h = 0;
    const uint16_t *A = S->edge_a + EB, *L = S->edge_l + EB;
    if constexpr (LOG2N == 5) {
      // 32x32: eight samples per lane and step - 128-bit LDS reads of the source row piece and of the above edge, v_sad_u16 on packed
      // pairs (sixteen steps of one sample each were 10 % of the block pass)
      typedef unsigned int u4 __attribute__((ext_vector_type(4)));
      const uint32_t dc2 = (uint32_t)dcv * 0x10001u;
      uint32_t a_dc = 0, a_v = 0, a_h = 0;
#pragma unroll
      for (int q = lane; q < 128; q += 64) {
        const int r = q >> 2, c8 = q & 3;
        const u4 sv = *reinterpret_cast<const u4 *>(&S->srcblk[r * 32 + 8 * c8]);
        const u4 av = *reinterpret_cast<const u4 *>(&A[8 * c8]);
        const uint32_t l2 = (uint32_t)L[r] * 0x10001u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          a_dc = __builtin_amdgcn_sad_u16(sv[k], dc2, a_dc);
          a_v = __builtin_amdgcn_sad_u16(sv[k], av[k], a_v);
          a_h = __builtin_amdgcn_sad_u16(sv[k], l2, a_h);
        }
      }
      s_dc = (int)a_dc; s_v = (int)a_v; s_h = (int)a_h;
    } else {
#pragma unroll 4
    for (int p = sl; p < N * N; p += G) {
      const int sv = S->srcblk[p];
      s_dc += iabs(sv - dcv);
      s_v += iabs(sv - (int)A[p & (N - 1)]);
      s_h += iabs(sv - (int)L[p >> LOG2N]);
    }
    }
    // candidates in mode order, first minimum wins; DC does not compete here (see the final trip)
    const unsigned mm = P->mode_mask;
    if (mm & 1u) sad_dc = wave_sum(s_dc);
    if (mm & 2u) { s_v = wave_sum(s_v); if (s_v < best_sad) { best_sad = s_v; best_mode = V_PRED; } }
    if (mm & 4u) { s_h = wave_sum(s_h); if (s_h < best_sad) { best_sad = s_h; best_mode = H_PRED; } }
    first = (mm & ~0x7u) ? 3 : 13;
  }
  STAMP(1);   // DC value + the default candidates' SADs
  // ---- chroma from luma (spec 7.11.5): key-frame blocks up to 32x32 luma.  The luma item left the block's
  // subsampled reconstruction minus its average (Q3) behind the two groups' staging areas; lanes 0-31 decide U's alpha and lanes
  // 32-63 V's at the same time: least-squares estimate, then the SAD of the estimate and its two neighbours.
  constexpr int CFL_ACO = 2 * 16 * 24;   // offset of that buffer in S->scratch (behind the two groups' staging areas of chroma blocks up to 16x16)
  int use_cfl = 0, cfl_alpha = 0;
  auto cfl_px = [&](int alpha, int ac) {
    const int sa = alpha * ac, rr = sa >= 0 ? (sa + 32) >> 6 : -((-sa + 32) >> 6);   // Round2Signed(alpha * ac, 6)
    const int v = dcv + rr, maxv = (1 << bd) - 1;
    return v < 0 ? 0 : (v > maxv ? maxv : v);
  };
  if constexpr (EXT && NPL == 2 && !INTER && PH == 0 && N <= 16) {
    if (P->cfl) {
      const int16_t *AC = S->scratch + CFL_ACO;
      auto group_sum = [&](int v) { return half_sum(v, lane); };   // (NPL == 2 here)
      int ang, dx, dy;
      dir_params(best_mode, best_delta, ang, dx, dy);
      const uint16_t *EA = nullptr, *EL = nullptr;
      int up_a = 0, up_l = 0;
      dir_edges(ang, EA, EL, up_a, up_l);
      int sad_reg = 0, num = 0, den = 0;
#pragma unroll 4
      for (int p = sl; p < N * N; p += G) {
        const int sv = S->srcblk[po + p];
        sad_reg += iabs(sv - pred_pixel<LOG2N, WV>(best_mode, p >> LOG2N, p & (N - 1), dcv, ang, dx, dy, EA, EL, up_a, up_l));
        const int a = (int)AC[p] >> 3, d = sv - dcv;
        num += a * d; den += a * a;
      }
      sad_reg = group_sum(sad_reg); num = group_sum(num); den = group_sum(den);
      int est = 0;
      if (den) {
        const long long t = 16ll * num + den, d2 = 2ll * den;   // round-half-up of 8 num / den
        est = (int)(t >= 0 ? t / d2 : -((-t + d2 - 1) / d2));
      }
      est = est < -16 ? -16 : (est > 16 ? 16 : est);
      int sad_cfl = 0x7FFFFFFF;
#pragma nounroll
      for (int k = 0; k < 3; k++) {
        int a = est + (k == 0 ? 0 : (k == 1 ? -1 : 1));
        a = a < -16 ? -16 : (a > 16 ? 16 : a);
        int sad = 0;
#pragma unroll 4
        for (int p = sl; p < N * N; p += G) sad += iabs((int)S->srcblk[po + p] - cfl_px(a, AC[p]));
        sad = group_sum(sad);
        if (sad < sad_cfl) { sad_cfl = sad; cfl_alpha = a; }
      }
      const int a_other = __shfl_xor(cfl_alpha, 32, 64);
      const int tot_cfl = sad_cfl + __shfl_xor(sad_cfl, 32, 64), tot_reg = sad_reg + __shfl_xor(sad_reg, 32, 64);
      use_cfl = uniform_i(((cfl_alpha | a_other) != 0) && tot_cfl + N < tot_reg);
      if (use_cfl) { best_mode = 13; best_delta = 0; }
    }
  }
#pragma nounroll
  for (int m = first; m <= 13; m++) {
    const bool final_trip = m == 13;
    if (PH != 1 && final_trip && NPL == 1 && plane0 == 0) {
      // DC is kept unless the best other candidate at least halves its SAD
      if (sad_dc >= 0 && (best_sad == 0x7FFFFFFF || 2 * (long)best_sad >= (long)sad_dc)) { best_mode = DC_PRED; best_sad = sad_dc; }
      // angle delta: a directional winner is refined over the deltas -1, +1, -2, +2, -3, +3; strictly smaller SAD wins
      best_delta = 0;
      if (P->angle_delta && best_mode >= V_PRED && best_mode <= D67_PRED) {
#pragma nounroll
        for (int k = 0; k < 6; k++) {
          const int delta = (k & 1) ? (k >> 1) + 1 : -((k >> 1) + 1);
          int ang, dx, dy;
          dir_params(best_mode, delta, ang, dx, dy);
          const uint16_t *EA, *EL;
          int up_a, up_l;
          dir_edges(ang, EA, EL, up_a, up_l);
          int sad = 0;
          if (PVOK && !(up_a | up_l)) sad = pv_run(best_mode, ang, dx, dy, EA, EL, false);
          else
#pragma unroll 4
          for (int p = sl; p < N * N; p += G)
            sad += iabs((int)S->srcblk[po + p] - pred_pixel<LOG2N, WV>(best_mode, p >> LOG2N, p & (N - 1), dcv, ang, dx, dy, EA, EL, up_a, up_l));
          sad = wave_sum(sad);
          if (sad < best_sad) { best_sad = sad; best_delta = delta; }
        }
      }
      // inter frames: motion compensation wins when its luma SAD is not larger
      if (INTER) ii.is_inter = ii.sad_inter <= best_sad;
      if constexpr (WV == 1) {   // split walk: the chroma wave starts on this block now
        if (lane == 0) q_post(&g_q.dec[post_idx], ((INTER ? ii.is_inter : 0) << 8) | ((best_delta + 3) << 4) | best_mode);
      }
      if constexpr (PH == 2) {
        if (ii.is_inter) { inter_done(); return (1 << 8) | (3 << 4) | best_mode; }
      }
    }
    const int mode = final_trip ? best_mode : m;
    if (!final_trip && !((P->mode_mask >> m) & 1)) continue;
    int ang, dx, dy;
    dir_params(mode, final_trip ? best_delta : 0, ang, dx, dy);
    const uint16_t *EA = nullptr, *EL = nullptr;
    int up_a = 0, up_l = 0;
    // (fe: the filtered edges occupy the staging area the residual goes to - it is then written in a pass of its own)
    const bool fe = ((INTER && final_trip && ii.is_inter) || use_cfl) ? false : dir_edges(ang, EA, EL, up_a, up_l);
    int sad = 0;
    bool vec_done = false;
    if constexpr (VEC) {
      // final trip of a luma 32x32 block / a chroma 16x16 pair predicted DC, V or H from the raw edges: prediction and residual eight
      // samples per lane and step (128-bit LDS accesses, packed 16-bit subtraction) instead of one
      if (final_trip && !(INTER && ii.is_inter) && !fe && !use_cfl && (mode == DC_PRED || ang == 90 || ang == 180)) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        typedef short s8 __attribute__((ext_vector_type(8)));
        const uint16_t *A0 = S->edge_a + EB + eo, *L0 = S->edge_l + EB + eo;
        const uint32_t dc2 = (uint32_t)dcv * 0x10001u;
        constexpr int CPR = N / 8;
#pragma unroll
        for (int q = sl; q < N * CPR; q += G) {
          const int r = q / CPR, c8 = q - r * CPR;
          const u4 sv = *reinterpret_cast<const u4 *>(&S->srcblk[po + r * N + 8 * c8]);
          u4 pv4;
          if (mode == DC_PRED) pv4 = (u4){ dc2, dc2, dc2, dc2 };
          else if (ang == 90) pv4 = *reinterpret_cast<const u4 *>(&A0[8 * c8]);
          else { const uint32_t l2 = (uint32_t)L0[r] * 0x10001u; pv4 = (u4){ l2, l2, l2, l2 }; }
          const u4 rs = __builtin_bit_cast(u4, (s8)(__builtin_bit_cast(s8, sv) - __builtin_bit_cast(s8, pv4)));   // eight 16-bit differences
          *reinterpret_cast<u4 *>(&S->blkpix[po + r * N + 8 * c8]) = pv4;
          *reinterpret_cast<u4 *>(&S->scratch[so + r * STR + 8 * c8]) = rs;
        }
        vec_done = true;
      }
    }
    // PVOK: an intra candidate's SAD, or the final intra prediction, by the piece-wise predictors (the residual then follows in a pass of
    // its own, as with filtered edges); upsampled edges (blocks up to 8x8 under the edge filter) keep the sample-by-sample path
    const bool pv_path = PVOK && !vec_done && !(INTER && final_trip && ii.is_inter) && !use_cfl && !(up_a | up_l);
    if (pv_path) {
      if (final_trip) wave_sync();   // (the candidates' reads of the transposed source tile, which the prediction overwrites)
      sad = pv_run(mode, ang, dx, dy, EA ? EA : S->edge_a + EB + eo, EL ? EL : S->edge_l + EB + eo, final_trip);
    }
    if (!vec_done && !pv_path)
#pragma unroll 4
    for (int p = sl; p < N * N; p += G) {
      const int r = p >> LOG2N, c = p & (N - 1);
      int pv;
      if (INTER && final_trip && ii.is_inter && mc_in_lds) {
        pv = S->blkpix[po + p];
      } else if (INTER && final_trip && ii.is_inter) {
        const int ss = plane0 > 0;
        const PIX *rp = static_cast<const PIX *>(ii.ref) + poff;
        // lastX / lastY of §7.11.3.3: the reference is clamped to the SIGNALLED frame size
        pv = mc_sample<PIX>(rp, gs, ((P->true_w + ss) >> ss) - 1, ((P->true_h + ss) >> ss) - 1,
                            ((gx + c) << 4) + ((2 * ii.mv_col) >> ss), ((gy + r) << 4) + ((2 * ii.mv_row) >> ss), (1 << bd) - 1);
      } else if (use_cfl) {
        pv = cfl_px(cfl_alpha, S->scratch[CFL_ACO + p]);
      } else {
        pv = pred_pixel<LOG2N, WV>(mode, r, c, dcv, ang, dx, dy, EA, EL, up_a, up_l);
      }
      const int sv = S->srcblk[po + p];
      if (final_trip) {
        S->blkpix[po + p] = (uint16_t)pv;
        if (!fe && LOG2N < 6) S->scratch[so + r * STR + c] = (int16_t)(sv - pv);
      } else {
        sad += iabs(sv - pv);
      }
    }
    if (final_trip && (fe || pv_path) && LOG2N < 6) {
      wave_sync();
      if constexpr (VEC) {   // eight differences per lane and step (the residual rows of these classes are 16-byte aligned)
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        typedef short s8 __attribute__((ext_vector_type(8)));
        constexpr int CPR = N / 8;
#pragma unroll
        for (int q = sl; q < N * CPR; q += G) {
          const int r = q / CPR, c8 = q - r * CPR;
          const u4 sv = *reinterpret_cast<const u4 *>(&S->srcblk[po + r * N + 8 * c8]), pv4 = *reinterpret_cast<const u4 *>(&S->blkpix[po + r * N + 8 * c8]);
          *reinterpret_cast<u4 *>(&S->scratch[so + r * STR + 8 * c8]) = __builtin_bit_cast(u4, (s8)(__builtin_bit_cast(s8, sv) - __builtin_bit_cast(s8, pv4)));
        }
      } else {
#pragma unroll 4
      for (int p = sl; p < N * N; p += G) S->scratch[so + (p >> LOG2N) * STR + (p & (N - 1))] = (int16_t)((int)S->srcblk[po + p] - (int)S->blkpix[po + p]);
      }
    }
    if (!final_trip) {
      sad = wave_sum(sad);
      if (m == DC_PRED) sad_dc = sad;
      else if (sad < best_sad) { best_sad = sad; best_mode = m; }
    }
  }
  wave_sync();
  STAMP(2);   // other candidates, decision, prediction + residual
  // ---- transform: fwd columns | fwd rows + quant + dequant + inv rows | inv columns
  // (Mode_To_Txfm, two bits per mode, out of a literal: the table read was a per-lane global load waited for on the spot)
  const int txt = (LOG2N <= 4 && !(INTER && ii.is_inter)) ? (int)((0x39DA724u >> (2 * best_mode)) & 3u) : 0;  // inter blocks: DCT_DCT
  // transform type search (tx_search; EXT instantiations): an intra luma block of up to 16x16 whose residual
  // is sparse - at most one sample in eight nonzero - takes the identity transform (IDTX) both ways
  int idtx = 0;
  if constexpr (EXT && NPL == 1 && LOG2N <= 4 && PH != 1) {
    if (plane0 == 0 && P->tx_search && !(INTER && ii.is_inter)) {
      int nz = 0;
      for (int p = sl; p < N * N; p += G) nz += S->scratch[so + (p >> LOG2N) * STR + (p & (N - 1))] != 0;
      idtx = uniform_i(wave_sum(nz) * 8 <= N * N);
    }
  }
  const int vt = idtx ? 3 : (txt & 1), ht = idtx ? 3 : ((txt >> 1) & 1);  // ADST_DCT(1): vertical ADST; DCT_ADST(2): horizontal ADST
  // 64-point transforms (64x64 build): only the 32x32 low-frequency corner is coded (CW = 32) - the column pass keeps its first
  // 32 outputs, the row pass runs on 32 lanes and keeps 32, the inverse passes take 32 inputs (zeros beyond) and give 64 outputs
  constexpr int CW = N > 32 ? 32 : N;
  constexpr int SH0 = LOG2N == 6 ? 0 : 2, SH1 = LOG2N == 2 ? 0 : (LOG2N == 3 ? 1 : (LOG2N == 4 ? 2 : (LOG2N == 5 ? 4 : 2))), SH2 = LOG2N == 6 ? 2 : 0;
  constexpr int RS = LOG2N == 2 ? 0 : (LOG2N == 3 ? 1 : 2);
  constexpr int TSH = LOG2N == 6 ? 2 : (LOG2N == 5 ? 1 : 0);  // dequant shift of the size class (§7.12.3)
  int32_t x[N];
  const bool tx_lane = sl < N, row_lane = sl < CW;
  int my_key = -1;  // (anti-diagonal << 6 | position inside it) of the last nonzero level in scan order
  uint32_t my_ext = 0;   // {last nonzero row + 1, last nonzero column + 1} of this lane's levels as packed 16-bit values (0: none)
  int16_t *lvl = reinterpret_cast<int16_t *>(S->srcblk) + po;  // source block is dead: reuse for the levels
  if constexpr (MM) {
    // ---- forward 32x32 DCT as Y = Cm * X * Cm^T on the matrix cores (v_mfma_i32_32x32x32_i8; tools/mfma_fwd32_ab.hip is the
    // measured A/B against the butterflies).  Values are split into signed bytes, v = 256 * hi + lo: four products per stage, the two
    // mixed ones into one accumulator.  Stage 1: A = X (lane = row r, its half h holds columns 16h .. 16h + 15), B = Cm^T; the result
    // U has its column on the lane and 16 of its rows in the accumulator registers, which is exactly the B operand of stage 2
    // (A = Cm with its k index in that register order) - no lane movement, no LDS between the stages.  The result again has the
    // column m (horizontal frequency) on the lane and the rows k = (reg & 3) + 8 * (reg >> 2) + 4 * h in 16 registers: the
    // quantiser runs on all 64 lanes, 16 coefficients each (the butterfly form: 32 lanes x 32).
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    typedef int v16i_t __attribute__((ext_vector_type(16)));
    const int mr = lane & 31, mh = lane >> 5;
    const v4i_t *fr = reinterpret_cast<const v4i_t *>(c_fdct32_frag[lane]);
    const v4i_t s1_lo = fr[0], s1_hi = fr[1], s2_lo = fr[2], s2_hi = fr[3];
    const v4i_t q0 = *reinterpret_cast<const v4i_t *>(&S->scratch[so + mr * 32 + 16 * mh]);
    const v4i_t q1 = *reinterpret_cast<const v4i_t *>(&S->scratch[so + mr * 32 + 16 * mh + 8]);
    v4i_t a_lo, a_hi;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const unsigned e0 = (unsigned)(i < 2 ? q0[2 * i] : q1[2 * i - 4]), e1 = (unsigned)(i < 2 ? q0[2 * i + 1] : q1[2 * i - 3]);
      a_lo[i] = (int)__builtin_amdgcn_perm(e1, e0, 0x06040200u);   // the low bytes of four residuals
      // (v + 128) >> 8 of each halfword: add 128 without carrying into the neighbour, take the high bytes
      const unsigned g0 = ((e0 & 0x7FFF7FFFu) + 0x00800080u) ^ (e0 & 0x80008000u);
      const unsigned g1 = ((e1 & 0x7FFF7FFFu) + 0x00800080u) ^ (e1 & 0x80008000u);
      a_hi[i] = (int)__builtin_amdgcn_perm(g1, g0, 0x07050301u);
    }
    v16i_t hh = {0}, mid = {0}, ll = {0};
    hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_hi, s1_hi, hh, 0, 0, 0);
    mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_hi, s1_lo, mid, 0, 0, 0);
    mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_lo, s1_hi, mid, 0, 0, 0);
    ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_lo, s1_lo, ll, 0, 0, 0);
    v4i_t b_lo = {0, 0, 0, 0}, b_hi = {0, 0, 0, 0};
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int u = ((hh[reg] << 16) + (mid[reg] << 8) + ll[reg] + 512) >> 10;
      b_lo[reg >> 2] |= (u & 255) << (8 * (reg & 3));
      b_hi[reg >> 2] |= (((u + 128) >> 8) & 255) << (8 * (reg & 3));
    }
    v16i_t hh2 = {0}, mid2 = {0}, ll2 = {0};
    hh2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(s2_hi, b_hi, hh2, 0, 0, 0);
    mid2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(s2_hi, b_lo, mid2, 0, 0, 0);
    mid2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(s2_lo, b_hi, mid2, 0, 0, 0);
    ll2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(s2_lo, b_lo, ll2, 0, 0, 0);
    STAMP(3);   // forward transform
    // ---- dead-zone quantiser of the lane's 16 coefficients (k, m): levels to LDS, the key of the last nonzero one in scan order
    {
      const int col = mr;
      const uint32_t acq = (uint32_t)P->ac_q, acr = P->ac_recip;
      const QmTab tab = QM ? qm_table(P) + (pc ? AV1MI_QM_PLANE : 0) + AV1MI_QM_32X32 + col : (QmTab)0;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * mh;
        const int v = ((hh2[reg] << 16) + (mid2[reg] << 8) + ll2[reg] + 2048) >> 12;
        uint32_t q, recip;
        if constexpr (QM) { q = tab[row * 32].q; recip = tab[row * 32].recip; }
        else if (reg == 0) { const bool dc = (row | col) == 0; q = dc ? (uint32_t)P->dc_q : acq; recip = dc ? P->dc_recip : acr; }
        else { q = acq; recip = acr; }
        const int d0 = row + col;
        const uint32_t rnd = d0 < 8 ? (3 * q) >> 3 : (d0 < 16 ? (q >> 2) : (q >> 3));
        const int sgn = v >> 31;
        const uint32_t a = ((uint32_t)((v ^ sgn) - sgn) << TSH) + rnd;
        uint32_t lv = __umulhi(a, recip);
        lv = lv > 0x7FFF ? 0x7FFF : lv;
        lvl[row * 32 + col] = (int16_t)(((int)lv ^ sgn) - sgn);
        if (lv) {
          const int key = (d0 << 6) | ((d0 & 1) ? row : col);
          my_key = key > my_key ? key : my_key;
          my_ext = ((uint32_t)(row + 1) << 16) | (uint32_t)(col + 1);   // (rows grow with reg)
        }
      }
    }
    wave_sync();
  } else {
    if (tx_lane) {
  #pragma unroll
      for (int i = 0; i < N; i++)   // (the residual's row stride; the passes' own tile is ST; 64x64: source minus prediction, straight from their tiles)
        x[i] = (LOG2N < 6 ? (int)S->scratch[so + i * STR + sl] : (int)S->srcblk[po + i * N + sl] - (int)S->blkpix[po + i * N + sl]) << SH0;
      Tx1d<LOG2N>::fwd(x, vt);
  #pragma unroll
      for (int i = 0; i < CW; i++) S->scratch[so + i * ST + sl] = (int16_t)rshift_round(x[i], SH1);
    }
    wave_sync();
    STAMP(3);   // forward columns
    // dead-zone quantiser + normative dequantiser (§7.12.3) of this lane's coefficient row.  QM: the step of every position
    // comes from the context's quantiser-matrix table {Round2(q * Quantizer_Matrix, 5), ceil(2^32 / that)}; the matrices are
    // symmetric, so lanes read entry [j][row] (consecutive addresses across the wave).
    if (row_lane) {
  #pragma unroll
      for (int j = 0; j < N; j++) x[j] = S->scratch[so + sl * ST + j];
      Tx1d<LOG2N>::fwd(x, ht);
      const int row = sl;
      if constexpr (!QM) {
        // One step for every AC coefficient, another for DC: the dead-zone class of a coefficient is two compares against per-lane
        // thresholds (no divergent three-way branch), signs go through the sign mask, the dequantiser's product fits a 24-bit
        // multiply (level < 2^15, step < 2^15), and of the scan key only the last nonzero column is tracked (the key grows with the
        // column inside a row).  The dequantiser stays behind `if (lv)`: a column with no level in any row costs the wave nothing.
        const uint32_t acq = (uint32_t)P->ac_q, acr = P->ac_recip;
        const uint32_t r0 = (3 * acq) >> 3, r1 = acq >> 2, r2 = acq >> 3;
        const int ta = (CW >> 2) - row, tb = (CW >> 1) - row;   // column j is in dead-zone class 0 below ta, 1 below tb, else 2
        const int lim = 1 << (7 + bd);
        int lastj = -1;
  #pragma unroll
        for (int j = 0; j < CW; j++) {
          const int v = rshift_round(x[j], SH2);
          uint32_t q = acq, recip = acr, rnd = j < ta ? r0 : (j < tb ? r1 : r2);
          if (j == 0) {
            const bool dc = row == 0;
            q = dc ? (uint32_t)P->dc_q : acq; recip = dc ? P->dc_recip : acr;
            rnd = 0 < ta ? (3 * q) >> 3 : (0 < tb ? (q >> 2) : (q >> 3));
          }
          const int sgn = v >> 31;
          const uint32_t a = ((uint32_t)((v ^ sgn) - sgn) << TSH) + rnd;
          uint32_t lv = __umulhi(a, recip);
          lv = lv > 0x7FFF ? 0x7FFF : lv;
          lvl[row * CW + j] = (int16_t)(((int)lv ^ sgn) - sgn);
          int d = 0;
          if (lv) {   // (most columns beyond the first few hold no level in any row: the wave skips the block)
            lastj = j;
            d = (int)((__umul24(lv, q) & 0xFFFFFF) >> TSH);
            d = (d ^ sgn) - sgn;
            d = d < -lim ? -lim : (d > lim - 1 ? lim - 1 : d);
          }
          x[j] = d;
        }
        if (lastj >= 0) { const int d0 = row + lastj; my_key = (d0 << 6) | ((d0 & 1) ? row : lastj); my_ext = ((uint32_t)(row + 1) << 16) | (uint32_t)(lastj + 1); }
      } else {
      constexpr int QM_OFF = LOG2N == 2 ? AV1MI_QM_4X4 : (LOG2N == 3 ? AV1MI_QM_8X8 : (LOG2N == 4 ? AV1MI_QM_16X16 : AV1MI_QM_32X32));
      const QmTab tab = qm_table(P) + (pc ? AV1MI_QM_PLANE : 0) + QM_OFF + row;
      // (the flat quantiser's four values as scalars first: selecting between a field of the parameter block and a table entry made the
      // compiler select between two ADDRESSES in different address spaces - FLAT loads)
      const uint32_t dcq = (uint32_t)P->dc_q, acq = (uint32_t)P->ac_q, dcr = P->dc_recip, acr = P->ac_recip;
  #pragma unroll
      for (int j = 0; j < CW; j++) {
        const int v = rshift_round(x[j], SH2);
        uint32_t q = tab[j * CW].q, recip = tab[j * CW].recip;
        if (idtx) {   // the matrices apply to the 2-D DCT / ADST types only (spec 7.12.3: PlaneTxType < IDTX)
          const bool dc = (row | j) == 0;
          q = dc ? dcq : acq; recip = dc ? dcr : acr;
        }
        // frequency-dependent dead zone: 3q/8 for row+col < n/4, q/4 below n/2, q/8 above (n = the coded width)
        const int d0 = row + j;
        const uint32_t rnd = d0 < (CW >> 2) ? (3 * q) >> 3 : (d0 < (CW >> 1) ? (q >> 2) : (q >> 3));
        const uint32_t a = ((uint32_t)iabs(v) << TSH) + rnd;
        uint32_t lv = __umulhi(a, recip);
        if (lv > 0x7FFF) lv = 0x7FFF;
        lvl[row * CW + j] = (int16_t)(v < 0 ? -(int)lv : (int)lv);
        int d = 0;
        if (lv) {
          // scan order: by anti-diagonal, odd ones by increasing row, even ones by increasing column
          const int key = (d0 << 6) | ((d0 & 1) ? row : j);
          my_key = key > my_key ? key : my_key;
          my_ext = ((uint32_t)(row + 1) << 16) | (uint32_t)(j + 1);   // (columns grow with j)
          d = (int)(((uint32_t)lv * q) & 0xFFFFFF) >> TSH;
          const int lim = 1 << (7 + bd);
          d = v < 0 ? -d : d;
          d = d < -lim ? -lim : (d > lim - 1 ? lim - 1 : d);
        }
        x[j] = d;
      }
      }
  #pragma unroll
      for (int j = CW; j < N; j++) x[j] = 0;
    }
  }
  my_key = NPL == 1 ? wave_max(my_key) : half_max(my_key, lane);
  int eob = 0;
  if (my_key >= 0) {
    const int d0 = my_key >> 6, w = my_key & 63;
    eob = scan_index((d0 & 1) ? w : d0 - w, (d0 & 1) ? d0 - w : w, CW) + 1;
  }
  // ---- the block's nonzero extent: rows / columns beyond it hold no level, so the inverse passes run the networks pruned for that many
  // inputs (txfm_gen.h: a rotation with a zero input is one multiply, sums with a zero are copies) and, in the matrix-core path, only
  // that many levels of a row are dequantised - most blocks' levels lie in a small low-frequency corner.  One class for the whole wave
  // (both planes of a chroma pair): ext = {columns, rows} as packed 16-bit maxima.
  int nzc, nzr;
  {
    uint32_t ext = my_ext;
    ext = pk_max(ext, (uint32_t)__builtin_amdgcn_update_dpp((int)ext, (int)ext, 0x111, 0xF, 0xF, false));
    ext = pk_max(ext, (uint32_t)__builtin_amdgcn_update_dpp((int)ext, (int)ext, 0x112, 0xF, 0xF, false));
    ext = pk_max(ext, (uint32_t)__builtin_amdgcn_update_dpp((int)ext, (int)ext, 0x114, 0xF, 0xF, false));
    ext = pk_max(ext, (uint32_t)__builtin_amdgcn_update_dpp((int)ext, (int)ext, 0x118, 0xF, 0xF, false));
    ext = pk_max(ext, (uint32_t)__builtin_amdgcn_update_dpp((int)ext, (int)ext, 0x142, 0xA, 0xF, false));
    ext = pk_max(ext, (uint32_t)__builtin_amdgcn_update_dpp((int)ext, (int)ext, 0x143, 0xC, 0xF, false));
    ext = (uint32_t)__builtin_amdgcn_readlane((int)ext, 63);
    const int ec = (int)(ext & 0xFFFF), er = (int)(ext >> 16);
    // the classes this size has pruned networks for (64 = the full network)
    auto cls = [](int v) { return LOG2N >= 5 ? (v <= 8 ? 8 : (v <= 16 ? 16 : 64)) : (LOG2N == 4 ? (v <= 4 ? 4 : (v <= 8 ? 8 : 64)) : (LOG2N == 3 && v <= 4 ? 4 : 64)); };
    nzc = cls(ec); nzr = cls(er);
#ifdef AV1MI_NO_PRUNE
    nzc = nzr = 64;
#endif
  }
  // row pass of one class: [matrix-core path: the row's levels back from LDS through the normative dequantiser (spec 7.12.3),] inverse rows
  auto row_pass = [&](auto tag) {
    constexpr int NZT = decltype(tag)::value, NZ = NZT < CW ? NZT : CW;
    if constexpr (MM) {
      const int row = sl, lim = 1 << (7 + bd);
      const QmTab tab = QM ? qm_table(P) + (pc ? AV1MI_QM_PLANE : 0) + AV1MI_QM_32X32 + row : (QmTab)0;
      const uint32_t *lw = reinterpret_cast<const uint32_t *>(lvl + row * 32);
#pragma unroll
      for (int j2 = 0; j2 < NZ / 2; j2++) {
        const uint32_t w = lw[j2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int j = 2 * j2 + e;
          const int lvs = e ? (int)w >> 16 : (int)(int16_t)(w & 0xFFFF);
          uint32_t q;
          if constexpr (QM) q = tab[j * 32].q;
          else q = (row | j) == 0 ? (uint32_t)P->dc_q : (uint32_t)P->ac_q;
          const int sgn = lvs >> 31;
          int d = (int)((__umul24((uint32_t)((lvs ^ sgn) - sgn), q) & 0xFFFFFF) >> TSH);
          d = (d ^ sgn) - sgn;
          x[j] = d < -lim ? -lim : (d > lim - 1 ? lim - 1 : d);
        }
      }
    }
#pragma unroll
    for (int j = NZ; j < N; j++) x[j] = 0;
    Tx1d<LOG2N>::template inv_nz<NZ>(x, ht);
#pragma unroll
    for (int j = 0; j < N; j++) S->scratch[so + sl * ST + j] = (int16_t)clamp_bits(rshift_round(x[j], RS), bd + 6 > 16 ? bd + 6 : 16);
  };
  if (row_lane && eob && sl < nzr) {   // (rows beyond the extent are all zero: the column pass does not read them)
#ifndef AV1MI_NO_PRUNE   /* (A/B switch: the build without the pruned networks) */
    if (LOG2N >= 4 && nzc == 8) row_pass(NzTag<8>{});
    else if (LOG2N >= 5 && nzc == 16) row_pass(NzTag<16>{});
    else if ((LOG2N == 3 || LOG2N == 4) && nzc == 4) row_pass(NzTag<4>{});
    else
#endif
    row_pass(NzTag<64>{});
  }
  wave_sync();
  STAMP(4);   // forward rows, quantiser, dequantiser, eob, inverse rows
  auto col_pass = [&](auto tag) {
    constexpr int NZT = decltype(tag)::value, NZ = NZT < CW ? NZT : CW;
    const int maxv = (1 << bd) - 1;
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = i < NZ ? (int)S->scratch[so + i * ST + sl] : 0;
    Tx1d<LOG2N>::template inv_nz<NZ>(x, vt);
#pragma unroll
    for (int i = 0; i < N; i++) {
      int v = S->blkpix[po + i * N + sl] + ((x[i] + 8) >> 4);
      S->blkpix[po + i * N + sl] = (uint16_t)(v < 0 ? 0 : (v > maxv ? maxv : v));
    }
  };
  if (tx_lane && eob) {
#ifndef AV1MI_NO_PRUNE
    if (LOG2N >= 4 && nzr == 8) col_pass(NzTag<8>{});
    else if (LOG2N >= 5 && nzr == 16) col_pass(NzTag<16>{});
    else if ((LOG2N == 3 || LOG2N == 4) && nzr == 4) col_pass(NzTag<4>{});
    else
#endif
    col_pass(NzTag<64>{});
  }
  if (eob) {  // levels out (32-bit words, coalesced inside the group)
    // (16 bytes = 8 levels per store: the block's area of the level buffer is 32-byte aligned, av1mi_levels_off)
    constexpr int PIECES = CW * CW / 8;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4 *d128 = reinterpret_cast<u4 *>(grp ? lv_out1 : lv_out0);
    const u4 *l128 = reinterpret_cast<const u4 *>(lvl);
    for (int i = sl; i < PIECES; i += G) d128[i] = l128[i];
  }
  wave_sync();
  STAMP(5);   // inverse columns, levels -> HBM
  // ---- reconstruction -> HBM (coalesced rows) and -> line buffers for the neighbours to come
  {
    PIX *pl = rec_frame + poff;
    bool wide_out = false;
    if constexpr (sizeof(PIX) == 2 && N >= 8) {   // 16-byte pieces of the rows, as the source came in
      wide_out = uniform_i(!overhang && ((gs | gx) & 7) == 0);
      if (wide_out) {
        constexpr int CPR = N / 8;
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = sl; q < N * CPR; q += G) {
          const int r = q / CPR, c8 = q - r * CPR;
          *reinterpret_cast<u4 *>(pl + (size_t)(gy + r) * gs + gx + 8 * c8) = *reinterpret_cast<const u4 *>(&S->blkpix[po + r * N + 8 * c8]);
        }
      }
    }
    if constexpr (sizeof(PIX) == 1 && N >= 16) {   // 8-bit samples: 16 per 16-byte store
      wide_out = uniform_i(!overhang && ((gs | gx) & 15) == 0);
      if (wide_out) {
        constexpr int CPR = N / 16;
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = sl; q < N * CPR; q += G) {
          const int r = q / CPR, c16 = q - r * CPR;
          const u4 lo = *reinterpret_cast<const u4 *>(&S->blkpix[po + r * N + 16 * c16]), hi = *reinterpret_cast<const u4 *>(&S->blkpix[po + r * N + 16 * c16 + 8]);
          u4 v;
          v[0] = __builtin_amdgcn_perm(lo[1], lo[0], 0x06040200u); v[1] = __builtin_amdgcn_perm(lo[3], lo[2], 0x06040200u);
          v[2] = __builtin_amdgcn_perm(hi[1], hi[0], 0x06040200u); v[3] = __builtin_amdgcn_perm(hi[3], hi[2], 0x06040200u);
          *reinterpret_cast<u4 *>(pl + (size_t)(gy + r) * gs + gx + 16 * c16) = v;
        }
      }
    }
    if (!wide_out) {
#pragma unroll 4
      for (int p = sl; p < N * N; p += G) {
        const int r = p >> LOG2N, c = p & (N - 1);
        if (!overhang || (r < ph_lim && c < pw_lim)) pl[(size_t)(gy + r) * gs + gx + c] = (PIX)S->blkpix[po + p];
      }
    }
    if constexpr (PH != 1) {
      if (sl < N) {
        LN.above(plane)[lx + sl] = S->blkpix[po + (N - 1) * N + sl];
        LN.left(plane)[ly + sl] = S->blkpix[po + sl * N + (N - 1)];
      }
      if (sl < (N >> cgs)) {  // corners at every 8-aligned (chroma: 4-aligned) position of the bottom row and right column
        const int j = sl + 1, q = (j << cgs) - 1;
        LN.corner[plane][(ly + N) >> cgs][(lx >> cgs) + j] = S->blkpix[po + (N - 1) * N + q];
        LN.corner[plane][(ly >> cgs) + j][(lx + N) >> cgs] = S->blkpix[po + q * N + (N - 1)];
      }
      if (lane < step) S->blkdec[pc][r4 + lane + 1] |= ((1u << step) - 1u) << (c4 + 1);   // (a lane per row of the block)
      if (EXT && lane < (plane0 ? N >> 2 : N >> 3)) {
        const uint8_t smf = (uint8_t)(!(INTER && ii.is_inter) && best_mode >= SMOOTH_PRED && best_mode <= SMOOTH_H_PRED);
        LN.sm_above[pc][(lx >> (plane0 ? 2 : 3)) + lane] = smf; LN.sm_left[pc][(ly >> (plane0 ? 2 : 3)) + lane] = smf;
      }
    }
  }
  if (sl == 0) eob_out[grp] = eob;
  wave_sync();
  STAMP(6);   // reconstruction -> HBM, line buffers, decoded-block map
  if constexpr (EXT && NPL == 1 && !INTER && PH == 0 && N >= 8 && N <= 32) {
    // chroma from luma: the block's reconstructed luma, subsampled 2x2 (sum of four << 1 = Q3) and centred on its rounded
    // average, for the chroma item that follows (it may predict from it - spec 7.11.5)
    if (plane0 == 0 && P->cfl) {
      constexpr int NC = N / 2, L2C = LOG2N - 1;
      int16_t *AC = S->scratch + 2 * 16 * 24;
      int sum = 0;
      for (int p = lane; p < NC * NC; p += 64) {
        const int i = p >> L2C, j = p & (NC - 1);
        const uint16_t *q = S->blkpix + (2 * i) * N + 2 * j;
        const int v = ((int)q[0] + q[1] + q[N] + q[N + 1]) << 1;
        AC[p] = (int16_t)v;
        sum += v;
      }
      sum = wave_sum(sum);
      const int avg = (sum + (1 << (2 * L2C - 1))) >> (2 * L2C);
      for (int p = lane; p < NC * NC; p += 64) AC[p] = (int16_t)(AC[p] - avg);
      wave_sync();
    }
  }
  if constexpr (NPL == 2) {   // chroma item: the chroma-from-luma decision goes back in bits 16 .. 28 (flag, alpha U + 16, alpha V + 16: 6 bits each)
    if (use_cfl) {
      const int au = __builtin_amdgcn_readlane(cfl_alpha, 0), av = __builtin_amdgcn_readlane(cfl_alpha, 32);
      return (1 << 16) | ((au + 16) << 17) | ((av + 16) << 23) | (3 << 4) | 13;
    }
  }
  return (idtx << 12) | ((INTER ? ii.is_inter : 0) << 8) | ((best_delta + 3) << 4) | best_mode;
}

// Leaf block size (log2) of the partition tree at superblock-local (bx, by), or 0 if (bx, by) is not the origin of a leaf: the
// geometry rule of DESIGN.md §3.2 and, with a split mask (content-driven partition, §3.2b), the mask - av1mi_dev.h has the rule.
__device__ __forceinline__ int leaf_bsl_at(const Av1miDevParams &P, int have_mask, uint32_t mask, int sb_x, int sb_y, int bx, int by) {
  return av1mi_leaf_bsl_at(P.width, P.height, P.min_bs_log2, P.max_bs_log2, have_mask, mask, sb_x, sb_y, bx, by);
}

template <typename PIX, bool INTER, int TSB, bool QM, bool SPLIT, bool EXT>
__device__ __forceinline__ void encode_superblock(const SbCtx &cx, const PIX *frame, PIX *rec_frame, int16_t *sb_levels,
                                                  Av1miBlkInfo *info, int b8_stride, const PIX *ref_frame,
                                                  const unsigned long long *me_best /* this superblock's first unit */, int si,
                                                  int have_mask, uint32_t mask /* the superblock's split mask (content-driven partition) */) {
  const Av1miDevParams &P = *cx.P;
  // SPLIT: wave 0 of the workgroup walks the luma blocks (and takes the decisions), wave 1 the chroma blocks
  constexpr int WL = SPLIT ? 1 : 0, WC = SPLIT ? 2 : 0;
  const int wv = SPLIT ? uniform_i((int)threadIdx.x >> 6) : 0;
#pragma nounroll
  for (int z = 0; z < 64; z++) {  // 8x8 units in Z (partition) order
    const int bx = (((z >> 0) & 1) | ((z >> 1) & 2) | ((z >> 2) & 4)) << 3;
    const int by = (((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4)) << 3;
    if (cx.sb_y + by >= P.height || cx.sb_x + bx >= P.width) continue;
    const int bsl = leaf_bsl_at(P, have_mask, mask, cx.sb_x, cx.sb_y, bx, by);
    if (bsl == 0) continue;
    const int n = 1 << bsl;
    const int qi = si * 64 + z;
    int mode = 3 << 4;   // (mode | (angle delta + 3) << 4)
    InterInfo ii;
    ii.ref = ref_frame; ii.is_inter = 0; ii.mv_row = ii.mv_col = 0; ii.sad_inter = 0;
    if constexpr (INTER) {
      // motion search result of this leaf: (cost << 16) | candidate index, cost = SAD + n * (|dx| + |dy|)
      const unsigned long long key = g_ws.key[(by >> 3) * 8 + (bx >> 3)];   // (staged per superblock: recon_sb_kernel)
      if (P.subpel) {  // refined (me_kernel.hip): (SAD << 36) | (u16 mv.row << 16) | u16 mv.col, 1/8 samples
        ii.mv_row = (int16_t)(key >> 16); ii.mv_col = (int16_t)key;
        ii.sad_inter = (int)(key >> 36);
      } else {
        int dyv, dxv, cost;
        av1mi_me_key_decode(key, P.me_range, &dyv, &dxv, &cost);
        ii.mv_row = dyv * 8; ii.mv_col = dxv * 8;
        ii.sad_inter = cost - n * ((dxv < 0 ? -dxv : dxv) + (dyv < 0 ? -dyv : dyv));
      }
      // the block's inter version is done (recon_inter_pre_kernel): its eobs, in case motion compensation wins
      const uint16_t *pre = g_ws.pre_eob[(by >> 3) * 8 + (bx >> 3)];
      ii.pre_eob[0] = pre[0]; ii.pre_eob[1] = pre[1]; ii.pre_eob[2] = pre[2];
    }
    int16_t *lv_y = sb_levels + av1mi_levels_off(0, bx, by);
    int16_t *lv_u = sb_levels + av1mi_levels_off(1, bx, by), *lv_v = sb_levels + av1mi_levels_off(2, bx, by);
    // luma (mode decision inside), then U and V together
    int dec = 0;
    if (!SPLIT || wv == 0) {
      int *eo = SbSel<WL>::get()->eobs;
      switch (bsl) {
#if AV1MI_RECON_BIG
        case 6: dec = tx_item<PIX, 6, 1, INTER, TSB, QM, (INTER ? 2 : 0), WL, EXT>(cx, frame, rec_frame, 0, bx, by, mode, ii, lv_y, lv_y, eo, qi); break;
#endif
        case 5: dec = tx_item<PIX, 5, 1, INTER, TSB, QM, (INTER ? 2 : 0), WL, EXT>(cx, frame, rec_frame, 0, bx, by, mode, ii, lv_y, lv_y, eo, qi); break;
        case 4: dec = tx_item<PIX, 4, 1, INTER, TSB, QM, (INTER ? 2 : 0), WL, EXT>(cx, frame, rec_frame, 0, bx, by, mode, ii, lv_y, lv_y, eo, qi); break;
        default: dec = tx_item<PIX, 3, 1, INTER, TSB, QM, (INTER ? 2 : 0), WL, EXT>(cx, frame, rec_frame, 0, bx, by, mode, ii, lv_y, lv_y, eo, qi); break;
      }
      dec = uniform_i(dec);
      if (SPLIT) {   // the luma block is finished: its eob (and identity-transform flag) for the block-info entry, which the chroma wave writes
        if (cx.lane == 0) q_post(&g_q.e0[qi], eo[0] | (((dec >> 12) & 1) << 12));
        continue;
      }
    } else {
      dec = q_wait(&g_q.dec[qi]);
    }
    int idtx = (dec >> 12) & 1;
    mode = dec & 0x7F; ii.is_inter = (dec >> 8) & 1;   // the luma pass decides (mode and angle delta); the chroma pass follows it
    int *eo = SbSel<WC>::get()->eobs;
    int cdec = 0;   // chroma item's return: chroma from luma in bits 16 .. 28
    switch (bsl) {
#if AV1MI_RECON_BIG
      case 6: tx_item<PIX, 5, 2, INTER, TSB, QM, (INTER ? 2 : 0), WC, EXT>(cx, frame, rec_frame, 1, bx >> 1, by >> 1, mode, ii, lv_u, lv_v, eo + 1, -1); break;
#endif
      case 5: cdec = tx_item<PIX, 4, 2, INTER, TSB, QM, (INTER ? 2 : 0), WC, EXT>(cx, frame, rec_frame, 1, bx >> 1, by >> 1, mode, ii, lv_u, lv_v, eo + 1, -1); break;
      case 4: cdec = tx_item<PIX, 3, 2, INTER, TSB, QM, (INTER ? 2 : 0), WC, EXT>(cx, frame, rec_frame, 1, bx >> 1, by >> 1, mode, ii, lv_u, lv_v, eo + 1, -1); break;
      default: cdec = tx_item<PIX, 2, 2, INTER, TSB, QM, (INTER ? 2 : 0), WC, EXT>(cx, frame, rec_frame, 1, bx >> 1, by >> 1, mode, ii, lv_u, lv_v, eo + 1, -1); break;
    }
    {  // the block's info into every 8x8 unit it covers: one lane per unit
      const int n8 = n >> 3;
      int e0 = SPLIT ? q_wait(&g_q.e0[qi]) : eo[0];
      if (SPLIT) { idtx = (e0 >> 12) & 1; e0 &= 0xFFF; }
      if (cx.lane < n8 * n8) {
        const int e1 = eo[1], e2 = eo[2];
        const int i = cx.lane / n8, j = cx.lane - i * n8;
        Av1miBlkInfo bi;
        const bool unit_inside = cx.sb_y + by + 8 * i < P.height && cx.sb_x + bx + 8 * j < P.width;   // an overhanging block's units beyond the frame have no entry
        bi.ymode = (uint8_t)(mode & 15); bi.skip = (uint8_t)((e0 | e1 | e2) == 0); bi.bsl = (uint8_t)bsl; bi.is_inter = (uint8_t)ii.is_inter;
        bi.eob[0] = (uint16_t)e0; bi.eob[1] = (uint16_t)e1; bi.eob[2] = (uint16_t)e2;
        bi.mv_row = (int16_t)(ii.is_inter ? ii.mv_row : 0); bi.mv_col = (int16_t)(ii.is_inter ? ii.mv_col : 0);
        // angle delta + 3 of the (luma and chroma) directional mode | luma identity transform << 3 | chroma from luma: alpha U, alpha V as
        // 6-bit two's complement << 4, << 10 (both zero = no chroma from luma: that pair is not codable)
        cdec = uniform_i(cdec);
        const int cflf = (cdec >> 16) & 1, au = cflf ? ((cdec >> 17) & 63) - 16 : 0, av = cflf ? ((cdec >> 23) & 63) - 16 : 0;
        bi.angle = (uint16_t)(((mode >> 4) & 7) | (idtx << 3) | ((au & 63) << 4) | ((av & 63) << 10));
        if (unit_inside) info[((by >> 3) + i) * b8_stride + (bx >> 3) + j] = bi;
      }
    }
  }
}

// The tile walk of an inter frame is SPLIT (two waves per tile, see encode_superblock): its duration is the chain of the tile with
// most intra-coded blocks, and the chroma blocks were 45 % of that chain.
#ifndef AV1MI_RECON_MIN_WAVES
#define AV1MI_RECON_MIN_WAVES 5
#endif
#ifndef AV1MI_SPLIT_INTRA
#define AV1MI_SPLIT_INTRA 0
#endif
template <bool INTER> struct WalkSplit { static constexpr bool value = INTER || AV1MI_SPLIT_INTRA; };
template <typename PIX, bool INTER, int TSB, bool QM, bool EXT>
__global__ void __launch_bounds__(WalkSplit<INTER>::value ? 128 : 64) __attribute__((amdgpu_waves_per_eu(AV1MI_RECON_BIG ? 2 : AV1MI_RECON_MIN_WAVES, AV1MI_RECON_BIG ? 2 : AV1MI_RECON_MIN_WAVES))) recon_sb_kernel(const Av1miDevParams *__restrict__ Pd, const PIX *__restrict__ src, PIX *__restrict__ rec,
                                                     int16_t *__restrict__ levels, Av1miBlkInfo *__restrict__ blk,
                                                     const PIX *__restrict__ ref /* inter frame: previous final reconstruction, one frame */,
                                                     const unsigned long long *__restrict__ me_best,
                                                     const uint32_t *__restrict__ part /* split masks of the launch's first frame on (null: partition by geometry) */) {
  // The parameters come through a pointer to device memory, not by value: the transform items are `noinline` and take the
  // address of the block, and the address of a by-value kernel argument is a private copy - every lane wrote the whole
  // structure (0.6 KB) to scratch at the start of the kernel and read its fields back from there.  Loads through a
  // uniform const pointer are scalar loads.
  const Av1miDevParams &P = *Pd;
  constexpr bool SPLIT = WalkSplit<INTER>::value;
  if constexpr (SPLIT) {
    for (int t = threadIdx.x; t < TSB * TSB * 64; t += 128) { g_q.dec[t] = 0; g_q.e0[t] = 0; }
  }
#ifdef AV1MI_STAMPS
  if (threadIdx.x < STAMP_CLASSES * STAMP_PHASES) g_stamp.acc[threadIdx.x] = 0;
  const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // one wave per TILE: TSB x TSB superblocks in raster order (TSB = 1 unless the frame needs more than 64 x 64 tiles)
  const int tiles_per_frame = P.tile_rows * P.tile_cols, sbs_per_frame = P.sb_rows * P.sb_cols;
  const int f = blockIdx.x / tiles_per_frame, tile = blockIdx.x % tiles_per_frame;
  const int tr = tile / P.tile_cols, tc = tile % P.tile_cols;
  // tile extent in 4x4 units, clipped to the frame
  const int tile_mi_r0 = tr * TSB * 16, tile_mi_c0 = tc * TSB * 16;
  const int tile_mi_r1 = (tile_mi_r0 + TSB * 16 < P.mi_rows) ? tile_mi_r0 + TSB * 16 : P.mi_rows;
  const int tile_mi_c1 = (tile_mi_c0 + TSB * 16 < P.mi_cols) ? tile_mi_c0 + TSB * 16 : P.mi_cols;
  const PIX *frame = src + (size_t)f * P.frame_samples;
#pragma nounroll
  for (int si = 0; si < TSB * TSB; si++) {
    const int sbr = tr * TSB + si / TSB, sbc = tc * TSB + si % TSB;
    if (sbr >= P.sb_rows || sbc >= P.sb_cols) continue;
    const int sb = sbr * P.sb_cols + sbc;
    SbCtx cx;
    cx.P = Pd; cx.lane = threadIdx.x & 63; cx.sb_x = sbc * 64; cx.sb_y = sbr * 64;
    cx.tox = (si % TSB) * 64; cx.toy = (si / TSB) * 64;
    SbLds *const sbl = (SPLIT && threadIdx.x >= 64) ? SbSel<2>::get() : SbSel<0>::get();   // each wave of a split walk keeps a map of its own
    // decoded-block map (clear_block_decoded_flags, spec §5.11.3): the row above and the column left of the superblock
    // are decoded as far as the TILE reaches (so the above-right superblock of a two-superblock tile counts)
    __syncthreads();
    {
      const int w4 = tile_mi_c1 - sbc * 16, h4 = tile_mi_r1 - sbr * 16;
      if (cx.lane < 2 * 19) {   // a lane per row of a plane's map
        const int pl = cx.lane / 19, y = cx.lane % 19 - 1;
        const int sz = 16 >> pl, sw = w4 >> pl, sh = h4 >> pl;
        uint32_t m = 0;
        for (int x = -1; x < 18; x++) {
          int v = 0;
          if (y <= sz && x <= sz) {
            if (y < 0 && x < sw) v = 1;
            else if (x < 0 && y < sh) v = 1;
            if (y == sz && x == -1) v = 0;
          }
          m |= (uint32_t)v << (x + 1);
        }
        sbl->blkdec[pl][y + 1] = m;
      }
    }
    if constexpr (INTER) {
      // stage what the walk's blocks would fetch from HBM one after the other (WalkStage): every load of the superblock in flight at once
      const int t = threadIdx.x, W = P.width, H = P.height, Wc = W >> 1, Hc = H >> 1;
      const int sx = sbc * 64, sy = sbr * 64, gsy = P.stride_y, gsc = P.stride_c;
      const PIX *recf = rec + (size_t)f * P.frame_samples;
      for (int q = t; q < 64 * 8; q += 128) {   // source luma, 8 samples per piece
        const int r = q >> 3, c8 = q & 7, y = sy + r < H ? sy + r : H - 1, x = sx + 8 * c8;
        const PIX *row = frame + (size_t)y * gsy;
        uint16_t *dst = &g_ws.src_y[r * 64 + 8 * c8];
        if (sizeof(PIX) == 2 && x + 8 <= W && ((gsy | x) & 7) == 0) {
          typedef unsigned int u4 __attribute__((ext_vector_type(4)));
          *reinterpret_cast<u4 *>(dst) = *reinterpret_cast<const u4 *>(row + x);
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) dst[j] = (uint16_t)row[x + j < W ? x + j : W - 1];
        }
      }
      for (int q = t; q < 8 * 64; q += 128) {   // provisional reconstruction: luma rows / columns 8 k + 7
        const int k = q >> 6, i = q & 63;
        const int yr = sy + 8 * k + 7 < H ? sy + 8 * k + 7 : H - 1, xr = sx + i < W ? sx + i : W - 1;
        const int yc = sy + i < H ? sy + i : H - 1, xc = sx + 8 * k + 7 < W ? sx + 8 * k + 7 : W - 1;
        g_ws.row_y[k][i] = (uint16_t)recf[(size_t)yr * gsy + xr];
        g_ws.col_y[k][i] = (uint16_t)recf[(size_t)yc * gsy + xc];
      }
      for (int q = t; q < 2 * 8 * 32; q += 128) {   // U, V: rows / columns 4 k + 3
        const int pl = q >> 8, k = (q >> 5) & 7, i = q & 31;
        const PIX *pc = recf + (pl ? P.plane_off_v : P.plane_off_u);
        const int cx0 = sx >> 1, cy0 = sy >> 1;
        const int yr = cy0 + 4 * k + 3 < Hc ? cy0 + 4 * k + 3 : Hc - 1, xr = cx0 + i < Wc ? cx0 + i : Wc - 1;
        const int yc = cy0 + i < Hc ? cy0 + i : Hc - 1, xc = cx0 + 4 * k + 3 < Wc ? cx0 + 4 * k + 3 : Wc - 1;
        g_ws.row_c[pl][k][i] = (uint16_t)pc[(size_t)yr * gsc + xr];
        g_ws.col_c[pl][k][i] = (uint16_t)pc[(size_t)yc * gsc + xc];
      }
      if (t < 64) {   // search keys and the inter version's eobs of the superblock's 8x8 units (units outside the frame are never asked for)
        const int uy = sbr * 8 + (t >> 3), ux = sbc * 8 + (t & 7);
        if (uy < P.b8_rows && ux < P.b8_cols) {
          const size_t u = (size_t)uy * P.b8_cols + ux;
          g_ws.key[t] = me_best[u];
          const Av1miBlkInfo bi = blk[(size_t)f * P.b8_rows * P.b8_cols + u];
          g_ws.pre_eob[t][0] = bi.eob[0]; g_ws.pre_eob[t][1] = bi.eob[1]; g_ws.pre_eob[t][2] = bi.eob[2];
        }
      }
    }
    __syncthreads();
    int16_t *sb_levels = levels + ((size_t)f * sbs_per_frame + sb) * AV1MI_SB_LEVELS;
    Av1miBlkInfo *info = blk + (size_t)f * P.b8_rows * P.b8_cols + (size_t)(sbr * 8) * P.b8_cols + sbc * 8;
    // (inter frames are launched one at a time: f == 0 then, and `ref` / `me_best` belong to that frame)
    encode_superblock<PIX, INTER, TSB, QM, SPLIT, EXT>(cx, frame, rec + (size_t)f * P.frame_samples, sb_levels, info, P.b8_cols, ref,
                                                  me_best ? me_best + (size_t)(sbr * 8) * P.b8_cols + sbc * 8 : nullptr, si,
                                                  part != nullptr, part ? part[(size_t)f * sbs_per_frame + sb] : 0u);
  }
#ifdef AV1MI_STAMPS
  __syncthreads();
  if (threadIdx.x < STAMP_CLASSES * STAMP_PHASES && g_stamp.acc[threadIdx.x]) atomicAdd(&g_stamp_sum[threadIdx.x], g_stamp.acc[threadIdx.x]);
  if (threadIdx.x == 0) { atomicAdd(&g_stamp_sum[STAMP_CLASSES * STAMP_PHASES], __builtin_amdgcn_s_memtime() - stamp_t0); atomicAdd(&g_stamp_sum[STAMP_CLASSES * STAMP_PHASES + 1], 1ull);
                          atomicAdd(&g_stamp_sum[STAMP_CLASSES * STAMP_PHASES + 2], __builtin_amdgcn_s_memrealtime() - stamp_r0); }
#endif
}

// First launch of an inter frame: every leaf block coded as an INTER block - motion compensation with the search's vector,
// transform, quantisation, reconstruction, levels - with one wave per cell of the largest block size, all blocks of the
// frame at once (luma and chroma in waves of their own): nothing here depends on a neighbour.  It leaves the reconstruction
// and the levels in place and the eobs in a provisional block-info entry; recon_sb_kernel then walks the tiles in order for the part
// that does depend on neighbours - the intra SAD of the decision and the blocks intra prediction wins.
// (As one kernel a P frame was 510 waves of six serial block passes: 290 us of latency on the chunk's serial chain.)
template <typename PIX, bool QM, bool SP /* sub-sample motion vectors (P->subpel) */>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AV1MI_RECON_BIG ? 2 : AV1MI_RECON_MIN_WAVES, AV1MI_RECON_BIG ? 2 : AV1MI_RECON_MIN_WAVES))) recon_inter_pre_kernel(const Av1miDevParams *__restrict__ Pd,
                                                     const PIX *__restrict__ src, PIX *__restrict__ rec, int16_t *__restrict__ levels,
                                                     Av1miBlkInfo *__restrict__ blk, const PIX *__restrict__ ref,
                                                     const unsigned long long *__restrict__ me_best, int cell_log2,
                                                     const uint32_t *__restrict__ part) {
  const Av1miDevParams &P = *Pd;
  const int u = 1 << (cell_log2 - 3);
  for (int uy = blockIdx.y * u; uy < (int)(blockIdx.y + 1) * u; uy++)
    for (int ux = blockIdx.x * u; ux < (int)(blockIdx.x + 1) * u; ux++) {
      const int x = ux * 8, y = uy * 8;
      if (x >= P.width || y >= P.height) continue;
      SbCtx cx;
      cx.P = Pd; cx.lane = threadIdx.x; cx.sb_x = x & ~63; cx.sb_y = y & ~63; cx.tox = 0; cx.toy = 0;
      const int bx = x - cx.sb_x, by = y - cx.sb_y;
      const int bsl = leaf_bsl_at(P, part != nullptr, part ? part[(cx.sb_y >> 6) * P.sb_cols + (cx.sb_x >> 6)] : 0u, cx.sb_x, cx.sb_y, bx, by);
      if (bsl == 0) continue;
      const int n = 1 << bsl;
      InterInfo ii;
      ii.ref = ref; ii.is_inter = 1; ii.pre_eob[0] = ii.pre_eob[1] = ii.pre_eob[2] = 0;
      const unsigned long long key = me_best[(size_t)uy * P.b8_cols + ux];
      if (P.subpel) {
        ii.mv_row = (int16_t)(key >> 16); ii.mv_col = (int16_t)key;
        ii.sad_inter = (int)(key >> 36);
      } else {
        int dyv, dxv, cost;
        av1mi_me_key_decode(key, P.me_range, &dyv, &dxv, &cost);
        ii.mv_row = dyv * 8; ii.mv_col = dxv * 8;
        ii.sad_inter = cost - n * ((dxv < 0 ? -dxv : dxv) + (dyv < 0 ? -dyv : dyv));
      }
      const int sb = (cx.sb_y >> 6) * P.sb_cols + (cx.sb_x >> 6);
      int16_t *sb_levels = levels + (size_t)sb * AV1MI_SB_LEVELS;
      int16_t *lv_y = sb_levels + av1mi_levels_off(0, bx, by);
      int16_t *lv_u = sb_levels + av1mi_levels_off(1, bx, by), *lv_v = sb_levels + av1mi_levels_off(2, bx, by);
      int mode = 3 << 4;
      __syncthreads();
      // blockIdx.z = 0: the luma block, 1: the two chroma blocks - separate waves, half the latency; each leaves its eobs
      // in the provisional block-info entry (the only fields recon_sb_kernel reads back)
      Av1miBlkInfo *bi = &blk[(size_t)uy * P.b8_cols + ux];
      if (blockIdx.z == 0) {
        switch (bsl) {
#if AV1MI_RECON_BIG
          case 6: tx_item<PIX, 6, 1, true, 1, QM, 1, 0, SP>(cx, src, rec, 0, bx, by, mode, ii, lv_y, lv_y, g_sb.eobs, -1); break;
#endif
          case 5: tx_item<PIX, 5, 1, true, 1, QM, 1, 0, SP>(cx, src, rec, 0, bx, by, mode, ii, lv_y, lv_y, g_sb.eobs, -1); break;
          case 4: tx_item<PIX, 4, 1, true, 1, QM, 1, 0, SP>(cx, src, rec, 0, bx, by, mode, ii, lv_y, lv_y, g_sb.eobs, -1); break;
          default: tx_item<PIX, 3, 1, true, 1, QM, 1, 0, SP>(cx, src, rec, 0, bx, by, mode, ii, lv_y, lv_y, g_sb.eobs, -1); break;
        }
        if (threadIdx.x == 0) bi->eob[0] = (uint16_t)g_sb.eobs[0];
      } else {
        switch (bsl) {
#if AV1MI_RECON_BIG
          case 6: tx_item<PIX, 5, 2, true, 1, QM, 1, 0, SP>(cx, src, rec, 1, bx >> 1, by >> 1, mode, ii, lv_u, lv_v, g_sb.eobs + 1, -1); break;
#endif
          case 5: tx_item<PIX, 4, 2, true, 1, QM, 1, 0, SP>(cx, src, rec, 1, bx >> 1, by >> 1, mode, ii, lv_u, lv_v, g_sb.eobs + 1, -1); break;
          case 4: tx_item<PIX, 3, 2, true, 1, QM, 1, 0, SP>(cx, src, rec, 1, bx >> 1, by >> 1, mode, ii, lv_u, lv_v, g_sb.eobs + 1, -1); break;
          default: tx_item<PIX, 2, 2, true, 1, QM, 1, 0, SP>(cx, src, rec, 1, bx >> 1, by >> 1, mode, ii, lv_u, lv_v, g_sb.eobs + 1, -1); break;
        }
        if (threadIdx.x == 0) { bi->eob[1] = (uint16_t)g_sb.eobs[1]; bi->eob[2] = (uint16_t)g_sb.eobs[2]; }
      }
    }
}
}  // namespace

#if defined(AV1MI_STAMPS) && !AV1MI_RECON_BIG && !AV1MI_RECON_PIX8
// diagnostic build: read (and clear) the phase sums.  out[class * 8 + phase] cycles, then total wave cycles, then waves
extern "C" int av1mi_debug_stamps(unsigned long long *out, int reset) {
  unsigned long long z[STAMP_CLASSES * STAMP_PHASES + 3] = {};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp_sum), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_sum), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif

// ref == nullptr: P->n_frames key frames in one launch.  ref != nullptr: ONE inter frame (P->n_frames must be 1),
// predicted from `ref` with the motion search results `me_best` of that frame.
// dP: the same parameters in device memory (what the kernel reads; n_frames and the loop-filter levels are not used by it).
#if AV1MI_RECON_BIG && AV1MI_RECON_PIX8
#define AV1MI_LAUNCH_RECON av1mi_launch_recon64_u8   /* leaf blocks up to 64x64 (P->max_bs_log2 == 6), 8-bit samples */
#elif AV1MI_RECON_BIG
#define AV1MI_LAUNCH_RECON av1mi_launch_recon64_u16
#elif AV1MI_RECON_PIX8
#define AV1MI_LAUNCH_RECON av1mi_launch_recon_u8     /* leaf blocks up to 32x32 */
#else
#define AV1MI_LAUNCH_RECON av1mi_launch_recon_u16
#endif
#if AV1MI_RECON_PIX8
typedef uint8_t ReconPix;
#else
typedef uint16_t ReconPix;
#endif
extern "C" hipError_t AV1MI_LAUNCH_RECON(const Av1miDevParams *P, const Av1miDevParams *dP, const void *src, void *rec, int16_t *levels,
                                         Av1miBlkInfo *blk, const void *ref, const unsigned long long *me_best, const uint32_t *part, hipStream_t stream) {
  const int grid = P->n_frames * P->tile_rows * P->tile_cols;
  const bool inter = ref != nullptr;
  if (inter) {  // first launch of an inter frame: every block as an inter block, all at once (see recon_inter_pre_kernel)
    const int g = P->max_bs_log2 > (AV1MI_RECON_BIG ? 6 : 5) ? (AV1MI_RECON_BIG ? 6 : 5) : P->max_bs_log2, cell = 1 << g;
    dim3 pgrid((P->width + cell - 1) / cell, (P->height + cell - 1) / cell, 2);   // z: luma | chroma
#define PRE_LAUNCH2(PIXT, SPV)                                                                                                               \
    do {                                                                                                                                     \
      if (P->qm_tab) hipLaunchKernelGGL((recon_inter_pre_kernel<PIXT, true, SPV>), pgrid, dim3(64), 0, stream, dP, (const PIXT *)src, (PIXT *)rec, \
                                        levels, blk, (const PIXT *)ref, me_best, g, part);                                                   \
      else hipLaunchKernelGGL((recon_inter_pre_kernel<PIXT, false, SPV>), pgrid, dim3(64), 0, stream, dP, (const PIXT *)src, (PIXT *)rec,          \
                              levels, blk, (const PIXT *)ref, me_best, g, part);                                                             \
    } while (0)
#define PRE_LAUNCH(PIXT) do { if (P->subpel) PRE_LAUNCH2(PIXT, true); else PRE_LAUNCH2(PIXT, false); } while (0)
    PRE_LAUNCH(ReconPix);
#undef PRE_LAUNCH
#undef PRE_LAUNCH2
  }
  // quantiser matrices (P->qm_tab): kernels of their own, so the plain quantiser's registers and scratch are what they were
#define RECON_LAUNCH2(PIXT, INTERV, TSBV, EXTV)                                                                                             \
  do {                                                                                                                                      \
    if (P->qm_tab) hipLaunchKernelGGL((recon_sb_kernel<PIXT, INTERV, TSBV, true, EXTV>), dim3(grid), dim3(WalkSplit<INTERV>::value ? 128 : 64), 0, stream, dP, (const PIXT *)src,  \
                                      (PIXT *)rec, levels, blk, (const PIXT *)ref, me_best, part);                                          \
    else hipLaunchKernelGGL((recon_sb_kernel<PIXT, INTERV, TSBV, false, EXTV>), dim3(grid), dim3(WalkSplit<INTERV>::value ? 128 : 64), 0, stream, dP, (const PIXT *)src,           \
                            (PIXT *)rec, levels, blk, (const PIXT *)ref, me_best, part);                                                    \
  } while (0)
  // the optional intra tools (edge filter, chroma from luma) live in instantiations of their own (EXT)
#define RECON_LAUNCH(PIXT, INTERV, TSBV)                                                                                                    \
  do { if (P->edge_filter || P->cfl || P->tx_search) RECON_LAUNCH2(PIXT, INTERV, TSBV, true); else RECON_LAUNCH2(PIXT, INTERV, TSBV, false); } while (0)
  if (P->tile_sb == 1) { if (inter) RECON_LAUNCH(ReconPix, true, 1); else RECON_LAUNCH(ReconPix, false, 1); }
  else { if (inter) RECON_LAUNCH(ReconPix, true, 2); else RECON_LAUNCH(ReconPix, false, 2); }
#undef RECON_LAUNCH
#undef RECON_LAUNCH2
  return hipGetLastError();
}
