// av1mi_dev.h - host/device shared definitions of the MI355X AV1 chunk encoder.
// Product code (never includes anything from oracle/).
#ifndef AV1MI_DEV_H
#define AV1MI_DEV_H
#include <stdint.h>
#include <stddef.h>
#ifdef __HIPCC__
#define AV1MI_HD __host__ __device__
#else
#define AV1MI_HD
#endif

// Uniform kernel parameters of one chunk (all frames of a chunk share them).
#define AV1MI_QM_4X4 0
#define AV1MI_QM_8X8 16
#define AV1MI_QM_16X16 80
#define AV1MI_QM_32X32 336
#define AV1MI_QM_PLANE 1360   /* entries per plane type */

struct Av1miQmEntry { uint32_t q, recip; };

struct Av1miDevParams {
  int width, height, bit_depth;   // CODED size: the signalled size rounded up to multiples of 8 (the source is edge-extended)
  int true_w, true_h;             // signalled size: what the decoder crops to, clamps references to and restores within
  int mi_rows, mi_cols;     // 4x4 units
  int sb_rows, sb_cols;     // 64x64 superblocks
  int tile_sb;              // tile size in superblocks, both ways: 1, or 2 when the frame has more than 64 superblock rows/columns
  int tile_rows, tile_cols; // tile grid (the last row/column of tiles may be one superblock short)
  int b8_rows, b8_cols;     // 8x8 units (block-info granularity)
  int n_frames;
  int base_q_idx, qctx;
  int dc_q, ac_q;
  uint32_t dc_recip, ac_recip;  // ceil(2^32 / q)
  // quantiser matrices (using_qmatrix): null, or per plane type (luma | chroma) and square transform size the dequantiser
  // step of every coefficient position, {q2 = Round2(q * Quantizer_Matrix[level][plane > 0][pos], 5), ceil(2^32 / q2)};
  // a plane whose level is 15 (flat) holds q itself.  Offsets AV1MI_QM_*; header fields using_qm, qm_y, qm_uv.
  const Av1miQmEntry *qm_tab;
  int using_qm, qm_y, qm_uv;
  int min_bs_log2, max_bs_log2;
  // content-driven partition (av1mi_params.partition_search): null, or per frame of the chunk and superblock the split mask
  // partition_kernel wrote (av1mi_node_split below); a node larger than min_bs_log2 and not larger than max_bs_log2 follows it
  const uint32_t *part_map;
  uint32_t mode_mask;
  int angle_delta;          // 1: directional winners of the luma mode decision are refined over the angle deltas -3 .. +3
  int edge_filter;          // enable_intra_edge_filter
  int cfl;                  // chroma from luma is a candidate (key frames, blocks up to 32x32)
  int tx_search;            // identity transform for sparse intra luma residuals
  int enable_cdef, cdef_y_pri, cdef_y_sec, cdef_uv_pri, cdef_uv_sec, cdef_damping;
  int disable_cdf_update;
  // plane geometry in samples
  int stride_y, stride_c;
  long plane_off_u, plane_off_v;  // sample offsets of U and V inside a frame
  long frame_samples;             // samples per frame
  // per-tile bitstream slot
  int tile_slot_bytes;
  int stream_cap;                 // 32-bit symbol-stream entries per tile (multiple of 4)
  // header blob: sequence header OBU, then one slot of `hdr_slot_bytes` per frame holding that frame's OBU_FRAME
  // payload prefix (frame header + alignment); frame_hdr_bytes = its length on key frames, inter_hdr_bytes on
  // inter frames (frames differ only in type and grain_seed)
  int seq_hdr_bytes, frame_hdr_bytes, inter_hdr_bytes, hdr_slot_bytes;
  int tile_size_bytes;
  // inter coding: key frame every `keyint` frames of the chunk (1 = all key frames); motion search range
  int keyint, me_range;
  int subpel;                     // inter frames: 1 = quarter-sample vectors (refined search) + EIGHTTAP filter; 0 = whole-sample vectors, BILINEAR
  int me_presearch;               // inter frames: 1 = the full search runs around the centre a quarter-resolution pre-search found per superblock
  // deblocking filter: loop_filter_level[0..3] (luma vertical edges, luma horizontal, U, V) of key / inter frames and sharpness
  int lf_level[4], lf_level_inter[4], lf_sharpness;
  // loop restoration (luma Wiener, 64x64 units): literal bits that code candidate k's coefficients against the
  // reference RefLrWiener (both passes), MSB first in the low `lr_code_len[r][k]` bits
  int enable_lr;
  int lr_code_len[4][3];                 // [reference: 0 = Wiener_Taps_Mid, r = candidate r-1][candidate]
  unsigned long long lr_code_bits[4][3];
  // enable_lr = 2: the same for a self-guided unit (lr_sgr_set + both weights against RefSgrXqd; reference 0 = Sgrproj_Xqd_Mid)
  int sgr_code_len[4][3];
  unsigned long long sgr_code_bits[4][3];
};

// ---- partition: does the node of size 2^bsl at superblock-local (ox, oy) split?  One rule for every kernel that
// walks blocks (reconstruction, motion search, symbolize).  The syntax forces a split where the node's half point is outside the
// frame (has_rows / has_cols of spec 5.11.4; a leaf may overhang the frame edge by less than half its size); nodes above max_bs_log2
// split, nodes at min_bs_log2 (and 8x8) never do; in between the superblock's split mask decides - bit 0: the 64x64 node, bits
// 1 .. 4: its 32x32 quadrants in Z order, bits 5 .. 20: the sixteen 16x16 nodes in Z order - or, without a mask, the node is a leaf.
AV1MI_HD inline int av1mi_node_split(int width, int height, int min_bs_log2, int max_bs_log2, int have_mask, uint32_t mask,
                                     int sb_x, int sb_y, int ox, int oy, int bsl) {
  const int n = 1 << bsl;
  if (bsl <= 3) return 0;
  if (sb_y + oy + (n >> 1) >= height || sb_x + ox + (n >> 1) >= width) return 1;
  if (bsl <= min_bs_log2) return 0;
  if (bsl > max_bs_log2) return 1;
  if (!have_mask) return 0;
  const int bit = bsl == 6 ? 0 : (bsl == 5 ? 1 + ((oy >> 5) << 1 | (ox >> 5))
                                           : 5 + (((ox >> 4) & 1) | (((oy >> 4) & 1) << 1) | (((ox >> 5) & 1) << 2) | (((oy >> 5) & 1) << 3)));
  return (int)((mask >> bit) & 1u);
}
// log2 size of the leaf whose origin is superblock-local (bx, by), or 0 if (bx, by) is not the origin of a leaf
AV1MI_HD inline int av1mi_leaf_bsl_at(int width, int height, int min_bs_log2, int max_bs_log2, int have_mask, uint32_t mask,
                                      int sb_x, int sb_y, int bx, int by) {
  for (int bsl = 6; bsl >= 3; bsl--) {
    const int n = 1 << bsl;
    const int ox = bx & ~(n - 1), oy = by & ~(n - 1);
    if (!av1mi_node_split(width, height, min_bs_log2, max_bs_log2, have_mask, mask, sb_x, sb_y, ox, oy, bsl)) return (ox == bx && oy == by) ? bsl : 0;
  }
  return 0;
}

// ---- motion search keys (me_kernel.hip -> recon_kernel.hip): per leaf a 64-bit key, the minimum over its candidates of
//   (centre code << 40) | (cost << 16) | candidate index,   cost = SAD + n (|dx| + |dy|) < 2^24,   index = (dy - Cy + R) (2 R + 1) + (dx - Cx + R)
// centre code = (Cy / 8 & 0xFFF) << 12 | (Cx / 8 & 0xFFF): the centre (Cx, Cy) of the superblock's search in units of 8 luma samples, 12-bit two's
// complement - zero without the quarter-resolution pre-search (av1mi_params.me_presearch), the same for every candidate of a leaf, so the
// minimum is taken over (cost, index) as before.
AV1MI_HD inline int av1mi_sext12(uint32_t v) { return (int)((v & 0xFFFu) ^ 0x800u) - 0x800; }
AV1MI_HD inline void av1mi_me_key_decode(unsigned long long key, int R, int *dy, int *dx, int *cost) {
  const int nc = 2 * R + 1, idx = (int)(key & 0xFFFF);
  *dy = idx / nc - R + 8 * av1mi_sext12((uint32_t)(key >> 52));
  *dx = idx % nc - R + 8 * av1mi_sext12((uint32_t)(key >> 40));
  *cost = (int)((key >> 16) & 0xFFFFFF);
}

// frame f of a chunk is a key frame iff f % keyint == 0
AV1MI_HD inline int av1mi_frame_is_inter(const Av1miDevParams &P, int f) { return P.keyint > 1 && (f % P.keyint) != 0; }

// Per 8x8-unit block info written by the recon kernel, read by entropy + CDEF kernels.
// Only the entry at a block's top-left 8x8 unit carries eobs; mode/skip are replicated over the
// block so neighbour-context lookups are direct.
struct Av1miBlkInfo {
  uint8_t ymode;
  uint8_t skip;
  uint8_t bsl;      // log2 block size in pixels (3..6)
  uint8_t is_inter; // inter frames: 1 = predicted from LAST_FRAME with `mv`
  uint16_t eob[3];
  int16_t mv_row, mv_col;  // 1/8 luma samples
  uint16_t angle;   // bits 0-2: angle delta + 3 of a directional intra mode (3 = none), luma and chroma alike; bit 3: luma IDTX;
                    // bits 4-9 / 10-15: chroma-from-luma alpha U / V (6-bit two's complement; both zero = not chroma from luma)
};
static_assert(sizeof(Av1miBlkInfo) == 16, "block info is one 16-byte record");

#define AV1MI_SB_LEVELS 6144  // int16 levels per superblock: 64*64 + 2*32*32
// Offset of a block's levels inside its superblock's 6144: blocks are stored in Morton (Z) order of their origin's
// 8x8 unit - a block of n x n samples at a quadtree-aligned origin covers (n/8)^2 consecutive Morton indices, so its
// n*n levels are contiguous and blocks of any mix of sizes never overlap.  Luma 64 levels per unit, then U and V with
// 16 per unit.  (bx, by): luma sample position of the block inside the superblock.
AV1MI_HD inline int av1mi_morton8(int ux, int uy) {
  return (ux & 1) | ((uy & 1) << 1) | ((ux & 2) << 1) | ((uy & 2) << 2) | ((ux & 4) << 2) | ((uy & 4) << 3);
}
AV1MI_HD inline int av1mi_levels_off(int plane, int bx, int by) {
  const int m = av1mi_morton8(bx >> 3, by >> 3);
  return plane == 0 ? m * 64 : 4096 + (plane - 1) * 1024 + m * 16;
}

// Default-CDF blob layout (uint16 inverted CDFs, each row n+1 entries: n-1 values, 0, counter)
// for one q context; offsets in uint16 units.  Mirrors the per-tile adaptive state.
struct Av1miCdfLayout {
  enum {
    PARTITION = 0,                         // [16][11]  (8x8 .. 64x64: no 128x128 superblocks)
    KF_Y_MODE = PARTITION + 16 * 11,       // [5][5][14]
    UV_MODE = KF_Y_MODE + 25 * 14,         // [2][13][15]
    ANGLE_DELTA = UV_MODE + 26 * 15,       // [8][8]
    SKIP = ANGLE_DELTA + 64,               // [3][3]
    TX_SET1 = SKIP + 9,                    // [2][13][8]
    TX_SET2 = TX_SET1 + 26 * 8,            // [3][13][6]
    TXB_SKIP = TX_SET2 + 39 * 6,           // [5][13][3]
    EOB16 = TXB_SKIP + 65 * 3,             // [2][2][6]
    EOB32 = EOB16 + 4 * 6,                 // [2][2][7]
    EOB64 = EOB32 + 4 * 7,
    EOB128 = EOB64 + 4 * 8,
    EOB256 = EOB128 + 4 * 9,
    EOB512 = EOB256 + 4 * 10,
    EOB1024 = EOB512 + 4 * 11,
    EOB_EXTRA = EOB1024 + 4 * 12,          // [5][2][9][3]
    DC_SIGN = EOB_EXTRA + 90 * 3,          // [2][3][3]
    COEFF_BASE_EOB = DC_SIGN + 6 * 3,      // [5][2][4][4]
    USE_WIENER = COEFF_BASE_EOB + 40 * 4,  // [3]
    RESTORE_SW = USE_WIENER + 3,           // [4] restoration_type of RESTORE_SWITCHABLE frames: NONE, WIENER, SGRPROJ
    CFL_SIGN = RESTORE_SW + 4,             // [9]
    CFL_ALPHA = CFL_SIGN + 9,              // [6][17]
    COEFF_BASE = CFL_ALPHA + 6 * 17,       // [5][2][42][5]
    COEFF_BR = COEFF_BASE + 420 * 5,       // [5][2][21][5]
    INTRA_TOTAL = COEFF_BR + 210 * 5,      // everything a key frame needs
    // inter frames
    INTER_BASE = INTRA_TOTAL,
    IF_Y_MODE = INTER_BASE,                // [4][14]
    IS_INTER = IF_Y_MODE + 4 * 14,         // [4][3]
    NEWMV = IS_INTER + 4 * 3,              // [6][3]
    GLOBALMV = NEWMV + 6 * 3,              // [2][3]
    REFMV = GLOBALMV + 2 * 3,              // [6][3]
    DRL = REFMV + 6 * 3,                   // [3][3]
    SINGLE_REF = DRL + 3 * 3,              // [6 p1..p6][3 ctx][3]
    INTER_TX1 = SINGLE_REF + 18 * 3,       // [2][17]
    INTER_TX2 = INTER_TX1 + 2 * 17,        // [13]
    INTER_TX3 = INTER_TX2 + 13,            // [4][3]
    MV_JOINT = INTER_TX3 + 4 * 3,          // [5]
    MV_COMP = MV_JOINT + 5,                // [2] x { class[12], class0_fp[2][5], fp[5], sign[3], class0_hp[3], hp[3], class0[3], bits[10][3] }
    MVC_CLASS = 0, MVC_CLASS0_FP = 12, MVC_FP = 22, MVC_SIGN = 27, MVC_CLASS0_HP = 30, MVC_HP = 33, MVC_CLASS0 = 36, MVC_BITS = 39,
    MVC_SIZE = 69,
    TOTAL = MV_COMP + 2 * MVC_SIZE
  };
};

#endif
