// av1mi_host.cpp - host side of libav1mi behind the C ABI of include/av1mi.h.
//
// Mirrors the reference's encode boundary: `run_av1an(&Av1anEncodeParams) -> Result<(), EncodeError>`
// (/root/reference/crates/daemon/src/encode/av1an.rs:126-139) becomes av1mi_encode_file(); the
// per-chunk work av1an farms out to SVT-AV1 workers (`--workers`, av1an.rs:100-101) becomes
// av1mi_encode_chunk() on one context (= one GPU + one HIP stream).  No CPU encode path exists
// here: without a HIP device every entry point fails with AV1MI_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <errno.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <cstddef>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/av1mi.h"
#include "av1_tables.h"
#include "av1mi_dev.h"

extern "C" {
hipError_t av1mi_launch_partition(const Av1miDevParams *P, const void *frames, uint32_t *part, hipStream_t stream);
// one translation unit per (largest leaf, sample type): recon_kernel.hip, recon8_kernel.hip, recon64_kernel.hip, recon64_8_kernel.hip
#define AV1MI_RECON_PROTO(name) hipError_t name(const Av1miDevParams *P, const Av1miDevParams *dP, const void *src, void *rec, int16_t *levels, \
                                               Av1miBlkInfo *blk, const void *ref, const unsigned long long *me_best, const uint32_t *part, hipStream_t s)
AV1MI_RECON_PROTO(av1mi_launch_recon_u16);
AV1MI_RECON_PROTO(av1mi_launch_recon_u8);
AV1MI_RECON_PROTO(av1mi_launch_recon64_u16);
AV1MI_RECON_PROTO(av1mi_launch_recon64_u8);
hipError_t av1mi_launch_subpel_refine(const Av1miDevParams *P, const void *frames, const unsigned long long *best, unsigned long long *refined,
                                      int me_range, int frame0, int count, hipStream_t stream);
hipError_t av1mi_launch_motion_search(const Av1miDevParams *P, const void *frames, unsigned long long *best, int me_range, int frame0,
                                      int count, uint32_t *acc64, const uint32_t *centre, hipStream_t stream);
hipError_t av1mi_launch_quarter_luma(const Av1miDevParams *P, const void *frames, uint16_t *quarter, hipStream_t stream);
hipError_t av1mi_launch_presearch(const Av1miDevParams *P, const uint16_t *quarter, uint32_t *centre, int frame0, int count, hipStream_t stream);
hipError_t av1mi_launch_luma_sad(const Av1miDevParams *P, const void *frames, const void *prev0, unsigned long long *sad, hipStream_t s);
hipError_t av1mi_launch_pad(const void *in, void *out, int w, int h, int cw, int ch, int bit_depth, int n_frames, int crop, hipStream_t s);
hipError_t av1mi_launch_deblock(const Av1miDevParams *P, void *rec, const Av1miBlkInfo *blk, hipStream_t s);
hipError_t av1mi_launch_lr(const Av1miDevParams *P, const void *pre, const void *cdef, const void *src, void *out, uint8_t *choice,
                           unsigned long long *unit_sse, int clear, hipStream_t s);
hipError_t av1mi_launch_entropy(const Av1miDevParams *P, const uint16_t *cdf_init, const int16_t *levels, const Av1miBlkInfo *blk,
                                uint32_t *streams, uint32_t *stream_len, uint32_t *tile_combos, uint8_t *slots, uint32_t *tile_bytes,
                                const uint8_t *lr_choice, uint32_t *tile_order, int frame0, int count, int phase,
                                hipStream_t s, hipEvent_t mid, hipStream_t aux, hipEvent_t fork, hipEvent_t join);
hipError_t av1mi_launch_cdef(const Av1miDevParams *P, const void *rec, void *fin, const Av1miBlkInfo *blk, const uint16_t *dirtab, const void *src,
                             unsigned long long *sse, hipStream_t s);
hipError_t av1mi_launch_cdef_dir(const Av1miDevParams *P, const void *rec, const Av1miBlkInfo *blk, uint16_t *dirtab, hipStream_t s);
hipError_t av1mi_launch_sse(const Av1miDevParams *P, const void *a, const void *b, unsigned long long *sse, hipStream_t s);
hipError_t av1mi_launch_pack(const Av1miDevParams *P, const uint8_t *slots, const uint32_t *tile_bytes, uint32_t *tile_off,
                             uint32_t *frame_size, uint32_t *payload_size, unsigned long long *frame_off, const uint8_t *hdr_blob,
                             uint8_t *out, int *overflow, int stage, hipStream_t s);
}

namespace {

// ------------------------------------------------------------------ header bit writer
struct BitWriter {
  std::vector<uint8_t> buf;
  size_t bits = 0;
  void put(uint32_t v, int n) {
    for (int i = n - 1; i >= 0; i--) {
      if ((bits & 7) == 0) buf.push_back(0);
      buf.back() |= (uint8_t)(((v >> i) & 1) << (7 - (bits & 7)));
      bits++;
    }
  }
  void align() { while (bits & 7) put(0, 1); }
  void trailing() { put(1, 1); align(); }
  // ns(n), AV1 spec §4.10.7
  void put_ns(int n, int v) {
    int w = 0;
    while ((1 << w) <= n) w++;  // w = floor(log2 n) + 1
    int m = (1 << w) - n;
    if (v < m) put((uint32_t)v, w - 1);
    else { int x = v + m; put((uint32_t)(x >> 1), w - 1); put((uint32_t)(x & 1), 1); }
  }
};

int tile_log2(int blk, int target) { int k = 0; while ((blk << k) < target) k++; return k; }
int bits_for(unsigned v) { int n = 0; while (v) { n++; v >>= 1; } return n ? n : 1; }

struct Resolved {
  av1mi_params p;
  int qidx;
  int cw, ch;                         // coded size: p.width / p.height rounded up to multiples of 8
  int sb_cols, sb_rows;
  int tile_sb, tile_cols, tile_rows;  // tiles of tile_sb x tile_sb superblocks (1, or 2 beyond 64 superblocks either way)
  int qm_level;                       // quantiser-matrix level of all planes (15 = flat); only meaningful with p.enable_qm
};

// aom's quantizer_to_qindex[] (CQ level -> base_q_idx); 30 -> 120
const uint8_t kQuantizerToQindex[64] = {
  0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76, 80, 84, 88, 92, 96, 100, 104, 108, 112, 116, 120, 124,
  128, 132, 136, 140, 144, 148, 152, 156, 160, 164, 168, 172, 176, 180, 184, 188, 192, 196, 200, 204, 208, 212, 216, 220, 224, 228, 232, 236, 240, 244, 249, 255 };

int resolve(const av1mi_params *in, Resolved *r) {
  if (!in) return AV1MI_E_INVALID_ARG;
  r->p = *in;
  av1mi_params &p = r->p;
  if (p.width < 8 || p.height < 8 || (p.width & 1) || (p.height & 1) || p.width > 65536 || p.height > 65536) return AV1MI_E_INVALID_ARG;
  r->cw = (int)((p.width + 7) & ~7u); r->ch = (int)((p.height + 7) & ~7u);
  if (p.bit_depth != 8 && p.bit_depth != 10) return AV1MI_E_INVALID_ARG;
  if (p.cq_level > 63 || p.film_grain > 50) return AV1MI_E_INVALID_ARG;
  if (p.keyint == 0) p.keyint = 1;
  if (p.me_range == 0) p.me_range = 8;
  if (p.me_range != 8 && p.me_range != 16) return AV1MI_E_INVALID_ARG;
  if (p.block_log2 == 0) p.block_log2 = 5;
  if (p.block_log2 < 3 || p.block_log2 > 6) return AV1MI_E_INVALID_ARG;
  if (p.cdef_damping == 0) { p.cdef_y_pri = 2; p.cdef_y_sec = 0; p.cdef_uv_pri = 1; p.cdef_uv_sec = 0; p.cdef_damping = 5; }
  if (p.cdef_damping < 3 || p.cdef_damping > 6 || p.cdef_y_pri > 15 || p.cdef_uv_pri > 15 || p.cdef_y_sec > 3 || p.cdef_uv_sec > 3) return AV1MI_E_INVALID_ARG;
  r->qidx = kQuantizerToQindex[p.cq_level];
  if (p.subpel > 1 || p.enable_lr > 2 || p.color_range > 1 || p.intra_angle_delta > 1 || p.intra_edge_filter > 1 || p.cfl > 1 || p.tx_search > 1) return AV1MI_E_INVALID_ARG;
  if (p.partition_search > 1 || p.me_presearch > 1) return AV1MI_E_INVALID_ARG;
  if (p.min_block_log2 == 0) p.min_block_log2 = 3;
  if (p.min_block_log2 < 3 || p.min_block_log2 > p.block_log2) return AV1MI_E_INVALID_ARG;
  if (p.color_primaries > 255 || p.transfer_characteristics > 255 || p.matrix_coefficients > 255) return AV1MI_E_INVALID_ARG;
  // CP_BT_709 / TC_SRGB / MC_IDENTITY switches the syntax to 4:4:4 with no color_range bit (spec 5.5.2): not a 4:2:0 description
  if (p.color_primaries == 1 && p.transfer_characteristics == 13 && p.matrix_coefficients == 0) return AV1MI_E_INVALID_ARG;
  if (p.enable_qm > 1 || p.qm_min > 15 || p.qm_max > 15 || (p.enable_qm && p.qm_min > p.qm_max)) return AV1MI_E_INVALID_ARG;
  // level from the quantiser index, as SVT-AV1 / libaom derive it ("--qm-min", "--qm-max")
  r->qm_level = p.enable_qm ? (int)(p.qm_min + (uint32_t)r->qidx * (p.qm_max + 1 - p.qm_min) / 256) : 15;
  r->sb_cols = (r->cw + 63) / 64;
  r->sb_rows = (r->ch + 63) / 64;
  // AV1 allows at most 64 x 64 tiles: frames beyond 64 superblocks either way (8K) use tiles of 2 x 2 superblocks
  if (p.tile_sb > 2) return AV1MI_E_INVALID_ARG;
  r->tile_sb = p.tile_sb ? (int)p.tile_sb : ((r->sb_cols > 64 || r->sb_rows > 64) ? 2 : 1);
  r->tile_cols = (r->sb_cols + r->tile_sb - 1) / r->tile_sb;
  r->tile_rows = (r->sb_rows + r->tile_sb - 1) / r->tile_sb;
  if (r->tile_cols > 64 || r->tile_rows > 64) return AV1MI_E_UNSUPPORTED;
  return AV1MI_OK;
}

// deblocking level of a frame (the same for all four filters): libaom's "pick from q" rule; 0 = filter off
int deblock_level(const Resolved &r, bool key) {
  if (!r.p.deblock) return 0;
  const int q = r.p.bit_depth == 8 ? av1_ac_q8[r.qidx] : av1_ac_q10[r.qidx];
  int g = r.p.bit_depth == 8 ? (q * 20723 + 1015158) >> 18 : (q * 20723 + 4060632) >> 20;
  if (key) g -= 4;
  return g < 0 ? 0 : (g > 63 ? 63 : g);
}

// sequence_header_obu (AV1 spec §5.5), complete OBU incl. header and size
std::vector<uint8_t> make_sequence_header(const Resolved &r) {
  const av1mi_params &p = r.p;
  BitWriter b;
  const int wbits = bits_for(p.width - 1), hbits = bits_for(p.height - 1);
  b.put(0, 3);   // seq_profile
  b.put(0, 1);   // still_picture
  b.put(0, 1);   // reduced_still_picture_header
  b.put(0, 1);   // timing_info_present_flag
  b.put(0, 1);   // initial_display_delay_present_flag
  b.put(0, 5);   // operating_points_cnt_minus_1
  b.put(0, 12);  // operating_point_idc[0]
  b.put(31, 5);  // seq_level_idx[0]: maximum parameters
  b.put(0, 1);   // seq_tier[0]
  b.put((uint32_t)(wbits - 1), 4);
  b.put((uint32_t)(hbits - 1), 4);
  b.put(p.width - 1, wbits);
  b.put(p.height - 1, hbits);
  b.put(0, 1);  // frame_id_numbers_present_flag
  b.put(0, 1);  // use_128x128_superblock
  b.put(0, 1);  // enable_filter_intra
  b.put(p.intra_edge_filter ? 1 : 0, 1);  // enable_intra_edge_filter
  b.put(0, 1);  // enable_interintra_compound
  b.put(0, 1);  // enable_masked_compound
  b.put(0, 1);  // enable_warped_motion
  b.put(0, 1);  // enable_dual_filter
  b.put(0, 1);  // enable_order_hint
  b.put(0, 1);  // seq_choose_screen_content_tools
  b.put(0, 1);  // seq_force_screen_content_tools
  b.put(0, 1);  // enable_superres
  b.put(p.enable_cdef ? 1 : 0, 1);
  b.put(p.enable_lr ? 1 : 0, 1);  // enable_restoration
  // color_config
  b.put(p.bit_depth > 8, 1);  // high_bitdepth
  b.put(0, 1);                // mono_chrome
  {
    const bool desc = p.color_primaries || p.transfer_characteristics || p.matrix_coefficients;
    b.put(desc, 1);           // color_description_present_flag (absent: CP / TC / MC = 2 "unspecified")
    if (desc) { b.put(p.color_primaries, 8); b.put(p.transfer_characteristics, 8); b.put(p.matrix_coefficients, 8); }
  }
  b.put(p.color_range ? 1 : 0, 1);  // color_range: 0 = studio (limited) range, 1 = full range
  b.put(0, 2);                // chroma_sample_position
  b.put(0, 1);                // separate_uv_delta_q
  b.put(p.film_grain ? 1 : 0, 1);  // film_grain_params_present
  b.trailing();
  std::vector<uint8_t> out;
  out.push_back((1 << 3) | 2);
  out.push_back((uint8_t)b.buf.size());  // < 128
  out.insert(out.end(), b.buf.begin(), b.buf.end());
  return out;
}

// OBU_FRAME payload up to the first tile: frame_header_obu (§5.9) + byte_alignment +
// tile_group_obu's tile_start_and_end_present_flag + byte_alignment (§5.11.1)
std::vector<uint8_t> make_frame_header(const Resolved &r, size_t *hdr_bits, uint32_t frame_number = 0, bool inter = false) {
  const av1mi_params &p = r.p;
  BitWriter b;
  b.put(0, 1);  // show_existing_frame
  b.put(inter ? 1 : 0, 2);  // frame_type KEY_FRAME / INTER_FRAME
  b.put(1, 1);  // show_frame
  if (inter) b.put(0, 1);  // error_resilient_mode (implied 1 on shown key frames)
  b.put(p.cdf_update ? 0 : 1, 1);  // disable_cdf_update
  b.put(0, 1);  // frame_size_override_flag
  if (inter) {
    b.put(7, 3);     // primary_ref_frame = NONE: every frame starts from the default CDFs (tiles of all frames stay independent)
    b.put(0x01, 8);  // refresh_frame_flags: this frame replaces slot 0
    for (int i = 0; i < 7; i++) b.put(0, 3);  // ref_frame_idx[i] = 0: every reference name -> the previous frame
  }
  b.put(0, 1);  // render_and_frame_size_different
  if (inter) {
    b.put(0, 1);  // allow_high_precision_mv
    b.put(0, 1);  // is_filter_switchable
    b.put(r.p.subpel ? 0 : 3, 2);  // interpolation_filter: EIGHTTAP with sub-sample vectors, else BILINEAR
    b.put(0, 1);  // is_motion_mode_switchable
  }
  if (p.cdf_update) b.put(1, 1);  // disable_frame_end_update_cdf
  // tile_info (§5.9.15): explicit spacing, tiles of tile_sb x tile_sb superblocks (the last row/column may be short)
  {
    const int sb_cols = r.sb_cols, sb_rows = r.sb_rows, ts = r.tile_sb;
    const int max_tile_width_sb = 4096 >> 6, max_tile_area_sb = (4096 * 2304) >> 12;
    int min_log2_tile_cols = tile_log2(max_tile_width_sb, sb_cols);
    int min_log2_tiles = tile_log2(max_tile_area_sb, sb_rows * sb_cols);
    if (min_log2_tiles < min_log2_tile_cols) min_log2_tiles = min_log2_tile_cols;
    b.put(0, 1);  // uniform_tile_spacing_flag
    int widest = 0;
    for (int start = 0; start < sb_cols; start += ts) {
      const int max_w = sb_cols - start < max_tile_width_sb ? sb_cols - start : max_tile_width_sb;
      const int sz = sb_cols - start < ts ? sb_cols - start : ts;
      b.put_ns(max_w, sz - 1);  // width_in_sbs_minus_1
      if (sz > widest) widest = sz;
    }
    int area = sb_rows * sb_cols;
    if (min_log2_tiles > 0) area >>= (min_log2_tiles + 1);
    int max_tile_height_sb = area / widest;
    if (max_tile_height_sb < 1) max_tile_height_sb = 1;
    for (int start = 0; start < sb_rows; start += ts) {
      const int max_h = sb_rows - start < max_tile_height_sb ? sb_rows - start : max_tile_height_sb;
      const int sz = sb_rows - start < ts ? sb_rows - start : ts;
      b.put_ns(max_h, sz - 1);  // height_in_sbs_minus_1
    }
    const int cl = tile_log2(1, r.tile_cols), rl = tile_log2(1, r.tile_rows);
    if (cl > 0 || rl > 0) {
      b.put(0, cl + rl);  // context_update_tile_id
      b.put(3, 2);        // tile_size_bytes_minus_1
    }
  }
  b.put((uint32_t)r.qidx, 8);
  b.put(0, 1);  // DeltaQYDc
  b.put(0, 1);  // DeltaQUDc
  b.put(0, 1);  // DeltaQUAc
  b.put(r.p.enable_qm ? 1 : 0, 1);  // using_qmatrix
  if (r.p.enable_qm) { b.put((uint32_t)r.qm_level, 4); b.put((uint32_t)r.qm_level, 4); }  // qm_y, qm_u (= qm_v: separate_uv_delta_q = 0)
  b.put(0, 1);  // segmentation_enabled
  if (r.qidx > 0) b.put(0, 1);  // delta_q_present
  {  // loop_filter_params (§5.9.11): level 0 = deblocking off
    const uint32_t lv = (uint32_t)deblock_level(r, !inter);
    b.put(lv, 6); b.put(lv, 6);          // loop_filter_level[0..1]
    if (lv) { b.put(lv, 6); b.put(lv, 6); }  // [2..3] (U, V)
  }
  b.put(0, 3);  // loop_filter_sharpness
  b.put(0, 1);  // loop_filter_delta_enabled
  if (p.enable_cdef) {
    b.put(p.cdef_damping - 3, 2);
    b.put(0, 2);  // cdef_bits
    b.put(p.cdef_y_pri, 4); b.put(p.cdef_y_sec, 2);
    b.put(p.cdef_uv_pri, 4); b.put(p.cdef_uv_sec, 2);
  }
  if (p.enable_lr) {  // lr_params (§5.9.20): luma RESTORE_WIENER (lr_type 2), chroma none, lr_unit_shift 0 = 64x64 units
    b.put(p.enable_lr == 2 ? 1 : 2, 2); b.put(0, 2); b.put(0, 2);  // luma lr_type: 1 = RESTORE_SWITCHABLE, 2 = RESTORE_WIENER
    b.put(0, 1);
  }
  b.put(0, 1);  // tx_mode_select = 0: TX_MODE_LARGEST
  if (inter) b.put(0, 1);  // reference_select = 0
  b.put(0, 1);  // reduced_tx_set
  if (inter) for (int i = 0; i < 7; i++) b.put(0, 1);  // global_motion_params: is_global = 0
  if (p.film_grain) {  // film_grain_params (§5.9.30): fixed table, per-frame seed
    const uint32_t sy = p.film_grain * 2 > 255 ? 255 : p.film_grain * 2, sc = p.film_grain;
    b.put(1, 1);                                              // apply_grain
    b.put((7391u + 173u * (p.first_frame + frame_number)) & 0xFFFFu, 16);  // grain_seed
    if (inter) b.put(1, 1);                                   // update_grain (implied on key frames)
    b.put(2, 4); b.put(0, 8); b.put(sy, 8); b.put(255, 8); b.put(sy, 8);  // num_y_points + points
    b.put(0, 1);                                              // chroma_scaling_from_luma
    for (int pl = 0; pl < 2; pl++) { b.put(2, 4); b.put(0, 8); b.put(sc, 8); b.put(255, 8); b.put(sc, 8); }
    b.put(3, 2);                                              // grain_scaling_minus_8
    b.put(0, 2);                                              // ar_coeff_lag
    b.put(128, 8); b.put(128, 8);                             // ar_coeffs_cb/cr_plus_128[0]
    b.put(0, 2);                                              // ar_coeff_shift_minus_6
    b.put(0, 2);                                              // grain_scale_shift
    for (int pl = 0; pl < 2; pl++) { b.put(128, 8); b.put(192, 8); b.put(256, 9); }  // mult, luma_mult, offset
    b.put(1, 1);                                              // overlap_flag
    b.put(0, 1);                                              // clip_to_restricted_range
  }
  if (hdr_bits) *hdr_bits = b.bits;
  b.align();
  if (r.tile_cols * r.tile_rows > 1) { b.put(0, 1); b.align(); }  // tile_start_and_end_present_flag
  return b.buf;
}

// ---- loop restoration unit syntax (§5.11.58): the literal bits that code a Wiener coefficient set against the
// tile-start reference Wiener_Taps_Mid = {3, -7, 15} (tiles are one superblock = one unit, so that is always the
// reference).  Mirrors decode_signed_subexp_with_ref_bool / decode_subexp_bool / NS / inverse_recenter.
struct BitString { unsigned long long bits = 0; int len = 0; void put(unsigned v, int n) { for (int i = n - 1; i >= 0; i--) { bits = (bits << 1) | ((v >> i) & 1); len++; } } };
void lr_put_ns(BitString &b, int n, int v) {
  int w = 0, x = n;
  while (x) { w++; x >>= 1; }
  const int m = (1 << w) - n;
  if (v < m) b.put((unsigned)v, w - 1);
  else { const int extra = v + m; b.put((unsigned)(extra >> 1), w - 1); b.put((unsigned)(extra & 1), 1); }
}
void lr_put_subexp(BitString &b, int num_syms, int k, int v) {
  int i = 0, mk = 0;
  for (;;) {
    const int b2 = i ? k + i - 1 : k, a = 1 << b2;
    if (num_syms <= mk + 3 * a) { lr_put_ns(b, num_syms - mk, v - mk); return; }
    if (v >= mk + a) { b.put(1, 1); i++; mk += a; }
    else { b.put(0, 1); b.put((unsigned)(v - mk), b2); return; }
  }
}
int lr_recenter(int r, int v) { return v > 2 * r ? v : (v >= r ? (v - r) << 1 : ((r - v) << 1) - 1); }
void lr_put_signed_ref(BitString &b, int low, int high, int k, int r, int v) {
  const int mx = high - low, x = v - low, rr = r - low;
  if ((rr << 1) <= mx) lr_put_subexp(b, mx, k, lr_recenter(rr, x));
  else lr_put_subexp(b, mx, k, lr_recenter(mx - 1 - rr, mx - 1 - x));
}
const int8_t kWienerCand[3][3] = { { 0, 0, -4 }, { 1, -3, -6 }, { 3, -7, 15 } };  // == lr_kernel.hip, oracle/av1o_lr.c
const int8_t kSgrCand[3][3] = { { 9, 31, 31 }, { 9, 0, 31 }, { 9, 31, 95 } };     // { lr_sgr_set, xqd0, xqd1 }: == lr_kernel.hip, oracle/av1o_lr.c
// self-guided unit (§5.11.58): lr_sgr_set L(4), then the two weights against RefSgrXqd (ref 0 = Sgrproj_Xqd_Mid at the tile
// start, r = candidate r-1: the previous self-guided unit of the tile).  Every candidate uses set 9, whose radii are both
// non-zero, so both weights are always coded.
BitString sgr_code_of(int ref, int cand) {
  static const int xmin[2] = { -96, -32 }, xmax[2] = { 31, 95 }, mid[2] = { -32, 31 };
  BitString b;
  b.put((unsigned)kSgrCand[cand][0], 4);
  for (int i = 0; i < 2; i++) lr_put_signed_ref(b, xmin[i], xmax[i] + 1, 4, ref ? kSgrCand[ref - 1][1 + i] : mid[i], kSgrCand[cand][1 + i]);
  return b;
}
// ref: 0 = Wiener_Taps_Mid (tile start), r = candidate r-1 (the previous unit of the tile that was coded with a filter)
BitString lr_code_of(int ref, int cand) {
  static const int tmin[3] = { -5, -23, -17 }, tmax[3] = { 10, 8, 46 }, tk[3] = { 1, 2, 3 }, mid[3] = { 3, -7, 15 };
  BitString b;
  for (int pass = 0; pass < 2; pass++)
    for (int j = 0; j < 3; j++) lr_put_signed_ref(b, tmin[j], tmax[j] + 1, tk[j], ref ? kWienerCand[ref - 1][j] : mid[j], kWienerCand[cand][j]);
  return b;
}

// default CDF blob for one q context in the layout of Av1miCdfLayout (inverted, 0, counter)
template <size_t W>
void emit_rows(std::vector<uint16_t> &v, size_t off, const uint16_t (*rows)[W], int nrows, int row_stride, const int *nsym, int nsym_const) {
  for (int i = 0; i < nrows; i++) {
    int n = nsym ? nsym[i] : nsym_const;
    for (int k = 0; k < n - 1; k++) v[off + (size_t)i * row_stride + k] = (uint16_t)(32768 - rows[i][k]);
  }
}
std::vector<uint16_t> make_cdf_blob(int qidx) {
  typedef Av1miCdfLayout CL;
  const int q = qidx <= 20 ? 0 : (qidx <= 60 ? 1 : (qidx <= 120 ? 2 : 3));
  std::vector<uint16_t> v(CL::TOTAL, 0);
  int pn[20];
  for (int i = 0; i < 20; i++) pn[i] = i < 4 ? 4 : (i < 16 ? 10 : 8);
  emit_rows(v, CL::PARTITION, av1_default_partition_cdf, 16, 11, pn, 0);
  emit_rows(v, CL::KF_Y_MODE, &av1_default_kf_y_mode_cdf[0][0], 25, 14, nullptr, 13);
  emit_rows(v, CL::UV_MODE, av1_default_uv_mode_nocfl_cdf, 13, 15, nullptr, 13);
  emit_rows(v, CL::UV_MODE + 13 * 15, av1_default_uv_mode_cfl_cdf, 13, 15, nullptr, 14);
  emit_rows(v, CL::ANGLE_DELTA, av1_default_angle_delta_cdf, 8, 8, nullptr, 7);
  emit_rows(v, CL::SKIP, av1_default_skip_cdf, 3, 3, nullptr, 2);
  emit_rows(v, CL::TX_SET1, &av1_default_intra_tx_set1_cdf[0][0], 26, 8, nullptr, 7);
  emit_rows(v, CL::TX_SET2, &av1_default_intra_tx_set2_cdf[0][0], 39, 6, nullptr, 5);
  emit_rows(v, CL::TXB_SKIP, &av1_default_txb_skip_cdf[q][0][0], 65, 3, nullptr, 2);
  emit_rows(v, CL::EOB16, &av1_default_eob_multi16_cdf[q][0][0], 4, 6, nullptr, 5);
  emit_rows(v, CL::EOB32, &av1_default_eob_multi32_cdf[q][0][0], 4, 7, nullptr, 6);
  emit_rows(v, CL::EOB64, &av1_default_eob_multi64_cdf[q][0][0], 4, 8, nullptr, 7);
  emit_rows(v, CL::EOB128, &av1_default_eob_multi128_cdf[q][0][0], 4, 9, nullptr, 8);
  emit_rows(v, CL::EOB256, &av1_default_eob_multi256_cdf[q][0][0], 4, 10, nullptr, 9);
  emit_rows(v, CL::EOB512, &av1_default_eob_multi512_cdf[q][0][0], 4, 11, nullptr, 10);
  emit_rows(v, CL::EOB1024, &av1_default_eob_multi1024_cdf[q][0][0], 4, 12, nullptr, 11);
  emit_rows(v, CL::EOB_EXTRA, &av1_default_eob_extra_cdf[q][0][0][0], 90, 3, nullptr, 2);
  emit_rows(v, CL::DC_SIGN, &av1_default_dc_sign_cdf[q][0][0], 6, 3, nullptr, 2);
  emit_rows(v, CL::COEFF_BASE_EOB, &av1_default_coeff_base_eob_cdf[q][0][0][0], 40, 4, nullptr, 3);
  emit_rows(v, CL::USE_WIENER, av1_default_use_wiener_cdf, 1, 3, nullptr, 2);
  emit_rows(v, CL::RESTORE_SW, av1_default_switchable_restore_cdf, 1, 4, nullptr, 3);
  emit_rows(v, CL::CFL_SIGN, av1_default_cfl_sign_cdf, 1, 9, nullptr, 8);
  emit_rows(v, CL::CFL_ALPHA, av1_default_cfl_alpha_cdf, 6, 17, nullptr, 16);
  // inter frames
  emit_rows(v, CL::IF_Y_MODE, av1_default_if_y_mode_cdf, 4, 14, nullptr, 13);
  emit_rows(v, CL::IS_INTER, av1_default_is_inter_cdf, 4, 3, nullptr, 2);
  emit_rows(v, CL::NEWMV, av1_default_newmv_cdf, 6, 3, nullptr, 2);
  emit_rows(v, CL::GLOBALMV, av1_default_globalmv_cdf, 2, 3, nullptr, 2);
  emit_rows(v, CL::REFMV, av1_default_refmv_cdf, 6, 3, nullptr, 2);
  emit_rows(v, CL::DRL, av1_default_drl_cdf, 3, 3, nullptr, 2);
  emit_rows(v, CL::SINGLE_REF, &av1_default_single_ref_cdf[0][0], 18, 3, nullptr, 2);
  emit_rows(v, CL::INTER_TX1, av1_default_inter_tx_set1_cdf, 2, 17, nullptr, 16);
  emit_rows(v, CL::INTER_TX2, av1_default_inter_tx_set2_cdf, 1, 13, nullptr, 12);
  emit_rows(v, CL::INTER_TX3, av1_default_inter_tx_set3_cdf, 4, 3, nullptr, 2);
  emit_rows(v, CL::MV_JOINT, av1_default_mv_joint_cdf, 1, 5, nullptr, 4);
  for (int c = 0; c < 2; c++) {
    const size_t b0 = CL::MV_COMP + (size_t)c * CL::MVC_SIZE;
    emit_rows(v, b0 + CL::MVC_CLASS, av1_default_mv_class_cdf, 1, 12, nullptr, 11);
    emit_rows(v, b0 + CL::MVC_CLASS0_FP, av1_default_mv_class0_fp_cdf, 2, 5, nullptr, 4);
    emit_rows(v, b0 + CL::MVC_FP, av1_default_mv_fp_cdf, 1, 5, nullptr, 4);
    emit_rows(v, b0 + CL::MVC_SIGN, av1_default_mv_sign_cdf, 1, 3, nullptr, 2);
    emit_rows(v, b0 + CL::MVC_CLASS0_HP, av1_default_mv_class0_hp_cdf, 1, 3, nullptr, 2);
    emit_rows(v, b0 + CL::MVC_HP, av1_default_mv_hp_cdf, 1, 3, nullptr, 2);
    emit_rows(v, b0 + CL::MVC_CLASS0, av1_default_mv_class0_cdf, 1, 3, nullptr, 2);
    emit_rows(v, b0 + CL::MVC_BITS, av1_default_mv_bits_cdf, 10, 3, nullptr, 2);
  }
  emit_rows(v, CL::COEFF_BASE, &av1_default_coeff_base_cdf[q][0][0][0], 420, 5, nullptr, 4);
  emit_rows(v, CL::COEFF_BR, &av1_default_coeff_br_cdf[q][0][0][0], 210, 5, nullptr, 4);
  return v;
}

}  // namespace

// ------------------------------------------------------------------ context
struct av1mi_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  bool pooled_streams = false;    // the streams go back to the process-wide pool (av1mi_ctx_create)
  hipStream_t stream2 = nullptr;  // CDEF + SSE run here, beside the entropy kernels on `stream`
  hipStream_t stream3 = nullptr;  // inter chunks: entropy coding of finished groups of frames, beside the frame-by-frame chain
  hipStream_t stream4 = nullptr;  // all-key-frame chunks pipelined over groups: the groups' entropy coding alternates between stream3 and this one
  hipEvent_t ev[14] = {};
  std::vector<hipEvent_t> me_ev;       // per frame: its motion search has finished (second stream -> main stream)
  std::vector<hipEvent_t> grp_ev;      // per group of frames of an inter chunk: reconstructed (main stream -> third stream)
  std::string err;
  // workspace (device)
  size_t cap_frames = 0;
  int cap_scale = 1;   // per-tile symbol-stream / bitstream-slot capacity multiplier (1, 2, 4, ... 64)
  int ws_scale = 0;    // the multiplier the workspace was allocated with
  Resolved res = {};
  void *d_src = nullptr, *d_rec = nullptr, *d_fin = nullptr;
  int16_t *d_levels = nullptr;
  Av1miBlkInfo *d_blk = nullptr;
  uint8_t *d_slots = nullptr, *d_out = nullptr, *d_hdr = nullptr;
  uint16_t *d_cdf = nullptr;
  Av1miDevParams *d_params = nullptr;  // copy of P for the kernels that read their parameters through a pointer (recon)
  uint32_t *d_tile_bytes = nullptr, *d_tile_off = nullptr, *d_frame_size = nullptr, *d_payload = nullptr, *d_sym = nullptr;
  uint32_t *d_streams = nullptr, *d_combos = nullptr;
  unsigned long long *d_frame_off = nullptr, *d_sse = nullptr;
  int *d_overflow = nullptr;
  unsigned long long *d_me = nullptr;  // motion search results per 8x8 unit per frame
  void *d_stage = nullptr;             // sizes that are not multiples of 8: frames in the caller's tight layout (input / reconstruction out)
  void *d_cd = nullptr;                // loop restoration on: CDEF output (d_fin then holds the restored frames)
  unsigned long long *d_me_sub = nullptr;  // sub-sample refinement: refined [frame][8x8 unit] keys (me_kernel.hip)
  uint16_t *d_cdefdir = nullptr;           // chunk-wide CDEF: {adjusted luma primary strength << 3 | direction} per 8x8 block (cdef_dir_kernel)
  uint16_t *d_quarter = nullptr;           // me_presearch: quarter-resolution luma of every frame of the chunk
  uint32_t *d_centre = nullptr;            // me_presearch: centre code of the full search per [frame][superblock]
  uint32_t *d_part = nullptr;              // content-driven partition: split mask per [frame][superblock] (partition_kernel)
  uint32_t *d_me64 = nullptr;              // 64x64 leaves: the search's [frame][superblock][candidate] SAD table
  size_t me64_bytes = 0;
  Av1miQmEntry *d_qm = nullptr;        // quantiser-matrix steps (Av1miDevParams::qm_tab), valid for qm_key = (level, qidx, bit depth)
  std::vector<Av1miQmEntry> h_qm;
  int qm_key = -1;
  uint8_t *d_lrc = nullptr;            // per restoration unit: 0 = off, k = candidate k-1
  unsigned long long *d_lrsse = nullptr;  // per restoration unit: the candidates' SSE sums (scratch of lr_kernel.hip's two phases)
  size_t out_cap = 0;
  // host staging (pinned)
  uint8_t *h_out = nullptr;
  size_t h_out_cap = 0;
  Av1miDevParams P = {};
};

namespace {

void set_err(av1mi_ctx *c, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
}

#define HIPCHK(c, call)                                                                          \
  do {                                                                                           \
    hipError_t e__ = (call);                                                                     \
    if (e__ != hipSuccess) {                                                                     \
      set_err((c), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__);   \
      return e__ == hipErrorOutOfMemory ? AV1MI_E_OOM : AV1MI_E_HIP;                             \
    }                                                                                            \
  } while (0)

void free_workspace(av1mi_ctx *c) {
  void *ptrs[] = { c->d_src, c->d_rec, c->d_fin, c->d_levels, c->d_blk, c->d_slots, c->d_out, c->d_hdr, c->d_cdf, c->d_tile_bytes,
                   c->d_tile_off, c->d_frame_size, c->d_payload, c->d_sym, c->d_frame_off, c->d_sse, c->d_overflow, c->d_streams, c->d_combos, c->d_me, c->d_cd, c->d_lrc, c->d_stage, c->d_qm, c->d_me_sub, c->d_lrsse, c->d_params, c->d_me64, c->d_cdefdir, c->d_part, c->d_quarter, c->d_centre };
  for (void *p : ptrs) if (p) (void)hipFree(p);
  c->d_src = c->d_rec = c->d_fin = nullptr; c->d_levels = nullptr; c->d_blk = nullptr; c->d_slots = c->d_out = c->d_hdr = nullptr;
  c->d_cdf = nullptr; c->d_tile_bytes = c->d_tile_off = c->d_frame_size = c->d_payload = c->d_sym = nullptr;
  c->d_frame_off = c->d_sse = nullptr; c->d_overflow = nullptr; c->d_streams = c->d_combos = nullptr; c->d_me = nullptr; c->d_cd = nullptr; c->d_lrc = nullptr; c->d_stage = nullptr; c->d_qm = nullptr; c->qm_key = -1; c->d_me_sub = nullptr; c->d_lrsse = nullptr; c->d_params = nullptr; c->d_me64 = nullptr; c->me64_bytes = 0; c->d_cdefdir = nullptr; c->d_part = nullptr; c->d_quarter = nullptr; c->d_centre = nullptr;
  if (c->h_out) (void)hipHostFree(c->h_out);
  c->h_out = nullptr; c->h_out_cap = 0;
  c->cap_frames = 0;
}

int ensure_workspace(av1mi_ctx *c, const Resolved &r, uint32_t n_frames) {
  const av1mi_params &p = r.p;
  // the per-tile buffers (symbol streams, bitstream slots, packed output) are sized by the tile grid and the per-tile
  // capacities, which follow tile_sb: a context that switches tile_sb at the same frame size must not reuse them
  // (648x360: 66 tiles x 4096 entries at tile_sb 1, but 18 x 16384 at tile_sb 2)
  bool same = c->cap_frames >= n_frames && c->res.p.width == p.width && c->res.p.height == p.height && c->res.p.bit_depth == p.bit_depth &&
              c->ws_scale == c->cap_scale && c->res.tile_sb == r.tile_sb;
  const int bps = p.bit_depth > 8 ? 2 : 1;
  const size_t frame_samples = (size_t)r.cw * r.ch * 3 / 2;   // coded size
  const size_t nsb = (size_t)r.sb_cols * r.sb_rows;
  const size_t ntile = (size_t)r.tile_cols * r.tile_rows;
  // Per-tile capacities (per superblock of the tile): 8192 symbol-stream entries and 4096 output bytes hold any tile of
  // ordinary content down to about CQ 20 (1080p clip at CQ 30: longest tile 4826 entries, ~1.2 KB); a tile that outgrows
  // them is detected on the device and the chunk is re-run at the next multiplier, which the context then keeps.
  // x32 / x64 reach the true worst case of a 64x64 4:2:0 tile (6144 coefficients x <= 37 stream entries: base + 4
  // range + sign + 31 Golomb bits; <= 10 output bytes per coefficient), so the retry ends.
  const int slot = 4096 * c->cap_scale * r.tile_sb * r.tile_sb;
  const int stream_cap = 8192 * c->cap_scale * r.tile_sb * r.tile_sb;
  if (!same) {
    free_workspace(c);
    const size_t nf = n_frames;
    HIPCHK(c, hipMalloc(&c->d_src, nf * frame_samples * bps));
    HIPCHK(c, hipMalloc(&c->d_rec, nf * frame_samples * bps));
    HIPCHK(c, hipMalloc(&c->d_fin, nf * frame_samples * bps));
    HIPCHK(c, hipMalloc((void **)&c->d_levels, nf * nsb * AV1MI_SB_LEVELS * sizeof(int16_t)));
    const size_t nb8 = (size_t)(r.cw / 8) * (r.ch / 8);
    HIPCHK(c, hipMalloc((void **)&c->d_blk, nf * nb8 * sizeof(Av1miBlkInfo)));
    // on the context's own stream: hipMemset would run on the null stream, which a non-blocking stream does not wait
    // for - the fill could land after the first reconstruction kernel had written its block info
    HIPCHK(c, hipMemsetAsync(c->d_blk, 0, nf * nb8 * sizeof(Av1miBlkInfo), c->stream));
    HIPCHK(c, hipMalloc((void **)&c->d_slots, nf * ntile * slot * 2));  // 16-bit pre-carry entries, one per output byte
    c->out_cap = nf * (ntile * (size_t)(slot + 4) + 256);
    HIPCHK(c, hipMalloc((void **)&c->d_out, c->out_cap));
    HIPCHK(c, hipMalloc((void **)&c->d_hdr, 256 + nf * 512));
    HIPCHK(c, hipMalloc((void **)&c->d_cdf, Av1miCdfLayout::TOTAL * sizeof(uint16_t)));
    HIPCHK(c, hipMalloc((void **)&c->d_params, sizeof(Av1miDevParams)));
    HIPCHK(c, hipMalloc((void **)&c->d_tile_bytes, nf * nsb * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_tile_off, nf * nsb * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_sym, nf * nsb * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_combos, nf * nsb * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_streams, nf * ntile * (size_t)stream_cap * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_frame_size, nf * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_payload, nf * 4));
    HIPCHK(c, hipMalloc((void **)&c->d_frame_off, (nf + 1) * 8));
    HIPCHK(c, hipMalloc((void **)&c->d_sse, nf * 3 * 8));
    HIPCHK(c, hipMalloc((void **)&c->d_overflow, 4));
    HIPCHK(c, hipMalloc((void **)&c->d_me, nf * nb8 * 8));
    HIPCHK(c, hipMalloc((void **)&c->d_cdefdir, nf * nsb * 64 * sizeof(uint16_t)));
    c->cap_frames = n_frames;
    c->ws_scale = c->cap_scale;
  }
  if ((r.cw != (int)p.width || r.ch != (int)p.height) && !c->d_stage)
    HIPCHK(c, hipMalloc(&c->d_stage, c->cap_frames * (size_t)p.width * p.height * 3 / 2 * bps));
  if (p.partition_search && !c->d_part) HIPCHK(c, hipMalloc((void **)&c->d_part, c->cap_frames * nsb * sizeof(uint32_t)));
  if (p.me_presearch && !c->d_quarter) {
    HIPCHK(c, hipMalloc((void **)&c->d_quarter, c->cap_frames * (size_t)(r.cw / 4) * (r.ch / 4) * sizeof(uint16_t)));
    HIPCHK(c, hipMalloc((void **)&c->d_centre, c->cap_frames * nsb * sizeof(uint32_t)));
  }
  if (p.subpel && !c->d_me_sub)   // output of the sub-sample refinement
    HIPCHK(c, hipMalloc((void **)&c->d_me_sub, (size_t)c->cap_frames * (r.cw / 8) * (r.ch / 8) * 8));
  if (p.block_log2 >= 6 && p.keyint > 1) {   // candidate table of the 64x64 leaves' motion search
    const size_t nc = 2 * (size_t)(p.me_range ? p.me_range : 8) + 1, need = c->cap_frames * nsb * nc * nc * sizeof(uint32_t);
    if (c->me64_bytes < need) {
      if (c->d_me64) (void)hipFree(c->d_me64);
      c->d_me64 = nullptr; c->me64_bytes = 0;
      HIPCHK(c, hipMalloc((void **)&c->d_me64, need));
      c->me64_bytes = need;
    }
  }
  if (p.enable_lr && !c->d_cd) {
    const size_t nf = c->cap_frames;
    HIPCHK(c, hipMalloc(&c->d_cd, nf * frame_samples * bps));
    HIPCHK(c, hipMalloc((void **)&c->d_lrc, nf * nsb + 64));
    HIPCHK(c, hipMalloc((void **)&c->d_lrsse, (nf * nsb + 64) * 8 * sizeof(unsigned long long)));
  }
  c->res = r;
  Av1miDevParams &P = c->P;
  memset(&P, 0, sizeof(P));
  P.width = r.cw; P.height = r.ch; P.bit_depth = p.bit_depth;
  P.true_w = (int)p.width; P.true_h = (int)p.height;
  P.mi_rows = r.ch / 4; P.mi_cols = r.cw / 4;
  P.sb_rows = r.sb_rows; P.sb_cols = r.sb_cols;
  P.tile_sb = r.tile_sb; P.tile_rows = r.tile_rows; P.tile_cols = r.tile_cols;
  P.b8_rows = r.ch / 8; P.b8_cols = r.cw / 8;
  P.n_frames = (int)n_frames;
  P.base_q_idx = r.qidx;
  P.qctx = r.qidx <= 20 ? 0 : (r.qidx <= 60 ? 1 : (r.qidx <= 120 ? 2 : 3));
  P.dc_q = p.bit_depth == 8 ? av1_dc_q8[r.qidx] : av1_dc_q10[r.qidx];
  P.ac_q = p.bit_depth == 8 ? av1_ac_q8[r.qidx] : av1_ac_q10[r.qidx];
  P.dc_recip = (uint32_t)((((uint64_t)1 << 32) + P.dc_q - 1) / P.dc_q);
  P.ac_recip = (uint32_t)((((uint64_t)1 << 32) + P.ac_q - 1) / P.ac_q);
  P.using_qm = p.enable_qm ? 1 : 0; P.qm_y = P.qm_uv = r.qm_level;
  P.qm_tab = nullptr;
  if (p.enable_qm && r.qm_level < 15) {
    // dequantiser step per coefficient position (§7.12.3) and its reciprocal, for the square transform sizes 4..32
    if (!c->d_qm) HIPCHK(c, hipMalloc((void **)&c->d_qm, 2 * AV1MI_QM_PLANE * sizeof(Av1miQmEntry)));
    if (c->qm_key != ((r.qm_level << 16) | (r.qidx << 4) | (int)p.bit_depth)) {
      static const int off[4] = { AV1MI_QM_4X4, AV1MI_QM_8X8, AV1MI_QM_16X16, AV1MI_QM_32X32 };
      c->h_qm.resize(2 * AV1MI_QM_PLANE);
      for (int pt = 0; pt < 2; pt++)
        for (int l2 = 2; l2 <= 5; l2++)
          for (int i = 0; i < (1 << (2 * l2)); i++) {
            const uint32_t q = (uint32_t)(i ? P.ac_q : P.dc_q);
            const uint32_t q2 = (q * av1_qm_iwt[r.qm_level][pt][off[l2 - 2] + i] + 16) >> 5;
            c->h_qm[pt * AV1MI_QM_PLANE + off[l2 - 2] + i] = { q2, (uint32_t)((((uint64_t)1 << 32) + q2 - 1) / q2) };
          }
      HIPCHK(c, hipMemcpyAsync(c->d_qm, c->h_qm.data(), c->h_qm.size() * sizeof(Av1miQmEntry), hipMemcpyHostToDevice, c->stream));
      c->qm_key = (r.qm_level << 16) | (r.qidx << 4) | (int)p.bit_depth;
    }
    P.qm_tab = c->d_qm;
  }
  P.max_bs_log2 = (int)p.block_log2;
  P.min_bs_log2 = p.partition_search ? (int)p.min_block_log2 : (int)p.block_log2;
  P.part_map = p.partition_search ? c->d_part : nullptr;
  P.mode_mask = p.intra_mode_mask ? (p.intra_mode_mask & 0x1FFF) : 0x0007;  // default candidates: DC, V, H
  P.angle_delta = p.intra_angle_delta ? 1 : 0;
  P.edge_filter = p.intra_edge_filter ? 1 : 0;
  P.cfl = p.cfl ? 1 : 0;
  P.tx_search = p.tx_search ? 1 : 0;
  P.enable_cdef = p.enable_cdef ? 1 : 0;
  P.cdef_y_pri = p.cdef_y_pri; P.cdef_y_sec = p.cdef_y_sec; P.cdef_uv_pri = p.cdef_uv_pri; P.cdef_uv_sec = p.cdef_uv_sec;
  P.cdef_damping = p.cdef_damping;
  P.disable_cdf_update = p.cdf_update ? 0 : 1;
  P.stride_y = r.cw; P.stride_c = r.cw / 2;
  P.plane_off_u = (long)r.cw * r.ch;
  P.plane_off_v = P.plane_off_u + (long)(r.cw / 2) * (r.ch / 2);
  P.frame_samples = (long)frame_samples;
  P.tile_slot_bytes = slot;
  P.stream_cap = stream_cap;
  P.tile_size_bytes = 4;
  P.keyint = (int)p.keyint;
  P.me_range = (int)p.me_range;
  P.subpel = p.subpel ? 1 : 0;
  P.me_presearch = p.me_presearch ? 1 : 0;
  P.hdr_slot_bytes = 512;
  for (int i = 0; i < 4; i++) { P.lf_level[i] = deblock_level(r, true); P.lf_level_inter[i] = deblock_level(r, false); }
  P.lf_sharpness = 0;
  P.enable_lr = (int)p.enable_lr;
  for (int rf = 0; rf < 4; rf++)
    for (int k = 0; k < 3; k++) {
      const BitString b = lr_code_of(rf, k); P.lr_code_len[rf][k] = b.len; P.lr_code_bits[rf][k] = b.bits;
      const BitString g = sgr_code_of(rf, k); P.sgr_code_len[rf][k] = g.len; P.sgr_code_bits[rf][k] = g.bits;
    }
  return AV1MI_OK;
}

}  // namespace

extern "C" {

uint32_t av1mi_abi_version(void) { return AV1MI_ABI_VERSION; }
uint32_t av1mi_struct_sizes(uint32_t *sizes, uint32_t cap) {
  const uint32_t v[AV1MI_LAYOUT_ENTRIES] = {
    (uint32_t)sizeof(av1mi_params), (uint32_t)sizeof(av1mi_job), (uint32_t)sizeof(av1mi_report), (uint32_t)sizeof(av1mi_buf),
    (uint32_t)sizeof(av1mi_clip_info), (uint32_t)sizeof(av1mi_scene_state), (uint32_t)sizeof(av1mi_exec_job), (uint32_t)sizeof(av1mi_job_metrics),
    (uint32_t)offsetof(av1mi_job, params), (uint32_t)offsetof(av1mi_report, ms_h2d), (uint32_t)offsetof(av1mi_exec_job, params),
    (uint32_t)offsetof(av1mi_job_metrics, frames_encoded) };
  for (uint32_t i = 0; i < AV1MI_LAYOUT_ENTRIES && i < cap && sizes; i++) sizes[i] = v[i];
  return AV1MI_LAYOUT_ENTRIES;
}
uint32_t av1mi_cq_to_qindex(uint32_t cq) { return kQuantizerToQindex[cq > 63 ? 63 : cq]; }

void av1mi_default_params(av1mi_params *p, uint32_t w, uint32_t h, uint32_t bd) {
  memset(p, 0, sizeof(*p));
  p->width = w; p->height = h; p->bit_depth = bd;
  p->cq_level = 30; p->keyint = 1; p->block_log2 = 5; p->cdf_update = 1; p->enable_cdef = 1;
  p->cdef_y_pri = 2; p->cdef_y_sec = 0; p->cdef_uv_pri = 1; p->cdef_uv_sec = 0; p->cdef_damping = 5;
  p->qm_min = 8; p->qm_max = 15;  // used when enable_qm = 1
}

int av1mi_write_headers(const av1mi_params *p, uint8_t *seq_hdr, size_t *seq_len, uint8_t *frame_hdr, size_t *frame_hdr_bits) {
  Resolved r;
  int rc = resolve(p, &r);
  if (rc) return rc;
  std::vector<uint8_t> s = make_sequence_header(r);
  size_t bits = 0;
  std::vector<uint8_t> f = make_frame_header(r, &bits);
  if (seq_hdr && seq_len) { if (*seq_len < s.size()) return AV1MI_E_INVALID_ARG; memcpy(seq_hdr, s.data(), s.size()); }
  if (seq_len) *seq_len = s.size();
  if (frame_hdr) memcpy(frame_hdr, f.data(), f.size());
  if (frame_hdr_bits) *frame_hdr_bits = bits;
  return AV1MI_OK;
}

// The four streams of a context come from a process-wide pool and go back to it when the context is destroyed.  The runtime maps
// streams onto a few hardware queues in creation order, and after contexts had been created and destroyed a number of times a new
// context's chain stream could share a queue with its own search stream, whose launches for the whole chunk are queued up front: a
// single 1080p IPPP chunk then took 20.3 ms instead of 13.7 (bench.py's configs run one after the other in one process; a daemon that
// lives for days is the same case).  Reused streams keep the mapping of the first contexts.
namespace {
struct StreamSet { hipStream_t s[4]; };
std::mutex g_stream_mu;
std::vector<std::pair<int, StreamSet>> g_stream_pool;   // (device, set)
}  // namespace
extern "C" void av1mi_host_release_streams(void) {
  std::vector<std::pair<int, StreamSet>> all;
  { std::lock_guard<std::mutex> lk(g_stream_mu); all.swap(g_stream_pool); }
  for (auto &e : all) {
    (void)hipSetDevice(e.first);
    for (auto st : e.second.s) if (st) (void)hipStreamDestroy(st);
  }
}

int av1mi_ctx_create(int device_id, av1mi_ctx **out) {
  if (!out) return AV1MI_E_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return AV1MI_E_NO_DEVICE;
  av1mi_ctx *c = new (std::nothrow) av1mi_ctx();
  if (!c) return AV1MI_E_OOM;
  c->device = device_id;
  // test hook: start at a capacity multiplier content would otherwise have to provoke (tests/test_gpu_parity.py: slots beyond 4 GB)
  if (const char *e = getenv("AV1MI_CAP_SCALE")) { const int k = atoi(e); if (k >= 1 && k <= 64 && (k & (k - 1)) == 0) c->cap_scale = k; }
  // the main stream carries the serial chain of a chunk (and the latency-bound range coder): highest priority, so that
  // the bulk work put beside it on the second stream (CDEF, SSE, the chunk-wide motion search) fills gaps instead of
  // taking its slots
  int prio_lo = 0, prio_hi = 0;
  if (hipSetDevice(device_id) == hipSuccess) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (const char *fp = getenv("AV1MI_STREAM_PRIORITIES")) {   // experiment knob: "flat" = every stream at the default priority, "hi" = all high
    if (!strcmp(fp, "flat")) prio_lo = prio_hi = 0;
    else if (!strcmp(fp, "hi")) prio_lo = prio_hi;
    else if (!strcmp(fp, "mid")) prio_lo = 0;   // chain high, the rest normal
  }
  // experiment knob: AV1MI_AUX_CU_MASK=<hex word> restricts the two auxiliary streams (bulk work beside the chain) to the CUs whose bit
  // is set in the word, repeated over the chip's 256 CUs (0x55555555: every other CU)
  bool aux_ok = true;
  if (const char *m = getenv("AV1MI_AUX_CU_MASK")) {
    uint32_t mask[8];
    const uint32_t word = (uint32_t)strtoul(m, nullptr, 16);
    for (auto &w : mask) w = word;
    aux_ok = hipSetDevice(device_id) == hipSuccess && hipExtStreamCreateWithCUMask(&c->stream2, 8, mask) == hipSuccess &&
             hipExtStreamCreateWithCUMask(&c->stream3, 8, mask) == hipSuccess;
  }
  if (aux_ok && !c->stream2) {   // a pooled set, if there is one for this device
    std::lock_guard<std::mutex> lk(g_stream_mu);
    for (size_t i = 0; i < g_stream_pool.size(); i++)
      if (g_stream_pool[i].first == device_id) {
        const StreamSet ss = g_stream_pool[i].second;
        g_stream_pool.erase(g_stream_pool.begin() + (long)i);
        c->stream = ss.s[0]; c->stream2 = ss.s[1]; c->stream3 = ss.s[2]; c->stream4 = ss.s[3];
        c->pooled_streams = true;
        break;
      }
  }
  if (c->pooled_streams) {
    if (hipSetDevice(device_id) != hipSuccess) { delete c; return AV1MI_E_NO_DEVICE; }
  } else
  if (!aux_ok || hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
      (!c->stream2 && hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_lo) != hipSuccess) ||
      (!c->stream3 && hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_lo) != hipSuccess) ||
      hipStreamCreateWithPriority(&c->stream4, hipStreamNonBlocking, prio_lo) != hipSuccess) {
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream4) (void)hipStreamDestroy(c->stream4);
    delete c;
    return AV1MI_E_NO_DEVICE;
  }
  if (!getenv("AV1MI_AUX_CU_MASK")) c->pooled_streams = true;   // (the masked streams of the experiment knob are not pooled)
  for (auto &e : c->ev) (void)hipEventCreate(&e);
  *out = c;
  return AV1MI_OK;
}

void av1mi_ctx_destroy(av1mi_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  free_workspace(c);
  for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
  for (auto &e : c->me_ev) (void)hipEventDestroy(e);
  for (auto &e : c->grp_ev) (void)hipEventDestroy(e);
  if (c->pooled_streams && c->stream && c->stream2 && c->stream3 && c->stream4) {
    (void)hipStreamSynchronize(c->stream2); (void)hipStreamSynchronize(c->stream3); (void)hipStreamSynchronize(c->stream4);
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_stream_pool.push_back({ c->device, StreamSet{ { c->stream, c->stream2, c->stream3, c->stream4 } } });
  } else {
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream4) (void)hipStreamDestroy(c->stream4);
  }
  delete c;
}

// A context that goes back to av1mi_encode_file's cache forgets the capacity multiplier its last job's content raised (it would
// otherwise size - or, at the limit, refuse - an unrelated job's workspace by it); the next chunk reallocates at scale 1.
void av1mi_host_ctx_recycle(av1mi_ctx *c) { if (c) c->cap_scale = 1; }

const char *av1mi_last_error(const av1mi_ctx *c) { return c ? c->err.c_str() : "no context"; }

// Bitstream buffers handed to the caller (av1mi_buf.data) are page-locked blocks from a process-wide pool: the packed chunk is
// copied device -> host straight into the block the caller gets, and av1mi_free() puts the block back for the next chunk - no
// staging copy and no first-touch page faults on a fresh malloc per chunk (2.35 MB per 1080p all-key-frame chunk, 23 MB at the
// production point: 0.2 / 1.1 ms of host time per chunk before).  Blocks that do not come from the pool are plain malloc.
namespace {
std::mutex g_pin_mu;
std::vector<std::pair<void *, size_t>> g_pin_out;    // handed out: (block, capacity)
std::vector<std::pair<void *, size_t>> g_pin_free;   // waiting for the next chunk
constexpr size_t PIN_FREE_MAX_BYTES = (size_t)1 << 30;
// parked blocks: a job with w workers has up to 2 w finished chunks waiting for the writer, so the bound follows the largest worker
// count seen (av1mi_host_expect_blocks) - freeing a block is hipHostFree, which synchronises the whole device
std::atomic<size_t> g_pin_free_max_blocks{16};

void *pin_acquire(size_t bytes) {
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    int best = -1;
    for (size_t i = 0; i < g_pin_free.size(); i++)
      if (g_pin_free[i].second >= bytes && (best < 0 || g_pin_free[i].second < g_pin_free[(size_t)best].second)) best = (int)i;
    if (best >= 0) {
      const std::pair<void *, size_t> b = g_pin_free[(size_t)best];
      g_pin_free.erase(g_pin_free.begin() + best);
      g_pin_out.push_back(b);
      return b.first;
    }
  }
  void *p = nullptr;
  const size_t cap = bytes + (bytes >> 2) + 4096;
  if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lk(g_pin_mu);
  g_pin_out.emplace_back(p, cap);
  return p;
}
}  // namespace

extern "C" void av1mi_host_expect_blocks(unsigned n) {
  size_t cur = g_pin_free_max_blocks.load();
  while (n > cur && n <= 256 && !g_pin_free_max_blocks.compare_exchange_weak(cur, n)) { }
}

extern "C" void av1mi_host_release_buffers(void) {
  std::vector<std::pair<void *, size_t>> all;
  { std::lock_guard<std::mutex> lk(g_pin_mu); all.swap(g_pin_free); }
  for (auto &b : all) (void)hipHostFree(b.first);
}

void av1mi_free(void *p) {
  if (!p) return;
  bool pooled = false, drop = false;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (size_t i = 0; i < g_pin_out.size(); i++)
      if (g_pin_out[i].first == p) {
        const std::pair<void *, size_t> b = g_pin_out[i];
        g_pin_out.erase(g_pin_out.begin() + (long)i);
        size_t held = 0;
        for (auto &f : g_pin_free) held += f.second;
        if (g_pin_free.size() < g_pin_free_max_blocks.load() && held + b.second <= PIN_FREE_MAX_BYTES) g_pin_free.push_back(b);
        else drop = true;
        pooled = true;
        break;
      }
  }
  if (!pooled) free(p);
  else if (drop) (void)hipHostFree(p);
}

// Scene-cut rule (include/av1mi.h: av1mi_scene_cuts).  Integer only: d = mean absolute luma difference at 8-bit
// scale in Q8.  Mirrored by oracle/scenecut.py.
static int scene_rule_step(av1mi_scene_state *st, int has_prev, uint64_t sad, uint64_t luma_samples, int bit_depth, uint32_t min_len) {
  if (!has_prev) { st->hist_n = 0; st->frames_since_cut = 1; return 1; }
  const uint64_t d = ((sad >> (bit_depth - 8)) << 8) / luma_samples;
  uint64_t sum = 0;
  for (uint32_t i = 0; i < st->hist_n; i++) sum += st->hist_q8[i];
  const bool strong = st->hist_n ? (2 * d * st->hist_n >= 5 * sum && d >= 8 * 256) : d >= 24 * 256;
  if (strong && st->frames_since_cut >= min_len) { st->hist_n = 0; st->frames_since_cut = 1; return 1; }
  if (!strong) {  // in-scene sample: keep the last 8
    if (st->hist_n == 8) { for (int i = 0; i < 7; i++) st->hist_q8[i] = st->hist_q8[i + 1]; st->hist_n = 7; }
    st->hist_q8[st->hist_n++] = (uint32_t)d;
  }
  st->frames_since_cut++;
  return 0;
}

int av1mi_scene_cuts(av1mi_ctx *c, const av1mi_params *params, const void *frames, uint32_t n_frames, int frames_on_device,
                     const void *prev_frame, av1mi_scene_state *state, uint32_t min_scene_len, uint64_t *sad_out, uint8_t *is_cut) {
  if (!c || !frames || !state || n_frames == 0) return AV1MI_E_INVALID_ARG;
  Resolved r;
  int rc = resolve(params, &r);
  if (rc) { set_err(c, "invalid parameters"); return rc; }
  HIPCHK(c, hipSetDevice(c->device));
  const int bps = r.p.bit_depth > 8 ? 2 : 1;
  const size_t frame_bytes = (size_t)r.p.width * r.p.height * 3 / 2 * bps;
  Av1miDevParams P;
  memset(&P, 0, sizeof(P));
  P.width = r.p.width; P.height = r.p.height; P.bit_depth = r.p.bit_depth; P.n_frames = (int)n_frames;  // tight input layout
  P.true_w = P.width; P.true_h = P.height;
  P.frame_samples = (long)(frame_bytes / bps);
  hipStream_t s = c->stream;
  // host input: stage [prev | frames] contiguously on the device (frame_bytes is a multiple of 32)
  void *d_stage = nullptr;
  unsigned long long *d_sad = nullptr;
  const uint8_t *d_frames, *d_prev = nullptr;
  auto cleanup = [&] { if (d_stage) (void)hipFree(d_stage); if (d_sad) (void)hipFree(d_sad); };
  hipError_t e = hipMalloc((void **)&d_sad, (size_t)n_frames * 8);
  if (e == hipSuccess) e = hipMemsetAsync(d_sad, 0, (size_t)n_frames * 8, s);
  if (e == hipSuccess && !frames_on_device) {
    e = hipMalloc(&d_stage, (size_t)(n_frames + 1) * frame_bytes);
    if (e == hipSuccess && prev_frame) e = hipMemcpyAsync(d_stage, prev_frame, frame_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync((uint8_t *)d_stage + frame_bytes, frames, (size_t)n_frames * frame_bytes, hipMemcpyHostToDevice, s);
    d_frames = (const uint8_t *)d_stage + frame_bytes;
    d_prev = prev_frame ? (const uint8_t *)d_stage : nullptr;
  } else {
    d_frames = (const uint8_t *)frames;
    d_prev = (const uint8_t *)prev_frame;
    if (((uintptr_t)d_frames | (uintptr_t)d_prev) & 15) { cleanup(); set_err(c, "device frames must be 16-byte aligned"); return AV1MI_E_INVALID_ARG; }
  }
  std::vector<unsigned long long> sad(n_frames, 0);
  if (e == hipSuccess) e = av1mi_launch_luma_sad(&P, d_frames, d_prev, d_sad, s);
  if (e == hipSuccess) e = hipMemcpyAsync(sad.data(), d_sad, (size_t)n_frames * 8, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  cleanup();
  if (e != hipSuccess) { set_err(c, "scene-cut pass failed: %s", hipGetErrorString(e)); return e == hipErrorOutOfMemory ? AV1MI_E_OOM : AV1MI_E_HIP; }
  const uint64_t luma = (uint64_t)r.p.width * r.p.height;
  for (uint32_t t = 0; t < n_frames; t++) {
    const int has_prev = t > 0 || prev_frame != nullptr;
    const int cut = scene_rule_step(state, has_prev, sad[t], luma, (int)r.p.bit_depth, min_scene_len ? min_scene_len : 1);
    if (is_cut) is_cut[t] = (uint8_t)cut;
    if (sad_out) sad_out[t] = sad[t];
  }
  return AV1MI_OK;
}

static int encode_chunk_once(av1mi_ctx *c, const av1mi_params *params, const void *frames, uint32_t n_frames, int frames_on_device,
                             av1mi_buf *out, uint32_t *frame_sizes, void *recon, av1mi_report *report);

int av1mi_encode_chunk(av1mi_ctx *c, const av1mi_params *params, const void *frames, uint32_t n_frames, int frames_on_device,
                       av1mi_buf *out, uint32_t *frame_sizes, void *recon, av1mi_report *report) {
  if (!c || !frames || !out || n_frames == 0) return AV1MI_E_INVALID_ARG;
  for (;;) {
    const int rc = encode_chunk_once(c, params, frames, n_frames, frames_on_device, out, frame_sizes, recon, report);
    if (rc != AV1MI_E_OVERFLOW || c->cap_scale >= 64) return rc;
    c->cap_scale *= 2;  // stays raised for the following chunks of this context (same content)
  }
}

static int encode_chunk_once(av1mi_ctx *c, const av1mi_params *params, const void *frames, uint32_t n_frames, int frames_on_device,
                             av1mi_buf *out, uint32_t *frame_sizes, void *recon, av1mi_report *report) {
  out->data = nullptr; out->size = 0;
  Resolved r;
  int rc = resolve(params, &r);
  if (rc) { set_err(c, "invalid parameters"); return rc; }
  HIPCHK(c, hipSetDevice(c->device));
  rc = ensure_workspace(c, r, n_frames);
  if (rc) return rc;
  Av1miDevParams &P = c->P;
  const int bps = P.bit_depth > 8 ? 2 : 1;
  const bool padded = r.cw != (int)r.p.width || r.ch != (int)r.p.height;
  const size_t chunk_bytes = (size_t)n_frames * P.frame_samples * bps;                       // coded layout
  const size_t user_bytes = (size_t)n_frames * r.p.width * r.p.height * 3 / 2 * bps;         // the caller's tight layout
  hipStream_t s = c->stream;
  // headers + CDFs
  // header blob: sequence header OBU, then one fixed-size slot per frame with that frame's header (key and inter
  // frames have different lengths; frames of one kind differ only in grain_seed)
  std::vector<uint8_t> seq = make_sequence_header(r), fh = make_frame_header(r, nullptr, 0, false), fhi = make_frame_header(r, nullptr, 0, true);
  P.seq_hdr_bytes = (int)seq.size();
  P.frame_hdr_bytes = (int)fh.size();
  P.inter_hdr_bytes = (int)fhi.size();
  if (seq.size() > 256 || fh.size() > (size_t)P.hdr_slot_bytes || fhi.size() > (size_t)P.hdr_slot_bytes) { set_err(c, "internal: header larger than its slot"); return AV1MI_E_OVERFLOW; }
  std::vector<uint8_t> blob(seq.size() + (size_t)n_frames * P.hdr_slot_bytes, 0);
  memcpy(blob.data(), seq.data(), seq.size());
  for (uint32_t f = 0; f < n_frames; f++) {
    const bool inter = av1mi_frame_is_inter(P, (int)f);
    std::vector<uint8_t> &h = inter ? fhi : fh;
    if (r.p.film_grain && f) h = make_frame_header(r, nullptr, f, inter);
    memcpy(blob.data() + seq.size() + (size_t)f * P.hdr_slot_bytes, h.data(), h.size());
  }
  std::vector<uint16_t> cdf = make_cdf_blob(r.qidx);
  HIPCHK(c, hipMemcpyAsync(c->d_hdr, blob.data(), blob.size(), hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->d_cdf, cdf.data(), cdf.size() * 2, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->d_params, &P, sizeof(P), hipMemcpyHostToDevice, s));  // the recon kernel reads its parameters from device memory
  HIPCHK(c, hipMemsetAsync(c->d_overflow, 0, 4, s));
  HIPCHK(c, hipMemsetAsync(c->d_sse, 0, (size_t)n_frames * 24, s));
  HIPCHK(c, hipEventRecord(c->ev[0], s));
  const void *d_src = frames;
  if (!padded) {
    if (!frames_on_device) {
      HIPCHK(c, hipMemcpyAsync(c->d_src, frames, chunk_bytes, hipMemcpyHostToDevice, s));
      d_src = c->d_src;
    }
  } else {
    // not a multiple of 8: edge-extend every frame to the coded size (the signalled size stays exact)
    const void *tight = frames;
    if (!frames_on_device) {
      HIPCHK(c, hipMemcpyAsync(c->d_stage, frames, user_bytes, hipMemcpyHostToDevice, s));
      tight = c->d_stage;
    }
    HIPCHK(c, av1mi_launch_pad(tight, c->d_src, (int)r.p.width, (int)r.p.height, r.cw, r.ch, P.bit_depth, (int)n_frames, 0, s));
    d_src = c->d_src;
  }
  // content-driven partition: the split masks of every superblock of the chunk, from the source, before anything walks blocks (the second
  // stream's motion search waits for ev[1] as well)
  if (P.part_map) HIPCHK(c, av1mi_launch_partition(&P, d_src, c->d_part, s));
  HIPCHK(c, hipEventRecord(c->ev[1], s));
  const bool inter_chunk = P.keyint > 1 && n_frames > 1;
  // 64x64 leaf blocks run the kernels of recon64_kernel.hip (64-point transforms, larger LDS tiles)
  auto launch_recon = P.max_bs_log2 >= 6 ? (P.bit_depth == 8 ? av1mi_launch_recon64_u8 : av1mi_launch_recon64_u16)
                                         : (P.bit_depth == 8 ? av1mi_launch_recon_u8 : av1mi_launch_recon_u16);
  const bool lr = P.enable_lr != 0;
  uint32_t entropy_from = 0;      // inter chunks: frames before this one are entropy-coded on the third stream, beside the chain
  bool sym_groups = false;        // ... or only symbolized there (AV1MI_SYM_GROUP)
  bool entropy_joined = false;
  void *cdef_out = lr ? c->d_cd : c->d_fin;   // with loop restoration CDEF writes d_cd and the restored frame goes to d_fin
  if (!inter_chunk && lr) {
    // the unit decisions are part of the tile syntax: CDEF and restoration must precede entropy coding
    HIPCHK(c, launch_recon(&P, c->d_params, d_src, c->d_rec, c->d_levels, c->d_blk, nullptr, nullptr, P.part_map, s));
    if (P.lf_level[0]) HIPCHK(c, av1mi_launch_deblock(&P, c->d_rec, c->d_blk, s));
    HIPCHK(c, av1mi_launch_cdef(&P, c->d_rec, c->d_cd, c->d_blk, nullptr, nullptr, nullptr, s));
    HIPCHK(c, av1mi_launch_lr(&P, c->d_rec, c->d_cd, d_src, c->d_fin, c->d_lrc, c->d_lrsse, 1, s));
  } else if (!inter_chunk) {
    // All-key-frame chunk: the frames are independent, so the chunk CAN run as a software pipeline over groups of frames
    // (AV1MI_INTRA_GROUPS=k) - the main stream reconstructs group g + 1 while auxiliary streams symbolize and range-code group
    // g.  Measured (1080p x 60, MI355X): 14.1 k frames/s as one group, 12.3 k as two, 10.7 k as three - the range coder holds
    // 145 KB of LDS on every CU it sits on, which leaves the reconstruction (10 KB per wave) one wave there instead of sixteen.
    // The default is therefore one group; the knob stays for the day the range coder's per-tile state leaves LDS.
    uint32_t groups = 1;
    if (const char *eg = getenv("AV1MI_INTRA_GROUPS")) { const int k = atoi(eg); if (k >= 1) groups = (uint32_t)k; }
    if (groups > n_frames) groups = n_frames;
    const uint32_t grp = (n_frames + groups - 1) / groups;
    const size_t fbytes = (size_t)P.frame_samples * bps, nb8 = (size_t)P.b8_rows * P.b8_cols, nsb = (size_t)P.sb_rows * P.sb_cols;
    uint32_t n_grp = 0;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += grp) {
      const uint32_t cnt = f0 + grp < n_frames ? grp : n_frames - f0;
      Av1miDevParams Pg = P;
      Pg.n_frames = (int)cnt;
      HIPCHK(c, launch_recon(&Pg, c->d_params, (const uint8_t *)d_src + f0 * fbytes, (uint8_t *)c->d_rec + f0 * fbytes,
                             c->d_levels + f0 * nsb * AV1MI_SB_LEVELS, c->d_blk + f0 * nb8, nullptr, nullptr, P.part_map ? P.part_map + f0 * nsb : nullptr, s));
      if (f0 + cnt < n_frames) {
        if (c->grp_ev.size() <= n_grp) {
          hipEvent_t e;
          HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
          c->grp_ev.push_back(e);
        }
        hipStream_t sg = (n_grp & 1) ? c->stream4 : c->stream3;   // alternate: group g's range coder runs beside group g + 1's symbolize
        HIPCHK(c, hipEventRecord(c->grp_ev[n_grp], s));
        HIPCHK(c, hipStreamWaitEvent(sg, c->grp_ev[n_grp], 0));
        HIPCHK(c, av1mi_launch_entropy(&P, c->d_cdf, c->d_levels, c->d_blk, c->d_streams, c->d_sym, c->d_combos, c->d_slots, c->d_tile_bytes, c->d_lrc,
                                       c->d_tile_off, (int)f0, (int)cnt, 3, sg, nullptr, nullptr, nullptr, nullptr));
        n_grp++;
        entropy_from = f0 + cnt;
      }
    }
    if (n_grp) HIPCHK(c, hipEventRecord(c->ev[10], c->stream3));
    if (n_grp > 1) { HIPCHK(c, hipEventRecord(c->ev[11], c->stream4)); HIPCHK(c, hipStreamWaitEvent(s, c->ev[11], 0)); }
    entropy_joined = n_grp != 0;
  } else {
    // Inter chunk: a P frame needs the previous frame's final (post-CDEF) reconstruction, so motion search,
    // reconstruction and CDEF run frame by frame; entropy coding of ALL frames follows in one pass (every frame
    // starts from the default CDFs, so tiles of different frames stay independent).
    Av1miDevParams P1 = P;
    P1.n_frames = 1;
    const size_t fbytes = (size_t)P.frame_samples * bps, nb8 = (size_t)P.b8_rows * P.b8_cols, nsb = (size_t)P.sb_rows * P.sb_cols;
    // Motion search is open loop (source against previous source): all inter frames at once, on the second stream,
    // beside the chain below; the first inter frame's reconstruction waits for it.
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev[1], 0));  // the source frames are in HBM
    HIPCHK(c, hipMemsetAsync(c->d_me, 0xFF, (size_t)n_frames * nb8 * 8, c->stream2));
    uint32_t *acc64 = P.max_bs_log2 >= 6 ? c->d_me64 : nullptr;
    if (acc64) HIPCHK(c, hipMemsetAsync(acc64, 0, (size_t)n_frames * nsb * (2 * P.me_range + 1) * (2 * P.me_range + 1) * sizeof(uint32_t), c->stream2));
    // one launch + one event per frame, enqueued from inside the frame loop below, behind the first frame's chain kernels: the chain's first
    // kernel (the key frame's reconstruction needs no vectors) no longer waits on the host for 120 - 180 API calls (under
    // the profiler the chain started 6.4 ms after the chunk's first kernel; unprofiled the calls take ~ 8 us each and one chunk gains nothing)
    while (c->me_ev.size() < n_frames) {
      hipEvent_t e;
      HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      c->me_ev.push_back(e);
    }
    // hierarchical search: the chunk's luma at a quarter of the resolution (one launch), then per frame the superblocks' search centres
    const uint32_t *centre = nullptr;
    if (P.me_presearch) {
      HIPCHK(c, av1mi_launch_quarter_luma(&P, d_src, c->d_quarter, c->stream2));
      centre = c->d_centre;
    }
    uint32_t me_next = 0;
    auto search_upto = [&](uint32_t upto) -> int {   // enqueue the search of the frames below `upto` that are not enqueued yet
      for (; me_next < upto && me_next < n_frames; me_next++) {
        const uint32_t f = me_next;
        if (!av1mi_frame_is_inter(P, (int)f)) continue;
        if (centre) HIPCHK(c, av1mi_launch_presearch(&P, c->d_quarter, c->d_centre, (int)f, 1, c->stream2));
        HIPCHK(c, av1mi_launch_motion_search(&P, d_src, c->d_me, P.me_range, (int)f, 1, acc64, centre, c->stream2));
        if (P.subpel) HIPCHK(c, av1mi_launch_subpel_refine(&P, d_src, c->d_me, c->d_me_sub, P.me_range, (int)f, 1, c->stream2));
        HIPCHK(c, hipEventRecord(c->me_ev[f], c->stream2));
      }
      return AV1MI_OK;
    };
    // Frames the search is enqueued ahead of the chain.  Default: everything, right behind the FIRST frame's chain kernels - measured (1080p
    // IPPP x 60, one chunk / four in flight, frames/s): all 5 420 / 7 560; 8 ahead 5 390 / 7 220; 4 ahead 5 410 / 6 970; 2 ahead 5 370 / 6 900 -
    // with several chunks in flight a search that is enqueued late competes with the other chunks' chains instead of filling their gaps.
    uint32_t me_ahead = n_frames;
    if (const char *ea = getenv("AV1MI_ME_AHEAD")) { const int k = atoi(ea); me_ahead = k > 0 ? (uint32_t)k : n_frames; }
    // Entropy coding beside the chain: every frame starts from the default CDFs, so a group of frames can be symbolized and
    // range-coded (third stream) as soon as its last frame is reconstructed (and, with restoration on, its unit choices are
    // known) while the chain reconstructs the following frames; only the last group's entropy coding is left after the chain.
    // (AV1MI_ENTROPY_GROUP=k: groups of k frames - tests force small groups on short chunks; 0 = no overlap)
    // Group size: since the chain's kernels got short (walk 76, inter pass 38 us per 1080p frame) a range-coder launch beside them costs
    // more than it hides - 60-frame chunks, groups of 10 / 20 / none: 5 320 / 5 450 / 5 530 frames/s for one chunk, 7 670 / 8 050 / 7 970
    // with four in flight, 2 440 / 2 460 / 2 540 at the production point - so chunks of up to 64 frames code everything after the
    // chain and longer ones in groups of 32.
    uint32_t grp = n_frames <= 64 ? n_frames : 32;
    if (const char *eg = getenv("AV1MI_ENTROPY_GROUP")) { const int k = atoi(eg); grp = k > 0 ? (uint32_t)k : n_frames; }
    // AV1MI_SYM_GROUP=k: groups of k frames are only SYMBOLIZED beside the chain; one range-coder launch for the whole chunk follows it
    if (const char *eg = getenv("AV1MI_SYM_GROUP")) { const int k = atoi(eg); if (k > 0) { grp = (uint32_t)k; sym_groups = true; } }
    uint32_t n_grp = 0;
    if (lr) {   // the restoration units' candidate sums of the whole chunk, cleared once (not a fill per frame on the chain)
      const size_t upf = (size_t)((P.true_h + 32) / 64 > 0 ? (P.true_h + 32) / 64 : 1) * ((P.true_w + 32) / 64 > 0 ? (P.true_w + 32) / 64 : 1);
      HIPCHK(c, hipMemsetAsync(c->d_lrsse, 0, (size_t)n_frames * upf * 8 * sizeof(unsigned long long), s));
    }
    for (uint32_t f = 0; f < n_frames; f++) {
      const uint8_t *srcf = (const uint8_t *)d_src + f * fbytes;
      uint8_t *recf = (uint8_t *)c->d_rec + f * fbytes, *finf = (uint8_t *)c->d_fin + f * fbytes, *cdf_ = (uint8_t *)cdef_out + f * fbytes;
      Av1miBlkInfo *blkf = c->d_blk + f * nb8;
      int16_t *lvf = c->d_levels + f * nsb * AV1MI_SB_LEVELS;
      if (!av1mi_frame_is_inter(P, (int)f)) {
        HIPCHK(c, launch_recon(&P1, c->d_params, srcf, recf, lvf, blkf, nullptr, nullptr, P.part_map ? P.part_map + f * nsb : nullptr, s));
      } else {
        const uint8_t *reff = (const uint8_t *)c->d_fin + (f - 1) * fbytes;
        unsigned long long *mef = (P.subpel ? c->d_me_sub : c->d_me) + f * nb8;
        { const int mrc = search_upto(f + 1); if (mrc) return mrc; }
        HIPCHK(c, hipStreamWaitEvent(s, c->me_ev[f], 0));
        HIPCHK(c, launch_recon(&P1, c->d_params, srcf, recf, lvf, blkf, reff, mef, P.part_map ? P.part_map + f * nsb : nullptr, s));
      }
      if (!av1mi_frame_is_inter(P, (int)f)) { for (int i = 0; i < 4; i++) P1.lf_level[i] = P.lf_level[i]; }
      else { for (int i = 0; i < 4; i++) P1.lf_level[i] = P.lf_level_inter[i]; }
      if (P1.lf_level[0]) HIPCHK(c, av1mi_launch_deblock(&P1, recf, blkf, s));
      HIPCHK(c, av1mi_launch_cdef(&P1, recf, cdf_, blkf, nullptr, nullptr, nullptr, s));
      if (lr) {
        const int upf = ((P.true_h + 32) / 64 > 0 ? (P.true_h + 32) / 64 : 1) * ((P.true_w + 32) / 64 > 0 ? (P.true_w + 32) / 64 : 1);
        HIPCHK(c, av1mi_launch_lr(&P1, recf, cdf_, srcf, finf, c->d_lrc + (size_t)f * upf, c->d_lrsse + (size_t)f * upf * 8, 0, s));
      }
      { const int mrc = search_upto(f + 1 + me_ahead); if (mrc) return mrc; }   // (after this frame's chain kernels are in the queue)
      if ((f + 1) % grp == 0 && f + 1 < n_frames) {   // a full group that is not the last: hand it to the third stream
        if (c->grp_ev.size() <= n_grp) {
          hipEvent_t e;
          HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
          c->grp_ev.push_back(e);
        }
        HIPCHK(c, hipEventRecord(c->grp_ev[n_grp], s));
        HIPCHK(c, hipStreamWaitEvent(c->stream3, c->grp_ev[n_grp], 0));
        HIPCHK(c, av1mi_launch_entropy(&P, c->d_cdf, c->d_levels, c->d_blk, c->d_streams, c->d_sym, c->d_combos, c->d_slots, c->d_tile_bytes, c->d_lrc,
                                       c->d_tile_off, (int)(f + 1 - grp), (int)grp, sym_groups ? 1 : 3, c->stream3, nullptr, nullptr, nullptr, nullptr));
        n_grp++;
      }
    }
    entropy_from = n_grp * grp;
    if (n_grp) {  // join: packing (main stream) needs every group's tile sizes and bytes
      HIPCHK(c, hipEventRecord(c->ev[10], c->stream3));
    }
    entropy_joined = n_grp != 0;
  }
  HIPCHK(c, hipEventRecord(c->ev[2], s));
  // CDEF (+SSE) depends only on the reconstruction.  The range-coding kernel is a latency-bound
  // serial chain (one lane per tile: 480 waves, half the SIMDs idle), so CDEF runs beside IT on a
  // second stream; symbolize and CDEF are both throughput-bound and would only slow each other.
  hipStream_t s2 = getenv("AV1MI_SERIAL") ? c->stream : c->stream2;  // AV1MI_SERIAL: single-stream timing experiments
  HIPCHK(c, hipEventRecord(c->ev[3], s));
  HIPCHK(c, av1mi_launch_entropy(&P, c->d_cdf, c->d_levels, c->d_blk, c->d_streams, c->d_sym, c->d_combos, c->d_slots, c->d_tile_bytes, c->d_lrc,
                                 c->d_tile_off /* scratch until the packing kernels fill it */, (int)entropy_from, (int)(n_frames - entropy_from),
                                 sym_groups && entropy_joined ? 1 : 3, s, c->ev[7],
                                 c->stream4, c->ev[12], c->ev[13]));   // the frame-edge tiles' symbolize variant beside the regular one
  if (entropy_joined) HIPCHK(c, hipStreamWaitEvent(s, c->ev[10], 0));
  if (sym_groups && entropy_joined)   // every frame is symbolized: one range-coder launch for the whole chunk
    HIPCHK(c, av1mi_launch_entropy(&P, c->d_cdf, c->d_levels, c->d_blk, c->d_streams, c->d_sym, c->d_combos, c->d_slots, c->d_tile_bytes, c->d_lrc,
                                   c->d_tile_off, 0, (int)n_frames, 2, s, nullptr, nullptr, nullptr, nullptr));
  HIPCHK(c, hipEventRecord(c->ev[4], s));
  // (AV1MI_CDEF_SPLIT: the direction search as a kernel of its own beside symbolize - measured slower overall: it takes symbolize's slots)
  const bool split_cdef = !inter_chunk && !lr && P.enable_cdef && c->d_cdefdir && getenv("AV1MI_CDEF_SPLIT");
  if (!inter_chunk && !lr) {  // deblocking and CDEF's direction search read only the reconstruction and block info: beside symbolize
    HIPCHK(c, hipStreamWaitEvent(s2, c->ev[2], 0));
    if (P.lf_level[0]) HIPCHK(c, av1mi_launch_deblock(&P, c->d_rec, c->d_blk, s2));
    if (split_cdef) HIPCHK(c, av1mi_launch_cdef_dir(&P, c->d_rec, c->d_blk, c->d_cdefdir, s2));
  }
  HIPCHK(c, hipStreamWaitEvent(s2, c->ev[7], 0));  // ev[7]: recorded between symbolize and range-code (measured: starting CDEF
                                                   // right after the reconstruction, beside symbolize, costs 8 % overall)
  HIPCHK(c, hipEventRecord(c->ev[8], s2));
  // (an all-key chunk's CDEF is one launch over the chunk: it sums the squared error itself - a second pass over the frames, 0.19 ms at
  // 1080p x 60, would end after the range coder CDEF runs beside; a one-frame chunk's CDEF runs in strips without the sum)
  const bool sse_in_cdef = !inter_chunk && !lr && n_frames > 1 && !getenv("AV1MI_CDEF_STRIPS");
  if (!inter_chunk && !lr) HIPCHK(c, av1mi_launch_cdef(&P, c->d_rec, c->d_fin, c->d_blk, split_cdef ? c->d_cdefdir : nullptr, sse_in_cdef ? d_src : nullptr,
                                                       sse_in_cdef ? c->d_sse : nullptr, s2));
  if (!sse_in_cdef) HIPCHK(c, av1mi_launch_sse(&P, d_src, c->d_fin, c->d_sse, s2));
  HIPCHK(c, hipEventRecord(c->ev[9], s2));
  // packing and the download of the bitstream need nothing from the second stream: they run beside the tail of CDEF / SSE;
  // the join comes before the reconstruction and the SSE are read (below)
  HIPCHK(c, av1mi_launch_pack(&P, c->d_slots, c->d_tile_bytes, c->d_tile_off, c->d_frame_size, c->d_payload, c->d_frame_off, c->d_hdr,
                              c->d_out, c->d_overflow, 0, s));
  HIPCHK(c, av1mi_launch_pack(&P, c->d_slots, c->d_tile_bytes, c->d_tile_off, c->d_frame_size, c->d_payload, c->d_frame_off, c->d_hdr,
                              c->d_out, c->d_overflow, 1, s));
  HIPCHK(c, hipEventRecord(c->ev[5], s));
  // sizes first, then exactly the bytes produced
  std::vector<unsigned long long> foff(n_frames + 1);
  int overflow = 0;
  HIPCHK(c, hipMemcpyAsync(foff.data(), c->d_frame_off, (n_frames + 1) * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(&overflow, c->d_overflow, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  if (overflow) {
    (void)hipStreamSynchronize(s2);  // the retry reuses the buffers the second stream is still reading
    set_err(c, "a tile outgrew its %d-byte bitstream slot or its %d-entry symbol stream", P.tile_slot_bytes, P.stream_cap);
    return AV1MI_E_OVERFLOW;
  }
  const size_t total = (size_t)foff[n_frames];
  if (total > c->out_cap) { (void)hipStreamSynchronize(s2); set_err(c, "internal: packed size exceeds buffer"); return AV1MI_E_OVERFLOW; }
  // the caller's buffer: a page-locked block of the pool (the copy lands in it directly); plain memory + this context's staging
  // buffer only if page-locked memory cannot be had
  uint8_t *host = (uint8_t *)pin_acquire(total ? total : 1);
  const bool direct = host != nullptr;
  if (!direct) {
    host = (uint8_t *)malloc(total ? total : 1);
    if (!host) { (void)hipStreamSynchronize(s2); return AV1MI_E_OOM; }
    if (c->h_out_cap < total) {
      if (c->h_out) (void)hipHostFree(c->h_out);
      c->h_out = nullptr; c->h_out_cap = 0;
      if (hipHostMalloc((void **)&c->h_out, total + (total >> 2) + 4096, hipHostMallocDefault) == hipSuccess) c->h_out_cap = total + (total >> 2) + 4096;
    }
  }
  uint8_t *dst = direct ? host : (c->h_out ? c->h_out : host);
  hipError_t e1 = hipMemcpyAsync(dst, c->d_out, total, hipMemcpyDeviceToHost, s);
  hipError_t e2 = hipEventRecord(c->ev[6], s);
  if (e2 == hipSuccess) e2 = hipStreamWaitEvent(s, c->ev[9], 0);  // join: the final reconstruction and the SSE come from the second stream
  if (e1 == hipSuccess && e2 == hipSuccess && recon) {
    const void *fin = c->d_fin;
    if (padded) {  // crop to the signalled size, in the caller's tight layout
      e1 = av1mi_launch_pad(c->d_fin, c->d_stage, (int)r.p.width, (int)r.p.height, r.cw, r.ch, P.bit_depth, (int)n_frames, 1, s);
      fin = c->d_stage;
    }
    if (e1 == hipSuccess) e1 = hipMemcpyAsync(recon, fin, padded ? user_bytes : chunk_bytes, frames_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s);
  }
  std::vector<unsigned long long> sse(n_frames * 3);
  std::vector<uint32_t> syms;
  if (e1 == hipSuccess) e1 = hipMemcpyAsync(sse.data(), c->d_sse, (size_t)n_frames * 24, hipMemcpyDeviceToHost, s);
  if (e1 == hipSuccess && report) {
    syms.resize((size_t)n_frames * P.tile_rows * P.tile_cols);
    e1 = hipMemcpyAsync(syms.data(), c->d_sym, syms.size() * 4, hipMemcpyDeviceToHost, s);
  }
  // the bitstream is on the host once ev[6] has passed: hand it over to the caller's buffer while the second stream finishes
  if (e1 == hipSuccess && e2 == hipSuccess) e1 = hipEventSynchronize(c->ev[6]);
  if (e1 == hipSuccess && e2 == hipSuccess && dst != host) memcpy(host, dst, total);
  if (e1 == hipSuccess) e1 = hipStreamSynchronize(s);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    (void)hipStreamSynchronize(s2);
    av1mi_free(host);
    set_err(c, "device-to-host copy failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return AV1MI_E_HIP;
  }
  out->data = host; out->size = total;
  if (frame_sizes) for (uint32_t f = 0; f < n_frames; f++) frame_sizes[f] = (uint32_t)(foff[f + 1] - foff[f]);
  if (report) {
    memset(report, 0, sizeof(*report));
    report->frames = n_frames;
    report->bytes = total;
    report->cap_scale = (uint32_t)c->cap_scale;
    report->gpus_used = 1u << (c->device & 31);
    report->chunks = 1;
    const double mx = (double)((1 << P.bit_depth) - 1);
    for (int pl = 0; pl < 3; pl++) {
      double t = 0;
      for (uint32_t f = 0; f < n_frames; f++) t += (double)sse[f * 3 + pl];
      report->sse[pl] = t;
      const double npx = (double)n_frames * P.width * P.height / (pl ? 4 : 1);
      report->psnr[pl] = t > 0 ? 10.0 * log10(mx * mx * npx / t) : 99.0;
    }
    for (uint32_t v : syms) { report->n_symbols += v; if (v > report->max_tile_symbols) report->max_tile_symbols = v; }
    (void)hipEventElapsedTime(&report->ms_h2d, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&report->ms_recon, c->ev[1], c->ev[2]);
    (void)hipEventElapsedTime(&report->ms_cdef, c->ev[8], c->ev[9]);
    (void)hipEventElapsedTime(&report->ms_entropy, c->ev[3], c->ev[4]);
    (void)hipEventElapsedTime(&report->ms_symbolize, c->ev[3], c->ev[7]);
    (void)hipEventElapsedTime(&report->ms_pack, c->ev[4], c->ev[5]);
    (void)hipEventElapsedTime(&report->ms_d2h, c->ev[5], c->ev[6]);
    (void)hipEventElapsedTime(&report->ms_total, c->ev[0], c->ev[6]);
  }
  return AV1MI_OK;
}

}  // extern "C"
