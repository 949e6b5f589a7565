/* av1o.h - CPU oracle ("golden model") for the MI355X AV1 chunk-encode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product path (av1-base_amd/, include/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * What it restates: the reference (IONIQ6000/av1-base) has NO codec arithmetic of its own -
 * its hot path is `run_av1an` (crates/daemon/src/encode/av1an.rs:126-139) which forks the
 * external av1an -> SVT-AV1 stack.  This oracle is therefore
 * a plain-C restatement of (1) the encoder algorithm this build defines for that path
 *  and (2) the NORMATIVE AV1 decoding processes the encoder must mirror
 * (AV1 Bitstream & Decoding Process Specification: §5.5 sequence header, §5.9 frame header,
 * §5.11 tile/block syntax incl. the inter syntax and §7.10.2 motion vector prediction, §7.11.2 intra
 * prediction, §7.11.3 motion compensation (BILINEAR and EIGHTTAP), §7.12 dequant incl. quantiser
 * matrices, §7.13 inverse transforms, §7.14 deblocking, §7.15 CDEF, §7.17 loop restoration (Wiener and
 * self-guided), §8.2 symbol coder).
 *
 * PARITY PIN: parity with SVT-AV1 output is UNPINNED (no av1an/SVT-AV1 anywhere, the
 * reference holds no media fixtures: SURVEY.md §8c).  The normative half IS pinned: streams
 * produced by this oracle are decoded by dav1d 1.5.3 (inside the container's Pillow/libavif)
 * and must reproduce the oracle's reconstruction bit-exactly; those streams + dav1d outputs are
 * committed under tests/golden/ (tools/make_golden.py).
 */
#ifndef AV1O_H
#define AV1O_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (AV1 spec §6.10.x symbol values) ------------------------------------------ */
enum { DC_PRED, V_PRED, H_PRED, D45_PRED, D135_PRED, D113_PRED, D157_PRED, D203_PRED, D67_PRED,
       SMOOTH_PRED, SMOOTH_V_PRED, SMOOTH_H_PRED, PAETH_PRED, UV_CFL_PRED, N_INTRA_MODES = 13 };
enum { PARTITION_NONE, PARTITION_HORZ, PARTITION_VERT, PARTITION_SPLIT };
enum { DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST, FLIPADST_DCT, DCT_FLIPADST, FLIPADST_FLIPADST,
       ADST_FLIPADST, FLIPADST_ADST, IDTX, V_DCT, H_DCT, V_ADST, H_ADST, V_FLIPADST, H_FLIPADST };
/* square transform sizes only in this build: log2 size 2..6 */
enum { TX_4X4, TX_8X8, TX_16X16, TX_32X32, TX_64X64 };

/* ---- encoder configuration ------------------------------------------------------------- */
typedef struct {
  int width, height;      /* luma; even, >= 8.  Sizes that are not multiples of 8 are coded at the next multiple of 8
                             (the source is extended by replicating its last column/row) and signalled exactly */
  int bit_depth;          /* 8 or 10 */
  int base_q_idx;         /* CQ 30 <-> 120 */
  int tile_w_sb, tile_h_sb; /* tile size in 64x64 superblocks */
  int min_bs_log2, max_bs_log2; /* leaf block size range, log2 (3..6) */
  int cdef_y_pri, cdef_y_sec, cdef_uv_pri, cdef_uv_sec, cdef_damping; /* one strength set; cdef_bits = 0 */
  int enable_cdef;
  uint32_t mode_mask;     /* bit m set -> luma intra mode m is a candidate */
  int still_picture;      /* 1: reduced still-picture headers (AVIF style) */
  int disable_cdf_update; /* 1: static default CDFs (no per-symbol adaptation) */
  /* film-grain table in every frame header (the reference runs `--film-grain 20`,
   * av1an.rs:14): fixed 2-point luma / chroma scaling functions, AR lag 0.  Synthesis is decoder-side. */
  int film_grain;         /* 1: film_grain_params_present + apply_grain */
  int fg_y_scaling, fg_c_scaling; /* 0..255 scaling value of both points */
  int fg_seed;            /* grain_seed of this frame (16 bits) */
  int deblock;            /* 1: deblocking filter on, levels picked from the quantiser; 2: levels given below */
  int lf_level[4];        /* loop_filter_level[0..3]: luma vertical edges, luma horizontal, U, V (deblock == 2) */
  int lf_sharpness;
  int enable_lr;          /* 1: loop restoration on luma: Wiener, 64x64 units, per-unit choice among {off, 3 filters};
                             2: RESTORE_SWITCHABLE - per unit {off, 3 Wiener filters, 3 self-guided filters (set, weights)} */
  int true_width, true_height; /* internal: set by the encoder when it runs at the padded size (0 = same as width/height) */
  int me_range;           /* inter frames: integer-pel full search, |dx|,|dy| <= me_range (default 8) */
  int subpel;             /* inter frames: 0 = whole-sample vectors, frame filter BILINEAR (chroma of odd vectors only); 1 = quarter-sample
                             vectors (half- then quarter-sample refinement of the full search) and the EIGHTTAP filter
                              */
  /* quantiser matrices (§5.9.12 using_qmatrix, §7.12.3; the reference runs `--enable-qm 1
   * --qm-min 1 --qm-max 15`, av1an.rs:14): level 0 (steepest) .. 14, 15 = flat (no matrix for that plane) */
  int enable_qm;          /* 1: using_qmatrix */
  int qm_y, qm_uv;        /* qm_y; qm_u = qm_v (separate_uv_delta_q is 0) */
  int angle_delta;        /* 1: a directional winner of the luma mode decision (V, H, D45 .. D67) is refined over the angle deltas
                             -3 .. +3 (3 degrees each, §7.11.2.4) by closed-loop SAD; chroma follows luma (same mode, same delta) */
  int color_range;        /* color_config.color_range: 0 = studio / limited (default: what Y4M and the reference's ffmpeg -> SVT-AV1
                             pipeline carry), 1 = full */
  int intra_edge_filter;  /* sequence header enable_intra_edge_filter: directional predictions use the filtered / upsampled
                           * edges of spec 7.11.2.9 - 7.11.2.12 (SVT-AV1 and libaom run with it on); 0 = the round-1 streams */
  int cfl;                /* 1: chroma-from-luma prediction (spec 7.11.5) is a candidate for the chroma planes of key-frame blocks up to 32x32 */
  int tx_search;          /* 1: transform type search - intra luma blocks up to 16x16 with a sparse residual take the identity transform (IDTX) */
  int color_primaries, transfer_characteristics, matrix_coefficients; /* color_config's colour description (§5.5.2, CICP code points): all 0 =
                           * none (color_description_present_flag 0); BASELINE config 5 "8K 10-bit HDR" = 9 / 16 / 9 (BT.2020, PQ, BT.2020 NCL) */
  int partition_search;   /* 1: content-driven partition - a node larger than min_bs_log2 and not larger than max_bs_log2 splits when its quadrants
                           * differ in level or activity (partition_wants_split in av1o_enc.c); 0 = every node of size <= max_bs_log2 is a leaf */
  int me_presearch;       /* inter frames: 1 = hierarchical motion search - a quarter-resolution search of +-64 luma samples per superblock gives the
                           * centre the full-resolution search of +-me_range runs around (presearch_centres in av1o_enc.c); 0 = around zero */
  /* test hooks (fuzzing the normative paths against dav1d) */
  int fuzz_coeffs;        /* !=0: replace quantised levels by pseudo-random ones (seeded by this) */
  int fuzz_density;       /* 1/N chance a coefficient is nonzero */
  int fuzz_maxlevel;
  int fuzz_modes;         /* !=0: choose modes pseudo-randomly instead of by cost */
} Av1oConfig;

void av1o_default_config(Av1oConfig *c, int w, int h, int bit_depth);

/* planes are uint16_t for every bit depth (8-bit data stored in the low byte) */
typedef struct {
  int w, h;               /* luma dims */
  uint16_t *p[3];
  int stride[3];
} Av1oFrame;

Av1oFrame *av1o_frame_alloc(int w, int h);
void av1o_frame_free(Av1oFrame *f);

typedef struct {
  uint64_t n_symbols;     /* arithmetic-coded symbols incl. literal bits */
  uint64_t n_blocks;
  uint64_t n_skip_blocks;
  uint64_t sse[3];        /* reconstruction vs source */
  uint64_t mode_hist[13];
  uint64_t bs_hist[7];
  uint64_t n_inter_blocks; /* blocks coded with motion compensation (inter frames) */
  uint64_t inter_mode_hist[4]; /* NEARESTMV, NEARMV, GLOBALMV, NEWMV */
} Av1oStats;

/* Encode one key frame.  Writes a temporal unit (TD [+ sequence header] + OBU_FRAME) to out.
 * recon (may be NULL) receives the decoder-identical reconstruction (post-CDEF).
 * Returns bytes written, or <0 on error. */
long av1o_encode_frame(const Av1oConfig *cfg, const Av1oFrame *src, int with_seq_hdr,
                       uint8_t *out, size_t out_cap, Av1oFrame *recon, Av1oStats *stats);

/* Same, with a reference: ref == NULL -> key frame (as av1o_encode_frame); ref != NULL -> INTER_FRAME predicted from
 * `ref` (the previous frame's final reconstruction, the only reference: LAST_FRAME, slot 0 refreshed every frame), with
 * motion vectors searched against `prev_src` (the previous SOURCE frame: open-loop search).
 * SURVEY.md §8a rows a13 (motion estimation) and a14 (motion compensation). */
long av1o_encode_frame2(const Av1oConfig *cfg, const Av1oFrame *src, const Av1oFrame *ref, const Av1oFrame *prev_src, int with_seq_hdr,
                        uint8_t *out, size_t out_cap, Av1oFrame *recon, Av1oStats *stats);

int av1o_qm_level(int base_q_idx, int qm_min, int qm_max);

/* size of the sequence header OBU etc. helpers used by the tests */
long av1o_write_sequence_header(const Av1oConfig *cfg, uint8_t *out, size_t cap);

/* ---- exported building blocks (tested individually / compared with the HIP kernels) ---- */
void av1o_fwd_txfm2d(const int32_t *resid, int stride, int32_t *coef, int log2n, int tx_type, int bd);
/* 32x32 DCT_DCT as U = (X * Cm^T + 512) >> 10, Y = (Cm * U + 2048) >> 12 with Cm = round(4096 * orthonormal DCT-II) (fdct32_matrix.h) */
void av1o_fwd_dct32x32_matrix(const int32_t *resid, int stride, int32_t *coef);
void av1o_inv_txfm2d_add(const int32_t *dq, uint16_t *dst, int stride, int log2n, int tx_type, int bd, int eob);
void av1o_inv_txfm2d(const int32_t *dq, int32_t *resid, int log2n, int tx_type, int bd);
void av1o_idct1d(int32_t *x, int log2n);
void av1o_fdct1d(int32_t *x, int log2n);
const int16_t *av1o_default_scan(int log2n); /* n*n entries (n<=32) */

/* edge arrays: above[-1..2n-1], left[-1..2n-1] (index 0 of the passed pointer = element -1) */
/* intra edge filter control of one prediction (spec 7.11.2): enable = enable_intra_edge_filter, filter_type = get_filter_type()
 * (a neighbour predicted with a smooth mode), n_top / n_left = Min(w, maxX - x + 1) / Min(h, maxY - y + 1) */
typedef struct { int enable, filter_type, n_top, n_left; } Av1oEdgeCtl;
void av1o_predict_intra_ef(uint16_t *dst, int stride, int log2n, int mode, int angle_delta,
                           const uint16_t *above_m1, const uint16_t *left_m1, int have_above, int have_left, int bd,
                           const Av1oEdgeCtl *ef);
void av1o_predict_intra(uint16_t *dst, int stride, int log2n, int mode, int angle_delta,
                        const uint16_t *above, const uint16_t *left, int have_above, int have_left, int bd);

void av1o_cdef_frame(const Av1oConfig *cfg, const Av1oFrame *in, Av1oFrame *out,
                     const uint8_t *skip_mi, int mi_stride, const int8_t *cdef_idx_sb);

/* ---- deblocking filter (av1o_deblock.c) */
void av1o_deblock_levels(const Av1oConfig *cfg, int is_key, int *levels /* [4] */);
void av1o_deblock_frame(const Av1oConfig *cfg, Av1oFrame *f, const uint8_t *mi_bsl, const uint8_t *mi_skip, const uint8_t *mi_is_inter,
                        int mi_stride, const int *levels, int sharpness);

/* ---- loop restoration (av1o_lr.c) */
typedef struct {
  int8_t type;            /* 0 RESTORE_NONE, 1 RESTORE_WIENER, 2 RESTORE_SGRPROJ (enable_lr == 2 only) */
  int8_t coef[2][3];      /* Wiener: [pass: 0 vertical, 1 horizontal][tap 0..2] */
  int8_t sgr_set;         /* self-guided: lr_sgr_set 0..15 */
  int8_t sgr_xqd[2];      /* LrSgrXqd: weights of the two passes' outputs (w2 = 128 - xqd[0] - xqd[1]) */
} Av1oLrUnit;
extern const int av1o_sgr_params[16][4];
extern const int8_t av1o_sgr_candidates[3][3]; /* { set, xqd0, xqd1 } */
void av1o_sgr_plane(const Av1oConfig *cfg, const Av1oFrame *pre, const Av1oFrame *cdef, Av1oFrame *out, int set, int w0, int w1);
extern const int8_t av1o_wiener_candidates[3][3];
int av1o_lr_units(int size);
void av1o_lr_frame(const Av1oConfig *cfg, const Av1oFrame *pre, const Av1oFrame *cdef, const Av1oFrame *src, Av1oFrame *out,
                   Av1oLrUnit *units, unsigned fuzz);

/* ---- synthclip v1: deterministic integer-only synthetic clip ------------ */
void av1o_synthclip_frame(Av1oFrame *f, int bit_depth, uint64_t seed, int t, int scene_len);

/* ---- range coder (exposed for the known-answer tests) ------------------------------------ */
typedef struct {
  uint32_t low;
  uint32_t rng;
  int cnt;
  uint8_t *buf;
  size_t cap, offs;
  int error;
  uint64_t nsym;
} Av1oRangeEnc;
void av1o_ec_init(Av1oRangeEnc *e, uint8_t *buf, size_t cap);
/* icdf: inverted cdf (32768 - cdf), n-1 entries + terminating 0 + counter; adapts in place */
void av1o_ec_encode_symbol(Av1oRangeEnc *e, int s, uint16_t *icdf, int nsyms);
void av1o_ec_encode_bool(Av1oRangeEnc *e, int val, unsigned f_q15);
void av1o_ec_encode_literal(Av1oRangeEnc *e, unsigned v, int bits);
size_t av1o_ec_finish(Av1oRangeEnc *e);

#ifdef __cplusplus
}
#endif
#endif
