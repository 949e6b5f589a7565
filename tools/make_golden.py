#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the build container; needs Pillow's
libavif = dav1d 1.5.3).

For every case: the oracle encodes a seeded synthclip frame, dav1d decodes the resulting OBU
stream, and the script REQUIRES dav1d's planes to equal the oracle's reconstruction bit for bit
before writing
    <case>.obu        the temporal unit (what both the oracle and the HIP path must reproduce)
    <case>.json       parameters + SHA-256 of dav1d's decoded Y/U/V planes + stats
This is the pin of the oracle's normative half (headers, symbol coder, default CDFs, contexts,
dequantiser, inverse DCT/ADST 4..64, intra predictors, CDEF): see oracle/av1o.h "PARITY PIN".
The reference (av1-base) holds no fixtures for this path, so these are the
golden vectors; inputs are regenerated from the seed, outputs are data, no reference source.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import av1o
import oracle_avif

OUT = os.path.join(ROOT, "tests", "golden")

# name, width, height, bit_depth, seed, t, config overrides
CASES = [
    ("k64_bs5", 64, 64, 8, 7, 0, dict(min_bs_log2=5, max_bs_log2=5)),
    ("k64_bs4", 64, 64, 8, 7, 0, dict(min_bs_log2=4, max_bs_log2=4)),
    ("k64_bs3", 64, 64, 8, 7, 0, dict(min_bs_log2=3, max_bs_log2=3)),
    ("k64_bs6", 64, 64, 8, 7, 0, dict(min_bs_log2=6, max_bs_log2=6)),
    ("k200x120_bs5", 200, 120, 8, 1080, 0, dict(min_bs_log2=5, max_bs_log2=5)),
    ("k200x120_bs4_all13", 200, 120, 8, 1080, 3, dict(min_bs_log2=4, max_bs_log2=4, mode_mask=0x1FFF)),
    ("k200x120_bs3_all13_10b", 200, 120, 10, 1080, 4, dict(min_bs_log2=3, max_bs_log2=3, mode_mask=0x1FFF)),
    ("k200x120_bs4", 200, 120, 8, 1080, 1, dict(min_bs_log2=4, max_bs_log2=4)),
    ("k200x120_bs3", 200, 120, 8, 1080, 2, dict(min_bs_log2=3, max_bs_log2=3)),
    ("k200x120_bs5_10b", 200, 120, 10, 1080, 0, dict(min_bs_log2=5, max_bs_log2=5)),
    ("k200x120_bs4_10b", 200, 120, 10, 1080, 1, dict(min_bs_log2=4, max_bs_log2=4)),
    ("k200x120_static", 200, 120, 8, 1080, 0, dict(min_bs_log2=5, max_bs_log2=5, disable_cdf_update=1)),
    ("k200x120_nocdef", 200, 120, 8, 1080, 0, dict(min_bs_log2=5, max_bs_log2=5, enable_cdef=0)),
    ("k200x120_cdef_strong", 200, 120, 8, 1080, 0, dict(min_bs_log2=4, max_bs_log2=4, cdef_y_pri=7, cdef_y_sec=3, cdef_uv_pri=4, cdef_uv_sec=2, cdef_damping=3)),
    ("k328x248_tiles2x1", 328, 248, 8, 3, 2, dict(min_bs_log2=4, max_bs_log2=4, tile_w_sb=2, tile_h_sb=1)),
    ("k136_onetile", 136, 136, 8, 3, 2, dict(min_bs_log2=5, max_bs_log2=5, tile_w_sb=64, tile_h_sb=64)),
    ("k72x56_q60", 72, 56, 8, 11, 0, dict(min_bs_log2=4, max_bs_log2=4, base_q_idx=60)),
    ("k72x56_q200", 72, 56, 10, 11, 0, dict(min_bs_log2=5, max_bs_log2=5, base_q_idx=200)),
    # film-grain table in the frame header: scaling 0 -> dav1d's output still equals the reconstruction (pins the
    # syntax); scaling 40/20 -> dav1d synthesises grain on top (bounded difference, its hash recorded separately)
    ("k200x120_grain_tab0_10b", 200, 120, 10, 1080, 2, dict(film_grain=1, fg_y_scaling=0, fg_c_scaling=0, fg_seed=1234)),
    ("k200x120_grain20_10b", 200, 120, 10, 1080, 2, dict(film_grain=1, fg_y_scaling=40, fg_c_scaling=20, fg_seed=7391)),
    # loop restoration (luma Wiener, 64x64 units): decision-driven, and fuzzed unit types / coefficients
    ("k200x120_lr", 200, 120, 8, 1080, 1, dict(min_bs_log2=5, max_bs_log2=5, enable_lr=1)),
    ("k328x248_lr_10b", 328, 248, 10, 7, 1, dict(min_bs_log2=4, max_bs_log2=4, enable_lr=1)),
    ("fuzz_lr", 200, 120, 8, 31, 0, dict(min_bs_log2=4, max_bs_log2=4, enable_lr=1, fuzz_modes=3)),
    ("fuzz_lr_10b_onetile", 264, 200, 10, 32, 0, dict(min_bs_log2=5, max_bs_log2=5, enable_lr=1, fuzz_modes=5, tile_w_sb=64, tile_h_sb=64)),
    # RESTORE_SWITCHABLE (enable_lr = 2): per unit off / Wiener / self-guided (all 16 parameter sets and random weights when fuzzed)
    ("k200x120_lr2", 200, 120, 8, 1080, 1, dict(min_bs_log2=5, max_bs_log2=5, enable_lr=2)),
    ("k328x248_lr2_deblock_10b", 328, 248, 10, 7, 0, dict(min_bs_log2=4, max_bs_log2=4, enable_lr=2, deblock=1)),
    ("k202x122_lr2_odd_q180", 202, 122, 8, 42, 1, dict(min_bs_log2=5, max_bs_log2=5, enable_lr=2, base_q_idx=180)),
    ("fuzz_lr2", 200, 120, 8, 31, 0, dict(min_bs_log2=4, max_bs_log2=4, enable_lr=2, fuzz_modes=3)),
    ("fuzz_lr2_10b_onetile", 264, 200, 10, 32, 0, dict(min_bs_log2=5, max_bs_log2=5, enable_lr=2, fuzz_modes=5, tile_w_sb=64, tile_h_sb=64)),
    # frame sizes that are not multiples of 8 (coded at the padded size, signalled exactly)
    ("k70x58_odd", 70, 58, 8, 41, 0, dict(min_bs_log2=5, max_bs_log2=5)),
    ("k202x122_odd_lr_10b", 202, 122, 10, 42, 1, dict(min_bs_log2=4, max_bs_log2=4, enable_lr=1)),
    # deblocking filter: levels picked from q, and explicit levels / sharpness
    ("k200x120_deblock", 200, 120, 8, 1080, 2, dict(min_bs_log2=5, max_bs_log2=5, deblock=1)),
    ("k216x88_deblock_lr_10b", 216, 88, 10, 51, 0, dict(min_bs_log2=4, max_bs_log2=4, deblock=1, enable_lr=1)),
    ("fuzz_deblock_bs3", 136, 72, 8, 52, 0, dict(min_bs_log2=3, max_bs_log2=3, fuzz_modes=6, deblock=2, lf_level=(63, 40, 17, 5), lf_sharpness=3)),
    # quantiser matrices (using_qmatrix): one level for all planes as the HIP path derives it, and split / fuzzed levels
    ("k200x120_qm4", 200, 120, 8, 1080, 1, dict(min_bs_log2=5, max_bs_log2=5, enable_qm=1, qm_y=4, qm_uv=4)),
    ("k232x120_qm9_deblock_10b", 232, 120, 10, 61, 0, dict(min_bs_log2=4, max_bs_log2=4, enable_qm=1, qm_y=9, qm_uv=9, deblock=1)),
    ("k136x72_qm0_bs3", 136, 72, 8, 62, 0, dict(min_bs_log2=3, max_bs_log2=3, enable_qm=1, qm_y=0, qm_uv=0)),
    ("k72x56_qm15_flat", 72, 56, 8, 11, 0, dict(min_bs_log2=5, max_bs_log2=5, enable_qm=1, qm_y=15, qm_uv=15)),
    ("fuzz_qm_y3_uv11_bs6_10b", 136, 136, 10, 63, 0, dict(min_bs_log2=6, max_bs_log2=6, enable_qm=1, qm_y=3, qm_uv=11, fuzz_coeffs=63, fuzz_density=6, fuzz_maxlevel=12, fuzz_modes=2)),
    ("fuzz_qm_y14_uv0_bs4", 136, 72, 8, 64, 0, dict(min_bs_log2=4, max_bs_log2=4, enable_qm=1, qm_y=14, qm_uv=0, fuzz_coeffs=64, fuzz_density=3, fuzz_maxlevel=20, mode_mask=0x1FFF)),
    # blocks that overhang the frame edge (a node is a leaf when its half point is inside: remainders of 56 / 24 samples with
    # 32x32 blocks, 40..56 with 64x64): right edge, bottom edge, the corner; decision-driven and fuzzed, with the loop filters
    ("k248x216_overhang", 248, 216, 8, 71, 0, dict(min_bs_log2=5, max_bs_log2=5, mode_mask=0x1FFF)),
    ("k184x176_overhang_bs6_10b", 184, 176, 10, 72, 1, dict(min_bs_log2=6, max_bs_log2=6, deblock=1, enable_lr=2)),
    ("k88x120_overhang_lr_deblock", 88, 120, 8, 73, 0, dict(min_bs_log2=5, max_bs_log2=5, deblock=1, enable_lr=1, cdef_y_sec=2, cdef_uv_sec=1)),
    ("fuzz_overhang_bs5", 216, 248, 10, 74, 0, dict(min_bs_log2=5, max_bs_log2=5, fuzz_modes=74, fuzz_coeffs=74, fuzz_density=6, fuzz_maxlevel=30)),
    ("fuzz_overhang_bs6_onetile", 248, 184, 8, 75, 0, dict(min_bs_log2=6, max_bs_log2=6, fuzz_modes=75, tile_w_sb=64, tile_h_sb=64, enable_lr=2)),
    # angle deltas chosen by the mode decision (directional winners refined over -3 .. +3), all 13 candidates and the default three
    ("k200x120_angle_all13", 200, 120, 8, 1080, 5, dict(min_bs_log2=4, max_bs_log2=4, mode_mask=0x1FFF, angle_delta=1)),
    ("k248x184_angle_bs5_10b", 248, 184, 10, 81, 0, dict(min_bs_log2=5, max_bs_log2=5, mode_mask=0x1FFF, angle_delta=1, deblock=1)),
    ("k136_angle_dcvh_bs3", 136, 136, 8, 82, 1, dict(min_bs_log2=3, max_bs_log2=3, angle_delta=1)),
    ("k184x176_angle_bs6", 184, 176, 8, 83, 2, dict(min_bs_log2=6, max_bs_log2=6, mode_mask=0x1FFF, angle_delta=1)),
    # enable_intra_edge_filter: filtered / upsampled edges for the directional modes (decision-driven with all 13 candidates and
    # angle deltas; fuzzed modes / angles so that smooth neighbours, every strength and both upsampling cases occur)
    ("k200x120_ef_all13_bs5", 200, 120, 8, 1080, 6, dict(min_bs_log2=5, max_bs_log2=5, mode_mask=0x1FFF, angle_delta=1, intra_edge_filter=1)),
    ("k200x120_ef_all13_bs4_10b", 200, 120, 10, 1080, 7, dict(min_bs_log2=4, max_bs_log2=4, mode_mask=0x1FFF, angle_delta=1, intra_edge_filter=1)),
    ("k202x122_ef_all13_bs3", 202, 122, 8, 85, 0, dict(min_bs_log2=3, max_bs_log2=3, mode_mask=0x1FFF, angle_delta=1, intra_edge_filter=1)),
    ("k184x176_ef_bs6_10b", 184, 176, 10, 86, 1, dict(min_bs_log2=6, max_bs_log2=6, mode_mask=0x1FFF, angle_delta=1, intra_edge_filter=1)),
    ("k136_ef_dirs_bs3", 136, 136, 8, 87, 2, dict(min_bs_log2=3, max_bs_log2=3, mode_mask=0x01FE, intra_edge_filter=1)),
    ("fuzz_ef_bs3", 200, 120, 8, 88, 0, dict(min_bs_log2=3, max_bs_log2=3, mode_mask=0x1FFF, fuzz_modes=3, intra_edge_filter=1)),
    ("fuzz_ef_bs4_10b", 264, 200, 10, 89, 0, dict(min_bs_log2=4, max_bs_log2=4, mode_mask=0x1FFF, fuzz_modes=5, intra_edge_filter=1)),
    ("fuzz_ef_bs5", 264, 200, 8, 90, 0, dict(min_bs_log2=5, max_bs_log2=5, mode_mask=0x1FFF, fuzz_modes=7, intra_edge_filter=1)),
    ("fuzz_ef_bs6", 264, 200, 8, 91, 0, dict(min_bs_log2=6, max_bs_log2=6, mode_mask=0x1FFF, fuzz_modes=9, intra_edge_filter=1)),
    ("fuzz_ef_tiles2x1_10b", 328, 248, 10, 92, 0, dict(min_bs_log2=3, max_bs_log2=3, mode_mask=0x1FFF, fuzz_modes=17, tile_w_sb=2, tile_h_sb=1, intra_edge_filter=1)),
    # chroma from luma (UV_CFL_PRED, spec 7.11.5): decision-driven on key frames (alpha by least squares + neighbours),
    # fuzzed alphas / signs with every block size that allows it, an overhanging block row, levels fuzzed as well
    ("k200x120_cfl_bs5", 200, 120, 8, 1080, 8, dict(min_bs_log2=5, max_bs_log2=5, cfl=1)),
    ("k200x120_cfl_all13_ef_bs4_10b", 200, 120, 10, 1080, 9, dict(min_bs_log2=4, max_bs_log2=4, mode_mask=0x1FFF, angle_delta=1, intra_edge_filter=1, cfl=1)),
    ("k202x122_cfl_bs3", 202, 122, 8, 95, 0, dict(min_bs_log2=3, max_bs_log2=3, mode_mask=0x1FFF, cfl=1)),
    ("k248x216_cfl_overhang_bs5_10b", 248, 216, 10, 96, 1, dict(min_bs_log2=5, max_bs_log2=5, cfl=1, deblock=1)),
    ("k184x176_cfl_bs6", 184, 176, 8, 97, 1, dict(min_bs_log2=6, max_bs_log2=6, cfl=1)),
    ("fuzz_cfl_bs3", 200, 120, 8, 98, 0, dict(min_bs_log2=3, max_bs_log2=3, mode_mask=0x1FFF, fuzz_modes=3, cfl=1)),
    ("fuzz_cfl_bs4_10b_ef", 264, 200, 10, 99, 0, dict(min_bs_log2=4, max_bs_log2=4, mode_mask=0x1FFF, fuzz_modes=5, intra_edge_filter=1, cfl=1)),
    ("fuzz_cfl_bs5", 264, 200, 8, 100, 0, dict(min_bs_log2=5, max_bs_log2=5, mode_mask=0x1FFF, fuzz_modes=7, cfl=1)),
    ("fuzz_cfl_tiles2x1_10b", 328, 248, 10, 101, 0, dict(min_bs_log2=3, max_bs_log2=3, fuzz_modes=17, tile_w_sb=2, tile_h_sb=1, cfl=1)),
    ("fuzz_cfl_levels_10b", 136, 136, 10, 102, 0, dict(min_bs_log2=3, max_bs_log2=3, fuzz_modes=14, fuzz_coeffs=19, fuzz_density=6, fuzz_maxlevel=30, cfl=1)),
    # identity transform (IDTX) for intra luma blocks with sparse residuals (tx_search): decision-driven on posterised sources, fuzzed
    # on regular ones (with quantiser matrices: they do not apply to IDTX blocks, spec 7.12.3)
    ("k200x120_idtx_bs4_post", 200, 120, 8, 1080, 10, dict(min_bs_log2=4, max_bs_log2=4, tx_search=1, src_shift=5)),
    ("k200x120_idtx_bs3_post_10b", 200, 120, 10, 1080, 11, dict(min_bs_log2=3, max_bs_log2=3, mode_mask=0x1FFF, tx_search=1, src_shift=7)),
    ("k248x184_idtx_cfl_bs4_post", 248, 184, 8, 105, 0, dict(min_bs_log2=4, max_bs_log2=4, tx_search=1, cfl=1, intra_edge_filter=1, mode_mask=0x1FFF, src_shift=5)),
    ("fuzz_idtx_bs3", 200, 120, 8, 106, 0, dict(min_bs_log2=3, max_bs_log2=3, mode_mask=0x1FFF, fuzz_modes=3, tx_search=1)),
    ("fuzz_idtx_bs4_qm_10b", 264, 200, 10, 107, 0, dict(min_bs_log2=4, max_bs_log2=4, mode_mask=0x1FFF, fuzz_modes=5, tx_search=1, enable_qm=1, qm_y=4, qm_uv=6)),
    ("fuzz_idtx_levels", 136, 136, 8, 108, 0, dict(min_bs_log2=4, max_bs_log2=4, fuzz_modes=15, fuzz_coeffs=21, fuzz_density=4, fuzz_maxlevel=20, tx_search=1)),
    # colour description in the sequence header (BASELINE config 5 "8K 10-bit HDR": BT.2020 primaries, PQ transfer, BT.2020 NCL matrix;
    # and a BT.709 full-range one): dav1d decodes the stream, libavif's own sequence-header parser reads the code points back
    ("k200x120_hdr_bt2020_pq_10b", 200, 120, 10, 1080, 12, dict(min_bs_log2=5, max_bs_log2=5, color_primaries=9, transfer_characteristics=16, matrix_coefficients=9)),
    ("k72x56_bt709_full", 72, 56, 8, 11, 1, dict(min_bs_log2=4, max_bs_log2=4, color_primaries=1, transfer_characteristics=1, matrix_coefficients=1, color_range=1)),
    # content-driven partition (partition_search): leaves of 8 .. 32 / 8 .. 64 mixed by the source's activity, with the other tools on top
    ("k328x248_part_5_3", 328, 248, 8, 111, 0, dict(min_bs_log2=3, max_bs_log2=5, partition_search=1)),
    ("k264x200_part_6_3_all13_10b", 264, 200, 10, 112, 1, dict(min_bs_log2=3, max_bs_log2=6, partition_search=1, mode_mask=0x1FFF, angle_delta=1)),
    ("k328x248_part_5_4_cfl_ef_lr2_deblock_10b", 328, 248, 10, 113, 2, dict(min_bs_log2=4, max_bs_log2=5, partition_search=1, cfl=1, intra_edge_filter=1, mode_mask=0x1FFF, enable_lr=2, deblock=1)),
    ("k202x122_part_6_3_odd_qm", 202, 122, 8, 114, 0, dict(min_bs_log2=3, max_bs_log2=6, partition_search=1, enable_qm=1, qm_y=5, qm_uv=5)),
    ("fuzz_modes", 136, 72, 8, 21, 0, dict(min_bs_log2=4, max_bs_log2=4, fuzz_modes=121)),
    ("fuzz_coefs_sparse", 64, 64, 8, 22, 0, dict(min_bs_log2=5, max_bs_log2=5, fuzz_coeffs=22, fuzz_density=30, fuzz_maxlevel=300, mode_mask=1)),
    ("fuzz_coefs_dense", 64, 64, 10, 23, 0, dict(min_bs_log2=3, max_bs_log2=3, fuzz_coeffs=23, fuzz_density=2, fuzz_maxlevel=16, mode_mask=1)),
    ("fuzz_both_bs6", 136, 136, 8, 24, 0, dict(min_bs_log2=6, max_bs_log2=6, fuzz_coeffs=24, fuzz_modes=124, fuzz_density=8, fuzz_maxlevel=40)),
]


def sha(planes):
    h = hashlib.sha256()
    for p in planes:
        h.update(np.ascontiguousarray(p.astype("<u2")).tobytes())
    return h.hexdigest()


# inter-coded sequences (key frame + P frames, each predicted from the previous reconstruction): dav1d decodes the
# AVIF image sequence and every frame must equal the oracle's reconstruction
SEQ_CASES = [
    ("p200x120_bs5", 200, 120, 8, 1080, 4, dict(min_bs_log2=5, max_bs_log2=5)),
    ("p200x120_bs4_10b", 200, 120, 10, 1080, 3, dict(min_bs_log2=4, max_bs_log2=4)),
    ("p136_bs3", 136, 136, 8, 3, 3, dict(min_bs_log2=3, max_bs_log2=3)),
    ("p328x248_bs5_me16", 328, 248, 8, 7, 3, dict(min_bs_log2=5, max_bs_log2=5, me_range=16)),
    ("p200x120_static_grain", 200, 120, 10, 1080, 3, dict(min_bs_log2=5, max_bs_log2=5, disable_cdf_update=1, film_grain=1, fg_y_scaling=0,
                                                        fg_c_scaling=0, fg_seed=99)),
    ("p200x120_lr", 200, 120, 8, 1080, 3, dict(min_bs_log2=5, max_bs_log2=5, enable_lr=1)),
    ("pfuzz_lr_10b", 200, 120, 10, 26, 3, dict(min_bs_log2=4, max_bs_log2=4, enable_lr=1, fuzz_modes=12)),
    ("p130x66_odd_lr", 130, 66, 8, 43, 3, dict(min_bs_log2=4, max_bs_log2=4, enable_lr=1)),
    ("pfuzz_90x100_odd_me16", 90, 100, 8, 44, 3, dict(min_bs_log2=3, max_bs_log2=3, fuzz_modes=5, me_range=16)),
    ("p200x120_deblock", 200, 120, 8, 1080, 3, dict(min_bs_log2=5, max_bs_log2=5, deblock=1)),
    ("pfuzz_deblock_bs6_odd", 130, 134, 10, 53, 3, dict(min_bs_log2=6, max_bs_log2=6, fuzz_modes=4, deblock=2, lf_level=(30, 30, 30, 30), lf_sharpness=0)),
    ("p200x120_qm6", 200, 120, 8, 1080, 3, dict(min_bs_log2=5, max_bs_log2=5, enable_qm=1, qm_y=6, qm_uv=6)),
    ("pfuzz_qm_y2_uv12_bs4_10b", 200, 120, 10, 27, 3, dict(min_bs_log2=4, max_bs_log2=4, enable_qm=1, qm_y=2, qm_uv=12, fuzz_modes=8)),
    # quarter-sample vectors + EIGHTTAP interpolation (subpel = 1): decision-driven and fuzzed (random fractional vectors)
    ("p200x120_subpel", 200, 120, 8, 1080, 4, dict(min_bs_log2=5, max_bs_log2=5, subpel=1)),
    ("p328x248_subpel_bs4_10b_me16", 328, 248, 10, 7, 3, dict(min_bs_log2=4, max_bs_log2=4, subpel=1, me_range=16, deblock=1)),
    ("pfuzz_subpel_bs3", 200, 120, 8, 28, 3, dict(min_bs_log2=3, max_bs_log2=3, subpel=1, fuzz_modes=9)),
    ("pfuzz_subpel_bs6_10b_odd", 130, 134, 10, 29, 3, dict(min_bs_log2=6, max_bs_log2=6, subpel=1, fuzz_modes=4)),
    ("pfuzz_subpel_onetile_bs5", 264, 200, 8, 30, 3, dict(min_bs_log2=5, max_bs_log2=5, subpel=1, fuzz_modes=6, tile_w_sb=64, tile_h_sb=64, enable_lr=1)),
    ("p200x120_lr2", 200, 120, 8, 1080, 3, dict(min_bs_log2=5, max_bs_log2=5, enable_lr=2)),
    ("pfuzz_lr2_10b", 200, 120, 10, 26, 3, dict(min_bs_log2=4, max_bs_log2=4, enable_lr=2, fuzz_modes=12)),
    # inter frames with blocks that overhang the frame edge: motion compensation, the candidate list and the filters at the edge
    ("p248x216_overhang", 248, 216, 8, 76, 3, dict(min_bs_log2=5, max_bs_log2=5)),
    ("p216x120_overhang_subpel_deblock_lr2_10b", 216, 120, 10, 77, 3, dict(min_bs_log2=5, max_bs_log2=5, subpel=1, deblock=1, enable_lr=2, me_range=16)),
    ("pfuzz_overhang_bs6_subpel", 184, 248, 8, 78, 3, dict(min_bs_log2=6, max_bs_log2=6, fuzz_modes=11, subpel=1)),
    ("pfuzz_overhang_bs5_onetile", 248, 184, 10, 79, 3, dict(min_bs_log2=5, max_bs_log2=5, fuzz_modes=13, tile_w_sb=64, tile_h_sb=64, deblock=1)),
    ("p200x120_angle_all13", 200, 120, 8, 84, 3, dict(min_bs_log2=5, max_bs_log2=5, mode_mask=0x1FFF, angle_delta=1)),
    ("p200x120_ef_all13", 200, 120, 8, 93, 3, dict(min_bs_log2=4, max_bs_log2=4, mode_mask=0x1FFF, angle_delta=1, intra_edge_filter=1)),
    ("pfuzz_ef_bs3_10b", 136, 120, 10, 94, 3, dict(min_bs_log2=3, max_bs_log2=3, mode_mask=0x1FFF, fuzz_modes=21, intra_edge_filter=1)),
    ("p200x120_cfl", 200, 120, 8, 103, 3, dict(min_bs_log2=5, max_bs_log2=5, cfl=1)),
    ("pfuzz_cfl_bs4_10b", 136, 120, 10, 104, 3, dict(min_bs_log2=4, max_bs_log2=4, mode_mask=0x1FFF, fuzz_modes=23, cfl=1)),
    ("p328x248_part_6_3_subpel", 328, 248, 8, 115, 3, dict(min_bs_log2=3, max_bs_log2=6, partition_search=1, subpel=1)),
    ("p264x200_part_5_3_lr2_deblock_10b", 264, 200, 10, 116, 3, dict(min_bs_log2=3, max_bs_log2=5, partition_search=1, enable_lr=2, deblock=1, me_range=16)),
    # hierarchical motion search (me_presearch): the clip sampled every 12th frame - the pan moves (24, 12) samples per frame, the rectangles
    # (+-36, +-24): beyond the +-8 / +-16 of the one-level search
    ("p328x248_presearch_fast", 328, 248, 8, 118, 3, dict(min_bs_log2=5, max_bs_log2=5, me_presearch=1, t_step=12)),
    ("p392x264_presearch_part_6_3_subpel_10b", 392, 264, 10, 119, 3, dict(min_bs_log2=3, max_bs_log2=6, partition_search=1, me_presearch=1, subpel=1, me_range=16, t_step=12)),
    ("pfuzz_bs4", 200, 120, 8, 21, 4, dict(min_bs_log2=4, max_bs_log2=4, fuzz_modes=7)),
    ("pfuzz_bs3_all13", 200, 120, 8, 22, 3, dict(min_bs_log2=3, max_bs_log2=3, fuzz_modes=9, mode_mask=0x1FFF)),
    ("pfuzz_bs6", 136, 136, 8, 23, 3, dict(min_bs_log2=6, max_bs_log2=6, fuzz_modes=3)),
    ("pfuzz_onetile_bs3", 328, 248, 8, 24, 3, dict(min_bs_log2=3, max_bs_log2=3, tile_w_sb=64, tile_h_sb=64, fuzz_modes=5)),
    ("pfuzz_tiles2x2_me16", 328, 248, 10, 25, 3, dict(min_bs_log2=4, max_bs_log2=4, tile_w_sb=2, tile_h_sb=2, fuzz_modes=6, me_range=16)),
]


def make_sequences():
    index = []
    for name, w, h, bd, seed, n, kw in SEQ_CASES:
        kw = dict(kw)
        t_step = kw.pop("t_step", 1)   # frame i of the sequence is frame i * t_step of the clip (fast motion)
        cfg = av1o.default_config(w, h, bd, **kw)
        tus, recs, ref, prev, modes, ninter = [], [], None, None, [0, 0, 0, 0], 0
        for t in range(n):
            src = av1o.synthclip_frame(w, h, bd, seed=seed, t=t * t_step)
            tu, rec, st = av1o.encode_frame(cfg, src, with_seq_hdr=(t == 0), ref=ref, prev_src=prev)
            tus.append(tu)
            recs.append(rec)
            ref, prev = rec, src
            ninter += int(st.n_inter_blocks)
            modes = [a + int(b) for a, b in zip(modes, st.inter_mode_hist)]
        dec = oracle_avif.decode_sequence(oracle_avif.wrap_avis(tus, w, h, bd), w, h)
        if len(dec) != n:
            raise SystemExit("%s: dav1d decoded %d of %d frames" % (name, len(dec), n))
        for t in range(n):
            for p in range(3):
                if not (dec[t][p].astype(np.uint16) == recs[t][p]).all():
                    raise SystemExit("%s: frame %d plane %d: dav1d output differs from the oracle reconstruction" % (name, t, p))
        open(os.path.join(OUT, name + ".obu"), "wb").write(b"".join(tus))
        meta = dict(name=name, width=w, height=h, bit_depth=bd, seed=seed, frames=n, config=kw, frame_bytes=[len(x) for x in tus],
                    **({"t_step": t_step} if t_step != 1 else {}),
                    dav1d_sha256=[sha(d) for d in dec], inter_blocks=ninter, inter_modes_nearest_near_global_new=modes,
                    decoder="dav1d 1.5.3 via libavif 1.4.1 (Pillow 12.2.0), AVIF image sequence")
        json.dump(meta, open(os.path.join(OUT, name + ".json"), "w"), indent=1, sort_keys=True)
        index.append(name)
        print("%-24s %s B  inter blocks %d modes %s  dav1d == oracle recon on %d frames" % (name, meta["frame_bytes"], ninter, modes, n))
    json.dump(index, open(os.path.join(OUT, "index_seq.json"), "w"), indent=1)


def main():
    os.makedirs(OUT, exist_ok=True)
    index = []
    for name, w, h, bd, seed, t, kw in CASES:
        kw = dict(kw)
        src_shift = kw.pop("src_shift", 0)   # posterised source (low bits cleared): flat areas with sharp edges, sparse residuals
        src = [(p >> src_shift) << src_shift for p in av1o.synthclip_frame(w, h, bd, seed=seed, t=t)]
        cfg = av1o.default_config(w, h, bd, **kw)
        tu, rec, st = av1o.encode_frame(cfg, src)
        dec = oracle_avif.decode_obus(tu, w, h, bd)
        grain = bool(kw.get("fg_y_scaling") or kw.get("fg_c_scaling"))
        for p in range(3):
            d = np.abs(dec[p].astype(np.int32) - rec[p].astype(np.int32))
            if grain:
                if d.max() == 0 or d.max() > 64 or d.mean() > 8:
                    raise SystemExit("%s: plane %d: dav1d's grain is not a bounded perturbation of the reconstruction (max %d)" % (name, p, d.max()))
            elif d.max() != 0:
                raise SystemExit("%s: dav1d output differs from the oracle reconstruction in plane %d" % (name, p))
        if kw.get("color_primaries"):
            cicp = oracle_avif.decode_colour(oracle_avif.wrap_avif(tu, w, h, bd))
            want = (kw["color_primaries"], kw["transfer_characteristics"], kw["matrix_coefficients"], 1 if kw.get("color_range") else 0)
            if tuple(cicp) != want:
                raise SystemExit("%s: libavif reads the colour description as %s, the oracle wrote %s" % (name, cicp, want))
        open(os.path.join(OUT, name + ".obu"), "wb").write(tu)
        meta = dict(name=name, width=w, height=h, bit_depth=bd, seed=seed, t=t, config=kw, src_shift=src_shift, bytes=len(tu),
                    dav1d_sha256=sha(dec), recon_sha256=sha(rec), dav1d_applies_grain=grain, src_sha256=sha(src), n_symbols=int(st.n_symbols),
                    **({"block_sizes_8_16_32_64": [int(st.bs_hist[k]) for k in (3, 4, 5, 6)]} if kw.get("partition_search") else {}),
                    psnr=[round(x, 3) for x in av1o.psnr(st, cfg)], decoder="dav1d 1.5.3 via libavif 1.4.1 (Pillow 12.2.0)")
        json.dump(meta, open(os.path.join(OUT, name + ".json"), "w"), indent=1, sort_keys=True)
        index.append(name)
        print("%-24s %6d B  psnr %s  dav1d == oracle recon" % (name, len(tu), meta["psnr"]))
    json.dump(index, open(os.path.join(OUT, "index.json"), "w"), indent=1)
    make_sequences()


if __name__ == "__main__":
    main()
