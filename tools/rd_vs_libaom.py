#!/usr/bin/env python3
"""Reported-only rate/quality context (BASELINE.json: "PSNR ... vs CPU ref"): the luma plane of synthclip frame 0 (1080p,
8-bit) coded as a key frame by libaom 3.13.2 (through Pillow's libavif: monochrome, end-usage=q, cq-level 30, speeds 8/6/3)
and by this build's algorithm (the oracle; the HIP path produces the same bytes) at several quantisers.  Needs the build
container's Pillow; results are quoted in DESIGN.md §5."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import av1o
import oracle_avif


def psnr(a, b):
    return 10 * np.log10(255.0 ** 2 / ((a.astype(float) - b.astype(float)) ** 2).mean())


def main():
    w, h = 1920, 1080
    src = av1o.synthclip_frame(w, h, 8, seed=1080, t=0)
    y = src[0].astype(np.uint8)
    for sp in (8, 6, 3):
        t0 = time.time()
        av = oracle_avif.libaom_encode_gray(y, cq=30, speed=sp, threads=8)
        dt = time.time() - t0
        p, _ = oracle_avif.decode_yuv(av)
        print("libaom cq 30 speed %d: %7d B  PSNR-Y %.2f dB  %.2f s (8 threads)" % (sp, len(av), psnr(p[0], y), dt))
    flat = [src[0], np.full_like(src[1], 128), np.full_like(src[2], 128)]   # chroma flat: the comparison is luma only
    for q in (120, 100, 92, 84, 76, 68):
        cfg = av1o.default_config(w, h, 8, min_bs_log2=5, max_bs_log2=5, base_q_idx=q)
        tu, rec, _ = av1o.encode_frame(cfg, flat)
        print("this build base_q_idx %3d: %7d B  PSNR-Y %.2f dB" % (q, len(tu), psnr(rec[0], y)))


if __name__ == "__main__":
    main()
