#!/usr/bin/env python3
"""End-to-end check, part 1 (on the MI355X box): synthesise a clip with scene cuts as Y4M, encode it through the
run_av1an drop-in (scene-cut chunks, IPPP with quarter-sample vectors, quantiser matrices, deblocking, switchable restoration, all 13 intra
modes with angle deltas, the intra edge filter, chroma from luma and the identity-transform search on 16x16 blocks, several
chunks in flight) and leave the IVF under gpurun_out/.
Part 2 (tools/e2e_decode.py, where dav1d is available) decodes it and compares with the source."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "av1-base_amd"))
import numpy as np
import bench, av1mi

def main():
    w, h, bd, n, scene_len = 640, 360, 10, 96, 30
    out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
    y4m = os.path.join(out_dir, "e2e.y4m")
    with open(y4m, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420p10\n" % (w, h))
        for t in range(n):
            fr = bench.synthclip_frame(w, h, bd, 4000 + t // scene_len, t % scene_len)   # re-seeded every scene_len frames
            f.write(b"FRAME\n" + b"".join(p.astype("<u2").tobytes() for p in fr))
    out = os.path.join(out_dir, "e2e.ivf")
    rep = av1mi.run_mi355x(av1mi.EncodeParams(y4m, out, out_dir, av1mi.derive_plan(16), chunk_frames=0, keyint=240, enable_lr=2, film_grain=0,
                                              deblock=1, subpel=1, enable_qm=1, qm_min=1, qm_max=15, block_log2=4, intra_mode_mask=0x1FFF, intra_angle_delta=1,
                                              intra_edge_filter=1, cfl=1, tx_search=1))   # every optional tool on (grain off: PSNR is compared)
    os.remove(y4m)
    print("frames %d chunks %d bytes %d psnr %.2f %.2f %.2f" % (rep.frames, rep.chunks, rep.bytes, rep.psnr[0], rep.psnr[1], rep.psnr[2]))

main()
