#!/usr/bin/env python3
"""End-to-end check, part 2: decode gpurun_out/e2e.ivf (written by tools/e2e_encode.py on the GPU box) with dav1d
(through libavif, as an AVIF image sequence) and compare every frame with the regenerated source."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bench, oracle_avif

def main():
    w, h, bd, scene_len = 640, 360, 10, 30
    blob = open(os.path.join(ROOT, "gpurun_out", "e2e.ivf"), "rb").read()
    tus, pos = [], 32
    while pos < len(blob):
        sz = int.from_bytes(blob[pos:pos + 4], "little"); tus.append(blob[pos + 12:pos + 12 + sz]); pos += 12 + sz
    keys = [i + 1 for i, t in enumerate(tus) if t[2] == 0x0A]   # temporal units that carry a sequence header
    dec = oracle_avif.decode_sequence(oracle_avif.wrap_avis(tus, w, h, bd, sync=keys), w, h)
    assert len(dec) == len(tus), (len(dec), len(tus))
    sse = [0.0, 0.0, 0.0]
    for t, d in enumerate(dec):
        src = bench.synthclip_frame(w, h, bd, 4000 + t // scene_len, t % scene_len)
        for p in range(3):
            sse[p] += float(((d[p].astype(np.int64) - src[p].astype(np.int64)) ** 2).sum())
    n = len(dec)
    psnr = [10 * math.log10(1023.0 ** 2 * n * w * h / (1 if p == 0 else 4) / sse[p]) for p in range(3)]
    print("dav1d decoded %d frames, key frames at %s, %d bytes; PSNR vs source %.2f %.2f %.2f dB" % (n, [k - 1 for k in keys], len(blob), *psnr))

main()
