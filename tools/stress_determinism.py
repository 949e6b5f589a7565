#!/usr/bin/env python3
"""Bring-up helper: run the same file encode many times with several chunks in flight and report any difference."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av1-base_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import av1o, av1mi

def tus_of(blob):
    out, pos = [], 32
    while pos < len(blob):
        sz = int.from_bytes(blob[pos:pos + 4], "little"); out.append(blob[pos + 12:pos + 12 + sz]); pos += 12 + sz
    return out

def main():
    w, h = 136, 72
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    d = tempfile.mkdtemp()
    y4m = os.path.join(d, "c.y4m")
    with open(y4m, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F25:1 Ip A1:1 C420jpeg\n" % (w, h))
        for t in range(n):
            fr = av1o.synthclip_frame(w, h, 8, seed=78, t=t)
            f.write(b"FRAME\n" + b"".join(p.astype(np.uint8).tobytes() for p in fr))
    for keyint in (1, 2):
        for workers in (1, 4):
            ref, bad = None, 0
            for i in range(reps):
                out = os.path.join(d, "o%d.ivf" % i)
                plan = av1mi.derive_plan(8, workers_override=workers)
                import time
                t0 = time.time()
                av1mi.run_mi355x(av1mi.EncodeParams(y4m, out, d, plan, chunk_frames=4, keyint=keyint))
                t = tus_of(open(out, "rb").read())
                if ref is None: ref = t
                elif t != ref:
                    bad += 1
                    diff = [k for k in range(n) if t[k] != ref[k]]
                    print("  keyint", keyint, "workers", workers, "run", i, "frames differ:", diff, [(len(t[k]), len(ref[k])) for k in diff][:4])
            print("keyint %d workers %d: %d of %d runs differ" % (keyint, workers, bad, reps))

main()
