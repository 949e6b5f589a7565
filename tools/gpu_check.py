#!/usr/bin/env python3
"""Ad-hoc GPU-vs-oracle comparison used during bring-up (the permanent version lives in tests/)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av1-base_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import av1mi, av1o

def planes_to_raw(fr, bd):
    dt = np.uint8 if bd == 8 else np.dtype("<u2")
    return b"".join(p.astype(dt).tobytes() for p in fr)

def check(w, h, bd, n, bs, cdf_update=1):
    frames = [av1o.synthclip_frame(w, h, bd, seed=1080, t=t) for t in range(n)]
    raw = b"".join(planes_to_raw(fr, bd) for fr in frames)
    params = av1mi.default_params(w, h, bd, block_log2=bs, cdf_update=cdf_update)
    with av1mi.Context(0) as ctx:
        t0 = time.time()
        data, sizes, rep, recon = ctx.encode_chunk(params, raw, n, want_recon=True)
        dt = time.time() - t0
    cfg = av1o.default_config(w, h, bd, min_bs_log2=bs, max_bs_log2=bs, disable_cdf_update=0 if cdf_update else 1)
    ref = b""; recs = []
    for fr in frames:
        tu, rec, st = av1o.encode_frame(cfg, fr)
        ref += tu; recs.append(rec)
    ok_bits = data == ref
    dtp = np.uint8 if bd == 8 else np.dtype("<u2")
    ref_rec = b"".join(planes_to_raw(r, bd) for r in recs)
    ok_rec = recon.tobytes() == ref_rec
    msg = "%dx%d bd%d n%d bs%d cdf%d: gpu %d B oracle %d B bits %s recon %s | recon %.2f cdef %.2f ec %.2f pack %.2f total %.2f ms (wall %.1f ms) psnr %.2f" % (
        w, h, bd, n, bs, cdf_update, len(data), len(ref), "OK" if ok_bits else "DIFF", "OK" if ok_rec else "DIFF",
        rep.ms_recon, rep.ms_cdef, rep.ms_entropy, rep.ms_pack, rep.ms_total, dt * 1e3, rep.psnr[0])
    print(msg, flush=True)
    if not ok_bits:
        m = min(len(data), len(ref))
        first = next((i for i in range(m) if data[i] != ref[i]), m)
        print("   first differing byte at", first, "sizes gpu", sizes[:4], flush=True)
    if not ok_rec:
        a = np.frombuffer(recon.tobytes(), dtype=dtp); b = np.frombuffer(ref_rec, dtype=dtp)
        bad = np.nonzero(a != b)[0]
        print("   recon mismatches:", len(bad), "first at sample", bad[:5], flush=True)
    return ok_bits and ok_rec

if __name__ == "__main__":
    ok = True
    ok &= check(64, 64, 8, 1, 5)
    ok &= check(64, 64, 8, 1, 4)
    ok &= check(64, 64, 8, 1, 3)
    ok &= check(200, 120, 8, 2, 5)
    ok &= check(200, 120, 8, 2, 4)
    ok &= check(200, 120, 10, 2, 5)
    ok &= check(200, 120, 8, 2, 5, cdf_update=0)
    ok &= check(1920, 1080, 8, 2, 5)
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)
