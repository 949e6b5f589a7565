#!/usr/bin/env python3
"""Diagnostic: where the reconstruction kernel's transform items spend their cycles (s_memtime stamps).

Builds av1-base_amd/build_stamps/libav1mi_stamps.so = the product's objects with recon_kernel.hip recompiled with
-DAV1MI_STAMPS (`--build`, runs without a GPU), then on the GPU box encodes the 1080p clip in three regimes - one key frame
(510 waves: latency), a 30-frame all-key-frame chunk (throughput), a 12-frame IPPP chunk (the P-frame chain) - and prints,
per item class (luma 32/16/8, chroma 16/8/4) and phase, the share of the wave-cycles.  Shares, not run times: the stamps'
waits forbid overlaps the real kernel has (MI355X guide, "In-kernel stamps").  The product library is never touched."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "av1-base_amd")
OUT = os.path.join(PKG, "build_stamps", "libav1mi_stamps.so")
PHASES = ["edges", "dc+3 SADs", "decide+pred+resid", "fwd", "q+dq+inv rows", "inv cols+levels", "recon out", "src load"]
CLASSES = ["luma 32", "luma 16", "luma 8", "chroma 16", "chroma 8", "chroma 4"]


def build():
    sys.path.insert(0, PKG)
    import importlib.util
    spec = importlib.util.spec_from_file_location("av1mi_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    obj = os.path.join(os.path.dirname(OUT), "recon_kernel.o")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc] + mod.FLAGS + ["-DAV1MI_STAMPS", "-c", os.path.join(mod.CSRC, "recon_kernel.hip"), "-o", obj])
    objs = [obj if os.path.basename(o) == "recon_kernel.o" else o for o in
            [os.path.join(mod.OBJDIR, s.rsplit(".", 1)[0] + ".o") for s in mod.SOURCES]]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-lpthread"])
    print("built", OUT)


def main():
    if "--build" in sys.argv:
        build()
        return
    sys.path.insert(0, ROOT)
    sys.path.insert(0, PKG)
    import torch
    import av1mi   # structures only; the encoder below is the stamps build
    import bench
    lib = C.CDLL(OUT)
    lib.av1mi_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.av1mi_encode_chunk.argtypes = [C.c_void_p, C.POINTER(av1mi.Params), C.c_void_p, C.c_uint32, C.c_int, C.POINTER(av1mi.Buf),
                                       C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(av1mi.Report)]
    lib.av1mi_debug_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    lib.av1mi_default_params.argtypes = [C.POINTER(av1mi.Params), C.c_uint32, C.c_uint32, C.c_uint32]
    lib.av1mi_free.argtypes = [C.c_void_p]
    lib.av1mi_ctx_destroy.argtypes = [C.c_void_p]
    h = C.c_void_p()
    assert lib.av1mi_ctx_create(0, C.byref(h)) == 0
    w, hh, bd = 1920, 1080, 10
    clip = bench.make_clip_torch(w, hh, bd, 30, 1080, torch.device("cuda", 0))
    torch.cuda.synchronize()
    for name, n, keyint, mask in (("one key frame (510 waves)", 1, 1, 0), ("30 key frames (15 300 waves)", 30, 1, 0), ("one key frame, all 13 candidates", 1, 1, 0x1FFF),
                                  ("12-frame IPPP chunk (walk + pre per P frame)", 12, 240, 0)):
        p = av1mi.Params()
        lib.av1mi_default_params(C.byref(p), w, hh, bd)
        p.keyint, p.intra_mode_mask = keyint, mask
        out, rep, sizes = av1mi.Buf(), av1mi.Report(), (C.c_uint32 * n)()
        st = (C.c_uint64 * 51)()
        lib.av1mi_debug_stamps(st, 1)
        rc = lib.av1mi_encode_chunk(h, C.byref(p), C.c_void_p(clip.data_ptr()), n, 1, C.byref(out), sizes, None, C.byref(rep))
        assert rc == 0, rc
        lib.av1mi_free(out.data)
        lib.av1mi_debug_stamps(st, 1)
        tot, waves, real = st[48], st[49], st[50]
        print("== %s: recon stage %.3f ms; %d tile-walk waves, %.0f cycles = %.1f us per wave (s_memrealtime): shader clock %.2f GHz; stamped %.0f %%" % (
            name, rep.ms_recon, waves, tot / max(waves, 1), real / max(waves, 1) / 100.0, tot / max(real, 1) * 0.1, 100.0 * sum(st[:48]) / max(tot, 1)))
        for c, cn in enumerate(CLASSES):
            row = [st[c * 8 + k] for k in range(8)]
            if sum(row) == 0:
                continue
            print("   %-10s %5.1f %% of the wave |" % (cn, 100.0 * sum(row) / max(tot, 1)), "  ".join("%s %4.1f%%" % (PHASES[k], 100.0 * row[k] / sum(row)) for k in range(8)))
    lib.av1mi_ctx_destroy(h)


if __name__ == "__main__":
    main()
