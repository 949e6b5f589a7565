#!/usr/bin/env python3
"""Build libav1mi.so (HIP kernels + C ABI host) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the resulting av1-base_amd/libav1mi.so is git-ignored but
travels with the tree to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# AV1MI_BUILD_VARIANT=<name> + AV1MI_EXTRA_FLAGS="-D...": a second library av1-base_amd/ab/libav1mi_<name>.so from objects of its own
# (same-box A/B runs: AV1MI_LIB selects the library at run time); the product build is untouched
_VARIANT = os.environ.get("AV1MI_BUILD_VARIANT", "")
OUT = os.path.join(HERE, "ab", "libav1mi_%s.so" % _VARIANT) if _VARIANT else os.path.join(HERE, "libav1mi.so")
OBJDIR = os.path.join(HERE, "build_" + _VARIANT if _VARIANT else "build")
SOURCES = ["recon64_kernel.hip", "recon64_8_kernel.hip", "recon_kernel.hip", "recon8_kernel.hip", "entropy_kernel.hip", "cdef_pack_kernels.hip", "scene_kernels.hip", "me_kernel.hip", "lr_kernel.hip", "deblock_kernel.hip", "av1mi_host.cpp", "av1mi_file.cpp", "av1mi_exec.cpp"]
# -fno-optimize-sibling-calls: keeps LLVM from marking the calls of the `noinline` transform items `tail`.  With the marker the
# backend's interprocedural register allocation treats the items as ordinary ABI functions that save and restore every
# callee-saved VGPR they touch - 33 stores + 29 loads of 256 B per call, 60 % of the reconstruction kernel's HBM traffic
# (profiles/r02_a_pmc_intra.txt vs r02_c); without it the items save nothing and their callers keep nothing in those registers.
FLAGS = os.environ.get("AV1MI_EXTRA_FLAGS", "").split() + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-optimize-sibling-calls", "-Wall", "-Wno-unused-function", "-Wno-missing-braces",
         "-Wno-pass-failed"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(not os.path.exists(d) or os.path.getmtime(d) > t for d in deps)


def _deps_of(obj, fallback):
    """the headers the object was built from (the compiler's -MMD depfile), or every header if there is no depfile yet"""
    d = obj + ".d"
    if not os.path.exists(d):
        return fallback
    txt = open(d).read().replace("\\\n", " ")
    parts = txt.split(":", 1)[1].split() if ":" in txt else []
    return [x for x in parts if x] or fallback


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(HERE, "..", "include", "av1mi.h"), os.path.join(CSRC, "recon_kernel.hip")]   # (recon64_kernel.hip includes recon_kernel.hip)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _newer(obj, _deps_of(obj, [src] + hdrs)):
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-MMD", "-MF", obj + ".d", "-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=int(os.environ.get("AV1MI_BUILD_JOBS", "6"))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or not os.path.exists(OUT):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-lpthread"])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
