#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X AV1 chunk-encode path.

Metric (BASELINE.json): encoded frames/s at CQ=30.  Headline workload at N=1 = BASELINE.json configs[1]:
1080p, 60-frame all-key-frame (intra-only) `synthclip v1` chunk, 10-bit
(the reference's pixel format, av1an.rs:90) unless --bit-depth 8.  One "step" = one complete
encode of one chunk per GPU: source frames already resident in HBM -> reconstruction, CDEF,
entropy coding, bitstream packing on the GPU -> complete OBU bitstream on the host.

The headline evaluates all 13 luma intra modes for every block (BASELINE config 2: "DCT/ADST + directional intra pred + CDEF").

At N=1 with default flags the same JSON line carries a `configs` object with the other BASELINE
configurations a single GPU can run (each with its own fps, roofline and stage times): config 2 with the cheapest
candidate set {DC, V, H}, at 8 bit, with 64x64 blocks and at the quantiser that matches libaom's quality; config 3 (1080p IPPP, 1 and
4 chunks in flight); config 4's per-GPU unit (4K 10-bit IPPP chunk); config 5's (8K 10-bit HDR chunk with a film-grain table) and the
reference's production operating point (av1an.rs:14).

Multi-GPU (`--gpus N`: the ranks are started by torchrun / torch.distributed.run, or by this script itself as child processes
when it is run without a launcher): scene-chunks are independent, every rank encodes its own chunk (seed 1080 + rank),
no data-path collective; weak scaling.  `--workload cfg4` instead shards BASELINE config 4's job of 8 4K chunks over the ranks with the
product's placement rule (strong scaling).  The only collectives are the timing barrier and the MAX over ranks.

CPU baseline (reported only; rank 0, N=1): runs BEFORE anything touches the GPU, in SURVEY §8d's order -
the reference's own av1an / SVT-AV1 if on PATH, else libaom 3.13.2 through the image's libavif (8-bit: this
libaom has no high-bit-depth build), else the build's own C restatement ("port").

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "av1-base_amd"))


_M64 = (1 << 64) - 1


def _splitmix64(x):
    """numpy uint64 arrays (or a Python int); arithmetic modulo 2^64"""
    import numpy as np
    if isinstance(x, int):
        x = (x + 0x9E3779B97F4A7C15) & _M64
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
        return x ^ (x >> 31)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _rects(w, h, seed, t):
    """the six moving rectangles of `synthclip v1`: (x, y, w, h, [value per plane])"""
    rects = []
    for k in range(6):
        hh = _splitmix64((seed * 977 + k) & _M64)
        vx, vy = (3 if hh & 1 else -3), (2 if hh & 2 else -2)
        rw, rh = 32 + (hh >> 8) % (w // 4 + 1), 32 + (hh >> 24) % (h // 4 + 1)
        rx, ry = (hh >> 40) % w + vx * t, (hh >> 52) % h + vy * t
        h2 = _splitmix64(hh)
        rects.append((rx % w, ry % h, rw, rh, [(h2 >> 3) & 255, (h2 >> 13) & 255, (h2 >> 23) & 255]))
    return rects


def synthclip_frame(w, h, bd, seed, t):
    """`synthclip v1` in numpy: gradients panning (2, 1) px/frame, 6 opaque rectangles moving (+-3, +-2),
    splitmix64 noise; byte for byte what oracle/av1o_synth.c generates (tests/test_oracle.py checks that), written here
    so that synthesising the benchmark input needs nothing from oracle/."""
    import numpy as np

    def tri(v, P):
        m = v % (2 * P)
        m = np.where(m > P, 2 * P - m, m)
        return (m * 64) // P
    sh, maxv, G = bd - 8, (1 << bd) - 1, (1 if bd == 8 else 4)
    rects = _rects(w, h, seed, t)
    planes = []
    for pl in range(3):
        ss = 1 if pl else 0
        pw, ph = w >> ss, h >> ss
        A1, A2, P1, P2 = (48, 32, 53, 41) if pl else (96, 64, 97, 61)
        base = 128 - (A1 + A2) // 2 if pl else 16
        x = np.arange(pw, dtype=np.int64)[None, :]
        y = np.arange(ph, dtype=np.int64)[:, None]
        v = base + (tri(x + ((2 * t) >> ss), P1) * A1) // 64 + (tri(y + (t >> ss), P2) * A2) // 64
        v = np.broadcast_to(v, (ph, pw)).copy()
        fx, fy = x << ss, y << ss
        for (px, py, rw, rh, rv) in rects:
            inside = (fx >= px) & (fx < px + rw) & (fy >= py) & (fy < py + rh)
            v = np.where(inside, rv[pl], v)
        v = v << sh
        key = (np.uint64(seed) ^ np.uint64((t << 40) & _M64)) ^ ((np.uint64(pl * 8192) + y.astype(np.uint64)) << np.uint64(20)) ^ x.astype(np.uint64)
        nz = _splitmix64(key)
        v = v + ((nz & np.uint64(15)).astype(np.int64) - 8) * G
        planes.append(np.clip(v, 0, maxv).astype(np.uint16))
    return planes


def make_clip(w, h, bd, n, seed):
    """synthclip v1 frames as one I420 byte string (numpy; the CPU legs and small cases)"""
    import numpy as np
    dt = np.uint8 if bd == 8 else np.dtype("<u2")
    out = []
    for t in range(n):
        fr = synthclip_frame(w, h, bd, seed, t)
        out.append(b"".join(p.astype(dt).tobytes() for p in fr))
    return b"".join(out)


def _s64(v):
    """a 64-bit pattern as the int64 torch works in"""
    v &= _M64
    return v - (1 << 64) if v >> 63 else v


def make_clip_torch(w, h, bd, n, seed, device):
    """The same clip, synthesised where it is needed (HBM) with torch integer arithmetic: a 60-frame 1080p clip takes
    a fraction of a second instead of 14 s of numpy (4K: 35 s).  int64 two's-complement multiplies wrap like uint64 ones;
    logical right shifts are arithmetic shifts with the sign extension masked off.  tests/test_oracle.py checks it against
    the numpy and the C generators on the CPU device.  Returns a uint8 tensor (I420, little-endian 16-bit above 8 bit)."""
    import torch

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    def mix(x):
        x = x + _s64(0x9E3779B97F4A7C15)
        x = (x ^ lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
        x = (x ^ lsr(x, 27)) * _s64(0x94D049BB133111EB)
        return x ^ lsr(x, 31)

    def tri(v, P):
        m = v % (2 * P)
        m = torch.where(m > P, 2 * P - m, m)
        return (m * 64) // P
    sh, maxv, G = bd - 8, (1 << bd) - 1, (1 if bd == 8 else 4)
    bps = 2 if bd > 8 else 1
    out = torch.empty((n, w * h * 3 // 2 * bps), dtype=torch.uint8, device=device)
    for t in range(n):
        rects = _rects(w, h, seed, t)
        off = 0
        for pl in range(3):
            ss = 1 if pl else 0
            pw, ph = w >> ss, h >> ss
            A1, A2, P1, P2 = (48, 32, 53, 41) if pl else (96, 64, 97, 61)
            base = 128 - (A1 + A2) // 2 if pl else 16
            x = torch.arange(pw, dtype=torch.int64, device=device)[None, :]
            y = torch.arange(ph, dtype=torch.int64, device=device)[:, None]
            v = base + (tri(x + ((2 * t) >> ss), P1) * A1) // 64 + (tri(y + (t >> ss), P2) * A2) // 64
            v = v.expand(ph, pw).clone()
            fx, fy = x << ss, y << ss
            for (px, py, rw, rh, rv) in rects:
                inside = (fx >= px) & (fx < px + rw) & (fy >= py) & (fy < py + rh)
                v = torch.where(inside, torch.full_like(v, rv[pl]), v)
            v = v << sh
            key = (_s64(seed ^ ((t << 40) & _M64))) ^ ((pl * 8192 + y) << 20) ^ x
            nz = mix(key)
            v = torch.clamp(v + ((nz & 15) - 8) * G, 0, maxv)
            nb = pw * ph * bps
            if bps == 1:
                out[t, off:off + nb] = v.to(torch.uint8).reshape(-1)
            else:
                out[t, off:off + nb] = torch.stack((v & 255, v >> 8), dim=-1).to(torch.uint8).reshape(-1)
            off += nb
    return out.reshape(-1)


# ------------------------------------------------------------------------------------------------ CPU baseline
def _psnr(sse, n, mx):
    return 99.0 if sse <= 0 else 10.0 * math.log10(mx * mx * n / sse)


def _port_worker(args):
    """the build's own C restatement (oracle/): same algorithm as the GPU path, one process per core"""
    w, h, bd, bs, seed, t0, cnt, keyint, mask = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import av1o
    cfg = av1o.default_config(w, h, bd, min_bs_log2=bs, max_bs_log2=bs, mode_mask=mask if mask else 0x7)
    frames = [av1o.synthclip_frame(w, h, bd, seed=seed, t=t0 + i) for i in range(cnt)]
    t = time.perf_counter()
    nbytes, sse, ref, prev = 0, [0, 0, 0], None, None
    for i, fr in enumerate(frames):
        key = keyint <= 1 or i % keyint == 0
        tu, rec, st = av1o.encode_frame(cfg, fr, with_seq_hdr=key, ref=None if key else ref, prev_src=None if key else prev)
        nbytes += len(tu)
        ref, prev = rec, fr
        for k in range(3):
            sse[k] += int(st.sse[k])
    return time.perf_counter() - t, nbytes, sse, cnt


def _libaom_worker(args):
    """libaom 3.13.2 through libavif's C API (tools/oracle_avif.py): I420 in, no RGB anywhere; all-key-frame = one still per
    frame, IPPP = one image sequence (libaom's own inter coding, key frame first only)"""
    w, h, bd, speed, seed, t0, cnt, keyint = args
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np
    import oracle_avif
    frames = [synthclip_frame(w, h, bd, seed, t0 + i) for i in range(cnt)]
    t = time.perf_counter()
    if keyint <= 1:
        avs = [oracle_avif.libaom_encode_yuv420([fr], bd, 30, speed, 1) for fr in frames]
    else:
        avs = [oracle_avif.libaom_encode_yuv420(frames, bd, 30, speed, 1, keyint=0)]
    dt = time.perf_counter() - t
    nbytes, sse = sum(len(a) for a in avs), [0, 0, 0]
    dec = [oracle_avif.decode_yuv(a)[0] for a in avs] if keyint <= 1 else oracle_avif.decode_sequence(avs[0], w, h)
    for fr, d in zip(frames, dec):
        for k in range(3):
            sse[k] += int(((d[k].astype(np.int64) - fr[k].astype(np.int64)) ** 2).sum())
    return dt, nbytes, sse, cnt


def _pool_run(fn, jobs, cores):
    import multiprocessing as mp
    t = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:   # nothing in this process has touched HIP yet (see main)
        res = pool.map(fn, jobs)
    wall = time.perf_counter() - t
    frames = sum(r[3] for r in res)
    nbytes = sum(r[1] for r in res)
    sse = [sum(r[2][k] for r in res) for k in range(3)]
    return wall, frames, nbytes, sse


def _reference_cli_leg(w, h, cores, keyint, frames):
    """SURVEY §8d choice 1: the reference's own path, if the box has it.  `av1an` with the reference's argv
    (build_av1an_command, av1an.rs:79-107; the one constant changed: --crf 30, --keyint as the config says, preset 8), or a
    bare SvtAv1EncApp.  Expected to be absent (the GPU box receives only this repository)."""
    av1an, svt = shutil.which("av1an"), shutil.which("SvtAv1EncApp")
    if not av1an and not svt:
        return None
    tmp = tempfile.mkdtemp(prefix="av1mi_cpu_")
    try:
        y4m = os.path.join(tmp, "clip.y4m")
        with open(y4m, "wb") as f:
            f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420p10 XYSCSS=420P10\n" % (w, h))
            for t in range(frames):
                f.write(b"FRAME\n" + b"".join(p.astype("<u2").tobytes() for p in synthclip_frame(w, h, 10, 1080, t)))
        out = os.path.join(tmp, "out.mkv" if av1an else "out.ivf")
        vp = "--crf 30 --preset 8 --keyint %d" % (1 if keyint <= 1 else keyint)
        workers = 8 if cores >= 32 else 4   # ConcurrencyPlan::derive, concurrency.rs:67-73
        if av1an:
            cmd = [av1an, "-i", y4m, "-o", out, "--encoder", "svt-av1", "--pix-format", "yuv420p10le", "--video-params", vp,
                   "--audio-params", "-c:a copy", "--workers", str(workers), "--temp", os.path.join(tmp, "chunks")]
        else:
            cmd = [svt, "-i", y4m, "-b", out] + vp.split()
        t = time.perf_counter()
        rc = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300).returncode
        wall = time.perf_counter() - t
        if rc != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
            return None
        return {"value": round(frames / wall, 3), "unit": "frames/s", "cores": cores, "kind": "reference",
                "encoder": "av1an + SVT-AV1 (reference argv, --crf 30 --preset 8)" if av1an else "SvtAv1EncApp --crf 30 --preset 8",
                "sample": "%d frames of the 1080p 10-bit synthclip chunk, whole encode incl. process start" % frames,
                "bytes_per_frame": round(os.path.getsize(out) / frames, 1), "psnr_db": None}
    except (OSError, subprocess.SubprocessError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(w, h, bd, bs, keyint, mask=0x7, budget_frames_per_core=4):
    """Reported-only CPU leg on a bounded sample of the headline workload (must run before the process initialises HIP:
    the worker pool forks).  Order of SURVEY §8d / BASELINE.md §3; the legs that ran besides the chosen one are kept
    under "others"."""
    # every core this process may run on (north_star: "timed on the box's own host cores, core count stated"); 64 bounds the worker
    # pool on very wide hosts (the GPU box's process guard), and the line says so when it bites
    avail = len(os.sched_getaffinity(0))
    quota = None   # the container's CPU share (cgroup v2 cpu.max / v1 cfs quota): what this process can actually run on at once
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per) + 0.5))
    except (OSError, ValueError):
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, int(q / per + 0.5))
        except (OSError, ValueError):
            pass
    cores = min(avail, quota if quota else avail, 64)
    legs = []
    ref = _reference_cli_leg(w, h, cores, keyint, 16)
    if ref:
        legs.append(ref)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import oracle_avif
        have_aom = oracle_avif.have_libavif()
    except Exception:
        have_aom = False
    if have_aom:
        for speed, per_core in ((8, budget_frames_per_core), (6, 1)):
            try:
                jobs = [(w, h, 8, speed, 1080, c * per_core, per_core, keyint) for c in range(cores)]
                wall, frames, nbytes, sse = _pool_run(_libaom_worker, jobs, cores)
                legs.append({"value": round(frames / wall, 3), "unit": "frames/s", "cores": cores, "kind": "libaom",
                             "encoder": "libaom 3.13.2 via libavif C API, end-usage=q cq-level=30, cpu-used %d, I420 8-bit (this libaom build has no "
                                        "high bit depth), one single-threaded encoder per core" % speed,
                             "sample": "%d frames of the same %dx%d synthclip chunk at 8 bit, %d per core (%s)" % (
                                 frames, w, h, per_core, "one still per frame" if keyint <= 1 else "one IPPP sequence per core"),
                             "bytes_per_frame": round(nbytes / frames, 1),
                             "psnr_db": [round(_psnr(sse[k], frames * w * h / (4 if k else 1), 255.0), 2) for k in range(3)]})
            except Exception as e:   # reported-only: a failing stand-in must not take the benchmark down
                legs.append({"kind": "libaom", "error": "%s: %s" % (type(e).__name__, e)})
    try:
        jobs = [(w, h, bd, bs, 1080, c * budget_frames_per_core, budget_frames_per_core, keyint, mask) for c in range(cores)]
        wall, frames, nbytes, sse = _pool_run(_port_worker, jobs, cores)
        legs.append({"value": round(frames / wall, 3), "unit": "frames/s", "cores": cores, "kind": "port",
                     "encoder": "oracle/: this build's algorithm restated in plain C (bit-identical output to the GPU path), one process per core",
                     "sample": "%d frames of the same %dx%d %d-bit synthclip chunk, %d per core" % (frames, w, h, bd, budget_frames_per_core),
                     "bytes_per_frame": round(nbytes / frames, 1),
                     "psnr_db": [round(_psnr(sse[k], frames * w * h / (4 if k else 1), float((1 << bd) - 1)), 2) for k in range(3)]})
    except Exception as e:
        legs.append({"kind": "port", "error": "%s: %s" % (type(e).__name__, e)})
    good = [l for l in legs if "value" in l]
    if not good:
        return {"value": None, "unit": "frames/s", "cores": cores, "kind": "none", "sample": "no CPU encoder could run", "others": legs}
    best = dict(good[0])
    best["host"] = {"nproc": os.cpu_count(), "affinity_cores": avail, "cgroup_cpu_quota": quota, "cores_used": cores}
    best["others"] = [l for l in legs if l is not good[0]]
    best["vmaf"] = "unavailable offline (no libvmaf)"
    return best


# ------------------------------------------------------------------------------------------------ GPU workloads
def workload_string(av1mi, a):
    mask = a["mode_mask"] if a["mode_mask"] else 0x7
    names = ["DC", "V", "H", "D45", "D135", "D113", "D157", "D203", "D67", "SMOOTH", "SMOOTH_V", "SMOOTH_H", "PAETH"]
    cand = "all 13 intra candidates" if mask == 0x1FFF else "intra candidates {%s} (mode mask 0x%X)" % (", ".join(n for i, n in enumerate(names) if (mask >> i) & 1), mask)
    kind = "all-key-frame" if a["keyint"] <= 1 else "IPPP (keyint %d, 1 reference, +-%d full search%s)" % (
        a["keyint"], a["me_range"], " + quarter-sample SATD refinement, EIGHTTAP" if a["subpel"] else "")
    s = "%dx%d %d-frame %s synthclip v1 chunk per GPU, %d-bit 4:2:0, CQ=%d (base_q_idx %d), %dx%d blocks, %s, 64x64 tiles, %s CDFs, CDEF on" % (
        a["width"], a["height"], a["frames"], kind, a["bit_depth"], a["cq"], av1mi.cq_to_qindex(a["cq"]), 1 << a["block_log2"], 1 << a["block_log2"],
        cand, "static" if a["static_cdf"] else "adaptive")
    s += (", deblocking on" if a["deblock"] else "") + (", quantiser matrices 1..15" if a["qm"] else "")
    s += ", loop restoration (Wiener + self-guided) on" if a["sgr"] else (", loop restoration (Wiener) on" if a["lr"] else "")
    s += (", film-grain table %d" % a["film_grain"]) if a["film_grain"] else ""
    if a.get("partition_min"):
        s = s.replace("%dx%d blocks" % (1 << a["block_log2"], 1 << a["block_log2"]), "content-driven partition with %dx%d .. %dx%d leaves" % (
            1 << a["partition_min"], 1 << a["partition_min"], 1 << a["block_log2"], 1 << a["block_log2"]))
    if a.get("presearch"):
        s = s.replace("full search", "full search around the centre of a quarter-resolution +-64 pre-search")
    if a.get("hdr"):
        s += ", colour description BT.2020 / PQ / BT.2020 NCL (HDR10) in the sequence header"
    if a["width"] > 4096 or a["height"] > 4096:
        s = s.replace("64x64 tiles", "tiles of 2x2 superblocks")
    if a.get("job_chunks"):
        s = s.replace("chunk per GPU", "chunks, a job of %d placed on the GPUs by av1mi_chunk_owner" % a["job_chunks"])
    if a["chunks_per_gpu"] > 1:
        s += ", %d chunks in flight per GPU on their own contexts" % a["chunks_per_gpu"]
    return s


def algorithmic_bytes(a, rep):
    """Algorithmic HBM bytes per step and per stage: N = samples per frame, b = bytes per sample"""
    w, h, n = a["width"], a["height"], a["frames"]
    b = 2 if a["bit_depth"] > 8 else 1
    N, L = w * h * 3 // 2, w * h
    inter = a["keyint"] > 1 and n > 1
    n_inter = (n - (n + a["keyint"] - 1) // a["keyint"]) if inter else 0
    lr = a["sgr"] or a["lr"]
    stage = {"symbolize": n * N * 2 + 4 * int(rep.n_symbols),        # levels read + 32-bit symbol entries written
             "rangecode": 4 * int(rep.n_symbols) + int(rep.bytes)}   # symbol entries read + bitstream written
    if not inter:
        stage["recon"] = n * N * (2 * b + 2)                          # source read + reconstruction written + int16 levels written
        stage["cdef"] = n * N * 2 * b                                 # reconstruction read + filtered frame written (timed with the SSE pass)
        if lr:
            stage["recon"] += n * N * 2 * b + n * (3 * L * b + N * b)   # CDEF and restoration precede entropy coding then
    else:
        # the frame-by-frame chain (timed as "recon"): per frame source + reconstruction + levels, CDEF in and out, and per
        # inter frame the reference read by motion compensation; restoration as above
        stage["recon"] = n * N * (2 * b + 2) + n * N * 2 * b + n_inter * N * b + (n * (3 * L * b + N * b) if lr else 0)
        stage["cdef"] = n * N * 2 * b                                 # the SSE pass: two frames read
    # whole step, SURVEY §8d: intra N(4b+2); inter N(7b+2) + 2Lb (search: source + previous source luma)
    step = n * N * (4 * b + 2) if not inter else (n - n_inter) * N * (4 * b + 2) + n_inter * (N * (7 * b + 2) + 2 * L * b)
    return stage, step


def run_workload(av1mi, torch, a, dev, local_rank, rank, world, steps, warmup, barrier, max_over_ranks):
    """`steps` timed steps of workload `a` on this rank; returns (elapsed seconds (max over ranks), stage ms per step, last report)"""
    w, h, bd, n = a["width"], a["height"], a["bit_depth"], a["frames"]
    # the job's scene-chunks over the ranks by the product's own placement rule (av1mi_chunk_owner): by default a job of `world` chunks, one
    # per GPU (weak scaling); job_chunks = J: a job of J chunks whatever the number of GPUs (strong scaling; BASELINE config 4: J = 8)
    job = int(a.get("job_chunks", 0))
    mine = av1mi.chunks_of_rank(job if job else world, world, rank)
    assert job or len(mine) == 1, mine
    # counter-collection runs (rocprofv3 --pmc, AV1MI_BENCH_CLIP_ON_CPU): torch's own GPU kernels crash under the profiler's counter service
    # on this image, so the same generator runs on the CPU device and the clip is copied over - no kernel but the library's is launched
    gen_dev = "cpu" if os.environ.get("AV1MI_BENCH_CLIP_ON_CPU") else dev
    clips = [make_clip_torch(w, h, bd, n, a["seed"] + c, gen_dev).to(dev) for c in mine]   # HBM-resident input
    d_frames = clips[0] if clips else None
    torch.cuda.synchronize(dev)
    params = av1mi.default_params(w, h, bd, block_log2=a["block_log2"], cdf_update=0 if a["static_cdf"] else 1, keyint=a["keyint"],
                                  me_range=a["me_range"], cq_level=a["cq"], film_grain=a["film_grain"],
                                  deblock=1 if a["deblock"] else 0, enable_lr=2 if a["sgr"] else (1 if a["lr"] else 0))
    params.subpel = 1 if a["subpel"] else 0
    if a["qm"]:
        params.enable_qm, params.qm_min, params.qm_max = 1, 1, 15
    params.intra_mode_mask = a["mode_mask"]
    if a.get("partition_min"):
        params.partition_search, params.min_block_log2 = 1, a["partition_min"]
    if a.get("presearch"):
        params.me_presearch = 1
    if a.get("hdr"):
        params.color_primaries, params.transfer_characteristics, params.matrix_coefficients = 9, 16, 9
    C_ = max(1, a["chunks_per_gpu"])
    ctxs = [av1mi.Context(local_rank) for _ in range(C_)]

    def merge(res):
        rep = max(res, key=lambda r: r.ms_total)   # stage times of the slowest part
        rep.n_symbols = sum(r.n_symbols for r in res)
        rep.max_tile_symbols = max(r.max_tile_symbols for r in res)
        rep.bytes = sum(r.bytes for r in res)
        return rep

    def step():
        import threading
        if job:   # this rank's chunks of the job, C_ of them in flight
            if not clips:
                return None
            res = [None] * len(clips)

            def work_job(i):
                for k in range(i, len(clips), C_):
                    res[k] = ctxs[i].encode_chunk(params, clips[k].data_ptr(), n, on_device=True, copy_out=False)[2]
            th = [threading.Thread(target=work_job, args=(i,)) for i in range(min(C_, len(clips)))]
            for t in th:
                t.start()
            for t in th:
                t.join()
            return merge(res)
        if C_ == 1:
            return ctxs[0].encode_chunk(params, d_frames.data_ptr(), n, on_device=True, copy_out=False)[2]
        res = [None] * C_

        def work(i):   # every context encodes the whole chunk (its own copy of the work)
            res[i] = ctxs[i].encode_chunk(params, d_frames.data_ptr(), n, on_device=True, copy_out=False)[2]
        th = [threading.Thread(target=work, args=(i,)) for i in range(C_)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return merge(res)
    try:
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        stage = {"recon": 0.0, "cdef": 0.0, "entropy": 0.0, "symbolize": 0.0, "pack": 0.0, "d2h": 0.0}
        rep = None
        for _ in range(steps):
            rep = step()
            if rep is None:   # a rank without a chunk of the job (more GPUs than chunks)
                continue
            stage["recon"] += rep.ms_recon; stage["cdef"] += rep.ms_cdef; stage["entropy"] += rep.ms_entropy
            stage["pack"] += rep.ms_pack; stage["d2h"] += rep.ms_d2h; stage["symbolize"] += rep.ms_symbolize
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
    finally:
        for c_ in ctxs:
            c_.close()
        del d_frames, clips
    for s in stage:
        stage[s] /= steps
    stage["rangecode"] = stage["entropy"] - stage["symbolize"]
    return elapsed, stage, rep


def result_of(av1mi, a, world, steps, elapsed, stage, rep, traffic_db):
    C_, n = max(1, a["chunks_per_gpu"]), a["frames"]
    job = int(a.get("job_chunks", 0))
    alg, step_bytes = algorithmic_bytes(a, rep)
    # frames and algorithmic bytes of one step over ALL ranks: a job of J chunks, or one chunk (x C_ copies in flight) per rank
    frames_per_step = job * n if job else world * C_ * n
    step_bytes_all = step_bytes * (job if job else world * C_)
    dom = max(alg, key=lambda s: stage[s])
    achieved = alg[dom] / (stage[dom] * 1e-3) / 1e9
    peak = 8000.0
    step_gbs = step_bytes_all / (elapsed / steps) / 1e9 / world   # per GPU
    traffic = traffic_raw = valu_busy = None
    ent = (traffic_db or {}).get(a["name"], {}).get(dom)
    if ent:
        # separate FETCH_SIZE / WRITE_SIZE passes of this same workload (profiles/); gfx950 correction: FETCH_SIZE counts 64 B per
        # 128-B request, so the fetch side is doubled (an upper estimate for narrow loads)
        traffic_raw = (ent["FETCH_SIZE"] + ent["WRITE_SIZE"]) * 1024
        traffic = (2 * ent["FETCH_SIZE"] + ent["WRITE_SIZE"]) * 1024
        if ent.get("SQ_INSTS_VALU"):
            # what actually bounds the kernel: its vector-instruction issue.  SQ_INSTS_VALU wave-instructions (the committed SQ pass of this
            # workload) x 4 cycles each on one of 1024 SIMDs, over the kernel's LIVE duration at the 2.4 GHz peak clock (the clock under load is
            # lower, so this is a lower estimate of the busy fraction)
            valu_busy = round(ent["SQ_INSTS_VALU"] * 4.0 / (1024 * stage[dom] * 1e-3 * 2.4e9), 4)
    return {
        "workload": workload_string(av1mi, a),
        "fps": round(frames_per_step * steps / elapsed, 2), "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps,
        "frames_per_step": frames_per_step,
        # `frac` is the contract's figure: the dominant kernel's algorithmic bytes per launch over its duration against the HBM peak.  The
        # kernel is not HBM bound: `bound` names what is (vector-instruction issue, `valu_busy` of the SIMDs' issue slots)
        "roofline": {"bound": "valu" if valu_busy is not None and valu_busy > 0.5 else "hbm", "valu_busy": valu_busy, "kernel": dom,
                     "achieved": round(achieved, 2), "peak": peak, "unit": "GB/s",
                     "frac": round(achieved / peak, 5), "traffic": traffic, "traffic_raw_counters": traffic_raw,
                     "algorithmic_bytes_per_launch": alg[dom], "kernel_ms": round(stage[dom], 3),
                     "step_algorithmic_bytes": step_bytes_all, "step_achieved": round(step_gbs, 2), "step_frac": round(step_gbs / peak, 5)},
        "stage_ms": {s: round(v, 3) for s, v in stage.items()},
        "bytes_per_frame": round(int(rep.bytes) / ((len(av1mi.chunks_of_rank(job, world, 0)) if job else C_) * n), 1),
        "psnr_db": [round(rep.psnr[i], 2) for i in range(3)],
        "symbols_per_frame": int(rep.n_symbols // ((len(av1mi.chunks_of_rank(job, world, 0)) if job else C_) * n)), "max_tile_symbols": int(rep.max_tile_symbols),
    }


def run_from_file(av1mi, torch, a, dev, n_frames, repeats=3):
    """End to end through the drop-in itself (av1mi_encode_file = the reference's run_av1an): a Y4M of the clip on tmpfs ->
    reader threads -> pinned host chunks -> H2D -> encode on `workers` contexts -> bitstream D2H -> Matroska on tmpfs.
    PCIe- and file-inclusive: reported beside the HBM-resident numbers, never as the headline value."""
    w, h, bd = a["width"], a["height"], a["bit_depth"]
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    tmp = tempfile.mkdtemp(prefix="av1mi_e2e_", dir=shm)
    try:
        y4m, out = os.path.join(tmp, "clip.y4m"), os.path.join(tmp, "clip.mkv")
        fb = w * h * 3 // 2 * (2 if bd > 8 else 1)
        with open(y4m, "wb") as f:
            f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C%s\n" % (w, h, b"420p10 XYSCSS=420P10" if bd > 8 else b"420jpeg"))
            for t0 in range(0, n_frames, 30):   # scenes of 30 frames (seed + scene), synthesised on the GPU, written frame by frame
                n = min(30, n_frames - t0)
                clip = make_clip_torch(w, h, bd, n, a["seed"] + t0 // 30, dev).cpu().numpy()
                for t in range(n):
                    f.write(b"FRAME\n")
                    f.write(clip[t * fb:(t + 1) * fb].tobytes())
        enc = dict(keyint=a["keyint"], block_log2=a["block_log2"], me_range=a["me_range"], intra_mode_mask=a["mode_mask"])
        best, rep = None, None
        for _ in range(repeats):
            t = time.perf_counter()
            rep = av1mi.run_mi355x(av1mi.EncodeParams(y4m, out, tmp, None, cq_level=a["cq"], chunk_frames=60, **enc))
            dt = time.perf_counter() - t
            best = dt if best is None or dt < best else best
        workers = len(av1mi.plan_workers(0, 0, 1))
        return {"workload": "av1mi_encode_file: %dx%d %d-bit Y4M on tmpfs (%d frames, %.2f GB) -> .mkv, chunks of 60 frames, %d contexts on one GPU, %s, "
                            "pinned host staging + parallel reader" % (w, h, bd, n_frames, os.path.getsize(y4m) / 1e9, workers,
                                                                     "all-key-frame" if a["keyint"] <= 1 else "IPPP keyint %d" % a["keyint"]),
                "fps": round(n_frames / best, 2), "ms_total": round(best * 1e3, 2), "frames": n_frames, "chunks": int(rep.chunks),
                "input_GBps": round(os.path.getsize(y4m) / best / 1e9, 2), "bytes_per_frame": round(os.path.getsize(out) / n_frames, 1),
                "psnr_db": [round(rep.psnr[i], 2) for i in range(3)], "note": "best of %d runs, file and PCIe inclusive" % repeats}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def metric_label(a):
    res = "1080p" if (a["width"], a["height"]) == (1920, 1080) else ("4K" if (a["width"], a["height"]) == (3840, 2160) else "%dx%d" % (a["width"], a["height"]))
    return "encoded frames/sec at CQ=%d (%s %s)" % (a["cq"], res, "intra-only" if a["keyint"] <= 1 else "IPPP")


def extra_configs(base):
    """the other BASELINE configurations one GPU runs (BASELINE.json configs[1..3] + the reference's production string)"""
    def mk(name, **kw):
        d = dict(base)
        d.update(kw)
        d["name"] = name
        return d
    # the inter configurations keep the library's default intra candidate set {DC, V, H} for the intra / inter decision of their P-frame
    # blocks (a workload string says which set ran)
    prod = dict(keyint=240, cq=8, qm=True, film_grain=20, subpel=True, deblock=True, sgr=True, mode_mask=0x7)
    # one-chunk configurations first, the ones with four chunks in flight last: a context created after many others' streams exist
    # can get hardware queues that serialise its chain behind its own search stream - latency figures are taken on
    # the first contexts of the process
    return [
        mk("cfg2_1080p_intra_dcvh", mode_mask=0x7),   # the encoder's cheapest candidate set (rounds 1-2's headline)
        mk("cfg2_1080p_intra_8bit", bit_depth=8),     # SURVEY 8d config 2 "run at 8-bit (dav1d-checkable) and 10-bit"
        mk("cfg2_1080p_intra_64x64", block_log2=6),
        mk("cfg2_1080p_intra_partition_8_64", block_log2=6, partition_min=3),   # content-driven partition: a third fewer bytes than 32x32 at equal PSNR
        mk("cfg3_1080p_ippp", keyint=240, mode_mask=0x7),
        mk("cfg3_1080p_ippp_all13", keyint=240),
        mk("cfg3_1080p_ippp_64x64", keyint=240, mode_mask=0x7, block_log2=6),   # 64x64 leaves: 40 % fewer bytes at -0.2 dB, and faster
        mk("cfg3_1080p_ippp_presearch_partition", keyint=240, mode_mask=0x7, presearch=True, block_log2=6, partition_min=3),
        mk("cfg4_4k_ippp_chunk", width=3840, height=2160, frames=30, keyint=240, seed=2160, mode_mask=0x7),
        # BASELINE config 5's per-GPU unit: one 8K 10-bit HDR scene-chunk of 16 frames, film-grain table in every frame header, tiles of 2x2
        # superblocks (AV1 allows at most 64 x 64 tiles)
        mk("cfg5_8k_ippp_chunk", width=7680, height=4320, frames=16, keyint=240, seed=4320, film_grain=20, hdr=True, mode_mask=0x7),
        # iso-quality point against libaom (tools/rd_vs_libaom.py): libaom cq-level 30 reaches 36.0 dB PSNR-Y on this clip at 8 bit; this
        # encoder reaches it at CQ 19 (base_q_idx 76)
        mk("cfg2_1080p_intra_8bit_cq19_isoquality", bit_depth=8, cq=19),
        mk("production_1080p", **prod),
        mk("cfg2_1080p_intra_x4", chunks_per_gpu=4),   # the product's default: AV1MI_DEFAULT_WORKERS_PER_GPU chunks in flight per GPU
        mk("cfg3_1080p_ippp_x4", keyint=240, chunks_per_gpu=4, mode_mask=0x7),
        mk("production_1080p_x4", chunks_per_gpu=4, **prod),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--keyint", type=int, default=1, help="1 = all key frames (the headline config); N > 1 = IPPP, key frame every N frames")
    ap.add_argument("--me-range", type=int, default=8)
    ap.add_argument("--cq", type=int, default=30, help="CQ level (the headline metric is quoted at 30; the reference's production string uses 8)")
    ap.add_argument("--film-grain", type=int, default=0)
    ap.add_argument("--sgr", action="store_true", help="loop restoration with switchable units: off / Wiener / self-guided (enable_lr = 2)")
    ap.add_argument("--subpel", action="store_true", help="inter frames: quarter-sample motion vectors + EIGHTTAP interpolation (default: whole-sample vectors)")
    ap.add_argument("--qm", action="store_true", help="quantiser matrices on, --qm-min 1 --qm-max 15 as in the reference's SVT_PARAMS (av1an.rs:14)")
    ap.add_argument("--deblock", action="store_true", help="deblocking filter on (default off: loop_filter_level 0, SURVEY.md §8a a19)")
    ap.add_argument("--lr", action="store_true", help="loop restoration on (default off)")
    ap.add_argument("--chunks-per-gpu", type=int, default=1,
                    help="independent full chunks encoded concurrently on their own contexts per GPU (the reference's `--workers`); "
                         "a step then processes chunks-per-gpu x frames frames")
    ap.add_argument("--bit-depth", type=int, default=10)
    ap.add_argument("--block-log2", type=int, default=5)
    ap.add_argument("--partition-min", type=int, default=0, help="content-driven partition with leaves from 2^k (3..) up to --block-log2")
    ap.add_argument("--static-cdf", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode-mask", type=lambda s: int(s, 0), default=0x1FFF,
                    help="intra mode candidate mask.  Default 0x1FFF: all 13 luma intra modes - BASELINE config 2 names \"directional intra pred\", so the "
                         "headline evaluates the directional, smooth and Paeth predictors for every block; 0x7 = {DC, V, H}, the library's own default")
    ap.add_argument("--configs", choices=["auto", "all", "none"], default="auto",
                    help="the `configs` object with the other BASELINE configurations: auto = at N=1 when the workload flags are the defaults")
    ap.add_argument("--workload", choices=["cfg2", "cfg4"], default="cfg2",
                    help="cfg2 (default, the driver's line): every rank encodes its own 1080p chunk (weak scaling).  cfg4: BASELINE config 4 as "
                         "worded - a job of 8 scene-chunks of 4K 10-bit x 30 frames (IPPP), placed on the ranks by av1mi_chunk_owner (strong scaling)")
    args = ap.parse_args()

    # `--gpus N` without a launcher: start the N ranks ourselves, as children, BEFORE anything in this process touches the GPU (a process
    # that has initialised HIP must not be replaced or forked on this pool), and leave with their exit code.  Under torchrun /
    # torch.distributed.run (the driver's way) WORLD_SIZE is set and this is one of the ranks.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    base = dict(name="cfg2_1080p_intra", width=args.width, height=args.height, frames=args.frames, keyint=args.keyint, me_range=args.me_range,
                cq=args.cq, film_grain=args.film_grain, sgr=args.sgr, subpel=args.subpel, qm=args.qm, deblock=args.deblock, lr=args.lr,
                chunks_per_gpu=args.chunks_per_gpu, bit_depth=args.bit_depth, block_log2=args.block_log2, static_cdf=args.static_cdf,
                mode_mask=args.mode_mask, seed=1080, partition_min=args.partition_min)
    if args.workload == "cfg4":   # BASELINE config 4: "4K30 10-bit, 8 scene-chunks sharded across 8xMI355X (one chunk/GPU, independent HIP streams)"
        base.update(name="cfg4_4k_job8", width=3840, height=2160, frames=30, keyint=240, seed=2160, bit_depth=10, job_chunks=8, mode_mask=0x7)
    default_flags = args.workload == "cfg2" and all(getattr(args, k) == ap.get_default(k) for k in (
        "width", "height", "frames", "keyint", "me_range", "cq", "film_grain", "sgr", "subpel", "qm", "deblock", "lr", "chunks_per_gpu",
        "bit_depth", "block_log2", "static_cdf", "mode_mask", "partition_min"))
    if not default_flags:
        base["name"] = "custom"
    with_configs = args.configs == "all" or (args.configs == "auto" and world == 1 and default_flags)

    # ---- CPU leg first: its worker pool forks, which must happen before this process (or torch) initialises HIP
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(base["width"], base["height"], base["bit_depth"], base["block_log2"], base["keyint"], base["mode_mask"])

    import torch
    dist = None
    # rehearsal on a one-GPU box (the 8-GPU runs are the driver's): AV1MI_BENCH_BACKEND=gloo AV1MI_BENCH_ONE_DEVICE=1 runs the
    # same multi-rank control flow with every rank on GPU 0 and the timing collectives over gloo
    backend = os.environ.get("AV1MI_BENCH_BACKEND", "nccl")
    if os.environ.get("AV1MI_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    import av1mi
    dev = torch.device("cuda", local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(elapsed):
        if dist is None:
            return elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    traffic_db = None
    try:
        traffic_db = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["workloads"]
    except (OSError, KeyError, ValueError):
        pass

    elapsed, stage, rep = run_workload(av1mi, torch, base, dev, local_rank, rank, world, args.steps, args.warmup, barrier, max_over_ranks)
    head = result_of(av1mi, base, world, args.steps, elapsed, stage, rep, traffic_db)
    configs = {}
    if with_configs:
        for a in extra_configs(base):
            k, wu = max(2, min(args.steps, 6)), max(1, min(args.warmup, 2))
            try:
                e2, s2, r2 = run_workload(av1mi, torch, a, dev, local_rank, rank, world, k, wu, barrier, max_over_ranks)
                configs[a["name"]] = result_of(av1mi, a, world, k, e2, s2, r2, traffic_db)
            except Exception as e:   # a secondary configuration must not take the headline down; it is reported as failed
                configs[a["name"]] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1:
            for name, kw in (("cfg2_e2e_y4m_to_mkv", dict()), ("cfg3_e2e_y4m_to_mkv_ippp", dict(keyint=240, mode_mask=0x7))):
                try:
                    d = dict(base)
                    d.update(kw)
                    configs[name] = run_from_file(av1mi, torch, d, dev, 480)
                except Exception as e:
                    configs[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        out = {
            "metric": metric_label(base), "value": head["fps"], "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "strong" if base.get("job_chunks") else "weak", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": head["workload"], "frames_per_chunk": base["frames"], "chunks_per_gpu": max(1, base["chunks_per_gpu"]),
                       "parallelism": ("job of %d chunks over %d GPUs (av1mi_chunk_owner)" % (base["job_chunks"], world)) if base.get("job_chunks")
                       else "chunk-per-gpu x%d" % world},
            "roofline": head["roofline"], "stage_ms": head["stage_ms"], "bytes_per_frame": head["bytes_per_frame"], "psnr_db": head["psnr_db"],
            "symbols_per_frame": head["symbols_per_frame"], "max_tile_symbols": head["max_tile_symbols"],
        }
        if configs:
            out["configs"] = configs
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
