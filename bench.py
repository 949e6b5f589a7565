#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X AV1 chunk-encode path.

Metric (BASELINE.json): encoded frames/s at CQ=30.  Workload at N=1 = BASELINE.json configs[1]:
1080p, 60-frame all-key-frame (intra-only) `synthclip v1` chunk, 10-bit
(the reference's pixel format, av1an.rs:90) unless --bit-depth 8.  One "step" = one complete
encode of one chunk per GPU: source frames already resident in HBM -> reconstruction, CDEF,
entropy coding, bitstream packing on the GPU -> complete OBU bitstream on the host.

Multi-GPU (torchrun, one rank per GPU): scene-chunks are independent, every
rank encodes its own chunk (seed 1080 + rank), no data-path collective; weak scaling.  The only
collectives are the timing barrier and the MAX over ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
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
    rects = []
    for k in range(6):
        hh = _splitmix64((seed * 977 + k) & _M64)
        vx, vy = (3 if hh & 1 else -3), (2 if hh & 2 else -2)
        rw, rh = 32 + (hh >> 8) % (w // 4 + 1), 32 + (hh >> 24) % (h // 4 + 1)
        rx, ry = (hh >> 40) % w + vx * t, (hh >> 52) % h + vy * t
        h2 = _splitmix64(hh)
        rects.append((rx % w, ry % h, rw, rh, [(h2 >> 3) & 255, (h2 >> 13) & 255, (h2 >> 23) & 255]))
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
    """synthclip v1 frames as one I420 byte string"""
    import numpy as np
    dt = np.uint8 if bd == 8 else np.dtype("<u2")
    out = []
    for t in range(n):
        fr = synthclip_frame(w, h, bd, seed, t)
        out.append(b"".join(p.astype(dt).tobytes() for p in fr))
    return b"".join(out)


def _cpu_worker(args):
    w, h, bd, bs, seed, t0, cnt = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import av1o
    cfg = av1o.default_config(w, h, bd, min_bs_log2=bs, max_bs_log2=bs)
    frames = [av1o.synthclip_frame(w, h, bd, seed=seed, t=t0 + i) for i in range(cnt)]
    t = time.perf_counter()
    nbytes = 0
    for fr in frames:
        tu, _, _ = av1o.encode_frame(cfg, fr)
        nbytes += len(tu)
    return time.perf_counter() - t, nbytes


def cpu_baseline(w, h, bd, bs, budget_s=20.0):
    """The CPU oracle (kind 'port': same algorithm, plain C, one process per host core) on a
    bounded sample of the same workload."""
    import multiprocessing as mp
    cores = min(len(os.sched_getaffinity(0)), 16)
    # one calibration frame on one core
    t1, _ = _cpu_worker((w, h, bd, bs, 1080, 0, 1))
    per_core = max(1, min(4, int(budget_s / max(t1, 1e-3))))
    jobs = [(w, h, bd, bs, 1080, c * per_core, per_core) for c in range(cores)]
    t = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t
    frames = cores * per_core
    return {"value": round(frames / wall, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frames of the same 1080p synthclip chunk, %d per core, oracle (plain C restatement), single-frame %.2f s" % (
                frames, per_core, t1)}


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
    ap.add_argument("--static-cdf", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode-mask", type=int, default=0, help="experiment: intra mode candidate mask (0 = default {DC, V, H})")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    w, h, bd, n = args.width, args.height, args.bit_depth, args.frames
    clip = make_clip(w, h, bd, n, 1080 + rank)
    d_frames = torch.frombuffer(bytearray(clip), dtype=torch.uint8).to(dev)  # HBM-resident input
    torch.cuda.synchronize(dev)
    params = av1mi.default_params(w, h, bd, block_log2=args.block_log2, cdf_update=0 if args.static_cdf else 1, keyint=args.keyint,
                                  me_range=args.me_range, cq_level=args.cq, film_grain=args.film_grain,
                                  deblock=1 if args.deblock else 0, enable_lr=2 if args.sgr else (1 if args.lr else 0))
    params.subpel = 1 if args.subpel else 0
    if args.qm:
        params.enable_qm, params.qm_min, params.qm_max = 1, 1, 15
    params.intra_mode_mask = args.mode_mask
    W_ = max(1, args.chunks_per_gpu)
    C_ = W_
    ctxs = [av1mi.Context(local_rank) for _ in range(C_)]
    ctx = ctxs[0]
    fbytes = w * h * 3 // 2 * (2 if bd > 8 else 1)
    parts = [(0, n)] * W_   # every context encodes the whole chunk (its own copy of the work)

    def step():
        if C_ == 1:
            return ctx.encode_chunk(params, d_frames.data_ptr(), n, on_device=True, copy_out=False)
        import threading
        res = [None] * C_

        def work(i):
            a, b = parts[i]
            res[i] = ctxs[i].encode_chunk(params, d_frames.data_ptr() + a * fbytes, b - a, on_device=True, copy_out=False)
        th = [threading.Thread(target=work, args=(i,)) for i in range(C_)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        # merge: bitstream in frame order, the report of the slowest part for the stage times
        rep = max((r[2] for r in res), key=lambda r: r.ms_total)
        rep.n_symbols = sum(r[2].n_symbols for r in res)
        rep.max_tile_symbols = max(r[2].max_tile_symbols for r in res)
        rep.bytes = sum(r[2].bytes for r in res)
        return None, [x for r in res for x in r[1]], rep, None

    for _ in range(args.warmup):
        step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    stage = {"recon": 0.0, "cdef": 0.0, "entropy": 0.0, "symbolize": 0.0, "pack": 0.0, "d2h": 0.0}
    last = None
    for _ in range(args.steps):
        data, sizes, rep, _ = step()
        last = (data, rep)
        stage["recon"] += rep.ms_recon; stage["cdef"] += rep.ms_cdef; stage["entropy"] += rep.ms_entropy
        stage["pack"] += rep.ms_pack; stage["d2h"] += rep.ms_d2h; stage["symbolize"] += rep.ms_symbolize
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        data, rep = last
        k = args.steps
        bps = 2 if bd > 8 else 1
        N = w * h * 3 // 2  # samples per frame
        for s in stage:
            stage[s] /= k
        # algorithmic HBM bytes per launch of each kernel
        stage["rangecode"] = stage["entropy"] - stage["symbolize"]
        alg = {"recon": n * N * (2 * bps + 2),           # source read + reconstruction write + int16 levels write
               "cdef": n * N * 2 * bps,                  # reconstruction read + filtered write (timed with the SSE kernel)
               "symbolize": n * N * 2 + 4 * int(rep.n_symbols),   # levels read + 32-bit symbol entries write
               "rangecode": 4 * int(rep.n_symbols) + int(rep.bytes)}   # symbol entries read + bitstream write
        dom = max(alg, key=lambda s: stage[s])
        achieved = alg[dom] / (stage[dom] * 1e-3) / 1e9
        peak = 8000.0
        # HBM traffic of the dominant kernel from the committed PMC passes (separate FETCH_SIZE / WRITE_SIZE
        # runs of this same workload, profiles/); gfx950 correction: FETCH_SIZE counts 64 B per 128-B request.
        traffic = traffic_raw = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_h_pmc_traffic.json")))
            if (w, h, n, bd, args.block_log2, args.static_cdf, args.keyint, args.cq) == (1920, 1080, 60, 10, 5, False, 1, 30) and dom in pm["kernels"]:
                kk = pm["kernels"][dom]
                traffic_raw = (kk["FETCH_SIZE"] + kk["WRITE_SIZE"]) * 1024
                traffic = (2 * kk["FETCH_SIZE"] + kk["WRITE_SIZE"]) * 1024
        except OSError:
            pass
        out = {
            "metric": "encoded frames/sec at CQ=30 (1080p intra-only)", "value": round(world * W_ * n * k / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": k, "warmup": args.warmup, "ms_per_step": round(elapsed / k * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": "%dx%d %d-frame %s synthclip v1 chunk per GPU, %d-bit 4:2:0, CQ=%d (base_q_idx %d), "
                                   "%dx%d blocks, 64x64 tiles, %s CDFs, CDEF on" % (
                                       w, h, n, "all-key-frame" if args.keyint <= 1 else "IPPP (keyint %d, 1 reference, +-%d full search%s)" % (args.keyint, args.me_range, " + quarter-sample refinement, EIGHTTAP" if args.subpel else ""),
                                       bd, args.cq, av1mi.cq_to_qindex(args.cq), 1 << args.block_log2, 1 << args.block_log2,
                                       "static" if args.static_cdf else "adaptive") + (", deblocking on" if args.deblock else "") + (", quantiser matrices 1..15" if args.qm else "") + (
                                           ", loop restoration (Wiener + self-guided) on" if args.sgr else (", loop restoration on" if args.lr else "")),
                       "frames_per_chunk": n, "chunks_per_gpu": W_,
                       "parallelism": "chunk-per-gpu x%d" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": peak, "unit": "GB/s",
                         "frac": round(achieved / peak, 5), "traffic": traffic, "traffic_raw_counters": traffic_raw,
                         "algorithmic_bytes_per_launch": alg[dom], "kernel_ms": round(stage[dom], 3)},
            "stage_ms": {s: round(v, 3) for s, v in stage.items()},
            "bytes_per_frame": round(int(rep.bytes) / n, 1),
            "psnr_db": [round(rep.psnr[i], 2) for i in range(3)],
            "symbols_per_frame": int(rep.n_symbols // n), "max_tile_symbols": int(rep.max_tile_symbols),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, h, bd, args.block_log2)  # (the oracle's all-key-frame path)
        print(json.dumps(out), flush=True)
    for c_ in ctxs:
        c_.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
