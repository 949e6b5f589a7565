#!/usr/bin/env python3
"""Probe: K contexts (HIP streams) on ONE GPU, each encoding its own chunk from its own host thread.
The serial range-coding tail of one chunk overlaps the throughput-bound kernels of the others."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av1-base_amd")); sys.path.insert(0, ROOT)
import torch, av1mi
from bench import make_clip
w, h, bd, n = 1920, 1080, 10, 60
clip = make_clip(w, h, bd, n, 1080)
dev = torch.device("cuda", 0)
d = torch.frombuffer(bytearray(clip), dtype=torch.uint8).to(dev)
params = av1mi.default_params(w, h, bd)
for K in (1, 2, 3, 4):
    ctxs = [av1mi.Context(0) for _ in range(K)]
    for c in ctxs: c.encode_chunk(params, d.data_ptr(), n, on_device=True)
    steps = 4
    def run(c):
        for _ in range(steps): c.encode_chunk(params, d.data_ptr(), n, on_device=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(c,)) for c in ctxs]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print("contexts", K, "fps", round(K * steps * n / dt, 1), flush=True)
    for c in ctxs: c.close()
