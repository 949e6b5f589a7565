"""Multi-process (gloo, world_size 2) test of the N>1 path on CPU: scene-chunks shard over ranks
with no data-path collective; the only collectives are the timing barrier / MAX
and the host-side gather of per-chunk streams in chunk order.  The encode itself is stood in for
by the oracle here (no GPU in this container) - what is tested is the sharding, the ordering and
that concatenated chunk streams equal the single-process result."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chunks, q):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import av1o
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, frames = 72, 56, 2
    cfg = av1o.default_config(w, h, 8, min_bs_log2=5, max_bs_log2=5)
    mine = {}
    for c in range(n_chunks):
        if c % world != rank:          # chunk i -> rank i mod G
            continue
        data = b""
        for t in range(frames):
            tu, _, _ = av1o.encode_frame(cfg, av1o.synthclip_frame(w, h, 8, seed=100 + c, t=t))
            data += tu
        mine[c] = data
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)     # the bench's max-over-ranks timing
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        merged = {}
        for g in gathered:
            merged.update(g)
        q.put((float(t.item()), b"".join(merged[c] for c in sorted(merged)), sorted(merged)))
    dist.destroy_process_group()


def test_two_rank_chunk_sharding():
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import av1o
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_chunks = 5
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_chunks, q)) for r in range(2)]
    for p in procs:
        p.start()
    tmax, stream, order = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert tmax == 2.0 and order == list(range(n_chunks))
    cfg = av1o.default_config(72, 56, 8, min_bs_log2=5, max_bs_log2=5)
    ref = b""
    for c in range(n_chunks):
        for t in range(2):
            tu, _, _ = av1o.encode_frame(cfg, av1o.synthclip_frame(72, 56, 8, seed=100 + c, t=t))
            ref += tu
    assert stream == ref
