import os, sys, time, tempfile, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "av1-base_amd"))
import torch, bench, av1mi
dev = torch.device("cuda", 0)
w, h, bd, n = 1920, 1080, 10, int(os.environ.get("E2E_FRAMES", "240"))
tmp = tempfile.mkdtemp(prefix="e2e_", dir="/dev/shm")
y4m, out = os.path.join(tmp, "c.y4m"), os.path.join(tmp, "c.mkv")
fb = w * h * 3
with open(y4m, "wb") as f:
    f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420p10\n" % (w, h))
    clip = bench.make_clip_torch(w, h, bd, 30, 1080, dev).cpu().numpy()
    for t in range(n):
        f.write(b"FRAME\n"); f.write(clip[(t % 30) * fb:(t % 30 + 1) * fb].tobytes())
if os.environ.get("E2E_TIMING"):
    os.environ["AV1MI_TIMING"] = "1"
for rt in os.environ.get("E2E_THREADS", "12").split(","):
    os.environ["AV1MI_READ_THREADS"] = rt
    for wk in os.environ.get("E2E_WORKERS", "4").split(","):
        best = 1e9
        for i in range(3):
            t = time.perf_counter()
            rep = av1mi.run_mi355x(av1mi.EncodeParams(y4m, out, tmp, av1mi.derive_plan(64, workers_override=int(wk)), chunk_frames=int(os.environ.get("E2E_CHUNK", "60"))))
            best = min(best, time.perf_counter() - t)
        print("read threads %s workers %s: best %.1f ms = %.0f fps" % (rt, wk, best * 1e3, n / best), flush=True)
shutil.rmtree(tmp)
