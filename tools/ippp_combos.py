import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/av1-base_amd")
import torch, av1mi, bench
dev = torch.device("cuda", 0)
w,h,bd,n = 1920,1080,10,60
clip = bench.make_clip_torch(w,h,bd,n,1080,dev); torch.cuda.synchronize()
import time
def run(label, **kw):
    p = av1mi.default_params(w,h,bd,keyint=240, **kw)
    with av1mi.Context(0) as c:
        for _ in range(2): c.encode_chunk(p, clip.data_ptr(), n, on_device=True, copy_out=False)
        t=time.perf_counter(); k=4
        for _ in range(k): rep=c.encode_chunk(p, clip.data_ptr(), n, on_device=True, copy_out=False)[2]
        dt=(time.perf_counter()-t)/k
    print("%-36s %7.0f fps  recon %.2f ms entropy %.2f  %6.1f KB/frame  %.2f dB" % (label, n/dt, rep.ms_recon, rep.ms_entropy, rep.bytes/n/1e3, rep.psnr[0]), flush=True)
run("bs5 fixed", block_log2=5)
run("bs6 fixed", block_log2=6)
run("bs5 partition 3..5", block_log2=5, partition_search=1, min_block_log2=3)
run("bs6 partition 3..6", block_log2=6, partition_search=1, min_block_log2=3)
run("bs6 partition 4..6", block_log2=6, partition_search=1, min_block_log2=4)
run("bs5 presearch", block_log2=5, me_presearch=1)
run("bs6 presearch", block_log2=6, me_presearch=1)
