#!/usr/bin/env python3
"""Copy what tools/collect_evidence.sh left under gpurun_out/<tag>/ into profiles/ (the tracked summaries):

  profiles/<tag>_kernel_stats_{intra,ippp,prod}.csv      rocprofv3 --kernel-trace --stats tables (only this library's kernels)
  profiles/<tag>_pmc_{intra,ippp}.txt                    per kernel: counter sums over the dispatches of ONE bench step
  profiles/pmc_traffic.json                              FETCH_SIZE / WRITE_SIZE in KiB per launch per bench stage (bench.py reads it)

usage: tools/evidence_to_profiles.py <tag>
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = ("recon_", "cdef_", "symbolize_", "rangecode", "motion_search", "subpel_refine", "me64_", "pack_", "sse_", "lr_", "deblock_",
        "luma_sad", "tile_order", "carry_", "scene_")
STAGE_OF = (("recon_sb_kernel", "recon"), ("recon_inter_pre_kernel", "recon_pre"), ("cdef_sb_kernel", "cdef"), ("symbolize_tile_kernel", "symbolize"),
            ("rangecode2_tiles_kernel", "rangecode"), ("rangecode4_tiles_kernel", "rangecode"), ("motion_search_kernel", "motion_search"))


def short(name):
    m = re.search(r"(\w+_kernel)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def stats(tag, what):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", tag, "stats_" + what, "**", "*kernel_stats.csv"), recursive=True)
    if not fs:
        return None
    rows = [r for r in csv.DictReader(open(fs[0])) if any(k in r["Name"] for k in OURS)]
    dst = os.path.join(ROOT, "profiles", "%s_kernel_stats_%s.csv" % (tag, what))
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    return dst


def counters(tag, dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for d in dirs:
        for f in glob.glob(os.path.join(ROOT, "gpurun_out", tag, d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if not any(k in r["Kernel_Name"] for k in OURS):
                    continue
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[(k, r["Counter_Name"])] += 1
    return acc, calls


def main():
    tag = sys.argv[1]
    for what in ("intra", "ippp", "prod"):
        print(stats(tag, what))
    traffic = {}
    for what, dirs, wl in (("intra", ("pmc_fetch", "pmc_write", "pmc_sq"), "cfg2_1080p_intra"),
                           ("ippp", ("pmc_fetch_ippp", "pmc_write_ippp", "pmc_sq_ippp"), "cfg3_1080p_ippp")):
        acc, calls = counters(tag, dirs)
        if not acc:
            continue
        dst = os.path.join(ROOT, "profiles", "%s_pmc_%s.txt" % (tag, what))
        with open(dst, "w") as f:
            f.write("# rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --configs none%s\n"
                    % (" --keyint 240 --mode-mask 7" if what == "ippp" else ""))
            f.write("# separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*; sums over the dispatches of one 60-frame 1080p 10-bit chunk; FETCH/WRITE in KiB (raw)\n")
            for k in sorted(acc):
                f.write(k + "\n")
                for c, v in sorted(acc[k].items()):
                    f.write("    %-24s %18.0f   (%d dispatches)\n" % (c, v, calls[(k, c)]))
        print(dst)
        ent = {}
        for k, cs in acc.items():
            for pat, stage in STAGE_OF:
                if k.startswith(pat) and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                    e = ent.setdefault(stage, {"FETCH_SIZE": 0, "WRITE_SIZE": 0, "dispatches": 0, "SQ_INSTS_VALU": 0})
                    e["FETCH_SIZE"] += int(cs["FETCH_SIZE"]); e["WRITE_SIZE"] += int(cs["WRITE_SIZE"]); e["dispatches"] += calls[(k, "FETCH_SIZE")]
                    e["SQ_INSTS_VALU"] += int(cs.get("SQ_INSTS_VALU", 0))   # wave-instructions (bench.py: valu_busy)
        if what == "ippp":   # the P-frame kernels run once per frame: per-launch figures
            for e in ent.values():
                if e["dispatches"] > 2:
                    e["FETCH_SIZE"] //= e["dispatches"]; e["WRITE_SIZE"] //= e["dispatches"]; e["SQ_INSTS_VALU"] //= e["dispatches"]; e["per"] = "launch (one frame)"
        traffic[wl] = ent
    if traffic:
        dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        json.dump({"source": "profiles/%s_pmc_*.txt (tools/collect_evidence.sh %s; rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % (tag, tag),
                   "unit": "FETCH_SIZE / WRITE_SIZE: KiB per launch, raw counter values (gfx950: FETCH_SIZE counts 64 B per 128-B request - bench.py doubles it); SQ_INSTS_VALU: wave-instructions per launch",
                   "workloads": traffic}, open(dst, "w"), indent=1)
        print(dst)


if __name__ == "__main__":
    main()
