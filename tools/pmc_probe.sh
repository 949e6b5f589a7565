#!/bin/bash
# usage (GPU box, repo root): tools/pmc_probe.sh <tag> [bench flags...]   - SQ issue counters + instruction-cache counters of one bench step
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export AV1MI_BENCH_CLIP_ON_CPU=1
B=$GRAFT_REPO_ROOT/bench.py
rocprofv3 -L > $out/counters.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/pmc_sq -- python3 $B --steps 1 --warmup 0 --no-cpu-baseline --configs none "$@" > $out/pmc_sq.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT --output-format csv -d $out/pmc_ic -- python3 $B --steps 1 --warmup 0 --no-cpu-baseline --configs none "$@" > $out/pmc_ic.log 2>&1
echo "probe $tag exit=$?"
python3 - <<PY
import csv, glob, collections
for d in ("pmc_sq", "pmc_ic"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for pat in ("recon_sb_kernel", "symbolize_tile", "rangecode_tiles", "cdef_sb"):
                if pat in k:
                    acc[pat][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, cs in acc.items():
        print(d, k, {c: int(v) for c, v in sorted(cs.items())})
PY
