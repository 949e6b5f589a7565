#!/bin/bash
# usage (on the MI355X box, from the repo root): tools/collect_evidence.sh <tag>      e.g. r03_a
# One pass per kind of evidence, each its own rocprofv3 run (kernel trace + stats; FETCH_SIZE; WRITE_SIZE; SQ counters) of
# `python3 bench.py` directly after `--`.  Everything lands under gpurun_out/<tag>/ ; tools/evidence_to_profiles.py copies the
# summaries into profiles/.
tag=${1:-r03_x}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
common="--no-cpu-baseline --configs none"
if [ -z "$PMC_ONLY" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_intra -- python3 $B --steps 5 --warmup 2 $common > $out/stats_intra.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_ippp -- python3 $B --steps 5 --warmup 2 $common --keyint 240 --mode-mask 7 > $out/stats_ippp.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_prod -- python3 $B --steps 3 --warmup 1 $common --keyint 240 --mode-mask 7 --subpel --qm --deblock --sgr --film-grain 20 --cq 8 > $out/stats_prod.log 2>&1 || exit 1
fi
export AV1MI_BENCH_CLIP_ON_CPU=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $B --steps 1 --warmup 0 $common > $out/pmc_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $B --steps 1 --warmup 0 $common > $out/pmc_write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/pmc_sq -- python3 $B --steps 1 --warmup 0 $common > $out/pmc_sq.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_ippp -- python3 $B --steps 1 --warmup 0 $common --keyint 240 --mode-mask 7 > $out/pmc_fetch_ippp.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_ippp -- python3 $B --steps 1 --warmup 0 $common --keyint 240 --mode-mask 7 > $out/pmc_write_ippp.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/pmc_sq_ippp -- python3 $B --steps 1 --warmup 0 $common --keyint 240 --mode-mask 7 > $out/pmc_sq_ippp.log 2>&1
rc=$?
# the traces themselves are large: keep the stats and the counter tables only
find $out -name '*kernel_trace.csv' -path '*stats_*' -delete
echo "evidence $tag exit=$rc"; ls $out
exit $rc
