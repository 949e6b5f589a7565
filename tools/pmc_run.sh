#!/bin/bash
# usage: tools/pmc_run.sh <tag> <counter list...>   (run on the GPU box from the repo root)
# collects the given PMC counters for one bench step; --kernel-trace only (no other trace domains)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames ${PMC_FRAMES:-20} > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag exit=$?"
