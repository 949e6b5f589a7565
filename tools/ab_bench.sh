#!/bin/bash
# usage (GPU box, repo root): tools/ab_bench.sh <libA.so> <libB.so> [bench flags...]   - the same bench line for two builds, interleaved
a=$1; b=$2; shift 2
for rep in 1 2 3; do
  for lib in $a $b; do
    AV1MI_LIB=$PWD/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --configs none "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['stage_ms']['recon'], d['stage_ms']['symbolize'], d['stage_ms']['rangecode'])" || exit 1
  done
done
