#!/bin/bash
# usage (GPU box, repo root): tools/ab_env.sh "<ENV=VAL ...|->" "<ENV=VAL ...|->" [bench flags...]   - one build, two environments, interleaved
a=$1; b=$2; shift 2
for rep in 1 2 3; do
  for e in "$a" "$b"; do
    ( [ "$e" != "-" ] && export $e; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --configs none "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['value'], d['stage_ms']['recon'], d['stage_ms']['symbolize'], d['stage_ms']['rangecode'])" ) || exit 1
  done
done
