#!/bin/bash
# usage (GPU box, repo root): tools/ab_e2e.sh <libA.so> <libB.so>   - the file -> file configurations of bench.py for two builds, interleaved
a=$1; b=$2
for rep in 1 2; do
  for lib in $a $b; do
    AV1MI_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --configs auto 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']; print('$lib', d['value'], c['cfg2_e2e_y4m_to_mkv']['fps'], c['cfg3_e2e_y4m_to_mkv_ippp']['fps'], c['cfg2_1080p_intra_x4']['fps'], c['production_1080p_x4']['fps'])" || exit 1
  done
done
