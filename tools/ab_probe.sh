#!/bin/bash
# same-box A/B of the product library against av1-base_amd/ab/libav1mi_<name>.so on a few bench workloads: tools/ab_probe.sh <name> [repeats]
v=$1; reps=${2:-2}
run() { local name=$1; shift
  python bench.py --configs none --no-cpu-baseline --steps 6 --warmup 2 "$@" > /tmp/ab.json 2>/tmp/ab.err || { echo "$name failed"; tail -2 /tmp/ab.err; return; }
  python - "$name" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1]); s = d["stage_ms"]
print("%-26s fps %8.0f  step %7.3f  recon %.3f sym %.3f rc %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], s["recon"], s["symbolize"], s["rangecode"]))
PY
}
for r in $(seq $reps); do
  for lib in product $v; do
    if [ $lib = product ]; then unset AV1MI_LIB; else export AV1MI_LIB=av1-base_amd/ab/libav1mi_$lib.so; fi
    run ${lib}_intra60
    run ${lib}_ippp60 --keyint 240 --mode-mask 7
    run ${lib}_intra60_x4 --chunks-per-gpu 4
    run ${lib}_ippp60_x4 --keyint 240 --mode-mask 7 --chunks-per-gpu 4
    run ${lib}_prod60 --keyint 240 --cq 8 --qm --film-grain 20 --subpel --deblock --sgr --mode-mask 7
  done
done
