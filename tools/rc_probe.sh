#!/bin/bash
# range-coder A/B on one box: its two forms (AV1MI_RC_STAGES) on bench workloads either side of 256 workgroups per launch
out=${1:-gpurun_out/rc_probe}; shift; mkdir -p $out
run() {  # name, bench args..., env via RCENV
  local name=$1; shift
  env $RCENV python bench.py --configs none --no-cpu-baseline --steps 5 --warmup 1 "$@" > $out/$name.json 2> $out/$name.err
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d["stage_ms"]
    print("%-28s fps %8.0f  step %7.3f  sym %.3f rc %.3f  max_tile_symbols %d" % (sys.argv[2], d["value"], d["ms_per_step"], s["symbolize"], s["rangecode"], d["max_tile_symbols"]))
except Exception as e:
    print("%-28s failed: %r" % (sys.argv[2], e))
PY
}
for st in 2 4; do
  export RCENV="AV1MI_RC_STAGES=$st"
  run intra30_s$st --frames 30
  run intra60_s$st --frames 60
  run intra120_s$st --frames 120
  run intra60_cq8_s$st --frames 60 --cq 8
  run ippp60_s$st --frames 60 --keyint 240 --mode-mask 7
  run ippp60_cq8_s$st --frames 60 --keyint 240 --mode-mask 7 --cq 8
  run prod60_s$st --frames 60 --keyint 240 --cq 8 --qm --film-grain 20 --subpel --deblock --sgr
  run intra4k30_s$st --frames 30 --width 3840 --height 2160
  run ippp4k30_s$st --frames 30 --width 3840 --height 2160 --keyint 240 --mode-mask 7
done
