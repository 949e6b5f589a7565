#!/bin/bash
# run bench.py with extra args and print a compact line (bring-up helper, GPU box)
tag=$1; shift
python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', d['value'], d['stage_ms'], d['bytes_per_frame'], d['psnr_db'][0], 'sym/frame', d['symbols_per_frame'], 'max_tile', d['max_tile_symbols'])"
