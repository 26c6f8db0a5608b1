#!/bin/bash
for cfg in "28 14" "28 10" "28 8" "21 10" "16 8" "28 16" "14 8"; do set -- $cfg; echo "first $1 later $2"; PTX_DEBUG_WG_FIRST=$1 PTX_DEBUG_WG_LATER=$2 timeout -k 10 300 python tools/gpu_c4_long.py 2>/dev/null
  PTX_DEBUG_WG_FIRST=$1 PTX_DEBUG_WG_LATER=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  20 steps', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['kernels_ms_per_step'].items() if x})"
done
