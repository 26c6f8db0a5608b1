#!/bin/bash
# does the wall time follow HBM bytes?  N = the product; UV = the same kernels carrying two unused fields (+8 % traffic, +0.7 % instructions)
for rep in 1 2 3 4; do for v in N UV; do
  PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v C4 20 steps', round(d['ms_per_step'],4), 'long', round(d.get('long_run',{}).get('ms_per_step',0),4) if isinstance(d.get('long_run'),dict) else d.get('long_run'), {k: round(x,4) for k,x in d['roofline']['kernels_ms_per_step'].items() if x})"
done; done
for rep in 1 2 3 4; do for v in N UV; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_c4_long.py 2>/dev/null; done; done
