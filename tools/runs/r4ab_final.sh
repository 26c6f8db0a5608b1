#!/bin/bash
# round 3's library against this round's on ONE box: the driver's bench command (C4, 20 steps) and the C5 leg (24 iterations as round 3 timed
# it, and 72), interleaved, five / three pairs
for rep in 1 2 3 4 5; do for v in R3 R4; do
  PTX_AB_LIBRARY=$PWD/.ab/lib$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v C4 20 steps', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['kernels_ms_per_step'].items() if x})"
done; done
for rep in 1 2 3; do for v in R3 R4; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so python tools/gpu_c5_leg.py 24 72 2>/dev/null; done; done
