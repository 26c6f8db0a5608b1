#!/bin/bash
# round 5: the driver's 20-step call cut into two launch sets (round 3's rule) or three -- re-measured on the round's kernels
mkdir -p gpurun_out
for rep in 1 2 3; do for n in 2 3; do
  PTX_DEBUG_NSETS=$n python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nsets $n: 20 steps', round(d['ms_per_step'],4))"
done; done
for rep in 1 2; do for n in 2 3; do
  PTX_DEBUG_NSETS=$n python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nsets $n: 40 steps', round(d['ms_per_step'],4))"
done; done
