#!/bin/bash
# C5-shaped kernel split with and without an environment knob: bash tools/ab_env_c5.sh VAR
VAR=$1
for rep in 1 2; do
  for v in off on; do
    if [ $v = on ]; then export $VAR=1; else unset $VAR; fi
    for sc in cornellSpaceship.txt cornellSpaceship20k.txt; do
      timeout -k 10 120 python tools/gpu_kernel_split.py $sc 3840 2160 depth_of_field=1 2>gpurun_out/ab_env_c5.err | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else None
print('$VAR', '$v', d and (d['scene'][:22], d['wall_ms_per_iter'], d['kernels_ms_per_iter']))"
    done
  done
done
