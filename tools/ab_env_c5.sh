#!/bin/bash
# C5-shaped kernel split under values of an environment knob: bash tools/ab_env_c5.sh VAR [v1 v2 ...]   (no values: unset, then 1)
VAR=$1; shift
VALS="$@"; [ -z "$VALS" ] && VALS="off 1"
for rep in 1 2; do
  for v in $VALS; do
    if [ $v = off ]; then unset $VAR; else export $VAR=$v; fi
    for sc in cornellSpaceship.txt cornellSpaceship20k.txt; do
      timeout -k 10 120 python tools/gpu_kernel_split.py $sc 3840 2160 depth_of_field=1 2>gpurun_out/ab_env_c5.err | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else None
print('$VAR', '$v', d and (d['scene'][:22], d['wall_ms_per_iter'], d['kernels_ms_per_iter']))"
    done
  done
done
