#!/bin/bash
# A/B of abx/lib*.so variants on the C5-shaped runs (per-kernel split): bash tools/ab_c5.sh A B ...   (AB_SCENES="a.txt b.txt" picks the scenes)
for rep in $(seq 1 ${AB_REPS:-2}); do
for v in "$@"; do
  for sc in ${AB_SCENES:-cornellSpaceship.txt cornellSpaceship20k.txt}; do
    PTX_DEV=1 PTX_AB_LIBRARY=$PWD/abx/lib$v.so python tools/gpu_kernel_split.py $sc 3840 2160 depth_of_field=1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['scene'][:22], d['wall_ms_per_iter'], d['kernels_ms_per_iter'])"
  done
done
done
