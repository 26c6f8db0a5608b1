#!/bin/bash
# A/B of .ab/lib*.so variants on the C5-shaped runs (per-kernel split): bash tools/ab_c5.sh A B ...
cp mygpuraytracer_amd/libmi355x_pathtracer.so /tmp/keep.so
for rep in 1 2; do
for v in "$@"; do
  cp .ab/lib$v.so mygpuraytracer_amd/libmi355x_pathtracer.so
  for sc in cornellSpaceship.txt cornellSpaceship20k.txt; do
    python tools/gpu_kernel_split.py $sc 3840 2160 depth_of_field=1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['scene'][:22], d['wall_ms_per_iter'], d['kernels_ms_per_iter'])"
  done
done
done
cp /tmp/keep.so mygpuraytracer_amd/libmi355x_pathtracer.so
