#!/bin/bash
# A/B of .ab/lib*.so variants on the short tile run (20 steps, rank 0 of 8 and of 4) and the full frame
cp mygpuraytracer_amd/libmi355x_pathtracer.so /tmp/keep.so
for rep in 1 2; do for v in "$@"; do
  cp .ab/lib$v.so mygpuraytracer_amd/libmi355x_pathtracer.so
  python tools/gpu_tile_short.py $v 2>/dev/null
done; done
cp /tmp/keep.so mygpuraytracer_amd/libmi355x_pathtracer.so
