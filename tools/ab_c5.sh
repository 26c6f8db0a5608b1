#!/bin/bash
# A/B of prebuilt library variants on the C5 20k-triangle case: bash tools/ab_c5.sh M5 M6 ...
for v in "$@"; do
  cp .ab/lib$v.so mygpuraytracer_amd/libmi355x_pathtracer.so
  echo "$v $(python tools/gpu_kernel_split.py cornellSpaceship20k.txt 3840 2160 depth_of_field=1 | cut -c70-260)"
done
