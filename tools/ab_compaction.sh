#!/bin/bash
# A/B of prebuilt library variants (.ab/lib*.so) on the scan / compaction rates: bash tools/ab_compaction.sh A B C ...
cp mygpuraytracer_amd/libmi355x_pathtracer.so .ab/lib_keep.so
for v in "$@"; do
  cp .ab/lib$v.so mygpuraytracer_amd/libmi355x_pathtracer.so
  echo "== $v"
  python tools/gpu_compaction_bw.py ${AB_SIZES:-8294400 67108864 268435456} 2>&1 | grep "^{" | cut -c1-150
done
cp .ab/lib_keep.so mygpuraytracer_amd/libmi355x_pathtracer.so
