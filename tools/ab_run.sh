#!/bin/bash
# bash tools/ab_run.sh CHECKVARIANT A B C ...: A/B timing of .ab/lib*.so variants, then tools/gpu_quick.py (HIP vs oracle) on CHECKVARIANT
set -e
CHK=$1; shift
cp mygpuraytracer_amd/libmi355x_pathtracer.so /tmp/keep.so
bash tools/ab_bench.sh "$@"
if [ "$CHK" != "-" ]; then
  cp .ab/lib$CHK.so mygpuraytracer_amd/libmi355x_pathtracer.so
  python tools/gpu_quick.py > gpurun_out/quick_$CHK.log 2>&1 || true
  tail -4 gpurun_out/quick_$CHK.log
fi
cp /tmp/keep.so mygpuraytracer_amd/libmi355x_pathtracer.so
