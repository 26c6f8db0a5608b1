#!/bin/bash
# bash tools/ab_run.sh CHECKVARIANT A B C ...: A/B timing of abx/lib*.so variants, then tools/gpu_quick.py (HIP vs oracle) on CHECKVARIANT
set -e
CHK=$1; shift
bash tools/ab_bench.sh "$@"
if [ "$CHK" != "-" ]; then
  PTX_DEV=1 PTX_AB_LIBRARY=$PWD/abx/lib$CHK.so python tools/gpu_quick.py > gpurun_out/quick_$CHK.log 2>&1 || true
  tail -4 gpurun_out/quick_$CHK.log
fi
