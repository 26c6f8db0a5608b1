#!/bin/bash
# A/B of abx/lib*.so variants on one box with tools/gpu_arith.py (C4 200-step wall + kernels alone, C5 72 iterations + kernels alone):
#   bash tools/ab_levels.sh A B ...      (AB_REPS=2 by default; AB_ARGS="--levels 0 --no-c5" etc.)
for rep in $(seq 1 ${AB_REPS:-2}); do
for v in "$@"; do
  echo "== $v (rep $rep)"
  PTX_DEV=1 PTX_AB_LIBRARY=$PWD/abx/lib$v.so python tools/gpu_arith.py ${AB_ARGS:---levels 0} 2>/dev/null | grep '^{'
done
done
