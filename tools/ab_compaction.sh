#!/bin/bash
# A/B of prebuilt library variants (abx/lib*.so) on the scan / compaction rates: bash tools/ab_compaction.sh A B C ...
for v in "$@"; do
  echo "== $v"
  PTX_DEV=1 PTX_AB_LIBRARY=$PWD/abx/lib$v.so python tools/gpu_compaction_bw.py ${AB_SIZES:-8294400 67108864 268435456} 2>&1 | grep "^{" | cut -c1-150
done
