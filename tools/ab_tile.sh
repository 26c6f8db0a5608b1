#!/bin/bash
# A/B of abx/lib*.so variants on the short tile run (20 steps, rank 0 of 8 and of 4) and the full frame
for rep in 1 2; do for v in "$@"; do
  PTX_DEV=1 PTX_AB_LIBRARY=$PWD/abx/lib$v.so python tools/gpu_tile_short.py $v 2>/dev/null
done; done
