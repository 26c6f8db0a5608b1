#!/bin/bash
# round 5: the grids (workgroups per CU of the camera bounce / of the later bounces) and (iterations per set x sets in flight) swept again on
# the round's kernels -- three barriers per tile instead of six, lighter tails: does the balance sit where round 4 left it (20 / 8; 12 x 3)?
mkdir -p gpurun_out
for rep in 1 2; do for cfg in "20 8" "28 8" "14 8" "20 12" "20 16" "28 16" "12 8"; do set -- $cfg; echo "first $1 later $2"
  PTX_DEBUG_WG_FIRST=$1 PTX_DEBUG_WG_LATER=$2 timeout -k 10 300 python tools/gpu_c4_long.py 2>/dev/null
done; done
python tools/gpu_batch_lanes_sweep.py 2>/dev/null
