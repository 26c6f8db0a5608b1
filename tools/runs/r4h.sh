#!/bin/bash
run() { # lib batch wgpercu
  PTX_AB_LIBRARY=$PWD/.ab/lib$1.so PTX_DEBUG_MESH_WG_PER_CU=$3 python tools/gpu_kernel_split.py cornellSpaceship20k.txt 3840 2160 depth_of_field=1 batch=$2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 batch=$2 wg/cu=$3', d['wall_ms_per_iter'], d['kernels_ms_per_iter'])"
}
for b in 5 12; do
run OLD $b 0
for v in q64 q128 q128all q128r32 q128r8 q128n16 q128n48 q128allr32; do run $v $b 0; done
run q128 $b 3; run q128 $b 4; run q128all $b 3
done
