#!/bin/bash
# k_mesh variants x iterations per launch set (batch) x k_mesh workgroups per CU on the C5 20k scene: wall ms per iteration and kernels
run() { # lib batch wgpercu
  PTX_AB_LIBRARY=$PWD/.ab/lib$1.so PTX_DEBUG_MESH_WG_PER_CU=$3 python tools/gpu_kernel_split.py cornellSpaceship20k.txt 3840 2160 depth_of_field=1 batch=$2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 batch=$2 wg/cu=$3', d['wall_ms_per_iter'], d['kernels_ms_per_iter'])"
}
run OLD 5 0; run OLD 8 0; run OLD 12 0
run c128 5 0; run c128 8 0; run c128 12 0
run c128 8 3; run c128 12 3; run c128 12 2
run c64 8 3; run c64 12 3; run c64r32 12 3
