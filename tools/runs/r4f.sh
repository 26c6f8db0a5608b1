#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sorted_stream or fences or cottage or production_intersect or split_mesh or random_scenes or ship" > gpurun_out/r4f_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r4f_tests.log; [ $rc -ne 0 ] && exit $rc
run() { # lib batch wgpercu
  PTX_AB_LIBRARY=$PWD/.ab/lib$1.so PTX_DEBUG_MESH_WG_PER_CU=$3 python tools/gpu_kernel_split.py cornellSpaceship20k.txt 3840 2160 depth_of_field=1 batch=$2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 batch=$2 wg/cu=$3', d['wall_ms_per_iter'], d['kernels_ms_per_iter'])"
}
for b in 5 12; do
run OLD $b 0
for v in n1 n1all n1r32 n1allr32 n1w8; do run $v $b 0; done
run n1 $b 3; run n1all $b 3; run n1r32 $b 3
done
