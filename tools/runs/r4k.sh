#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sorted_stream or fences or cottage or production_intersect or split_mesh or random_scenes or ship" > gpurun_out/r4k_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r4k_tests.log; [ $rc -ne 0 ] && exit $rc
AB_REPS=2 AB_SCENES=cornellSpaceship20k.txt bash tools/ab_c5.sh NEW S16 S12 S20w6 S16w6 S16r32
for v in NEW S16 NEW S16; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so python tools/gpu_c5_leg.py 72 2>/dev/null; done
