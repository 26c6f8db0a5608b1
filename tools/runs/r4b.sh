#!/bin/bash
# round 4: first run of the refilling k_mesh -- the split-scene parity tests under a short timeout, then the C5 kernel split
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sorted_stream or fences or cottage or production_intersect or split_mesh or random_scenes or c5_ship or ship" > gpurun_out/r4b_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r4b_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/gpu_kernel_split.py cornellSpaceship20k.txt 3840 2160 depth_of_field=1 > gpurun_out/r4b_c5split.json 2>&1 && tail -1 gpurun_out/r4b_c5split.json
timeout -k 10 200 python tools/gpu_kernel_split.py cornellSpaceship.txt 3840 2160 depth_of_field=1 > gpurun_out/r4b_c5split320.json 2>&1 && tail -1 gpurun_out/r4b_c5split320.json
