#!/bin/bash
# round 5: the direct tile epilogue -- GPU tier, then C4 / C5 wall and kernel times at the exact level (compare: gpurun_out/r5b_arith.log)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r5d_tests.log 2>&1 || { tail -40 gpurun_out/r5d_tests.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r5d_tests.log
python tools/gpu_arith.py --levels ${LEVELS:-0} > gpurun_out/r5d_time.log 2>&1; cat gpurun_out/r5d_time.log
