#!/bin/bash
# round 5: the arithmetic levels -- per-stage differences (to set the stated bounds), then the new GPU tests
mkdir -p gpurun_out
python tools/gpu_arith.py --errors --no-timing --no-c5 > gpurun_out/r5b_arith2.log 2>&1 || { tail -20 gpurun_out/r5b_arith2.log; exit 1; }
cat gpurun_out/r5b_arith2.log
python -m pytest tests/test_gpu_arith.py -q > gpurun_out/r5b_tests.log 2>&1; tail -40 gpurun_out/r5b_tests.log | cut -c1-400
