#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4d2_tests.log 2>&1 || { tail -30 gpurun_out/r4d2_tests.log; exit 1; }
tail -2 gpurun_out/r4d2_tests.log
for rep in 1 2 3 4; do for v in N D2; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_c5_leg.py 72 2>/dev/null; done; done
