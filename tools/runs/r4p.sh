#!/bin/bash
# the ranking pass with two histograms and two barriers per tile (P) against the library before it (S): split-scene tests, then C5 A/B
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "c5 or split or spaceship or random_scenes or mesh or stream" > gpurun_out/r4p_tests.log 2>&1 || { tail -30 gpurun_out/r4p_tests.log; exit 1; }
tail -2 gpurun_out/r4p_tests.log
for rep in 1 2 3 4; do for v in S P; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_c5_leg.py 72 2>/dev/null; done; done
for v in S P; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so C5_ITERS=36 timeout -k 10 300 python tools/gpu_c5_profile.py 2>/dev/null | tail -3; done
