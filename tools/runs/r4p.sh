#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4p_tests.log 2>&1; tail -3 gpurun_out/r4p_tests.log
AB_REPS=2 AB_BENCH_ARGS="--no-extra-legs" bash tools/ab_bench.sh TGE FX
for v in TGE FX TGE FX; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so python tools/gpu_c5_leg.py 72 2>/dev/null; done
