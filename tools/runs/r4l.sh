#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4l_tests.log 2>&1; tail -3 gpurun_out/r4l_tests.log
AB_REPS=2 AB_BENCH_ARGS="--no-extra-legs" bash tools/ab_bench.sh NEW TG
