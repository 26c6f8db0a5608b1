#!/bin/bash
# round 5: "lit" flags (only radiance-carrying path ends stored / gathered) -- GPU tier, then A/B against the previous build on this box
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r5j_tests.log 2>&1 || { tail -40 gpurun_out/r5j_tests.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r5j_tests.log
bash tools/ab_levels.sh Prev Idx16 > gpurun_out/r5j_ab.log 2>&1; cat gpurun_out/r5j_ab.log
