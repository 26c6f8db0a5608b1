#!/bin/bash
# round 5 profiling pass A (C4): kernel stats + PMC traffic, SQ counters at the exact level, the VALU group at arithmetic levels 1 and 2
set -e
bash tools/profile_round.sh > gpurun_out/r5prof_round.log 2>&1; tail -2 gpurun_out/r5prof_round.log
bash tools/pmc_sq.sh round5 > gpurun_out/r5prof_sq.log 2>&1; tail -1 gpurun_out/r5prof_sq.log
PMC_EXTRA_ARGS="--arith 1" PMC_GROUPS="a d" bash tools/pmc_sq.sh round5_arith1 > gpurun_out/r5prof_sq1.log 2>&1; tail -1 gpurun_out/r5prof_sq1.log
PMC_EXTRA_ARGS="--arith 2" PMC_GROUPS="a d" bash tools/pmc_sq.sh round5_arith2 > gpurun_out/r5prof_sq2.log 2>&1; tail -1 gpurun_out/r5prof_sq2.log
