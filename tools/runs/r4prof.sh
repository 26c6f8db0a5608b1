#!/bin/bash
# round 4 profiling pass: C4 kernel stats + PMC traffic, SQ counters, then the same for the C5-shaped run
set -e
bash tools/profile_round.sh > gpurun_out/r4prof_round.log 2>&1; tail -2 gpurun_out/r4prof_round.log
bash tools/pmc_sq.sh round4 > gpurun_out/r4prof_sq.log 2>&1; tail -1 gpurun_out/r4prof_sq.log
bash tools/profile_c5.sh > gpurun_out/r4prof_c5.log 2>&1; tail -1 gpurun_out/r4prof_c5.log
bash tools/pmc_c5.sh round4 > gpurun_out/r4prof_c5sq.log 2>&1; tail -1 gpurun_out/r4prof_c5sq.log
