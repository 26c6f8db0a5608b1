#!/bin/bash
# round 5 profiling pass B (the C5-shaped run: 3840x2160, DoF, textured 20448-triangle BVH mesh): kernel stats + PMC traffic, SQ counters
set -e
bash tools/profile_c5.sh > gpurun_out/r5prof_c5.log 2>&1; tail -1 gpurun_out/r5prof_c5.log
bash tools/pmc_c5.sh round5 > gpurun_out/r5prof_c5sq.log 2>&1; tail -1 gpurun_out/r5prof_c5sq.log
