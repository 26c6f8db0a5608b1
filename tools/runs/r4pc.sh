#!/bin/bash
# PC sampling (beta) of the C4 bench: which instructions of k_bounce the waves are at
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pcs
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${1:-host_trap} --pc-sampling-unit ${2:-time} --pc-sampling-interval ${3:-1} --kernel-trace --output-format csv -d $R/gpurun_out/pcs -- python3 $R/bench.py --lanes 1 --steps 24 --warmup 12 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/pcs.log 2>&1
echo rc=$?
tail -5 $R/gpurun_out/pcs.log
find $R/gpurun_out/pcs -type f | head; du -sh $R/gpurun_out/pcs
