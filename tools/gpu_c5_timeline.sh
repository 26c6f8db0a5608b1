#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace) of the C5-shaped run with the default three launch sets in flight: which kernel ran when
# on which queue -> tools/timeline_concurrency.py gpurun_out/c5_tl
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/c5_tl
export C5_ITERS=72 C5_LANES=3
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c5_tl -- python3 $R/tools/gpu_c5_profile.py > $R/gpurun_out/c5_tl.log 2>&1
tail -1 $R/gpurun_out/c5_tl.log
