#!/bin/bash
# SQ counter pass over the C5-shaped run (tools/gpu_c5_profile.py, 20 448-triangle mesh, 24 iterations, one launch set at a time):
# what bounds k_mesh.  bash tools/pmc_c5.sh TAG  ->  gpurun_out/pmc_c5_TAG_a/ ; condense with tools/collect_sq_c5.py TAG
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export C5_ITERS=${C5_ITERS:-24}
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
rm -rf $R/gpurun_out/pmc_c5_${TAG}_a
echo "rocprofv3 --pmc $A (tools/pmc_c5.sh, no tracing) -- python3 tools/gpu_c5_profile.py: cornellSpaceship20k.txt 3840x2160 depth 8, AA + DoF, $C5_ITERS iterations, lanes = 1; a launch covers the iterations of one launch set at 4K (batch = 12 since round 4: $C5_ITERS iterations = whole sets of 12 and a remainder)" > $R/gpurun_out/pmc_c5_${TAG}_how.txt
rocprofv3 --pmc $A --output-format csv -d $R/gpurun_out/pmc_c5_${TAG}_a -- python3 $R/tools/gpu_c5_profile.py > $R/gpurun_out/pmc_c5_${TAG}_a.log 2>&1
echo "pass a done"
