#!/bin/bash
# rocprofv3 passes over the C5-shaped run (3840x2160, depth 8, DoF, textured BVH mesh; tools/gpu_c5_profile.py SCENE): kernel stats,
# FETCH_SIZE, WRITE_SIZE in runs of their own.  bash tools/profile_c5.sh [scene]  ->  gpurun_out/c5_{stats,fetch,write}
set -e
R=$GRAFT_REPO_ROOT
SC=${1:-cornellSpaceship20k.txt}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/c5_stats $R/gpurun_out/c5_fetch $R/gpurun_out/c5_write
ITERS=${C5_ITERS:-60}
export C5_ITERS=$ITERS
for k in fetch write; do
  echo "rocprofv3 --pmc $(echo $k | tr a-z A-Z)_SIZE (a pass of its own, no tracing) -- python3 tools/gpu_c5_profile.py $SC: $SC 3840x2160 depth 8, AA + DoF, $ITERS iterations, lanes = 1 (one launch set at a time); a launch covers the iterations of one launch set at 4K (batch = 12 since round 4: $ITERS iterations = $((ITERS / 12)) sets of 12 and a remainder of $((ITERS % 12))), pass 1 / k_mesh / pass 2 of the split bounce apart" > $R/gpurun_out/c5_${k}_how.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5_stats -- python3 $R/tools/gpu_c5_profile.py $SC > $R/gpurun_out/c5_stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/c5_fetch -- python3 $R/tools/gpu_c5_profile.py $SC > $R/gpurun_out/c5_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/c5_write -- python3 $R/tools/gpu_c5_profile.py $SC > $R/gpurun_out/c5_write.log 2>&1
echo "write done"
