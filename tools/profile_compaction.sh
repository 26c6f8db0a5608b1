#!/bin/bash
# Profiling pass for the scan / compaction library (k_onepass) on a 2^28-int array resident in HBM: kernel trace + stats,
# then FETCH_SIZE and WRITE_SIZE in runs of their own (MI355X_MICROARCH.md).  Output under gpurun_out/sc_*.
#   gpurun --timeout 900 -- 'bash tools/profile_compaction.sh'   then   python tools/collect_compaction_profile.py TAG
set -e
R=$GRAFT_REPO_ROOT
N=${1:-268435456}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/sc_stats $R/gpurun_out/sc_fetch $R/gpurun_out/sc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sc_stats -- python3 $R/tools/gpu_compaction_bw.py $N > $R/gpurun_out/sc_bw.json 2> $R/gpurun_out/sc_stats.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/sc_fetch -- python3 $R/tools/gpu_compaction_bw.py $N > /dev/null 2> $R/gpurun_out/sc_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/sc_write -- python3 $R/tools/gpu_compaction_bw.py $N > /dev/null 2> $R/gpurun_out/sc_write.err
echo "write pass done"
cd $R && python3 tools/gpu_compaction_bw.py | grep "^{" > gpurun_out/sc_bw_plain.json
cat gpurun_out/sc_bw_plain.json | cut -c1-220
