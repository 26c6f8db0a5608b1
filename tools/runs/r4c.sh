#!/bin/bash
# kernel trace + SQ counters of the C5-shaped run with the current library (k_mesh / k_finish apart)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export C5_ITERS=24
rm -rf $R/gpurun_out/r4c_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4c_stats -- python3 $R/tools/gpu_c5_profile.py > $R/gpurun_out/r4c_stats.log 2>&1
python3 - <<P
import csv,glob
f=max(glob.glob("$R/gpurun_out/r4c_stats/**/*kernel_stats.csv",recursive=True))
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "k_bounce" in n or "k_mesh" in n or "k_gather" in n or "k_finish" in n:
        short=n.replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        print("%-40s calls %5s avg_us %9.1f total_ms %8.2f" % (short[:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
P
cd $R && bash tools/pmc_c5.sh r4c && python3 tools/collect_sq_c5.py r4c
