#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of the C5-shaped run for abx/lib*.so variants: bash tools/ab_c5_profile.sh A B ...
# (the variant is loaded through PTX_AB_LIBRARY, exported BEFORE rocprofv3 starts: no `env` hop between the profiler and python)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export PTX_DEV=1; export PTX_AB_LIBRARY=$R/abx/lib$v.so
  rm -rf $R/gpurun_out/c5prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5prof_$v -- python3 $R/tools/gpu_c5_profile.py > $R/gpurun_out/c5prof_$v.log 2>&1
  echo "== $v"; tail -1 $R/gpurun_out/c5prof_$v.log
  python3 - <<P
import csv,glob
f=max(glob.glob("$R/gpurun_out/c5prof_$v/**/*kernel_stats.csv",recursive=True))
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "k_bounce" in n or "k_mesh" in n or "k_gather" in n:
        short=n.replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
        print("%-40s calls %5s avg_us %9.1f total_ms %8.2f" % (short[:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
P
done
unset PTX_AB_LIBRARY
