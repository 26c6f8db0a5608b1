#!/bin/bash
# SQ_INSTS_VALU / lane utilisation of the product library's C4 kernels (one counter pass)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_q_a
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmc_q_a -- python3 $R/bench.py --lanes 1 --steps 24 --warmup 12 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/pmc_q_a.json 2> $R/gpurun_out/pmc_q_a.err
cd $R && python3 - <<P
import csv,glob,collections,sys
sys.path.insert(0,"tools")
from profile_meta import kernel_label
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_q_a/**/*_counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        lab=kernel_label(r["Kernel_Name"])
        if lab: acc[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    m={c:sum(v)/len(v) for c,v in cs.items()}
    print(k, "INSTS_VALU %.4e"%m["SQ_INSTS_VALU"], "lanes %.3f"%(m["SQ_THREAD_CYCLES_VALU"]/(64*m["SQ_ACTIVE_INST_VALU"])), "launches", len(cs["SQ_INSTS_VALU"]))
P
