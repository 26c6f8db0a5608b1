#!/bin/bash
# one-off: the other SQ counter groups over the C5-shaped run (k_mesh's scalar / LDS / memory / branch counts)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export C5_ITERS=24
B="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"
C="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"
for g in B C; do
  eval "CNT=\$$g"
  rm -rf $R/gpurun_out/pmc_c5_more_$g
  rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_c5_more_$g -- python3 $R/tools/gpu_c5_profile.py > $R/gpurun_out/pmc_c5_more_$g.log 2>&1
  echo "pass $g done"
done
python3 - <<P
import csv, glob, collections, sys
sys.path.insert(0, "$R/tools")
from profile_meta import kernel_label
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for g in "BC":
    for f in glob.glob("$R/gpurun_out/pmc_c5_more_%s/**/*_counter_collection.csv" % g, recursive=True):
        for r in csv.DictReader(open(f)):
            lab = kernel_label(r["Kernel_Name"])
            if lab: acc[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("k_mesh", "k_finish", "k_bounce pass1", "k_bounce pass2"):
    print(k, {c: round(sum(v)/len(v)/1e6, 2) for c, v in acc[k].items()})
P
