#!/bin/bash
# SQ / SQC counter passes over `bench.py --lanes 1` (kernels back to back), one rocprofv3 --pmc run per group of <= 8 counters,
# no tracing in the same run.  bash tools/pmc_sq.sh TAG   ->  gpurun_out/pmc_TAG_{a,b,c}/ ; condense with tools/collect_sq.py TAG
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# (PMC_EXTRA_ARGS="--arith 1": the same passes over another arithmetic level's code object; PMC_GROUPS="a": only that group of counters)
ARGS="--lanes 1 --steps 24 --warmup 12 --no-cpu-baseline --no-extra-legs $PMC_EXTRA_ARGS"
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
B="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
C="SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_LDS SQ_WAIT_INST_LDS"
D="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH"
echo "rocprofv3 --pmc <8 counters per pass, four passes> (tools/pmc_sq.sh, no tracing) -- python3 bench.py $ARGS: cornellObj.txt 1920x1080 depth 8 (C4), one launch set at a time; a k_bounce launch = 12 iterations of one bounce" > $R/gpurun_out/pmc_${TAG}_how.txt
for g in ${PMC_GROUPS:-a b c d}; do
  eval "CNT=\$$(echo $g | tr a-d A-D)"
  rm -rf $R/gpurun_out/pmc_${TAG}_$g
  rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_${TAG}_$g -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_$g.json 2> $R/gpurun_out/pmc_${TAG}_$g.err
  echo "pass $g done"
done
