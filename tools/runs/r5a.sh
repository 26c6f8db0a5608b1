#!/bin/bash
# round 5, first call: the GPU tier on the hygiene/advisor commit, the driver's bench command, the compaction library at frame sizes
# (DPP wave scans instead of __shfl_up: round 4 measured 13.7 us at n = 2 M)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r5a_tests.log 2>&1 || { tail -30 gpurun_out/r5a_tests.log; exit 1; }
tail -3 gpurun_out/r5a_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err || { tail -c 600 gpurun_out/r5a_bench.err; exit 1; }
python - <<P
import json
d=json.loads(open("gpurun_out/r5a_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["c5_ms_per_iteration"], d["dropin_per_call_ms"], d["long_run"], d["roofline"]["frac"], d["roofline"]["profiles"]["stale"])
P
python tools/gpu_compaction_bw.py 2073600 8294400 268435456 > gpurun_out/r5a_compaction.log 2>&1; cat gpurun_out/r5a_compaction.log
