#!/bin/bash
# round 5: GPU tier with the arithmetic levels in, the driver's bench command with its new legs (arith_levels, physical_frac_wall)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r5c_tests.log 2>&1 || { tail -40 gpurun_out/r5c_tests.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r5c_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5c_bench.json 2> gpurun_out/r5c_bench.err || { tail -c 800 gpurun_out/r5c_bench.err; exit 1; }
python - <<P
import json
d=json.loads(open("gpurun_out/r5c_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["c5_ms_per_iteration"], d["roofline"]["frac"], d["roofline"]["physical_frac_wall"], d["roofline"]["profiles"]["stale"])
print(json.dumps(d["arith_levels"])[:1500])
P
