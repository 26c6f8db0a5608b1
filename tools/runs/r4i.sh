#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4i_tests.log 2>&1; tail -3 gpurun_out/r4i_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4i_bench.json 2> gpurun_out/r4i_bench.err; tail -c 200 gpurun_out/r4i_bench.err
python - <<P
import json
d=json.loads(open("gpurun_out/r4i_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["c5_ms_per_iteration"], d["dropin_per_call_ms"], d["long_run"])
P
