#!/bin/bash
# end-of-round check: the GPU tier, smoke(), the driver's bench command
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4final_tests.log 2>&1; tail -3 gpurun_out/r4final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4final_bench.json 2> gpurun_out/r4final_bench.err; tail -c 300 gpurun_out/r4final_bench.err
python - <<P
import json
d=json.loads(open("gpurun_out/r4final_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["c5_ms_per_iteration"], d["dropin_per_call_ms"], d["long_run"], d["roofline"]["frac"], d["roofline"]["valu_issue"]["frac"], d["roofline"]["profiles"]["stale"], d["primary_rays_per_s"])
P
