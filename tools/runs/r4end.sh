#!/bin/bash
# end of round 4: GPU tier, smoke, profiling passes, all configs, scale prediction, the driver's bench command
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4end_tests.log 2>&1; tail -3 gpurun_out/r4end_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/runs/r4prof.sh
python tools/gpu_configs.py 2>/dev/null > gpurun_out/r4_all_configs.txt; cat gpurun_out/r4_all_configs.txt | cut -c1-160
PTX_BENCH_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/gpu_scale_predict.py 20 5 > gpurun_out/r4_scale_predict.txt 2> gpurun_out/r4_scale_predict.err; grep "^|" gpurun_out/r4_scale_predict.txt
python tools/gpu_tile_scaling.py 2>/dev/null > gpurun_out/r4_tile_scaling.txt; tail -4 gpurun_out/r4_tile_scaling.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4end_bench.json 2> gpurun_out/r4end_bench.err
python - <<P
import json
d=json.loads(open("gpurun_out/r4end_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["c5_ms_per_iteration"], d["dropin_per_call_ms"], d["long_run"], d["roofline"]["frac"], d["roofline"]["profiles"]["stale"])
P
