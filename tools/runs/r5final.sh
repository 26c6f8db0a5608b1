#!/bin/bash
# round 5, summary call: the GPU tier, smoke(), the driver's bench command, every single-GPU config, the arithmetic levels side by side,
# what one GPU can say about the 8-GPU run (short run with its exchange; long runs), the compaction library at frame sizes
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r5final_tests.log 2>&1; tail -3 gpurun_out/r5final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5final_bench.json 2> gpurun_out/r5final_bench.err; tail -c 300 gpurun_out/r5final_bench.err
python - <<P
import json
d=json.loads(open("gpurun_out/r5final_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["c5_ms_per_iteration"], d["dropin_per_call_ms"], d["long_run"], d["roofline"]["frac"], d["roofline"]["physical_frac_wall"], d["roofline"]["valu_issue"]["frac"], d["roofline"]["profiles"]["stale"], d["value_contracted"], d["value_fast"])
P
python tools/gpu_configs.py > gpurun_out/r5final_configs.txt 2>/dev/null; cat gpurun_out/r5final_configs.txt
python tools/gpu_arith.py --errors > gpurun_out/r5final_arith.txt 2>/dev/null; grep config gpurun_out/r5final_arith.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/gpu_scale_predict.py 20 5 > gpurun_out/r5final_predict.txt 2>&1; grep "^|" gpurun_out/r5final_predict.txt
python tools/gpu_tile_scaling.py C4 > gpurun_out/r5final_tile_scaling.txt 2>/dev/null; cat gpurun_out/r5final_tile_scaling.txt
python tools/gpu_compaction_bw.py 2073600 8294400 268435456 > gpurun_out/r5final_compaction.txt 2>/dev/null; cat gpurun_out/r5final_compaction.txt
