#!/bin/bash
# the N-rank bench line rehearsed on the final code: 1 RCCL rank, 2 gloo ranks on one GPU (structure only: the numbers mean nothing)
PTX_BENCH_DIST=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_rccl1.json 2> gpurun_out/r4_bench_rccl1.err; tail -c 200 gpurun_out/r4_bench_rccl1.err
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > gpurun_out/r4_bench_gloo2.json 2> gpurun_out/r4_bench_gloo2.err; tail -c 200 gpurun_out/r4_bench_gloo2.err
python - <<P
import json
for f in ("gpurun_out/r4_bench_rccl1.json","gpurun_out/r4_bench_gloo2.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["n_gpus"], d["value"], d["ms_per_step"], sorted(k for k in d if k in ("long_run","exchange_alt_ms","c5")), d["config"].get("backend"), d["config"].get("exchange"))
P
