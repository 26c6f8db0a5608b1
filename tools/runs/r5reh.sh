#!/bin/bash
# round 5: bench.py's N-rank line rehearsed on the one-GPU box -- 1 RCCL rank, then 2 gloo ranks sharing the GPU (the collective failure protocol
# of the C5 leg walks both), then the same 2 gloo ranks with a tracer allocation made to FAIL on every rank at 4K (PTX_DEBUG_MEM_BUDGET_MB does not
# fail anything: the failing phase is forced by an impossible scene path through PTX_BENCH_FAIL_C5_RANK, see bench.py)
mkdir -p gpurun_out
PTX_BENCH_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_rccl1.json 2> gpurun_out/r5_bench_rccl1.err; tail -c 300 gpurun_out/r5_bench_rccl1.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > gpurun_out/r5_bench_gloo2.json 2> gpurun_out/r5_bench_gloo2.err; tail -c 300 gpurun_out/r5_bench_gloo2.err
PTX_BENCH_FAIL_C5_RANK=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > gpurun_out/r5_bench_gloo2_fail.json 2> gpurun_out/r5_bench_gloo2_fail.err; tail -c 300 gpurun_out/r5_bench_gloo2_fail.err
python - <<P
import json
for f in ("r5_bench_rccl1", "r5_bench_gloo2", "r5_bench_gloo2_fail"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["value"]), round(d["ms_per_step"], 4), "long", d.get("long_run", {}).get("ms_per_step"), "c5", json.dumps(d.get("c5"))[:200], "exch", d.get("exchange_alt_ms", {}).get("gather"))
    except Exception as e:
        print(f, "no line:", e)
P
