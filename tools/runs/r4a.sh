#!/bin/bash
# round 4, first GPU pass: full GPU tier, then bench.py's N-rank line rehearsed with 1 RCCL rank and with 2 gloo ranks, then the plain bench
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4a_tests.log 2>&1; tail -3 gpurun_out/r4a_tests.log
PTX_BENCH_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_rccl1.json 2> gpurun_out/r4_bench_rccl1.err; tail -c 300 gpurun_out/r4_bench_rccl1.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > gpurun_out/r4_bench_gloo2.json 2> gpurun_out/r4_bench_gloo2.err; tail -c 300 gpurun_out/r4_bench_gloo2.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; tail -c 200 gpurun_out/r4a_bench.err
