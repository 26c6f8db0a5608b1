#!/bin/bash
# round 5: what one GPU can say about the 8-GPU run after the direct epilogue / lit flags (profiles/round4_scale_predict.txt is the comparison)
mkdir -p gpurun_out
PREDICT_ONLY=C4 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/gpu_scale_predict.py 20 5 > gpurun_out/r5h_predict.log 2>&1; tail -25 gpurun_out/r5h_predict.log
