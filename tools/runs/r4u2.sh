#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4u2_tests.log 2>&1 || { tail -40 gpurun_out/r4u2_tests.log; exit 1; }
tail -2 gpurun_out/r4u2_tests.log
for rep in 1 2 3 4; do for v in V U; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_c4_long.py 2>/dev/null; done; done
for rep in 1 2 3; do for v in V U; do
  PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v C4 20 steps', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['kernels_ms_per_step'].items() if x})"
done; done
