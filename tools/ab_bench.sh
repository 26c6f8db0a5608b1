#!/bin/bash
# A/B timing of prebuilt library variants (abx/lib*.so) on one GPU box: bash tools/ab_bench.sh A B C ...
# (the variant is loaded through PTX_AB_LIBRARY: the in-tree product library is never overwritten)
for rep in $(seq 1 ${AB_REPS:-2}); do
for v in "$@"; do
  PTX_DEV=1 PTX_AB_LIBRARY=$PWD/abx/lib$v.so python bench.py --no-cpu-baseline ${AB_BENCH_ARGS:-} > gpurun_out/ab_$v.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v.log").read().strip().splitlines()[-1])
print("$v", round(d["ms_per_step"],4), {k: round(x,4) for k,x in d["roofline"]["kernels_ms_per_step"].items()}, "c5", d.get("c5_ms_per_iteration"))
PY
done
done
