#!/bin/bash
# timing of one library under different values of an environment variable: bash tools/ab_env.sh VAR v1 v2 ...
VAR=$1; shift
for rep in 1 2; do
for v in "$@"; do
  env $VAR=$v python bench.py --no-cpu-baseline > gpurun_out/abenv_$v.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/abenv_$v.log").read().strip().splitlines()[-1])
print("$VAR=$v", round(d["ms_per_step"],4), {k: round(x,4) for k,x in d["roofline"]["kernels_ms_per_step"].items()})
PY
done
done
