#!/bin/bash
# wall time of bench.py per number of launch sets in flight: bash tools/ab_lanes.sh 2 3 4
for rep in 1 2; do for l in "$@"; do
  python bench.py --lanes $l --no-cpu-baseline > gpurun_out/abl.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/abl.log").read().strip().splitlines()[-1])
print("lanes=$l", round(d["ms_per_step"],4))
PY
done; done
