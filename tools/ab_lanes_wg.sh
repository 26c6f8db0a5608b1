#!/bin/bash
# wall time of bench.py over (launch sets in flight) x (workgroups per CU): bash tools/ab_lanes_wg.sh "1 2 3" "5 10 20"
for rep in 1 2; do for l in $1; do for w in $2; do
  env PTX_DEBUG_WG_PER_CU=$w python bench.py --lanes $l --no-cpu-baseline > gpurun_out/ablw.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ablw.log").read().strip().splitlines()[-1])
print("lanes=$l wg/cu=$w", round(d["ms_per_step"],4), {k: round(x,4) for k,x in d["roofline"]["kernels_ms_per_step"].items()})
PY
done; done; done
