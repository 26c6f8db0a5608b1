#!/bin/bash
# 512-path tiles (eight waves per workgroup) against the product's 256: parity on the variant, then the bench command at several grids
set -o pipefail
PTX_AB_LIBRARY=$PWD/.ab/libT512.so timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_camera_tile_masks_are_supersets -k "not tile_masks and not build_resources and not veneer" > gpurun_out/r4t512_tests.log 2>&1
tail -4 gpurun_out/r4t512_tests.log
for rep in 1 2 3; do
  for cfg in "T256 0" "T512 4" "T512 7" "T512 10"; do
    set -- $cfg
    if [ "$2" = "0" ]; then unset PTX_DEBUG_TOTAL_WG_PER_CU; else export PTX_DEBUG_TOTAL_WG_PER_CU=$2; fi
    PTX_AB_LIBRARY=$PWD/.ab/lib$1.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 wg/cu $2: C4 20 steps', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['roofline']['kernels_ms_per_step'].items() if x})"
  done
done
