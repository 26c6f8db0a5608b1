#!/bin/bash
# how much does a workgroup per CU less cost now?  C4 long run with 0 / 1.5 / 5 / 9 KB more LDS per workgroup (7 / 7 / 6 / 5 workgroups per CU)
for rep in 1 2; do for x in 0 1536 5120 9216; do echo "extra LDS $x"; PTX_DEBUG_EXTRA_LDS=$x timeout -k 10 300 python tools/gpu_c4_long.py 2>/dev/null; done; done
for wg in 7 10 14 21 28; do echo "total wg per cu $wg"; PTX_DEBUG_TOTAL_WG_PER_CU=$wg timeout -k 10 300 python tools/gpu_c4_long.py 2>/dev/null; done
