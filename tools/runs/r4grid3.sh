#!/bin/bash
for rep in 1 2; do for wg in 0 16 24 32 48; do echo "split kernels: total wg per cu $wg (0 = the rule: 32)"; if [ $wg = 0 ]; then unset PTX_DEBUG_TOTAL_WG_PER_CU; else export PTX_DEBUG_TOTAL_WG_PER_CU=$wg; fi; timeout -k 10 300 python tools/gpu_c5_leg.py 72 2>/dev/null; done; done
unset PTX_DEBUG_TOTAL_WG_PER_CU
for mw in 2 3 4 5; do echo "k_mesh wg per cu $mw"; PTX_DEBUG_MESH_WG_PER_CU=$mw timeout -k 10 300 python tools/gpu_c5_leg.py 72 2>/dev/null; done
