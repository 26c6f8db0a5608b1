#!/bin/bash
# the specialised kernel at 7 waves / 64-run window / 17 rows (W7), at 7 waves only (W7b), at 8 waves with the 64-run window (W8w64) against the product (Cur)
for rep in 1 2 3; do for v in Cur W7 W7b W8w64; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_tile_short.py $v 2>/dev/null | tail -1; done; done
for rep in 1 2 3; do for v in Cur W7 W7b W8w64; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_c4_long.py 2>/dev/null; done; done
