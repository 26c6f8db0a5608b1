#!/bin/bash
for rep in 1 2 3 4; do for v in N E F G Z; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_tile_short.py $v 2>/dev/null | tail -1; done; done
