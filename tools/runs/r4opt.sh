#!/bin/bash
for rep in 1 2 3; do for v in Cur O2 OsL O3; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_c4_long.py 2>/dev/null; done; done
for rep in 1 2; do for v in Cur O2 OsL O3; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_c5_leg.py 72 2>/dev/null; done; done
