#!/bin/bash
for rep in 1 2 3; do for v in S P; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_c5_kernels.py 2 2>/dev/null; done; done
