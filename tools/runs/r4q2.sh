#!/bin/bash
# pass 1 of the split bounce with a smaller LDS queue (seven workgroups' LDS per CU instead of six?)
for rep in 1 2 3; do for v in Z Q640 Q512; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_c5_leg.py 72 2>/dev/null; done; done
PTX_AB_LIBRARY=$PWD/.ab/libQ640.so timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "c5 or split or spaceship or random_scenes or mesh" 2>&1 | tail -2
