#!/bin/bash
for rep in 1 2 3; do for v in Z V; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so timeout -k 10 300 python tools/gpu_tile_short.py $v 2>/dev/null | tail -1; done; done
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
