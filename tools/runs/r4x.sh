#!/bin/bash
for rep in 1 2 3; do
  PTX_AB_LIBRARY=$PWD/.ab/libZ.so timeout -k 10 300 python tools/gpu_tile_short.py Z 2>/dev/null | tail -1
  PTX_DEBUG_NO_DIR_SKIP=1 PTX_AB_LIBRARY=$PWD/.ab/libZ.so timeout -k 10 300 python tools/gpu_tile_short.py Z-nomasks 2>/dev/null | tail -1
  PTX_DEBUG_NO_NORMAL_CODES=1 PTX_AB_LIBRARY=$PWD/.ab/libZ.so timeout -k 10 300 python tools/gpu_tile_short.py Z-nocodes 2>/dev/null | tail -1
  PTX_AB_LIBRARY=$PWD/.ab/libN.so timeout -k 10 300 python tools/gpu_tile_short.py N 2>/dev/null | tail -1
done
