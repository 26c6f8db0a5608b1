#!/bin/bash
for rep in 1 2 3; do
  python tools/gpu_tile_short.py rule 2>/dev/null | tail -1
  PTX_DEBUG_WG_FIRST=7 PTX_DEBUG_WG_LATER=7 python tools/gpu_tile_short.py 7-7 2>/dev/null | tail -1
  PTX_DEBUG_WG_FIRST=6 PTX_DEBUG_WG_LATER=6 python tools/gpu_tile_short.py 6-6 2>/dev/null | tail -1
  PTX_DEBUG_WG_FIRST=8 PTX_DEBUG_WG_LATER=7 python tools/gpu_tile_short.py 8-7 2>/dev/null | tail -1
  PTX_DEBUG_WG_FIRST=7 PTX_DEBUG_WG_LATER=5 python tools/gpu_tile_short.py 7-5 2>/dev/null | tail -1
done
