#!/bin/bash
# short tile runs of one library under different values of an environment variable: bash tools/ab_env_tile.sh VAR v1 v2 ...
VAR=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $VAR=$v python tools/gpu_tile_short.py "$VAR=$v" 2>/dev/null
done; done
