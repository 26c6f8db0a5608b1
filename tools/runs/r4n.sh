#!/bin/bash
# the driver's 20-step run: two launch sets (the rule) against three, and an uneven first set
for rep in 1 2 3; do for cfg in "rule" "NSETS=3" "FIRST_SET=12" "FIRST_SET=8" "FIRST_SET=6"; do
  unset PTX_DEBUG_NSETS PTX_DEBUG_FIRST_SET
  case $cfg in NSETS=3) export PTX_DEBUG_NSETS=3;; FIRST_SET=*) export PTX_DEBUG_FIRST_SET=${cfg#FIRST_SET=};; esac
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg: 20 steps', round(d['ms_per_step'],4))"
done; done
