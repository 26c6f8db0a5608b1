#!/bin/bash
for rep in 1 2; do for x in 0 176 512 1024 1536 2560; do echo "extra LDS $x"; PTX_DEBUG_EXTRA_LDS=$x timeout -k 10 300 python tools/gpu_c4_long.py 2>/dev/null; done; done
