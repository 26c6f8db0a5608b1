#!/usr/bin/env python3
"""bench.py's C5 leg alone (cornellSpaceship20k.txt 3840x2160 depth 8, AA + DoF, default options): ms per iteration for the given
numbers of timed iterations, three repetitions each.   python tools/gpu_c5_leg.py [iters ...]      (PTX_AB_LIBRARY picks a variant build)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
import bench
for iters in [int(a) for a in sys.argv[1:]] or [24, 72]:
    r = [round(bench.c5_per_iteration(pt, 0, iters)[0], 4) for _ in range(3)]
    print(json.dumps(dict(lib=os.environ.get("PTX_AB_LIBRARY", "product"), iters=iters, ms_per_iteration=r)), flush=True)
