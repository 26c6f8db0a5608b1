#!/usr/bin/env python3
"""How a short run (few iterations per ptx_render call) is cut into launch sets: wall time of render(steps) for the full
C4 frame and for one rank's tile of an 8-way split, per value of PTX_DEBUG_SPLIT_MIN (smallest launch set, in primary
rays; a huge value = never cut a run shorter than lanes x kmax, the behaviour before the even split)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
from mygpuraytracer_amd import multigpu
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
STEPS = [int(a) for a in sys.argv[1:]] or [4, 8, 12, 20, 36, 64, 100]
for label, kw in (("full", {}), ("tile 1/8", dict(tile_rows=multigpu.TILE_ROWS, tile_rank=0, tile_world=8))):
    for split_min in (1 << 40, 8 << 20, 4 << 20, 2 << 20, 1 << 20, 1 << 18):
        os.environ["PTX_DEBUG_SPLIT_MIN"] = str(split_min)
        with pt.Tracer(s, **kw) as T:
            T.render(1, 64); T.synchronize()
            row = {}
            for steps in STEPS:
                best = 1e9
                for rep in range(5):
                    t0 = time.perf_counter(); T.render(1000, steps); T.synchronize(); best = min(best, time.perf_counter() - t0)
                row[steps] = round(best * 1e3, 3)
        print(json.dumps(dict(frame=label, split_min=split_min, ms_total_by_steps=row)), flush=True)
