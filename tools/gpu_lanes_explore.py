#!/usr/bin/env python3
"""Timing of C4 (full frame and one rank's tile of an 8-way split) for several (batch, lanes) choices on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
for tag, kw in (("full frame", {}), ("tile 1/8", dict(tile_rows=8, tile_rank=3, tile_world=8))):
    for batch, lanes in ((0, 1), (0, 2), (0, 3), (0, 4)):
        with pt.Tracer(s, batch=batch, lanes=lanes, **kw) as T:
            T.render(1, 256); T.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter(); T.render(1000, 384); T.synchronize(); best = min(best, (time.perf_counter() - t0) / 384 * 1e3)
            print("%-10s lanes %d: %.4f ms/iter" % (tag, lanes, best), flush=True)
