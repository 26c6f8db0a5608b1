#!/usr/bin/env python3
"""Timing of C4 for several (batch, lanes) choices on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
for batch, lanes in ((8, 1), (8, 2), (4, 2), (6, 2), (12, 2), (16, 2), (16, 1)):
    with pt.Tracer(s, batch=batch, lanes=lanes) as T:
        T.render(1, 48); T.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); T.render(100, 192); T.synchronize(); best = min(best, (time.perf_counter() - t0) / 192 * 1e3)
        print("batch %2d lanes %d: %.4f ms/iter" % (batch, lanes, best), flush=True)
