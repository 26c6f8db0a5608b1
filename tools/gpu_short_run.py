#!/usr/bin/env python3
"""Does a short timed region (what bench.py --gpus 8 gives every rank) run at the same rate as a long one?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
for warm, steps in ((20, 200), (20, 64), (400, 200), (2000, 200)):
    with pt.Tracer(s, tile_rows=8, tile_rank=3, tile_world=8) as T:
        time.sleep(0.5)
        T.render(1, warm); T.synchronize()
        t0 = time.perf_counter(); T.render(warm + 1, steps); T.synchronize(); dt = time.perf_counter() - t0
        print("tile 1/8: warmup %4d, %3d timed steps: %.4f ms/step (%.1f ms region)" % (warm, steps, dt / steps * 1e3, dt * 1e3), flush=True)
for warm, steps in ((20, 200), (400, 200)):
    with pt.Tracer(s) as T:
        time.sleep(0.5)
        T.render(1, warm); T.synchronize()
        t0 = time.perf_counter(); T.render(warm + 1, steps); T.synchronize(); dt = time.perf_counter() - t0
        print("full frame: warmup %4d, %3d timed steps: %.4f ms/step" % (warm, steps, dt / steps * 1e3), flush=True)
