#!/usr/bin/env python3
"""ms/step of one rank's tile of an 8-way split for several batch sizes (iterations per launch set)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
for world in (8,):
    for batch in (32, 48, 64):
        with pt.Tracer(s, tile_rows=8, tile_rank=3, tile_world=world, batch=batch) as T:
            T.render(1, 256); T.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter(); T.render(1000, 384); T.synchronize(); best = min(best, time.perf_counter() - t0)
            print("world %d batch %2d: %.4f ms/step" % (world, batch, best / 384 * 1e3), flush=True)
