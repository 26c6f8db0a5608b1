#!/usr/bin/env python3
"""Timing exploration: ms/iteration of config 4 for different batch sizes, whole frame and a 1/8 row-tile share."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt

s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8)
s.apply_runcuda_camera()
for world in (1, 2, 4, 8):
    for batch in (1, 2, 4, 8, 16):
        kw = dict(batch=batch)
        if world > 1:
            kw.update(tile_rows=16, tile_rank=0, tile_world=world)
        T = pt.Tracer(s, **kw)
        T.render(1, 16); T.synchronize()
        n = 96
        t0 = time.perf_counter(); T.render(17, n); T.synchronize(); dt = time.perf_counter() - t0
        st = T.stats()
        print("world %d batch %2d: %.4f ms/iter wall, %.4f ms/iter device, rays/iter %d -> %.0f Mrays/s (x%d ranks = %.0f)" % (
            world, batch, dt / n * 1e3, T.last_loop_ms() / n, sum(st["rays_per_bounce"]), sum(st["rays_per_bounce"]) / (dt / n) / 1e6,
            world, world * sum(st["rays_per_bounce"]) / (dt / n) / 1e6), flush=True)
        T.close()
