#!/usr/bin/env python3
"""ms/iteration of small frames (launch-bound regime): C1 and tiny Cornell frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
for scene, res, depth in (("sphere.txt", (256, 256), 4), ("cornellObj.txt", (256, 256), 8), ("cornellObj.txt", (64, 64), 8), ("cornellObj.txt", (640, 360), 8)):
    s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=res, depth=depth); s.apply_runcuda_camera()
    with pt.Tracer(s) as T:
        T.render(1, 512); T.synchronize()
        best = 1e9
        for rep in range(3):
            r0 = T.stats()["rays_total"]
            t0 = time.perf_counter(); T.render(1000, 1024); T.synchronize(); dt = time.perf_counter() - t0
            best = min(best, dt)
            rays = (T.stats()["rays_total"] - r0) / 1024
        print("%s %dx%d d%d: %.4f ms/iter, %.2f Mrays/iter, %.2f Grays/s" % (scene, res[0], res[1], depth, best / 1024 * 1e3, rays / 1e6, rays / (best / 1024) / 1e9), flush=True)
