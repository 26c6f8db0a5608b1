#!/usr/bin/env python3
"""ms/iteration of frames beyond 4K (one iteration per launch set), with and without launch sets in flight on three streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
for res in ((5120, 2880), (7680, 4320)):
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=res, depth=8); s.apply_runcuda_camera()
    for kw in (dict(), dict(lanes=1), dict(lanes=2), dict(lanes=3)):
        with pt.Tracer(s, **kw) as T:
            T.render(1, 12); T.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter(); T.render(100, 24); T.synchronize(); best = min(best, time.perf_counter() - t0)
            print(res, kw, "ms/iter", round(best / 24 * 1e3, 3), flush=True)
