#!/usr/bin/env python3
"""C4 wall time per iteration for (iterations per launch set, launch sets in flight): the data behind the defaults."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
for batch, lanes in [(12, 3), (8, 3), (8, 4), (6, 4), (6, 6), (16, 3), (16, 2), (24, 2), (4, 6), (4, 8), (12, 4), (3, 8)]:
    with pt.Tracer(s, batch=batch, lanes=lanes) as T:
        T.render(1, 400); T.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); T.render(1000, 240); T.synchronize(); best = min(best, time.perf_counter() - t0)
    print(json.dumps(dict(batch=batch, lanes=lanes, ms_per_iter=round(best / 240 * 1e3, 4))), flush=True)
