#!/usr/bin/env python3
"""Wall time per iteration for (iterations per launch set, launch sets in flight): the data behind the defaults.
    gpu_batch_lanes_sweep.py            C4 (cornellObj.txt 1920x1080)
    gpu_batch_lanes_sweep.py c5         C5 (cornellSpaceship20k.txt 3840x2160, depth of field)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
C5 = len(sys.argv) > 1 and sys.argv[1] == "c5"
if C5:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import ensure_standin_assets
    ensure_standin_assets()
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship20k.txt"), res=(3840, 2160), depth=8); s.apply_runcuda_camera()
    plans, warm, n, opt = [(12, 3), (5, 3), (8, 3), (16, 3), (12, 2), (12, 4), (8, 4), (16, 2), (24, 2), (6, 6)], 48, 96, dict(depth_of_field=1)
else:
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
    plans, warm, n, opt = [(12, 3), (8, 3), (8, 4), (6, 4), (6, 6), (16, 3), (16, 2), (24, 2), (24, 3), (4, 8), (12, 4), (16, 4), (32, 2)], 400, 240, {}
for batch, lanes in plans:
    with pt.Tracer(s, batch=batch, lanes=lanes, **opt) as T:
        T.render(1, warm); T.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); T.render(1000, n); T.synchronize(); ts.append(time.perf_counter() - t0)
    print(json.dumps(dict(scene="C5" if C5 else "C4", batch=batch, lanes=lanes, ms_per_iter_best=round(min(ts) / n * 1e3, 4), ms_per_iter_median=round(sorted(ts)[1] / n * 1e3, 4))), flush=True)
