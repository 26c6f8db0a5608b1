#!/usr/bin/env python3
"""ms/iteration and Mrays/s of every BASELINE config that runs on one GPU (C2-C5), one JSON line each."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mygpuraytracer_amd as pt
from conftest import ensure_standin_assets
ensure_standin_assets()
CASES = [
    ("C2", "cornell.txt", (800, 800), 8, dict(antialiasing=0), 512),                      # first-bounce cache applies
    ("C2-aa", "cornell.txt", (800, 800), 8, {}, 512),
    ("C3", "cornellGlass.txt", (1920, 1080), 12, {}, 192),
    ("C4", "cornellObj.txt", (1920, 1080), 8, {}, 192),
    ("C5-320", "cornellSpaceship.txt", (3840, 2160), 8, dict(depth_of_field=1), 72),
    ("C5-20k", "cornellSpaceship20k.txt", (3840, 2160), 8, dict(depth_of_field=1), 72),
]
for tag, scene, res, depth, opt, iters in CASES:
    s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=res, depth=depth); s.apply_runcuda_camera()
    with pt.Tracer(s, **opt) as T:
        # the GPU needs ~100 ms of work to reach its clocks: long warm-up, then the best of three timed runs
        T.render(1, 4 * iters); T.synchronize()
        dt = 1e9
        for rep in range(3):
            r0 = T.stats()["rays_total"]
            t0 = time.perf_counter(); T.render(1000 + rep * iters, iters); T.synchronize(); dt = min(dt, time.perf_counter() - t0)
            rays = (T.stats()["rays_total"] - r0) / iters
        print(json.dumps(dict(config=tag, scene=scene, res=res, depth=depth, opt=opt, ms_per_iter=round(1e3 * dt / iters, 4),
                              mrays_per_iter=round(rays / 1e6, 3), grays_s=round(rays / (dt / iters) / 1e9, 3),
                              contract_436B_loop_ratio=round(436 * rays / (dt / iters) / 8e12, 3))), flush=True)
