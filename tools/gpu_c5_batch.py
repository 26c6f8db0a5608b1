#!/usr/bin/env python3
"""C5 (3840x2160, DoF, textured BVH mesh) wall time per iteration for several (iterations per launch set, launch sets in flight)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mygpuraytracer_amd as pt
from conftest import ensure_standin_assets
ensure_standin_assets()
scene = sys.argv[1] if len(sys.argv) > 1 else "cornellSpaceship.txt"
s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=(3840, 2160), depth=8); s.apply_runcuda_camera()
for batch, lanes in [(0, 0), (4, 3), (6, 3), (8, 3), (12, 3), (8, 2), (4, 4)]:
    with pt.Tracer(s, depth_of_field=1, batch=batch, lanes=lanes) as T:
        T.render(1, 48); T.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); T.render(1000, 48); T.synchronize(); best = min(best, time.perf_counter() - t0)
    print(json.dumps(dict(scene=scene, batch=batch, lanes=lanes, ms_per_iter=round(best / 48 * 1e3, 4))), flush=True)
