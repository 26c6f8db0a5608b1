#!/usr/bin/env python3
"""Timing of the textured-mesh configuration (BASELINE configs[4] / C5) on one GPU: 3840x2160, depth 8, AA + DoF, with
the 320-triangle and the 20448-triangle stand-in mesh, BVH on and off.  Prints one JSON line per case."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mygpuraytracer_amd as pt
from conftest import ensure_standin_assets
ensure_standin_assets()
res = (3840, 2160)
cases = [("cornellSpaceship.txt", {}), ("cornellSpaceship.txt", dict(no_bvh=1)), ("cornellSpaceship20k.txt", {}),
         ("cornellSpaceship20k.txt", dict(no_bvh=1))]
if len(sys.argv) > 1 and sys.argv[1] == "--quick":
    cases = cases[:3]
for scene, opt in cases:
    s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=res, depth=8); s.apply_runcuda_camera()
    slow = scene.endswith("20k.txt") and opt.get("no_bvh")
    iters = 2 if slow else 16
    with pt.Tracer(s, depth_of_field=1, **opt) as T:
        T.render(1, 2 if slow else 8); T.synchronize()
        t0 = time.time(); T.render(100, iters); T.synchronize(); dt = time.time() - t0
        st = T.stats()
        img = T.read_image()
        rays_iter = st["rays_total"] / max(st["iterations"], 1)
        print(json.dumps(dict(scene=scene, opt=opt, res=res, ms_per_iter=1e3 * dt / iters, iters=iters, rays_per_iter=rays_iter,
                              mrays_s=rays_iter / (dt / iters) / 1e6, checksum=float(np.float64(img).sum()))), flush=True)
