#!/usr/bin/env python3
"""Per-kernel time split (hipEvents around every launch, one launch set at a time) of a config: scene W H [opts k=v ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mygpuraytracer_amd as pt
from conftest import ensure_standin_assets
ensure_standin_assets()
scene, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
opt = {k: int(v) for k, v in (a.split("=") for a in sys.argv[4:])}
s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=(W, H), depth=8); s.apply_runcuda_camera()
with pt.Tracer(s, **opt) as T:
    T.render(1, 64); T.synchronize()
    t0 = time.perf_counter(); T.render(100, 64); T.synchronize(); wall = (time.perf_counter() - t0) / 64 * 1e3
    T.set_kernel_timing(True)
    T.render(200, 64)
    kt = T.kernel_times()
    T.set_kernel_timing(False)
    print(json.dumps(dict(scene=scene, opt=opt, wall_ms_per_iter=round(wall, 4),
                          kernels_ms_per_iter={k: round(v[0] / 64, 4) for k, v in kt.items()},
                          launches={k: v[1] for k, v in kt.items()})))
