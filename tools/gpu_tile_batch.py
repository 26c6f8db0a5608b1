#!/usr/bin/env python3
"""ms/step of one rank's tile of an 8-way split (tuning experiments: batch[:lanes] pairs via argv, grid via PTX_DEBUG_WG_PER_CU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
WORLD = int(os.environ.get("TILE_WORLD", "8"))
for arg in sys.argv[1:] or ["0"]:
    batch, lanes = (int(v) for v in (arg.split(":") + ["0"])[:2])
    with pt.Tracer(s, tile_rows=8, tile_rank=3 % WORLD, tile_world=WORLD, batch=batch, lanes=lanes) as T:
        T.render(1, 256); T.synchronize()
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter(); T.render(1000, 384); T.synchronize(); best = min(best, time.perf_counter() - t0)
        print("world %d WG_PER_CU=%s batch %2d lanes %d: %.4f ms/step" % (WORLD, os.environ.get("PTX_DEBUG_WG_PER_CU", "-"), batch, lanes, best / 384 * 1e3), flush=True)
