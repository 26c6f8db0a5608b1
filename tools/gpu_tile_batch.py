#!/usr/bin/env python3
"""ms/step of one rank's tile of an 8-way split (tuning experiments: batch via argv, grid via PTX_DEBUG_WG_PER_CU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
batches = [int(a) for a in sys.argv[1:]] or [0]
for batch in batches:
    with pt.Tracer(s, tile_rows=8, tile_rank=3, tile_world=8, batch=batch) as T:
        T.render(1, 256); T.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); T.render(1000, 384); T.synchronize(); best = min(best, time.perf_counter() - t0)
        print("WG_PER_CU=%s batch %2d: %.4f ms/step" % (os.environ.get("PTX_DEBUG_WG_PER_CU", "-"), batch, best / 384 * 1e3), flush=True)
