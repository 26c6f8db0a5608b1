#!/usr/bin/env python3
"""Host time to enqueue the launch sets of a ptx_render call against the GPU time they take (C4 full frame and one tile of an
8-way split): what a hipGraph of the bounce loop could save at most is the enqueue time that is not already hidden."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
from mygpuraytracer_amd import multigpu
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
for label, kw, steps in (("full frame", {}, 36), ("tile 1/8", dict(tile_rows=multigpu.TILE_ROWS, tile_rank=0, tile_world=8), 96)):
    with pt.Tracer(s, **kw) as T:
        T.render(1, steps); T.synchronize()
        enq, tot = [], []
        for rep in range(5):
            t0 = time.perf_counter(); T.render(1000, steps); t1 = time.perf_counter(); T.synchronize(); t2 = time.perf_counter()
            enq.append(t1 - t0); tot.append(t2 - t0)
        print(json.dumps(dict(frame=label, iterations=steps, enqueue_ms=round(min(enq) * 1e3, 3), total_ms=round(min(tot) * 1e3, 3))), flush=True)
