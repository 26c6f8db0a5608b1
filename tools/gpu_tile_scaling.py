#!/usr/bin/env python3
"""Predicted strong scaling of the pixel-row tile split, measured on ONE GPU: every rank's tile of an N-way split is
traced in turn (same kernels, same batching as a real rank) and compared with the full frame.  Excludes the one RCCL
reduce per run (24.9 MB at 1080p).  Iteration sharding keeps the full-frame rate per GPU by construction."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
from mygpuraytracer_amd import multigpu
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
STEPS = 200
def run(**kw):
    with pt.Tracer(s, **kw) as T:
        T.render(1, 100); T.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); T.render(1000, STEPS); T.synchronize(); best = min(best, time.perf_counter() - t0)
        return best / STEPS * 1e3
full = run()
print(json.dumps(dict(world=1, ms_per_step=round(full, 4))), flush=True)
ROWS = [int(a) for a in sys.argv[1:]] or [multigpu.TILE_ROWS]
for world, rows in [(w, r) for r in ROWS for w in (2, 4, 8)]:
    ms = [run(tile_rows=rows, tile_rank=r, tile_world=world) for r in range(world)]
    print(json.dumps(dict(world=world, tile_rows=rows, ms_per_step_by_rank=[round(m, 4) for m in ms], slowest=round(max(ms), 4),
                          predicted_speedup=round(full / max(ms), 2))), flush=True)
