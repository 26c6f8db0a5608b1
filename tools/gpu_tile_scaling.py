#!/usr/bin/env python3
"""Predicted strong scaling of the pixel-row tile split, measured on ONE GPU: every rank's tile of an N-way split is
traced in turn (same kernels, same batching as a real rank) and compared with the full frame.  Excludes the one RCCL
reduce per run (24.9 MB at 1080p, 99.5 MB at 4K).  Iteration sharding keeps the full-frame rate per GPU by construction.
    gpu_tile_scaling.py [C4|C5|C5-20k] [tile_rows ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mygpuraytracer_amd as pt
from mygpuraytracer_amd import multigpu
args = sys.argv[1:]
cfg = args.pop(0) if args and args[0].startswith("C") else "C4"
if cfg == "C4":
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); opt = {}; STEPS = 200
else:
    from conftest import ensure_standin_assets
    ensure_standin_assets()
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship20k.txt" if cfg == "C5-20k" else "cornellSpaceship.txt"), res=(3840, 2160), depth=8)
    opt = dict(depth_of_field=1); STEPS = 64
s.apply_runcuda_camera()
def run(**kw):
    with pt.Tracer(s, **opt, **kw) as T:
        T.render(1, STEPS // 2); T.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); T.render(1000, STEPS); T.synchronize(); best = min(best, time.perf_counter() - t0)
        return best / STEPS * 1e3
full = run()
print(json.dumps(dict(config=cfg, world=1, ms_per_step=round(full, 4))), flush=True)
ROWS = [int(a) for a in args] or [multigpu.TILE_ROWS]
for world, rows in [(w, r) for r in ROWS for w in (2, 4, 8)]:
    ms = [run(tile_rows=rows, tile_rank=r, tile_world=world) for r in range(world)]
    print(json.dumps(dict(config=cfg, world=world, tile_rows=rows, ms_per_step_by_rank=[round(m, 4) for m in ms], slowest=round(max(ms), 4),
                          predicted_speedup=round(full / max(ms), 2))), flush=True)
