#!/usr/bin/env python3
"""Wall time of the driver's short scaling run (`bench.py --steps 20 --warmup 5`) on the full C4 frame and on one rank's tile of a
2/4/8-way row split, per (launch sets in flight, smallest launch set): the data behind the defaults for short runs.
    gpu_short_tile_sweep.py [steps] [worlds ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
from mygpuraytracer_amd import multigpu
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
WORLDS = [int(a) for a in sys.argv[2:]] or [1, 8, 4, 2]
def run(lanes, split_min, **kw):
    os.environ["PTX_DEBUG_SPLIT_MIN"] = str(split_min)
    with pt.Tracer(s, lanes=lanes, **kw) as T:
        t0 = time.perf_counter()
        T.render(1, 5); T.synchronize()
        while time.perf_counter() - t0 < 0.15:
            T.render(10_000, 36); T.synchronize()
        ts = []
        for rep in range(9):
            t0 = time.perf_counter(); T.render(1000, STEPS); T.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[0] * 1e3, ts[len(ts) // 2] * 1e3
for world in WORLDS:
    kw = dict(tile_rows=multigpu.TILE_ROWS, tile_rank=0, tile_world=world) if world > 1 else {}
    for lanes, split_min in ((1, 1 << 20), (2, 1 << 20), (3, 1 << 20), (1, 1 << 20), (2, 1 << 20), (3, 1 << 20)):
        best, med = run(lanes, split_min, **kw)
        print(json.dumps(dict(world=world, steps=STEPS, lanes=lanes, split_min=split_min, best_ms=round(best, 3), median_ms=round(med, 3))), flush=True)
