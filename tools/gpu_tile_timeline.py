#!/usr/bin/env python3
"""Target for `rocprofv3 --kernel-trace`: warm-up, then ONE render(20) of rank 0's tile of an 8-way split (the driver's scaling
run), bracketed by two marker launches (k_pbo of a tiny... no: by a long sleep) so that the 20-step region is easy to find."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
from mygpuraytracer_amd import multigpu
world = int(os.environ.get("TILE_WORLD", "8"))
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
kw = dict(tile_rows=multigpu.TILE_ROWS, tile_rank=0, tile_world=world) if world > 1 else {}
with pt.Tracer(s, lanes=int(os.environ.get("LANES", "0")), **kw) as T:
    t0 = time.perf_counter()
    T.render(1, 5); T.synchronize()
    while time.perf_counter() - t0 < 0.15:
        T.render(10_000, 36); T.synchronize()
    time.sleep(0.05)
    t0 = time.perf_counter(); T.render(1000, 20); T.synchronize(); print("render(20) ms", (time.perf_counter() - t0) * 1e3)
    time.sleep(0.05)
