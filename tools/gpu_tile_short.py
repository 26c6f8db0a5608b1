"""Short tile runs (the driver's scaling command: 20 steps in one call) of rank 0 of 8, of 4 and the full frame; median of 9, ms.
usage: python tools/gpu_tile_short.py [label]"""
import os, sys, time
sys.path.insert(0, ".")
import mygpuraytracer_amd as pt

s = pt.Scene("scenes/cornellObj.txt", res=(1920, 1080), depth=8); s.apply_runcuda_camera()
out = {}
for world in (8, 4, 1):
    kw = dict(tile_rows=8, tile_rank=0, tile_world=world) if world > 1 else {}
    if os.environ.get("LANES"): kw["lanes"] = int(os.environ["LANES"])
    if os.environ.get("BATCH"): kw["batch"] = int(os.environ["BATCH"])
    with pt.Tracer(s, **kw) as T:
        t0 = time.perf_counter(); T.render(1, 5); T.synchronize()
        while time.perf_counter() - t0 < 0.15: T.render(10000, 36); T.synchronize()
        ts = []
        for rep in range(9):
            t0 = time.perf_counter(); T.render(1000, 20); T.synchronize(); ts.append(time.perf_counter() - t0)
        out[world] = round(sorted(ts)[4] * 1e3, 3)
print(sys.argv[1] if len(sys.argv) > 1 else "", out)
