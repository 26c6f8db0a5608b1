"""Short runs (render(20) in one call, the driver's scaling command) of one rank's tile per launch plan: number of launch sets the
call is cut into (PTX_DEBUG_NSETS) x workgroups per CU of a whole launch (PTX_DEBUG_TOTAL_WG_PER_CU; 0 = the library's rule).
Median of 9, ms.   usage: python tools/gpu_tile_grid_sweep.py [world=8] [rank=0]"""
import os, sys, time, json
sys.path.insert(0, ".")
import mygpuraytracer_amd as pt

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
s = pt.Scene("scenes/cornellObj.txt", res=(1920, 1080), depth=8); s.apply_runcuda_camera()
kw = dict(tile_rows=8, tile_rank=rank, tile_world=world) if world > 1 else {}
rows = []
for nsets in (0, 1, 2, 3):
    for wg in (0, 3, 4, 5, 6, 7, 8, 10, 14):
        os.environ["PTX_DEBUG_NSETS"] = str(nsets); os.environ["PTX_DEBUG_TOTAL_WG_PER_CU"] = str(wg)
        with pt.Tracer(s, **kw) as T:
            t0 = time.perf_counter(); T.render(1, 5); T.synchronize()
            while time.perf_counter() - t0 < 0.15: T.render(10000, 36); T.synchronize()
            ts = []
            for rep in range(9):
                t0 = time.perf_counter(); T.render(1000, 20); T.synchronize(); ts.append(time.perf_counter() - t0)
        rows.append(dict(nsets=nsets, wg_per_cu=wg, ms=round(sorted(ts)[4] * 1e3, 3)))
        print(json.dumps(rows[-1]), flush=True)
best = min(rows, key=lambda r: r["ms"])
print("best", json.dumps(best))
