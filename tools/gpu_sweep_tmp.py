import json, os, sys, time
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
from mygpuraytracer_amd import multigpu
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
STEPS = 20
def run(lanes, split_min, batch=0, **kw):
    os.environ["PTX_DEBUG_SPLIT_MIN"] = str(split_min)
    with pt.Tracer(s, lanes=lanes, batch=batch, **kw) as T:
        t0 = time.perf_counter()
        T.render(1, 5); T.synchronize()
        while time.perf_counter() - t0 < 0.15:
            T.render(10_000, 36); T.synchronize()
        ts = []
        for rep in range(7):
            t0 = time.perf_counter(); T.render(1000, STEPS); T.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[0] * 1e3, ts[len(ts) // 2] * 1e3
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
for world in (8, 1):
    kw = dict(tile_rows=multigpu.TILE_ROWS, tile_rank=0, tile_world=world) if world > 1 else {}
    for lanes, split_min in ((1, 1 << 20), (2, 1 << 20), (3, 1 << 20), (4, 1 << 20), (6, 1 << 19), (8, 1<<18)):
        best, med = run(lanes, split_min, **kw)
        print(json.dumps(dict(world=world, lanes=lanes, split_min=split_min, best_ms=round(best, 3), median_ms=round(med, 3))), flush=True)
