import json, os, sys, time
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mygpuraytracer_amd as pt
from conftest import ensure_standin_assets
ensure_standin_assets()
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship20k.txt"), res=(3840, 2160), depth=8); s.apply_runcuda_camera()
for rep in range(2):
    for batch, lanes in [(12, 3), (24, 2), (20, 2), (21, 2), (32, 2), (14, 3), (24, 3)]:
        with pt.Tracer(s, batch=batch, lanes=lanes, depth_of_field=1) as T:
            T.render(1, 48); T.synchronize()
            ts = []
            for r in range(3):
                t0 = time.perf_counter(); T.render(1000, 96); T.synchronize(); ts.append(time.perf_counter() - t0)
        print(json.dumps(dict(batch=batch, lanes=lanes, ms=[round(x / 96 * 1e3, 4) for x in sorted(ts)])), flush=True)
