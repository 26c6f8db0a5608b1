"""C4 long run (240 steps per call, three launch sets in flight), median of 5 calls, ms per step -- for A/B of library variants (PTX_AB_LIBRARY)."""
import json, os, sys, time
sys.path.insert(0, ".")
import mygpuraytracer_amd as pt
s = pt.Scene("scenes/cornellObj.txt", res=(1920, 1080), depth=8); s.apply_runcuda_camera()
with pt.Tracer(s) as T:
    T.render(1, 36); T.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); T.render(1000, 240); T.synchronize(); ts.append((time.perf_counter() - t0) * 1e3 / 240)
    print(json.dumps({"lib": os.path.basename(os.environ.get("PTX_AB_LIBRARY", "product")), "ms_per_step_long_run": [round(x, 4) for x in sorted(ts)]}))
