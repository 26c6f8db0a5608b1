import sys, time
sys.path.insert(0, ".")
import mygpuraytracer_amd as pt
s = pt.Scene("scenes/cornellObj.txt", res=(1920, 1080), depth=8); s.apply_runcuda_camera()
for apps in (0, 1):
    with pt.Tracer(s, apps_variant=apps) as T:
        t0 = time.perf_counter(); T.render(1, 36); T.synchronize()
        while time.perf_counter() - t0 < 0.3: T.render(100, 72); T.synchronize()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter(); T.render(1000, 360); T.synchronize(); ts.append((time.perf_counter() - t0) / 360)
        print("apps_variant", apps, "ms/iter", round(sorted(ts)[2] * 1e3, 4))
