import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
def timeit(T, first, n):
    t0 = time.perf_counter(); T.render(first, n); T.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for kw in ({}, dict(batch=8, lanes=2), dict(lanes=1), dict(batch=8), dict(device=0)):
    with pt.Tracer(s, **kw) as T:
        T.render(1, 200); T.synchronize()
        print(kw, timeit(T, 100, 192), timeit(T, 100, 192), flush=True)
import torch
img = torch.zeros(1920 * 1080 * 3, dtype=torch.float32, device="cuda:0")
with pt.Tracer(s, external_image_ptr=img.data_ptr(), device=0) as T:
    T.render(1, 200); T.synchronize()
    print("torch ext image", timeit(T, 100, 192), timeit(T, 100, 192), flush=True)
with pt.Tracer(s) as T:
    T.render(1, 200); T.synchronize()
    print("after torch import, plain", timeit(T, 100, 192), timeit(T, 100, 192), flush=True)
