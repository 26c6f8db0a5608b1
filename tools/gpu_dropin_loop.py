#!/usr/bin/env python3
"""What the reference-shaped call pattern costs on C4: one ptx_iterate per call (= pathtrace(pbo, frame, iter) of the C++
veneer), alone, with the fp32 frame copied to a reused host buffer after every iteration (src/pathtrace.cu:555-556), with
the 8-bit preview as well -- against ptx_render over the same iterations in one call."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
N = 120
vp = C.c_void_p
AHEAD = os.environ.get("RENDER_AHEAD", "0") == "1"
with pt.Tracer(s) as T:
    lib, h = T.lib, T.h
    T.set_render_ahead(AHEAD)
    img = np.zeros((1920 * 1080, 3), np.float32); pbo = np.zeros((1920 * 1080, 4), np.uint8)
    T.render(1, 24); T.synchronize()
    res = {"render_ahead": AHEAD}
    t0 = time.perf_counter(); T.render(100, N); T.synchronize(); res["ptx_render_bulk"] = (time.perf_counter() - t0) / N * 1e3
    t0 = time.perf_counter()
    for i in range(N): lib.ptx_iterate(h, 200 + i); lib.ptx_synchronize(h)
    res["iterate_sync"] = (time.perf_counter() - t0) / N * 1e3
    t0 = time.perf_counter()
    for i in range(N): lib.ptx_iterate(h, 300 + i); lib.ptx_read_image(h, img.ctypes.data_as(vp))
    res["iterate_read_image"] = (time.perf_counter() - t0) / N * 1e3
    t0 = time.perf_counter()
    for i in range(N): lib.ptx_iterate(h, 400 + i); lib.ptx_write_pbo(h, 400 + i, pbo.ctypes.data_as(vp)); lib.ptx_read_image(h, img.ctypes.data_as(vp))
    res["iterate_pbo_read_image"] = (time.perf_counter() - t0) / N * 1e3
    t0 = time.perf_counter()
    for i in range(N): lib.ptx_read_image(h, img.ctypes.data_as(vp))
    res["read_image_alone"] = (time.perf_counter() - t0) / N * 1e3
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}))
