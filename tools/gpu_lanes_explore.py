#!/usr/bin/env python3
"""Timing of C4 (full frame) for several (batch, lanes) choices on one GPU, interleaved repetitions."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
cases = [(8, 2), (8, 3), (12, 3), (16, 3), (8, 4), (12, 4), (10, 3)]
acc = {c: [] for c in cases}
for rep in range(3):
    for batch, lanes in cases:
        with pt.Tracer(s, batch=batch, lanes=lanes) as T:
            T.render(1, 192); T.synchronize()
            t0 = time.perf_counter(); T.render(1000, 384); T.synchronize(); acc[(batch, lanes)].append((time.perf_counter() - t0) / 384 * 1e3)
for c in cases:
    print("batch %2d lanes %d: %s  mean %.4f" % (c[0], c[1], " ".join("%.4f" % v for v in acc[c]), sum(acc[c]) / len(acc[c])), flush=True)
