#!/usr/bin/env python3
"""Experiment: do two tracers on two streams (each half of the iterations) fill each other's kernel tails?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
N = 256
def run(tracers, n_each):
    for T in tracers: T.render(1, 16)
    for T in tracers: T.synchronize()
    t0 = time.perf_counter()
    # enqueue in slices so the two streams' launches interleave in the queues
    done = 0
    while done < n_each:
        for T in tracers: T.render(100 + done, 16)
        done += 16
    for T in tracers: T.synchronize()
    return (time.perf_counter() - t0) / (n_each * len(tracers)) * 1e3
for batch in (8, 4, 2):
    for ntr in (1, 2, 3, 4):
        Ts = [pt.Tracer(s, batch=batch) for _ in range(ntr)]
        print("%d tracer(s) batch %d: %.4f ms/iter" % (ntr, batch, run(Ts, (N // ntr) // 16 * 16)), flush=True)
        for T in Ts: T.close()
