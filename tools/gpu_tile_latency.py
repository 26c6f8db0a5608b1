#!/usr/bin/env python3
"""Diagnostic (library built with make EXTRA=-DPT_WGCLOCK; the per-phase cycle sums need -DPT_STAMPS as well and perturb short kernels): mean wall-clock latency of a k_bounce workgroup's phases -- prologue (scene
tables to LDS + run search), tile loop, tail (local move) -- on one rank's tile of an N-way split, render(20) calls.
usage: python tools/gpu_tile_latency.py [world=8]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mygpuraytracer_amd as pt
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
kw = dict(tile_rows=8, tile_rank=0, tile_world=world) if world > 1 else {}
L = pt.load_library()
L.ptx_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros(48, np.uint64)
with pt.Tracer(s, **kw) as T:
    T.render(1, 40); T.synchronize(); L.ptx_debug_read_stamps(T.h, out.ctypes.data_as(C.c_void_p))
    for rep in range(10): T.render(1000, 20); T.synchronize()
    L.ptx_debug_read_stamps(T.h, out.ctypes.data_as(C.c_void_p))
names = {0: "load+shade/gen", 1: "isect-rest", 2: "classify+deposit", 3: "ranking", 4: "sort+write", 5: "cull+list", 6: "items", 7: "decode", 11: "load-wait", 12: "rank-ballots", 13: "rank-wait1", 14: "rank-counts"}
for base, tag, wb in ((0, "k_bounce<first>", 32), (16, "k_bounce", 40)):
    tiles = max(float(out[wb + 4]), 1.0)
    print(tag, "s_memtime ticks per wave and tile:", " ".join("%s %.0f" % (names[k], float(out[base + k]) / tiles / 4) for k in list(range(8)) + [11, 12, 13, 14]))
for base, tag in ((32, "k_bounce<first>"), (40, "k_bounce")):
    n = max(float(out[base + 3]), 1.0)
    print("%-16s workgroups %8d  tiles/wg %.2f  prologue %.2f us  tile loop %.2f us (%.2f per tile)  tail %.2f us" % (
        tag, n, float(out[base + 4]) / n, float(out[base]) / n / 100.0, float(out[base + 1]) / n / 100.0,
        float(out[base + 1]) / max(float(out[base + 4]), 1.0) / 100.0, float(out[base + 2]) / n / 100.0))
