#!/usr/bin/env python3
"""Diagnostic (needs the library built with make EXTRA=-DPT_STAMPS): share of k_bounce's wave cycles per phase."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mygpuraytracer_amd as pt
scene = sys.argv[1] if len(sys.argv) > 1 else "cornellObj.txt"
res = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ensure_standin_assets
ensure_standin_assets()
s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=res, depth=8); s.apply_runcuda_camera()
WORLD = int(os.environ.get("TILE_WORLD", "1"))          # > 1: rank 3's tile of such a split, in 20-step calls (the latency-bound case)
opt = dict(tile_rows=8, tile_rank=3 % WORLD, tile_world=WORLD) if WORLD > 1 else dict(lanes=1)
T = pt.Tracer(s, depth_of_field=1 if "Spaceship" in scene else 0, **opt)
L = pt.load_library()
L.ptx_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros(48, np.uint64)
T.render(1, 16); L.ptx_debug_read_stamps(T.h, out.ctypes.data_as(C.c_void_p))
if WORLD > 1:
    tot = np.zeros(48, np.uint64)
    for rep in range(20):
        T.render(100 + 20 * rep, 20); T.synchronize(); L.ptx_debug_read_stamps(T.h, out.ctypes.data_as(C.c_void_p)); tot += out
    out = tot
else:
    T.render(17, 64); L.ptx_debug_read_stamps(T.h, out.ctypes.data_as(C.c_void_p))
names = {0: "load+shade/gen", 1: "isect-rest", 2: "classify+deposit", 3: "ranking", 4: "sort+write", 5: "cull+list", 6: "items", 7: "decode", 11: "load-wait", 12: "rank-ballots", 13: "rank-wait1", 14: "rank-counts"}
for base, tag in ((0, "k_bounce<first>"), (16, "k_bounce")):
    tot = float(out[base:base + 8].sum() + out[base + 11:base + 15].sum())
    print(tag, " ".join("%s %.1f%%" % (names[k], 100 * float(out[base + k]) / max(tot, 1)) for k in list(range(8)) + [11, 12, 13, 14]), "total cycles %.3g" % tot)
    passes = max(float(out[base + 10]), 1)
    print("   per tile-pass: prim items %.1f mesh items %.1f (passes %d)" % (float(out[base + 8]) / passes, float(out[base + 9]) / passes, passes))
