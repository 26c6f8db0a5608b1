#!/usr/bin/env python3
"""The C5-shaped run alone (for rocprofv3 --kernel-trace --stats): 3840x2160, depth 8, AA + DoF, 20448-triangle stand-in,
64 iterations, one launch set at a time so that kernel durations are the kernels' own."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mygpuraytracer_amd as pt
from conftest import ensure_standin_assets
ensure_standin_assets()
SCENE = sys.argv[1] if len(sys.argv) > 1 else "cornellSpaceship20k.txt"
ITERS = int(os.environ.get("C5_ITERS", "64"))
s = pt.Scene(os.path.join(ROOT, "scenes", SCENE), res=(3840, 2160), depth=8); s.apply_runcuda_camera()
LANES = int(os.environ.get("C5_LANES", "1"))        # (3: the default plan, for the timeline; 1: kernels alone, for durations and counters)
with pt.Tracer(s, depth_of_field=1, lanes=LANES) as T:
    if LANES > 1:
        T.render(1, 36); T.synchronize()
    T.render(1000, ITERS); T.synchronize()
    print("loop ms per iteration", T.last_loop_ms() / ITERS)
