#!/usr/bin/env python3
"""The C5-shaped run's kernels alone (tracer's own event pairs, one launch set at a time): ms per iteration by kernel kind, REPS times.
usage: [PTX_AB_LIBRARY=...] python tools/gpu_c5_kernels.py [reps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mygpuraytracer_amd as pt
from conftest import ensure_standin_assets
ensure_standin_assets()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship20k.txt"), res=(3840, 2160), depth=8); s.apply_runcuda_camera()
with pt.Tracer(s, depth_of_field=1) as T:
    T.render(1, 24); T.synchronize()
    T.set_kernel_timing(True)
    for r in range(reps):
        T.kernel_times()
        T.render(1000 + 100 * r, 24); T.synchronize()
        kt = T.kernel_times()
        print(json.dumps({"lib": os.path.basename(os.environ.get("PTX_AB_LIBRARY", "product")), **{k: round(v[0] / 24, 4) for k, v in kt.items()}}), flush=True)
