#!/bin/bash
# A/B of .ab/lib*.so variants on the short tile run (20 steps, rank 0 of 8 and of 4) and the full frame
cp mygpuraytracer_amd/libmi355x_pathtracer.so /tmp/keep.so
for rep in 1 2; do for v in "$@"; do
  cp .ab/lib$v.so mygpuraytracer_amd/libmi355x_pathtracer.so
  python - <<PY
import json, os, sys, time
sys.path.insert(0, ".")
import mygpuraytracer_amd as pt
s = pt.Scene("scenes/cornellObj.txt", res=(1920, 1080), depth=8); s.apply_runcuda_camera()
out = {}
for world in (8, 4, 1):
    kw = dict(tile_rows=8, tile_rank=0, tile_world=world) if world > 1 else {}
    with pt.Tracer(s, **kw) as T:
        t0 = time.perf_counter(); T.render(1, 5); T.synchronize()
        while time.perf_counter() - t0 < 0.15: T.render(10000, 36); T.synchronize()
        ts = []
        for rep in range(9):
            t0 = time.perf_counter(); T.render(1000, 20); T.synchronize(); ts.append(time.perf_counter() - t0)
        out[world] = round(sorted(ts)[4] * 1e3, 3)
print("$v", out)
PY
done; done
cp /tmp/keep.so mygpuraytracer_amd/libmi355x_pathtracer.so
