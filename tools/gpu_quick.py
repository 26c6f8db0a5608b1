#!/usr/bin/env python3
"""Quick GPU bring-up check (not a test): HIP path vs the CPU oracle on a few cases, then a timing run."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mygpuraytracer_amd as pt  # noqa: E402
from cpulibs import OracleLib  # noqa: E402


def beq(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def check_scene(name, res, depth, iters, **opt):
    s = pt.Scene(os.path.join(ROOT, "scenes", name), res=res, depth=depth)
    s.apply_runcuda_camera()
    d = s.dump()
    O = OracleLib()
    O.set_libm(1)
    O.create(d, d["textures"])
    O.set_options(aa=opt.get("antialiasing", 1), dof=opt.get("depth_of_field", 0), sort=opt.get("sort_by_material", 1),
                  cache=opt.get("cache_first_bounce", 1))
    O.pt_init()
    T = pt.Tracer(s, **opt)
    ok = True
    # stage parity
    O.pt_generate(1)
    gp = T.generate(1)
    op = O.paths()
    e = beq(gp, op)
    print("  generate parity:", e)
    ok &= e
    oi = O.compute_intersections(op)
    gi = T.compute_intersections(op)
    hit = oi["t"] > 0
    e = beq(gi["t"], oi["t"]) and beq(gi["normal"][hit], oi["normal"][hit]) and beq(gi["materialId"], oi["materialId"]) \
        and beq(gi["geomId"][hit], oi["geomId"][hit])
    print("  intersect parity:", e, "hits", int(hit.sum()), "of", len(hit))
    ok &= e
    idx = np.arange(len(op), dtype=np.int32)
    osd = O.shade(1, 1, idx, oi, op)
    gsd = T.shade(1, idx, oi, op)
    e = beq(osd, gsd)
    print("  shade parity:", e)
    if not e:
        bad = np.nonzero([not beq(osd[k:k + 1], gsd[k:k + 1]) for k in range(len(osd))])[0]
        print("   first bad", bad[:5], osd[bad[:2]], gsd[bad[:2]], oi[bad[:2]])
    ok &= e
    for it in range(1, iters + 1):
        O.iterate(it)
        T.pathtrace(it)
    gi_img = T.read_image()
    oi_img = O.image()
    nd = int((gi_img != oi_img).any(axis=1).sum())
    st = T.stats()
    lc = O.live_counts()
    print("  image: differing pixels", nd, "of", len(gi_img), "max abs", float(np.abs(gi_img - oi_img).max()),
          "mean", float(gi_img.mean()), float(oi_img.mean()))
    print("  rays/bounce gpu", st["rays_per_bounce"], "oracle", lc.tolist())
    ok &= nd == 0
    T.close()
    return ok


def main():
    allok = True
    for name, res, depth, iters, opt in [
        ("sphere.txt", (64, 64), 4, 2, {}),
        ("cornell.txt", (64, 64), 8, 3, dict(antialiasing=0)),
        ("cornell.txt", (64, 64), 8, 2, dict(depth_of_field=1)),
        ("cornellGlass.txt", (96, 54), 12, 3, {}),
        ("cornellObj.txt", (96, 54), 8, 3, {}),
        ("cornellObj.txt", (96, 54), 8, 2, dict(sort_by_material=0)),
    ]:
        print(name, res, depth, opt)
        allok &= check_scene(name, res, depth, iters, **opt)
    if "--full" in sys.argv:
        print("cornellObj 1920x1080 depth 8, 1 iteration vs oracle")
        allok &= check_scene("cornellObj.txt", (1920, 1080), 8, 1)
    # timing
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8)
    s.apply_runcuda_camera()
    T = pt.Tracer(s)
    T.render(1, 3)
    T.synchronize()
    t0 = time.time()
    n = 20
    T.render(4, n)
    T.synchronize()
    dt = time.time() - t0
    st = T.stats()
    rays = sum(st["rays_per_bounce"])
    print("C4 timing: %.3f ms/iter wall, %.3f ms/iter device, %.1f Mrays/s, rays/iter %d" % (
        dt / n * 1e3, T.last_loop_ms() / n, rays / (T.last_loop_ms() / n * 1e-3) / 1e6, rays))
    print("ALL OK" if allok else "MISMATCH")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
