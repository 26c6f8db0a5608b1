#!/usr/bin/env python3
"""The arithmetic levels (ptx_options.arith 0 exact / 1 contracted / 2 fast) side by side on one box: wall time per step of C4 (200 steps,
three launch sets in flight) and per iteration of C5 (72 after 36), and the per-stage differences against the exact level on identical
inputs (what tests/test_gpu_arith.py bounds).  python tools/gpu_arith.py [--no-c5] [--errors]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mygpuraytracer_amd as pt
from conftest import ensure_standin_assets

def timed(T, first, count, reps=3):
    ts = []
    for r in range(reps):
        T.synchronize(); a = time.perf_counter(); T.render(first + r * count, count); T.synchronize(); ts.append((time.perf_counter() - a) / count * 1e3)
    return sorted(ts)

def main():
    ensure_standin_assets()
    levels = (0, 1, 2) if "--no-timing" not in sys.argv else ()
    if "--levels" in sys.argv: levels = tuple(int(x) for x in sys.argv[sys.argv.index("--levels") + 1].split(","))
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8); s.apply_runcuda_camera()
    for lv in levels:
        with pt.Tracer(s, arith=lv) as T:
            T.render(1, 600); T.synchronize()
            t = timed(T, 1000, 200)
            st = T.stats()
            T.set_kernel_timing(True); T.render(5000, 24); kt = T.kernel_times(); T.set_kernel_timing(False)
            print(json.dumps(dict(config="C4", arith=lv, ms_per_step=[round(x, 4) for x in t], kernels_ms_per_step={k: round(v[0] / 24, 4) for k, v in kt.items() if v[1]})), flush=True)
    if "--no-c5" not in sys.argv:
        s5 = pt.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship20k.txt"), res=(3840, 2160), depth=8); s5.apply_runcuda_camera()
        for lv in levels:
            with pt.Tracer(s5, arith=lv, depth_of_field=1) as T:
                T.render(1, 36); T.synchronize()
                t = timed(T, 100, 72, reps=2)
                T.set_kernel_timing(True); T.render(5000, 24); kt = T.kernel_times(); T.set_kernel_timing(False)
                print(json.dumps(dict(config="C5", arith=lv, ms_per_iteration=[round(x, 4) for x in t], fenced=T.stats()["fenced"],
                                      kernels_ms_per_iteration={k: round(v[0] / 24, 4) for k, v in kt.items() if v[1]})), flush=True)
    if "--errors" in sys.argv:
        ULP = float(np.finfo(np.float32).eps)
        for scene, res, depth, opt in (("cornellObj.txt", (96, 54), 8, {}), ("cornellGlass.txt", (96, 54), 12, dict(depth_of_field=1)), ("cornellSpaceship.txt", (96, 54), 8, {})):
            sc = pt.Scene(os.path.join(ROOT, "scenes", scene), res=res, depth=depth); sc.apply_runcuda_camera()
            T0 = pt.Tracer(sc, **opt)
            p0 = T0.generate(3); i0 = T0.compute_intersections(p0)
            idx = np.random.default_rng(9).integers(0, 4_000_000, len(p0)).astype(np.int32)
            s0 = T0.shade(3, idx, i0, p0)
            k0 = T0.tile_intersect(p0) if scene != "cornellSpaceship.txt" else None
            for lv in (1, 2):
                T1 = pt.Tracer(sc, arith=lv, **opt)
                p1 = T1.generate(3); i1 = T1.compute_intersections(p0); s1 = T1.shade(3, idx, i0, p0)
                same = ((i0["t"] > 0) == (i1["t"] > 0)) & np.where(i0["t"] > 0, i0["geomId"] == i1["geomId"], True)
                both = same & (i0["t"] > 0)
                live = (s0["remainingBounces"] > 0) & (s1["remainingBounces"] > 0)
                ddir = np.abs(s0["direction"][live] - s1["direction"][live]).max(axis=1)
                okd = ddir <= 1e-4
                col = np.abs(s0["color"][live][okd] - s1["color"][live][okd]) / np.maximum(np.abs(s0["color"][live][okd]), 1e-3)
                r = dict(scene=scene, arith=lv, gen_dir_ulp=float(np.abs(p0["direction"] - p1["direction"]).max() / ULP),
                         gen_origin_abs=float(np.abs(p0["origin"] - p1["origin"]).max()),
                         decisions_differ=float(1 - same.mean()), t_rel_ulp=float((np.abs(i0["t"][both] - i1["t"][both]) / np.abs(i0["t"][both])).max() / ULP),
                         t_rel_ulp_p99_p999=[float(np.quantile(np.abs(i0["t"][both] - i1["t"][both]) / np.abs(i0["t"][both]), q) / ULP) for q in (0.99, 0.999)],
                         normal_ulp=float(np.abs(i0["normal"][both] - i1["normal"][both]).max() / ULP),
                         normal_ulp_p99_p999=[float(np.quantile(np.abs(i0["normal"][both] - i1["normal"][both]).max(axis=1), q) / ULP) for q in (0.99, 0.999)],
                         t_abs_max=float(np.abs(i0["t"][both] - i1["t"][both]).max()), mat_same=bool(np.array_equal(i0["materialId"][both], i1["materialId"][both])),
                         uv_abs=float(np.abs(i0["texcoord"][both] - i1["texcoord"][both]).max()),
                         branch_differs=int((s0["remainingBounces"] != s1["remainingBounces"]).sum()), dir_abs_p50=float(np.median(ddir)), dir_abs_p999=float(np.quantile(ddir, 0.999)),
                         dir_abs_max=float(ddir.max()), dir_over_1e5=float((ddir > 1e-5).mean()), dir_over_2e5=float((ddir > 2e-5).mean()), col_rel_ulp_max=float(col.max() / ULP),
                         origin_abs=float(np.abs(s0["origin"][live] - s1["origin"][live]).max()))
                if k0 is not None:
                    k1 = T1.tile_intersect(p0)
                    samek = ((k0["t"] > 0) == (k1["t"] > 0)) & np.where(k0["t"] > 0, k0["geomId"] == k1["geomId"], True)
                    bk = samek & (k0["t"] > 0)
                    r.update(tile_decisions_differ=float(1 - samek.mean()), tile_t_rel_ulp=float((np.abs(k0["t"][bk] - k1["t"][bk]) / np.abs(k0["t"][bk])).max() / ULP))
                print(json.dumps(r), flush=True)
                T1.close()
            T0.close()
        sc = pt.Scene(os.path.join(ROOT, "scenes", "sphere.txt"), res=(16, 16), depth=2); sc.apply_runcuda_camera()
        rng = np.random.default_rng(2); n = 200000
        x = (rng.random(n) * np.float32(6.2831855)).astype(np.float32); pw = rng.uniform(0, 1, n); pxy = np.stack([rng.random(n).astype(np.float32), rng.uniform(0, 60, n).astype(np.float32)], 1)
        T0 = pt.Tracer(sc); a = T0.libm(x, pw, pxy)
        for lv in (1, 2):
            T1 = pt.Tracer(sc, arith=lv); b = T1.libm(x, pw, pxy)
            print(json.dumps(dict(libm=lv, sin_abs=float(np.abs(a[0] - b[0]).max()), cos_abs=float(np.abs(a[1] - b[1]).max()), pow5_same=bool(np.array_equal(a[2], b[2])),
                                  powf_rel=float((np.abs(a[3] - b[3]) / np.maximum(np.abs(a[3]), 1e-30)).max()))), flush=True)
            T1.close()
        T0.close()

main()
