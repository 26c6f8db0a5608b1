#!/usr/bin/env python3
"""The stated fp32 tolerance of the opt-in arithmetic levels, MEASURED on the GPU (tests/test_gpu_arith.py asserts the bounds; this prints
the figures): BASELINE configs 2-4 at 480x270, exact code object against level 1 (contracted) and level 2 (fast) -- rays per bounce of
iteration 1, pixels that differ at all at 1 spp, frame means / their difference in standard errors / per-pixel RMS difference against the
Monte-Carlo noise at 16 and 64 spp.  The CPU counterpart (oracle against its FMA-contracted build) is tools/fp_tolerance_report.py ->
profiles/fp_tolerance_round2.json.      python tools/gpu_arith_tolerance.py > profiles/fp_tolerance_round5_gpu.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mygpuraytracer_amd as pt
import fp_tolerance
MARKS = (1, 16, 64)


def frames(cfg, level):
    s = pt.Scene(os.path.join(ROOT, "scenes", cfg["scene"]), res=fp_tolerance.RES, depth=cfg["depth"]); s.apply_runcuda_camera()
    out, counts = {}, None
    with pt.Tracer(s, arith=level, antialiasing=cfg["aa"]) as T:
        prev = s1 = s2 = None
        for it in range(1, max(MARKS) + 1):
            T.pathtrace(it)
            img = T.read_image().astype(np.float64)
            one = img if prev is None else img - prev
            prev = img
            s1 = one if s1 is None else s1 + one
            s2 = one * one if s2 is None else s2 + one * one
            if it == 1:
                counts = T.stats()["rays_per_bounce"]
            if it in MARKS:
                var = np.maximum(s2 / it - (s1 / it) ** 2, 0.0) * (it / max(it - 1, 1))
                out[it] = (img / it, np.sqrt(var / it))
        fenced = T.stats()["fenced"]
    return out, counts, fenced


res = {}
npx = fp_tolerance.RES[0] * fp_tolerance.RES[1]
for name, cfg in fp_tolerance.CONFIGS.items():
    ra, ca, _ = frames(cfg, 0)
    res[name] = dict(scene=cfg["scene"], res=list(fp_tolerance.RES), depth=cfg["depth"], rays_per_bounce_iter1_exact=ca, levels={})
    for lv in (1, 2):
        rb, cb, fenced = frames(cfg, lv)
        d = dict(rays_per_bounce_iter1=cb, fenced=fenced, spp={})
        for n in MARKS:
            (a, sa), (b, sb) = ra[n], rb[n]
            diff = b - a
            se = np.sqrt((sa ** 2).sum(axis=0) + (sb ** 2).sum(axis=0)) / npx
            ma, mb = a.mean(axis=0), b.mean(axis=0)
            noise = float(np.sqrt((sa ** 2).mean())) if n > 1 else None
            d["spp"][n] = dict(flipped_pixel_fraction=float(np.any(diff != 0, axis=1).mean()),
                               frame_mean_relative_difference=[float(x) for x in np.abs(mb - ma) / np.maximum(np.abs(ma), 1e-12)],
                               frame_mean_difference_in_standard_errors=None if n == 1 else [float(x) for x in np.abs(mb - ma) / np.maximum(se, 1e-30)],
                               pixel_rms_difference_over_mc_noise=None if n == 1 else float(np.sqrt((diff ** 2).mean()) / max(noise, 1e-30)))
        res[name]["levels"][str(lv)] = d
print(json.dumps(res, indent=1))
