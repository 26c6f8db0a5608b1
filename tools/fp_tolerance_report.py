#!/usr/bin/env python3
"""Writes profiles/fp_tolerance_round2.json: C2-C4 at 480x270, 1 / 16 / 64 spp, oracle (no contraction, glibc libm) against the same
source built with FMA contraction and the other libm (tests/fp_tolerance.py) -- the stated fp32 tolerance against an arithmetic
like the reference's CUDA build.  CPU only."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mygpuraytracer_amd as pt
import fp_tolerance
out = {c: fp_tolerance.measure(pt, c, spp_marks=(1, 16, 64)) for c in ("C2", "C3", "C4")}
out["_what"] = fp_tolerance.__doc__
json.dump(out, open(os.path.join(ROOT, "profiles", "fp_tolerance_round2.json"), "w"), indent=1)
for c in ("C2", "C3", "C4"):
    for n, v in out[c]["spp"].items():
        print(c, n, "flipped %.4f" % v["flipped_pixel_fraction"], "mean rel diff", ["%.2e" % x for x in v["frame_mean_relative_difference"]],
              "in SE", None if v["frame_mean_difference_in_standard_errors"] is None else ["%.2f" % x for x in v["frame_mean_difference_in_standard_errors"]],
              "rms/noise", v["pixel_rms_difference_over_mc_noise"])
