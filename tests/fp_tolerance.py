"""How far apart two LEGAL arithmetics of the reference's source are (test infrastructure; see test_fp_tolerance.py).

A = the oracle as everything else here uses it: no FMA contraction, glibc libm (what the reference's host pass computes).
B = the same C source built with -ffp-contract=fast -mfma (what nvcc's default --fmad=true does to the reference's kernels)
    and the oracle's other libm (correctly rounded sin/cos/pow instead of glibc's: CUDA's sinf/cosf/powf differ from glibc's
    in the last place in just this way).
The shading RNG is seeded by a path's position in the sorted stream (src/pathtrace.cu:373), so one hit/miss that flips in the
last bit re-seeds every later path of that bounce: per-pixel differences are O(1) wherever that happens, and the only
meaningful end-to-end tolerance between two such builds is statistical.  This module measures it."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpulibs  # noqa: E402

CONFIGS = {      # BASELINE configs 2-4 at 480 x 270 (SURVEY 7 "hard parts" used the same frame for its probe)
    "C2": dict(scene="cornell.txt", depth=8, aa=0),
    "C3": dict(scene="cornellGlass.txt", depth=12, aa=1),
    "C4": dict(scene="cornellObj.txt", depth=8, aa=1),
}
RES = (480, 270)


def cpu_has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


def build_fma_oracle():
    subprocess.check_call(["make", "-s", "-C", cpulibs.ORACLE_DIR, "oracle_fma"])
    return os.path.join(cpulibs.ORACLE_DIR, "libptoracle_fma.so")


def _run(lib, libm, dump, cfg, spp_marks, threads):
    lib.set_libm(libm)
    lib.set_threads(threads)
    lib.create(dump, dump["textures"])
    lib.set_options(aa=cfg["aa"], dof=0, sort=1, cache=1)
    lib.pt_init()
    out, counts = {}, None
    s1 = s2 = None
    prev = None
    for it in range(1, max(spp_marks) + 1):
        lib.iterate(it)
        img = lib.image().astype(np.float64)
        one = img if prev is None else img - prev          # this iteration's radiance (sums are exact enough in fp64 here)
        prev = img
        s1 = one if s1 is None else s1 + one
        s2 = one * one if s2 is None else s2 + one * one
        if it == 1:
            counts = lib.live_counts().tolist()
        if it in spp_marks:
            n = it
            var = np.maximum(s2 / n - (s1 / n) ** 2, 0.0) * (n / max(n - 1, 1))
            out[it] = dict(mean=img / n, sem=np.sqrt(var / n))
    lib.set_threads(1)
    lib.set_libm(0)
    return out, counts


def measure(product, config, spp_marks=(1, 16, 64), threads=8):
    """-> {spp: {...}} for one config: flipped-pixel fraction, frame means, difference of the means against the Monte-Carlo
    standard error of that difference, per-pixel RMS difference against the per-pixel Monte-Carlo noise."""
    cfg = CONFIGS[config]
    s = product.Scene(os.path.join(ROOT, "scenes", cfg["scene"]), res=RES, depth=cfg["depth"])
    s.apply_runcuda_camera()
    dump = s.dump()
    A = cpulibs.OracleLib()
    B = cpulibs.OracleLib(build_fma_oracle())
    ra, ca = _run(A, 0, dump, cfg, spp_marks, threads)
    rb, cb = _run(B, 1, dump, cfg, spp_marks, threads)
    res = dict(config=config, scene=cfg["scene"], res=list(RES), depth=cfg["depth"], rays_per_bounce_iter1=dict(A=ca, B=cb), spp={})
    npx = RES[0] * RES[1]
    for n in spp_marks:
        a, b = ra[n]["mean"], rb[n]["mean"]
        diff = b - a
        flipped = np.any(diff != 0, axis=1)
        # standard error of the frame mean of each build from its per-pixel standard errors (pixels are independent estimates)
        se_frame = np.sqrt((ra[n]["sem"] ** 2).sum(axis=0) + (rb[n]["sem"] ** 2).sum(axis=0)) / npx
        mean_a, mean_b = a.mean(axis=0), b.mean(axis=0)
        noise_rms = float(np.sqrt((ra[n]["sem"] ** 2).mean())) if n > 1 else None
        res["spp"][n] = dict(
            flipped_pixel_fraction=float(flipped.mean()),
            max_abs_pixel_difference=float(np.abs(diff).max()),
            frame_mean_A=[float(x) for x in mean_a], frame_mean_B=[float(x) for x in mean_b],
            frame_mean_relative_difference=[float(x) for x in np.abs(mean_b - mean_a) / np.maximum(np.abs(mean_a), 1e-12)],
            frame_mean_difference_in_standard_errors=None if n == 1 else [float(x) for x in np.abs(mean_b - mean_a) / np.maximum(se_frame, 1e-30)],
            pixel_rms_difference=float(np.sqrt((diff ** 2).mean())),
            pixel_rms_difference_over_mc_noise=None if n == 1 else float(np.sqrt((diff ** 2).mean()) / max(noise_rms, 1e-30)))
    return res
