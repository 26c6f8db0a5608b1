"""CPU, dev container only: the plain-C oracle against the LIVE reference build (oracle/_ref/libptref.so, the
reference's own headers compiled from /root/reference) on fresh seeded inputs and on larger frames than the
committed fixtures hold.  Skipped where /root/reference (hence the _ref build) does not exist."""
import os

import numpy as np
import pytest

from conftest import beq
from cpulibs import REFERENCE_ROOT, scene_text_with


def _pair(R, O, name, res, depth, libm=0, **opt):
    R.load_text(scene_text_with(open(os.path.join(REFERENCE_ROOT, "scenes", name)).read(), res, depth))
    O.set_libm(libm)
    O.create(R.dump())
    R.apply_runcuda_camera(); O.apply_runcuda_camera()
    if opt:
        R.set_options(**opt); O.set_options(**opt)
    R.pt_init(); O.pt_init()


@pytest.mark.parametrize("name,res,depth,opt", [
    ("cornell.txt", (160, 160), 8, dict(aa=0, dof=0, sort=1, cache=1)),
    ("cornellGlass.txt", (192, 108), 12, dict(aa=1, dof=0, sort=1, cache=1)),
    ("cornellObj.txt", (192, 108), 8, dict(aa=1, dof=1, sort=1, cache=0)),
    ("cornellObj.txt", (192, 108), 8, dict(aa=1, dof=0, sort=0, cache=0)),
])
def test_frames_bit_identical_to_reference(ref_lib, oracle_lib, name, res, depth, opt):
    _pair(ref_lib, oracle_lib, name, res, depth, 0, **opt)
    for it in (1, 2, 3, 4):
        ref_lib.iterate(it); oracle_lib.iterate(it)
        assert beq(ref_lib.live_counts(), oracle_lib.live_counts())
    assert beq(ref_lib.image(), oracle_lib.image())
    assert beq(ref_lib.paths(), oracle_lib.paths())          # final permutation of the whole stream
    assert beq(ref_lib.pbo(4), oracle_lib.pbo(4))


def test_own_libm_mode_keeps_every_discrete_decision(ref_lib, oracle_lib):
    """With the portable sin/cos/pow (what the GPU runs) instead of glibc's, a 480x270 depth-8 frame still makes
    the same hit/miss/material decisions as the reference: identical live counts and identical image."""
    _pair(ref_lib, oracle_lib, "cornellObj.txt", (480, 270), 8, 1)
    try:
        for it in (1, 2):
            ref_lib.iterate(it); oracle_lib.iterate(it)
            assert beq(ref_lib.live_counts(), oracle_lib.live_counts())
        a, b = ref_lib.image(), oracle_lib.image()
        assert int((a != b).any(axis=1).sum()) == 0
    finally:
        oracle_lib.set_libm(0)


def test_random_stage_inputs(ref_lib, oracle_lib):
    """Fresh random rays through computeIntersections and arbitrary stream indices through shadeFakeMaterial."""
    _pair(ref_lib, oracle_lib, "cornellGlass.txt", (64, 64), 8, 0)
    rng = np.random.default_rng(77)
    from cpulibs import PATH_DTYPE
    p = np.zeros(6000, PATH_DTYPE)
    p["origin"] = rng.uniform(-4.5, 4.5, (6000, 3)) + [0, 5, 0]
    d = rng.normal(size=(6000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    p["direction"] = d
    p["color"] = rng.uniform(0, 1, (6000, 3))
    p["pixelIndex"] = np.arange(6000); p["remainingBounces"] = rng.integers(1, 9, 6000)
    ri, oi = ref_lib.compute_intersections(p), oracle_lib.compute_intersections(p)
    assert beq(ri, oi)
    idx = rng.integers(0, 3_000_000, 6000).astype(np.int32)
    assert beq(ref_lib.shade(9, 1, idx, ri, p), oracle_lib.shade(9, 1, idx, oi, p))


def _cottage_scene_text(res=(96, 54), depth=6):
    """the reference's cornellObj.txt with its cube swapped for models/cottage_obj.obj (5 objects, 3 materials, 32
    triangles + 227 quads = 486 triangles), scaled into the box.  Read where it lies under /root/reference."""
    text = open(os.path.join(REFERENCE_ROOT, "scenes", "cornellObj.txt")).read()
    assert "../models/cube.obj" in text
    text = text.replace("../models/cube.obj", "../models/cottage_obj.obj")
    head, tail = text.rsplit("TRANS", 1)
    tail = "       0.5 1.2 0\nROTAT       0 30 0\nSCALE       .02 .02 .02\n"
    return scene_text_with(head + "TRANS" + tail, res, depth)


def test_cottage_mesh_loader_oracle_and_bvh(ref_lib, oracle_lib, product, tmp_path):
    """A second real mesh, never copied into this repo: the reference loader, this repo's loader, the
    oracle's frame and the BVH are compared on it in place."""
    import ctypes as C
    text = _cottage_scene_text()
    ref_lib.load_text(text)
    rd = ref_lib.dump()
    (tmp_path / "scenes").mkdir()
    os.symlink(os.path.join(REFERENCE_ROOT, "models"), tmp_path / "models")
    (tmp_path / "scenes" / "cottage.txt").write_text(text)
    d = product.Scene(str(tmp_path / "scenes" / "cottage.txt")).dump()
    for k in ("geom_ints", "geom_trs", "geom_mats", "materials", "cam_floats"):
        assert beq(d[k], rd[k]), k
    assert len(d["faces"]) == len(rd["faces"])
    for a, b in zip(d["faces"], rd["faces"]):
        assert beq(a, b)
    mesh = [f for f in d["faces"] if len(f)]
    assert len(mesh) == 1 and mesh[0].size == 486 * 15
    # frames: oracle == reference build
    oracle_lib.set_libm(0)
    oracle_lib.create(rd)
    ref_lib.apply_runcuda_camera(); oracle_lib.apply_runcuda_camera()
    ref_lib.pt_init(); oracle_lib.pt_init()
    for it in (1, 2):
        ref_lib.iterate(it); oracle_lib.iterate(it)
        assert beq(ref_lib.live_counts(), oracle_lib.live_counts())
    assert beq(ref_lib.image(), oracle_lib.image())
    # BVH == the loop over its 486 faces, on rays around the object-space mesh
    from test_bvh import run_check, rays_around
    rng = np.random.default_rng(5)
    faces = np.asarray(mesh[0], np.float32).reshape(-1, 15)
    ext = float(np.abs(faces[:, [0, 1, 2, 5, 6, 7, 10, 11, 12]]).max())
    rays = np.concatenate([rays_around(rng, 20000, 3 * ext, ext), rays_around(rng, 5000, 0.3 * ext, ext)])
    fl, tl, fb, tb, st = run_check(product, faces, rays)
    assert beq(fl, fb) and beq(tl, tb) and (fl >= 0).sum() > 2000


def test_dead_triangle_functions_live(ref_lib, oracle_lib):
    """SURVEY 8(a10) on fresh rays: the reference's own objTriIntersectionTest (src/intersections.h:284-315, dead code there) against
    the oracle's restatement, 20 000 rays around cube.obj in cornellObj.txt -- bit for bit, hits and misses."""
    _pair(ref_lib, oracle_lib, "cornellObj.txt", (32, 32), 4, 0)
    d = ref_lib.dump()
    gi = [i for i in range(len(d["geom_ints"])) if d["geom_ints"][i][0] == 3][0]
    rng = np.random.default_rng(77)
    centre = d["geom_trs"][gi][:3].astype(np.float64)
    o = centre + rng.normal(size=(20000, 3)) * 3.0
    t = centre + rng.normal(size=(20000, 3)) * 0.8
    dirs = t - o
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rays = np.concatenate([o, dirs], 1).astype(np.float32)
    a, b = ref_lib.obj_tri_test(gi, rays), oracle_lib.obj_tri_test(gi, rays)
    assert (a[:, 0] > 0).sum() > 2000
    assert beq(a, b)


def test_dead_jittered_sampler_live(ref_lib, oracle_lib):
    """SURVEY 8(a13) on fresh samples: the reference's own calculateJitteredDirectionHemisphere (src/interactions.h:46-85, dead code
    there) against the oracle's restatement, 50 000 samples, three values of max_iter -- bit for bit (glibc mode)."""
    rng = np.random.default_rng(78)
    n = 50000
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    seeds = np.stack([rng.integers(1, 20000, n), rng.integers(0, 3840 * 2160, n), rng.integers(0, 13, n)], 1).astype(np.int32)
    oracle_lib.set_libm(0)
    for mi in (5000, 1, 1000):
        assert beq(ref_lib.jittered_test(nrm, seeds, mi), oracle_lib.jittered_test(nrm, seeds, mi))

