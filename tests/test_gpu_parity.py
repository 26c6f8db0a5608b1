"""GPU: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

The oracle runs in its portable-libm mode (the sin/cos/pow routines the kernels run; pinned to the reference in
tests/test_own_libm.py / test_oracle_pin.py); everything else about it is pinned to the reference by
tests/test_oracle_golden.py.  Tolerance: none -- integer work and fp32 alike must match exactly, because the
shading RNG is seeded by stream position and any flipped decision decorrelates the frame (hard parts).
"""
import os

import numpy as np
import pytest

from conftest import ROOT, beq, golden
from cpulibs import PATH_DTYPE

pytestmark = pytest.mark.gpu


def make_pair(pt, O, scene, res, depth, **opt):
    s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=res, depth=depth)
    s.apply_runcuda_camera()
    d = s.dump()
    O.set_libm(1)
    O.create(d, d["textures"])
    O.set_options(aa=opt.get("antialiasing", 1), dof=opt.get("depth_of_field", 0), sort=opt.get("sort_by_material", 1),
                  cache=opt.get("cache_first_bounce", 1))
    O.pt_init()
    return s, pt.Tracer(s, **opt)


@pytest.fixture()
def O(oracle_lib):
    oracle_lib.set_libm(1)
    yield oracle_lib
    oracle_lib.set_libm(0)


@pytest.fixture(autouse=True)
def fences_stay_silent(gpu_product):
    """Every index the kernels take from a table another launch wrote (mesh-search queue, parked ray's owner, sort index) is
    range-checked before it becomes an address, and what a check stops is COUNTED (ptx_stats.fenced).  Every Tracer a test of this
    module closes must have counted nothing: a corrupted queue shows as a number here, not only as a wrong pixel somewhere."""
    pt = gpu_product
    orig, seen = pt.Tracer.close, []

    def close(self):
        if getattr(self, "h", None) and not getattr(self, "expect_fenced", False):
            try:
                seen.append(self.stats()["fenced"])
            except Exception:            # (a tracer left in an error state by a test of the error paths)
                pass
        orig(self)
    pt.Tracer.close = close
    yield
    pt.Tracer.close = orig
    assert not any(seen), "the kernels fenced %s indices: internal tables were corrupt" % seen


def test_fences_count_what_they_stop(gpu_product, monkeypatch):
    """The counter itself: with the fence limit lowered to one tile (PTX_DEBUG_FENCE_SLOTS = 256, tests only) every queue entry,
    owner and sort-index entry beyond slot 255 is stopped -- skipped or clamped, so nothing is read or written out of range --
    and counted.  Without the knob the same run counts 0."""
    pt = gpu_product
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship.txt"), res=(96, 54), depth=6)
    s.apply_runcuda_camera()
    with pt.Tracer(s) as T:
        T.render(1, 3)
        assert T.stats()["fenced"] == 0
        T.reset_image()
        T.synchronize()
        assert T.stats()["fenced"] == 0
    monkeypatch.setenv("PTX_DEBUG_FENCE_SLOTS", "256")
    for opt in ({}, dict(no_mesh_split=1)):
        with pt.Tracer(s, **opt) as T:
            T.expect_fenced = True
            T.render(1, 3)
            n = T.stats()["fenced"]
            assert n > 0, opt
            T.reset_image()                 # the counter is part of what reset clears
            T.synchronize()
            assert T.stats()["fenced"] == 0


def test_device_libm_bit_identical(gpu_product, O):
    s, T = make_pair(gpu_product, O, "sphere.txt", (16, 16), 2)
    rng = np.random.default_rng(1)
    n = 200000
    x = np.concatenate([(rng.random(n - 7) * np.float32(6.2831855)).astype(np.float32),
                        np.float32([0, 6.2831855, 3.1415927, 1.5707964, -0.7853982, 2.3561945, 1e-30])])
    pw = rng.uniform(-0.5, 2.0, n)
    pxy = np.stack([rng.random(n).astype(np.float32), rng.uniform(0, 120, n).astype(np.float32)], 1)
    pxy[:6] = [[0, 0], [0, 2], [1, 5], [.5, 2], [2, 10], [.3, 0]]
    sn, cs, p5, po = T.libm(x, pw, pxy)
    for k in range(0, n, 37):
        a, b = O.own_sincosf(x[k])
        assert a.tobytes() == sn[k].tobytes() and b.tobytes() == cs[k].tobytes()
        assert O.lib.o_own_pow5(float(pw[k])) == p5[k]
        assert np.float32(O.lib.o_own_powf(float(pxy[k, 0]), float(pxy[k, 1]))).tobytes() == po[k].tobytes()
    T.close()


def test_core_sqrt_and_reciprocal_equal_the_ieee_expansions_on_every_operand(gpu_product):
    """pt_device.h (round 4): sqrtf, 1 / sqrtf and the triangle test's 1 / a run the CORE of the compiler's correctly rounded
    expansions behind one range compare and leave every other operand to the expansion itself.  All 2^32 bit patterns through both, on
    the device: not one differs (NaNs compare as equal).  (The kernels' parity with the oracle rests on these being IEEE results.)"""
    pt = gpu_product
    s = pt.Scene(os.path.join(ROOT, "scenes", "sphere.txt"), res=(32, 32), depth=2)
    s.apply_runcuda_camera()
    with pt.Tracer(s) as T:
        assert T.kat_fast_exact() == [0, 0, 0]


@pytest.mark.parametrize("scene", ["cornellGlass", "cornellObj", "cornellSpaceship"])
def test_intersection_kats_on_device(gpu_product, O, scene):
    """The golden per-geom vectors (produced by the reference's own box/sphere/mesh tests) through the device functions."""
    k = golden("isect_kat_%s.npz" % scene)
    s, T = make_pair(gpu_product, O, scene + ".txt", (16, 16), 8)
    for gi in range(s.num_geoms):
        if "rays_%d" % gi not in k.files:
            continue
        out = T.geom_test(gi, k["rays_%d" % gi])
        ref = k["out_%d" % gi]
        hit = ref[:, 0] > 0
        assert beq(out[:, 0], ref[:, 0]) and beq(out[hit], ref[hit])
    T.close()


def test_dead_triangle_functions_on_device(gpu_product, O):
    """objTriIntersectionTest -> triangleIntersectionLocalTest (src/intersections.h:175-205, 284-315; dead code in the
    reference, on no path here) as device functions behind ptx_kat_obj_tri_test, against what the reference's own functions returned
    ([direct] fixture dead_tri_kat.npz): cube.obj inside cornellObj.txt from the scene file, cottage_obj.obj from the loader's vectors."""
    k = golden("dead_tri_kat.npz")
    s, T = make_pair(gpu_product, O, "cornellObj.txt", (16, 16), 8)
    cases = [("obj", T)]
    from conftest import dump_from_golden
    Tc = gpu_product.Tracer.from_pod(dump_from_golden(golden("loader_cottage.npz")))
    cases.append(("cottage", Tc))
    for which, tr in cases:
        out = tr.obj_tri_test(int(k[which + "_geom"]), k[which + "_rays"])
        ref = k[which + "_out"]
        hit = ref[:, 0] > 0
        assert hit.sum() > 500
        assert beq(out[:, 0], ref[:, 0]) and beq(out[hit], ref[hit]), which
        assert beq(out, O.obj_tri_test(int(k[which + "_geom"]), k[which + "_rays"])) if which == "obj" else True
    T.close(); Tc.close()


def test_dead_jittered_sampler_on_device(gpu_product, O):
    """calculateJitteredDirectionHemisphere (src/interactions.h:46-85; dead code in the reference, on no path here) as a
    device function behind ptx_kat_jittered_hemisphere: bit-identical to the oracle in the kernels' libm mode, and within 1 ulp of what
    the reference's own function returned ([direct] fixture jitter_kat.npz; glibc's sinf / cosf there, the portable ones here)."""
    k = golden("jitter_kat.npz")
    s, T = make_pair(gpu_product, O, "sphere.txt", (16, 16), 2)
    for mi in (5000, 64):
        got = T.jittered_hemisphere(k["normals"], k["seeds"], mi)
        assert beq(got, O.jittered_test(k["normals"], k["seeds"], mi))          # (O: libm mode 1, the routines the kernels run)
        want = k["dir_%d" % mi]
        assert np.abs(got - want).max() <= 2 * np.finfo(np.float32).eps and (got == want).mean() > 0.9
    T.close()


@pytest.mark.parametrize("scene,res,depth,opt", [
    ("sphere.txt", (64, 64), 4, {}),
    ("cornell.txt", (64, 64), 8, dict(antialiasing=0)),
    ("cornellGlass.txt", (96, 54), 12, dict(depth_of_field=1)),
    ("cornellObj.txt", (96, 54), 8, {}),
    ("cornellSpaceship.txt", (96, 54), 8, dict(depth_of_field=1)),
    ("cornellSpaceship.txt", (96, 54), 8, dict(no_bvh=1)),
    ("cornellSpaceship20k.txt", (160, 90), 8, {}),                     # 20448 triangles through the BVH vs the oracle's loop
])
def test_stage_parity(gpu_product, O, scene, res, depth, opt):
    """generateRayFromCamera, computeIntersections and shadeFakeMaterial one at a time on identical inputs."""
    s, T = make_pair(gpu_product, O, scene, res, depth, **opt)
    O.pt_generate(3)
    op = O.paths()
    assert beq(T.generate(3), op)
    oi, gi = O.compute_intersections(op), T.compute_intersections(op)
    hit = oi["t"] > 0
    assert beq(gi["t"], oi["t"]) and beq(gi["materialId"], oi["materialId"])
    assert beq(gi["normal"][hit], oi["normal"][hit]) and beq(gi["geomId"][hit], oi["geomId"][hit])
    obj = hit & (s.dump()["geom_ints"][oi["geomId"], 0] == 3)
    assert beq(gi["texcoord"][obj], oi["texcoord"][obj])
    idx = np.random.default_rng(9).integers(0, 4_000_000, len(op)).astype(np.int32)
    assert beq(T.shade(3, idx, oi, op), O.shade(3, 1, idx, oi, op))
    # last bounce (remainingBounces == 1) and second scatter
    op2 = O.shade(3, 1, idx, oi, op)
    live = op2["remainingBounces"] > 0
    p2 = op2[live]
    p2["remainingBounces"][::3] = 1
    oi2 = O.compute_intersections(p2)
    assert beq(T.compute_intersections(p2)["t"], oi2["t"])
    assert beq(T.shade(4, idx[: len(p2)], oi2, p2), O.shade(4, 2, idx[: len(p2)], oi2, p2))
    T.close()


@pytest.mark.parametrize("tag", ["glass", "obj", "mirror0", "mirror20", "ship"])
def test_shade_golden_on_device(gpu_product, O, tag):
    """Golden (path, intersection) -> shaded path vectors captured from reference renders.  The expected values were
    produced with glibc's libm, the device runs the portable one: identical wherever no sin/cos/pow is involved
    (mirror with exponent 0, refraction, lights, misses), within 2e-6 otherwise."""
    from test_oracle_golden import _scene_for_shade
    from test_loader import product_dump_from_text
    import tempfile
    k = golden("shade_kat_%s.npz" % tag)
    if tag.startswith("mirror"):
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            f.write(bytes(k["scene_text"]).decode())
        s = gpu_product.Scene(f.name, base_dir=os.path.join(ROOT, "scenes"))
        os.unlink(f.name)
    else:
        s = gpu_product.Scene(os.path.join(ROOT, "scenes", dict(glass="cornellGlass.txt", obj="cornellObj.txt", ship="cornellSpaceship.txt")[tag]))
    T = gpu_product.Tracer(s)
    _scene_for_shade(O, tag)
    O.set_libm(1)
    for key in sorted(x[:-6] for x in k.files if x.endswith("_paths")):
        got = T.shade(int(key[2]), k[key + "_idx"], k[key + "_isects"], k[key + "_paths"])
        assert beq(got, O.shade(int(key[2]), 1, k[key + "_idx"], k[key + "_isects"], k[key + "_paths"]))
        want = k[key + "_shaded"]
        assert beq(got["remainingBounces"], want["remainingBounces"]) and beq(got["pixelIndex"], want["pixelIndex"])
        for f in ("origin", "direction", "color"):
            assert np.allclose(got[f], want[f], rtol=2e-6, atol=2e-6), (key, f)    # fp32 tolerance for the libm swap
            assert (got[f] == want[f]).mean() > 0.95                               # identical in the vast majority
    T.close()


def oracle_pending_stream(O, it, bounce):
    """What the HIP path stores after bounce `bounce`: the paths that will scatter at bounce+1, in the reference's
    sorted order, with their stream index (= position after intersect + sort) and pending intersection."""
    O.pt_generate(it)
    for _ in range(bounce + 1):
        n = O.num_paths()
        O.pt_bounce(it, 3)
        paths, isects = O.paths()[:n], O.isects()[:n]
        O.pt_bounce(it, 12)
    return n, paths, isects


@pytest.mark.parametrize("scene,res,depth,opt", [
    ("cornellGlass.txt", (96, 54), 12, {}),
    ("cornellObj.txt", (96, 54), 8, {}),
    ("cornellObj.txt", (96, 54), 8, dict(sort_by_material=0)),
    ("cornell.txt", (64, 64), 8, dict(antialiasing=0)),
    ("cornellSpaceship.txt", (96, 54), 8, dict(depth_of_field=1)),
    ("cornellSpaceship.txt", (96, 54), 8, dict(no_lds_triangles=1)),
    ("cornellSpaceship.txt", (96, 54), 8, dict(no_cull=1)),
    ("cornellSpaceship.txt", (96, 54), 8, dict(no_bvh=1)),                 # the reference's loop over all faces
    ("cornellSpaceship20k.txt", (64, 36), 8, dict(depth_of_field=1)),      # 20448 triangles: BVH, tables in global memory
    ("cornellSpaceship20k.txt", (64, 36), 8, dict(no_cull=1)),
    ("cornellSpaceship20k.txt", (64, 36), 8, dict(no_mesh_split=1)),      # mesh search inside the bounce kernel
    ("cornellSpaceship.txt", (96, 54), 8, dict(no_mesh_split=1, depth_of_field=1)),
    ("cornellGlass.txt", (96, 54), 12, dict(no_cull=1)),
])
def test_sorted_stream_parity(gpu_product, O, scene, res, depth, opt):
    """The permutation is the observable: after each bounce the device stream holds exactly the reference's sorted
    survivors -- same pixels in the same order, same RNG stream index, same ray / colour / hit bits."""
    s, T = make_pair(gpu_product, O, scene, res, depth, **opt)
    check_sorted_streams(T, O, s.dump(), depth)
    T.close()


def records_with_direction(d, material_ids):
    """the rule of ptx_create's dir_bins: a stored path carries its incoming direction iff scatterRay can read it for the material hit --
    reflective, refractive, or the material of an OBJ geom"""
    mats, gi = d["materials"], d["geom_ints"]
    need = (mats[:, 7] > 0) | (mats[:, 8] > 0)
    need[gi[gi[:, 0] == 3, 1]] = True
    return need[material_ids]


def records_with_normal_code(d, material_ids):
    """... and of its ntab_bins: a stored path of a material that only cubes have carries a 3-bit code of the cube's tabulated normal (bits
    28-30 of its pixel word) instead of the normal"""
    gi = d["geom_ints"]
    nm = len(d["materials"])
    on_cube = np.zeros(nm, bool); on_other = np.zeros(nm, bool)
    on_cube[gi[gi[:, 0] == 1, 1]] = True
    on_other[gi[gi[:, 0] != 1, 1]] = True
    return (on_cube & ~on_other)[material_ids]


def check_sorted_streams(T, O, d, depth, directions_where_needed_only=False):
    """d = the scene's POD dict (Scene.dump() layout)"""
    mats = d["materials"]
    it = 1
    for bounce in range(min(depth - 1, 5)):
        T.reset_image()
        T.debug_capture(bounce)
        T.pathtrace(it)
        g = T.debug_stream()
        n, paths, isects = oracle_pending_stream(O, it, bounce)
        # bounce b's sorted array is shaded at shade index b: remainingBounces there = depth - b
        pend = (isects["t"] > 0) & (mats[isects["materialId"], 10] <= 0) & (depth - bounce != 1)
        want_idx = np.nonzero(pend)[0].astype(np.int32)
        assert len(g["pix"]) == len(want_idx), (bounce, len(g["pix"]), len(want_idx))
        assert beq(g["idx"], want_idx)
        coded = records_with_normal_code(d, isects["materialId"][pend]) if directions_where_needed_only else np.zeros(int(pend.sum()), bool)
        assert beq(g["pix"] & 0x0fffffff if directions_where_needed_only else g["pix"], paths["pixelIndex"][pend])
        if coded.any():      # the code names the side of the cube whose tabulated normal the oracle's normal is: axis + 1 | side << 2
            code = (g["pix"][coded] >> 28) & 7
            assert np.all((code & 3) >= 1) and np.all((g["pix"][~coded] >> 28) == 0)
        assert beq(g["mat"], isects["materialId"][pend])
        # the stream carries the point that will be shaded, origin + t * direction (src/pathtrace.cu:392), in fp32
        sp = paths["origin"][pend] + isects["t"][pend][:, None] * paths["direction"][pend]
        for k, nm in enumerate(("px", "py", "pz")):
            assert beq(g[nm], sp[:, k])
        # (a capture makes every record carry its direction; with PTX_DEBUG_KEEP_DIR_SKIP the kernels run as they do otherwise, and the
        # words of the records that travel without one are whatever the slot held)
        with_dir = records_with_direction(d, isects["materialId"][pend]) if directions_where_needed_only else np.ones(int(pend.sum()), bool)
        for k, nm in enumerate(("dx", "dy", "dz")):
            assert beq(g[nm][with_dir], paths["direction"][pend][with_dir, k])
        if directions_where_needed_only and bounce == 0 and (~with_dir).any():
            # ... and the skip is live: what a fresh stage holds in those words is not the paths' directions
            assert not all(beq(g[nm][~with_dir], paths["direction"][pend][~with_dir, k]) for k, nm in enumerate(("dx", "dy", "dz")))
        for k, nm in enumerate(("cr", "cg", "cb")):
            assert beq(g[nm], paths["color"][pend][:, k])
        for k, nm in enumerate(("nx", "ny", "nz")):
            assert beq(g[nm][~coded], isects["normal"][pend][~coded, k])
        if directions_where_needed_only and bounce == 0 and coded.any():      # (live: a fresh stage does not hold the normals)
            assert not all(beq(g[nm][coded], isects["normal"][pend][coded, k]) for k, nm in enumerate(("nx", "ny", "nz")))
        if d.get("textures"):                                      # texcoords travel only when some texture exists
            obj = d["geom_ints"][isects["geomId"][pend], 0] == 3
            assert beq(g["u"][obj], isects["texcoord"][pend][obj, 0]) and beq(g["v"][obj], isects["texcoord"][pend][obj, 1])


@pytest.mark.parametrize("scene,res,depth,opt", [("cornellObj.txt", (192, 108), 8, {}), ("cornellSpaceship.txt", (160, 120), 6, {}),
                                                  ("cornellSpaceship.txt", (96, 54), 6, dict(no_mesh_split=1)), ("cornellGlass.txt", (96, 96), 8, {})])
def test_records_without_their_direction_lose_nothing_else(gpu_product, O, monkeypatch, scene, res, depth, opt):
    """A stored path whose material scatterRay never asks for the incoming direction (a diffuse hit on a cube or sphere) travels without
    those three words, and one whose material only cubes have with a 3-bit code instead of its normal.  With the kernels running exactly as they do outside a capture (PTX_DEBUG_KEEP_DIR_SKIP): every other field of
    every record, and the direction of every record that needs one, equal the oracle's after every bounce -- in the fused bounce, the
    split bounce (parked rays keep theirs), the textured mesh inside the bounce kernel and a scene of mostly reflective / refractive
    hits; both kinds of record occur; and the frame equals the one of a tracer that carries every direction (PTX_DEBUG_NO_DIR_SKIP)."""
    monkeypatch.setenv("PTX_DEBUG_KEEP_DIR_SKIP", "1")
    s, T = make_pair(gpu_product, O, scene, res, depth, **opt)
    d = s.dump()
    check_sorted_streams(T, O, d, depth, directions_where_needed_only=True)
    T.debug_capture(-1); T.reset_image()
    T.render(1, 6)
    img = T.read_image()
    T.close()
    n, paths, isects = oracle_pending_stream(O, 1, 0)
    need = records_with_direction(d, isects["materialId"][isects["t"] > 0])
    assert need.any() and (~need).any() and records_with_normal_code(d, isects["materialId"][isects["t"] > 0]).any()
    # ptx_stats counts the stored paths by what their records weigh: the oracle's streams of iteration 1, bounce by bounce
    want = np.zeros(3, np.int64)
    for b in range(depth - 1):
        n, paths, isects = oracle_pending_stream(O, 1, b)
        pend = (isects["t"] > 0) & (d["materials"][isects["materialId"], 10] <= 0)
        m = isects["materialId"][pend]
        want += (len(m), int(records_with_direction(d, m).sum()), int(records_with_normal_code(d, m).sum()))
    with gpu_product.Tracer(s, **opt) as T3:
        T3.render(1, 1)
        st = T3.stats()
        assert (st["stored_paths"], st["stored_with_direction"], st["stored_with_normal_code"]) == tuple(int(x) for x in want)
    monkeypatch.setenv("PTX_DEBUG_NO_DIR_SKIP", "1")
    with gpu_product.Tracer(s, **opt) as T2:
        T2.render(1, 6)
        assert beq(T2.read_image(), img)
        st = T2.stats()
        assert st["stored_with_direction"] == st["stored_paths"] > 0 and st["stored_with_normal_code"] == 0


def test_cottage_mesh_from_vectors_on_device(gpu_product, O):
    """SURVEY 8(f)-3 on the GPU without shipping the file: the reference's models/cottage_obj.obj (486 triangles) as the REFERENCE
    LOADER returned it (tests/golden/loader_cottage.npz: geoms, 486 x 15 face floats, materials, camera) goes through ptx_create
    as plain arrays (Tracer.from_pod), with the BVH + split mesh search, fused, and with the reference's loop over all faces:
    meshIntersectionTest's golden rays, the sorted stream after every bounce, image and ray counts after 1 and 4 iterations --
    against the oracle on the same arrays bit for bit, and against render_cottage.npz -- what oracle/_ref rendered: the reference's own
    intersection / scatter functions inside the RESTATED bounce loop (make_golden.py PROVENANCE "restated"), not the reference's CUDA build."""
    from conftest import dump_from_golden
    g, r, k = golden("loader_cottage.npz"), golden("render_cottage.npz"), golden("isect_kat_cottage.npz")
    d = dump_from_golden(g, cam="cam_floats_runcuda")
    depth = int(d["cam_ints"][3])
    for opt in ({}, dict(no_mesh_split=1), dict(no_bvh=1)):
        O.set_libm(1)
        O.create(d, {})
        O.set_options(aa=1, dof=0, sort=1, cache=1)
        O.pt_init()
        with gpu_product.Tracer.from_pod(d, **opt) as T:
            gi = [int(x[5:]) for x in k.files if x.startswith("rays_")][0]
            out, ref = T.geom_test(gi, k["rays_%d" % gi]), k["out_%d" % gi]
            hit = ref[:, 0] > 0
            assert hit.sum() > 200 and beq(out[:, 0], ref[:, 0]) and beq(out[hit], ref[hit])
            check_sorted_streams(T, O, d, depth)
            T.reset_image(); T.debug_capture(-1)
            O.pt_init()
            for it in range(1, 5):
                O.iterate(it); T.pathtrace(it)
                if it in (1, 4):
                    img = T.read_image()
                    assert beq(img, O.image()) and np.array_equal(img, r["image_spp%d" % it])
                    assert T.stats()["rays_per_bounce"][:depth] == r["counts_it%d" % it].tolist()


def _kat_paths(rays):
    p = np.zeros(len(rays), PATH_DTYPE)
    p["origin"], p["direction"] = rays[:, :3], rays[:, 3:6]
    p["color"] = 1.0
    p["pixelIndex"] = np.arange(len(rays)); p["remainingBounces"] = 4
    return p


@pytest.mark.parametrize("scene,fixture,opt,split", [
    ("cornellGlass.txt", "cornellGlass", {}, False), ("cornellObj.txt", "cornellObj", {}, False),
    ("cornellSpaceship.txt", "cornellSpaceship", {}, True), ("cornellSpaceship.txt", "cornellSpaceship", dict(no_mesh_split=1), False),
    ("cornellSpaceship.txt", "cornellSpaceship", dict(no_bvh=1), False), (None, "cottage", {}, True), (None, "cottage", dict(no_mesh_split=1), False)])
def test_production_intersect_on_the_golden_rays(gpu_product, O, scene, fixture, opt, split):
    """The functions the bounce kernels intersect with -- cullMask -> pair lists -> primKey / meshKey -> 64-bit minimum ->
    decodeKey (tileIntersect), and for BVH scenes the split search's pass 1 / stack traversal / pass 2 -- get a test of their own
    (ptx_kat_tile_intersect): on the rays of the per-geom golden vectors, all geoms' rays in one batch, the nearest hit over the whole
    scene equals the oracle's computeIntersections bit for bit, and wherever the winner is the geom a ray set was made for, distance,
    normal and texcoords are the REFERENCE's own box / sphere / meshIntersectionTest outputs (isect_kat_*.npz, [direct])."""
    from conftest import dump_from_golden
    k = golden("isect_kat_%s.npz" % fixture)
    if scene:
        s, T = make_pair(gpu_product, O, scene, (16, 16), 8, **opt)
        d = s.dump()
    else:
        d = dump_from_golden(golden("loader_%s.npz" % fixture), cam="cam_floats_runcuda")
        O.set_libm(1); O.create(d, {}); O.pt_init()
        T = gpu_product.Tracer.from_pod(d, **opt)
    sets = sorted(int(x[5:]) for x in k.files if x.startswith("rays_"))
    rays = np.concatenate([k["rays_%d" % gi] for gi in sets])
    p = _kat_paths(rays)
    got, want = T.tile_intersect(p, split=split), O.compute_intersections(p)
    hit = want["t"] > 0
    assert hit.sum() > 100 and beq(got["t"], want["t"])
    # (texcoords travel only when the scene has a texture to look up with them: DESIGN 4)
    for f in ("normal", "materialId", "geomId") + (("texcoord",) if d.get("textures") else ()):
        assert beq(got[f][hit], want[f][hit]), f
    assert beq(T.compute_intersections(p)["t"], want["t"])           # (the plain loop, for completeness)
    off = 0
    pinned = 0
    for gi in sets:
        ref = k["out_%d" % gi]
        sl = got[off:off + len(ref)]
        mine = (sl["t"] > 0) & (sl["geomId"] == gi)
        assert beq(sl["t"][mine], ref[mine, 0]) and beq(sl["normal"][mine], ref[mine, 4:7])
        if d["geom_ints"][gi][0] == 3 and d.get("textures"):
            assert beq(sl["texcoord"][mine], ref[mine, 7:9])
        pinned += int(mine.sum())
        off += len(ref)
    # (the cottage is scaled by 0.02 and the reference compares a mesh's OBJECT-space distance with the other geoms' world-space ones
    # -- src/intersections.h:233, SURVEY 8(a9) -- so a wall 5 units away beats the roof 2 units away: few rays keep the mesh)
    assert pinned > (100 if fixture != "cottage" else 0)
    T.close()


def test_full_size_c5_tile_of_an_8_way_split_against_oracle(gpu_product, O):
    """BASELINE configs[4] names 8 GPUs: what ONE rank of that run computes -- its interleaved 8-row blocks of the 3840x2160 frame of
    the textured-mesh scene with depth of field, split mesh search and BVH on a tile of owned rows -- against the oracle restricted
    to the same rows (its loop over all faces; 16 threads): identical partial frame, foreign rows zero, identical ray counts."""
    from mygpuraytracer_amd import multigpu
    rank = 5
    s, T = make_pair(gpu_product, O, "cornellSpaceship.txt", (3840, 2160), 8, depth_of_field=1, tile_rows=multigpu.TILE_ROWS, tile_rank=rank, tile_world=8)
    O.set_tile(multigpu.TILE_ROWS, rank, 8); O.pt_init()
    try:
        assert T.owned_pixels() == O.pixelcount()
        O.set_threads(16)
        for it in range(1, 4):
            O.iterate(it)
        T.render(1, 3)
        img = T.read_image()
        assert beq(img, O.image())
        assert T.stats()["rays_per_bounce"] == O.live_counts().tolist()
        mine = np.repeat((np.arange(2160) // multigpu.TILE_ROWS) % 8 == rank, 3840)
        assert not img[~mine].any() and img[mine].any()
    finally:
        O.set_threads(1)
        O.set_tile(0, 0, 1)
        T.close()


def test_full_size_c5_20k_mesh_tile_of_an_8_way_split_against_oracle(gpu_product, O):
    """The workload bench.py times as `c5` -- cornellSpaceship20k.txt (the 20 448-triangle stand-in: BVH, four-wide walk, split
    mesh search) at 3840x2160 with depth of field -- compared with the ORACLE at that size, not only with the GPU's own plain loop:
    one rank's interleaved 8-row blocks of the 8-way split (1/8 of 21.9 M ray-bounces x 20 448 triangles for the oracle's loop over all
    faces, src/intersections.h:213-233; 16 threads, about the cost of the 1080p test above).  Identical partial frame, foreign rows
    zero, identical rays per bounce."""
    from mygpuraytracer_amd import multigpu
    rank = 2
    s, T = make_pair(gpu_product, O, "cornellSpaceship20k.txt", (3840, 2160), 8, depth_of_field=1, tile_rows=multigpu.TILE_ROWS, tile_rank=rank, tile_world=8)
    O.set_tile(multigpu.TILE_ROWS, rank, 8); O.pt_init()
    try:
        assert T.owned_pixels() == O.pixelcount()
        O.set_threads(16)
        O.iterate(1)
        T.render(1, 1)
        img = T.read_image()
        assert beq(img, O.image())
        assert T.stats()["rays_per_bounce"] == O.live_counts().tolist()
        mine = np.repeat((np.arange(2160) // multigpu.TILE_ROWS) % 8 == rank, 3840)
        assert not img[~mine].any() and img[mine].any()
    finally:
        O.set_threads(1)
        O.set_tile(0, 0, 1)
        T.close()


RENDER_CASES = [
    ("c1_sphere", "sphere.txt", (64, 64), 4), ("c2_cornell_cache", "cornell.txt", (64, 64), 8), ("c3_glass", "cornellGlass.txt", (96, 54), 12),
    ("c4_obj", "cornellObj.txt", (96, 54), 8), ("c5_dof", "cornellGlass.txt", (96, 54), 8), ("nosort_obj", "cornellObj.txt", (96, 54), 8),
    ("c5_ship", "cornellSpaceship.txt", (96, 54), 8),
]


@pytest.mark.parametrize("tag,scene,res,depth", RENDER_CASES)
def test_images_match_reference_golden(gpu_product, O, tag, scene, res, depth):
    """Accumulated radiance after 1, 2 and 16 iterations equals the golden frames (oracle/_ref: the reference's device functions inside the
    RESTATED bounce loop, glibc libm -- frame-level parity is pinned against that restatement, not against the reference's CUDA build) and the
    oracle's bit for bit; so do the per-bounce ray counts and the 8-bit preview."""
    r = golden("render_%s.npz" % tag)
    aa, dof, sort, cache = map(int, r["options"])
    s, T = make_pair(gpu_product, O, scene, res, depth, antialiasing=aa, depth_of_field=dof, sort_by_material=sort, cache_first_bounce=cache)
    for it in range(1, 17):
        O.iterate(it)
        T.pathtrace(it)
        if it in (1, 2, 16):
            img = T.read_image()
            assert beq(img, O.image())
            assert np.array_equal(img, r["image_spp%d" % it])
            counts = r["counts_it%d" % it]
            got = T.stats()["rays_per_bounce"][: len(counts)]
            if not (cache and not aa and not dof and it > 1):       # cached iterations do not re-trace bounce 0
                assert got == counts.tolist()
            else:
                assert got[1:] == counts.tolist()[1:] and got[0] == 0
    assert np.array_equal(T.pbo(16), r["pbo_spp16"])
    T.close()


def test_full_size_c4_counts_and_properties(gpu_product, O):
    """BASELINE config 4 at full size (cornellObj 1920x1080 depth 8).  Rays per bounce of iteration 1 equal the
    reference's (SURVEY 8(c) anchor); beyond that, size-independent properties: reruns are bit-identical, every pixel
    is written at most once per iteration (image sum = per-iteration sums), batched == one-at-a-time."""
    f = golden("fullres_counts.npz")
    s = gpu_product.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8)
    s.apply_runcuda_camera()
    T = gpu_product.Tracer(s)
    T.pathtrace(1)
    img1 = T.read_image()
    assert T.stats()["rays_per_bounce"] == f["c4_counts"].tolist() == [2073600, 952877, 636958, 492327, 394775, 323704, 268423, 224735]
    assert np.array_equal(img1.sum(axis=0, dtype=np.float64), f["c4_image_sum"])
    assert np.array_equal(img1.reshape(1080, 1920, 3).sum(axis=(1, 2), dtype=np.float64), f["c4_image_rowsum"])
    T.pathtrace(2)
    img12 = T.read_image()
    T.reset_image(); T.pathtrace(2)
    img2 = T.read_image()
    assert np.array_equal(img12, img1 + img2)                       # one add per pixel per iteration, in order
    T.reset_image(); T.render(1, 2)
    assert np.array_equal(T.read_image(), img12)                    # batched enqueue == two calls
    T2 = gpu_product.Tracer(s)
    T2.render(1, 2)
    assert np.array_equal(T2.read_image(), img12)                   # a fresh tracer reproduces it
    assert (img12 >= 0).all() and np.isfinite(img12).all()
    T.close(); T2.close()


@pytest.mark.parametrize("tag,scene,res,depth,opt", [
    ("c1", "sphere.txt", (256, 256), 4, {}),
    ("c2", "cornell.txt", (800, 800), 8, dict(antialiasing=0)),
    ("c3", "cornellGlass.txt", (1920, 1080), 12, {}),
])
def test_full_size_c1_c2_c3_against_the_reference(gpu_product, tag, scene, res, depth, opt):
    """The other BASELINE configs at their full sizes: rays per bounce of iteration 1 and the image's channel and row
    sums (float64 sums of the fp32 frame) equal what the reference build produced (tests/golden/fullres_counts.npz)."""
    f = golden("fullres_counts.npz")
    s = gpu_product.Scene(os.path.join(ROOT, "scenes", scene), res=res, depth=depth)
    s.apply_runcuda_camera()
    with gpu_product.Tracer(s, **opt) as T:
        T.pathtrace(1)
        img = T.read_image()
        want = f[tag + "_counts"].tolist()               # the reference's loop stops at the first empty bounce
        got = T.stats()["rays_per_bounce"]
        assert got[:len(want)] == want and not any(got[len(want):])
        assert np.array_equal(img.sum(axis=0, dtype=np.float64), f[tag + "_image_sum"])
        assert np.array_equal(img.reshape(res[1], res[0], 3).sum(axis=(1, 2), dtype=np.float64), f[tag + "_image_rowsum"])
        T.render(2, 9)                                   # iterations 2..10 batched on two streams (C2: from the cache)
        ten = T.read_image()
    with gpu_product.Tracer(s, batch=1, lanes=1, **opt) as T:
        T.render(1, 10)
        assert beq(T.read_image(), ten)


def test_full_size_c5_tree_and_split_equal_the_plain_loop(gpu_product, monkeypatch):
    """BASELINE config 5 at full size (3840x2160, depth 8, AA + DoF, textured 20448-triangle stand-in): minutes for
    the CPU oracle (it runs at 1920x1080 in test_c5_with_the_20k_triangle_mesh_against_oracle, and at full size with the
    320-triangle mesh in test_full_size_c5_against_oracle), so here the size-independent property is checked -- the fast path (BVH, mesh search as a kernel of
    its own, two launch sets in flight) and the reference-shaped path (loop over all faces inside the bounce kernel, one
    iteration at a time) give the same image bits and ray counts.  Small frames of the same scene are checked against
    the oracle in test_sorted_stream_parity / test_stage_parity."""
    s = gpu_product.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship20k.txt"), res=(3840, 2160), depth=8)
    s.apply_runcuda_camera()
    with gpu_product.Tracer(s, depth_of_field=1) as T:
        T.render(1, 2)
        fast, rays = T.read_image(), T.stats()["rays_total"]
    with gpu_product.Tracer(s, depth_of_field=1, no_bvh=1, no_mesh_split=1, batch=1, lanes=1) as T:
        T.render(1, 2)
        assert beq(T.read_image(), fast) and T.stats()["rays_total"] == rays
    assert np.isfinite(fast).all() and (fast >= 0).all() and fast.max() > 0
    # k_mesh walks the four-wide quantised nodes by default (round 3); the binary front-to-back walk gives the same frame
    monkeypatch.setenv("PTX_DEBUG_NO_WIDE_BVH", "1")
    with gpu_product.Tracer(s, depth_of_field=1) as T:
        T.render(1, 2)
        assert beq(T.read_image(), fast) and T.stats()["rays_total"] == rays


def test_tuning_knobs_do_not_change_results(gpu_product, O, monkeypatch):
    """The environment knobs of the library only choose between equivalent execution plans: split mesh search forced on a
    scene whose mesh has no BVH (k_mesh then runs the plain loop), small-mesh loops not spread over lanes, another grid, the
    general bounce kernel instead of the one specialised for the common option set (k_bounce<.., FAST>)."""
    s, T = make_pair(gpu_product, O, "cornellObj.txt", (160, 90), 8)
    for it in (1, 2, 3):
        O.iterate(it)
    want = O.image()
    T.render(1, 3)
    assert beq(T.read_image(), want)
    T.close()
    for var, val in (("PTX_DEBUG_FORCE_SPLIT", "1"), ("PTX_DEBUG_NO_CHUNKS", "1"), ("PTX_DEBUG_WG_PER_CU", "3"), ("PTX_DEBUG_NO_FAST", "1")):
        monkeypatch.setenv(var, val)
        with gpu_product.Tracer(s) as T2:
            T2.render(1, 3)
            assert beq(T2.read_image(), want), var
        monkeypatch.delenv(var)


@pytest.mark.parametrize("scene,res,depth,opt,iters", [("cornellObj.txt", (1920, 1080), 8, {}, 2),                    # C4
                                                       ("cornellGlass.txt", (1920, 1080), 12, {}, 2),                 # C3
                                                       ("cornell.txt", (800, 800), 8, dict(antialiasing=0), 64),      # C2: 64 spp, cache
                                                       ("sphere.txt", (256, 256), 4, {}, 2)])                         # C1
def test_full_size_frames_against_oracle(gpu_product, O, scene, res, depth, opt, iters):
    """Whole frames of BASELINE configs 1-4 at their full sizes against the CPU oracle (its per-path loops on 16 threads,
    a few seconds each): identical accumulated image and rays per bounce -- after two iterations, and for config 2 after the
    64 spp its name gives (one bulk call: 64 iterations in batches from the first-bounce cache)."""
    s, T = make_pair(gpu_product, O, scene, res, depth, **opt)
    O.set_threads(16)
    try:
        for it in range(1, iters + 1):
            O.iterate(it)
    finally:
        O.set_threads(1)
    T.render(1, iters)
    assert beq(T.read_image(), O.image())
    want = O.live_counts().tolist()                      # of the last iteration; the oracle stops at the first empty bounce
    got = T.stats()["rays_per_bounce"]
    if opt.get("antialiasing", 1) == 0:                  # first-bounce cache: later iterations start from the cached bounce-0 stream
        assert got[0] == 0
        got[0] = want[0]
    assert got[:len(want)] == want and not any(got[len(want):])
    T.close()


def test_full_size_c5_against_oracle(gpu_product, O):
    """BASELINE config 5 at full size -- 3840x2160, depth 8, AA + DoF, the textured stand-in mesh (320 triangles, BVH and
    split mesh search on the GPU) -- one whole iteration against the CPU oracle, which loops over all faces for every ray
    like the reference (its per-path loops on 16 threads: same bits, tests/test_oracle_golden.py; about 10 s): identical
    image and rays per bounce, 21.9 M ray-bounces."""
    s, T = make_pair(gpu_product, O, "cornellSpaceship.txt", (3840, 2160), 8, depth_of_field=1)
    O.set_threads(16)
    try:
        O.iterate(1)
    finally:
        O.set_threads(1)
    T.pathtrace(1)
    assert beq(T.read_image(), O.image())
    assert T.stats()["rays_per_bounce"] == O.live_counts().tolist() and sum(O.live_counts().tolist()) > 21_000_000
    T.close()


def test_8k_frame_against_oracle(gpu_product, O):
    """A frame far beyond BASELINE's sizes -- 7680x4320, 33 M primary rays, where the per-iteration buffers leave room for
    one iteration per launch set only (no batching, no lanes: the plain path) -- two iterations against the oracle
    (16 threads, about 7 s): identical image and rays per bounce."""
    s, T = make_pair(gpu_product, O, "cornellObj.txt", (7680, 4320), 8)
    O.set_threads(16)
    try:
        O.iterate(1); O.iterate(2)
    finally:
        O.set_threads(1)
    T.render(1, 2)
    assert beq(T.read_image(), O.image())
    assert T.stats()["rays_per_bounce"] == O.live_counts().tolist() and T.stats()["rays_per_bounce"][0] == 7680 * 4320
    T.close()


def test_c5_with_the_20k_triangle_mesh_against_oracle(gpu_product, O):
    """Config 5's scene with the 20 448-triangle stand-in at 1920x1080 (the oracle's loop over all faces for every ray sets
    the size: 5.5 M ray-bounces x 20 448 triangles, 16 threads, about 20 s): the BVH + split mesh search on the GPU give the image
    and the ray counts of the reference's brute-force loop, bit for bit."""
    s, T = make_pair(gpu_product, O, "cornellSpaceship20k.txt", (1920, 1080), 8, depth_of_field=1)
    O.set_threads(16)
    try:
        O.iterate(1)
    finally:
        O.set_threads(1)
    T.pathtrace(1)
    assert beq(T.read_image(), O.image())
    assert T.stats()["rays_per_bounce"] == O.live_counts().tolist()
    T.close()


@pytest.mark.parametrize("rank", [0, 3, 7])
def test_full_size_tile_of_an_8_way_split_against_oracle(gpu_product, O, rank):
    """What one rank of the 8-GPU run of BASELINE config 4 computes -- its interleaved 8-row blocks of the 1920x1080 frame,
    as a stream of its own, 40 iterations in the batching a rank uses -- against the oracle restricted to the same rows:
    identical partial frame (foreign rows zero) and ray counts."""
    from mygpuraytracer_amd import multigpu
    s, T = make_pair(gpu_product, O, "cornellObj.txt", (1920, 1080), 8, tile_rows=multigpu.TILE_ROWS, tile_rank=rank, tile_world=8)
    O.set_tile(multigpu.TILE_ROWS, rank, 8); O.pt_init()
    try:
        assert T.owned_pixels() == O.pixelcount()
        O.set_threads(16)
        for it in range(1, 41):
            O.iterate(it)
        T.render(1, 40)
        img = T.read_image()
        assert beq(img, O.image())
        assert T.stats()["rays_per_bounce"] == O.live_counts().tolist()
        mine = np.repeat((np.arange(1080) // multigpu.TILE_ROWS) % 8 == rank, 1920)
        assert not img[~mine].any() and img[mine].any()
    finally:
        O.set_threads(1)
        O.set_tile(0, 0, 1)
        T.close()


def test_tile_split_matches_oracle_on_the_same_tile(gpu_product, O):
    """Multi-GPU row tiles: each tile is its own stream (local stream indices), so a tile must equal the oracle run
    on that tile; the tiles together cover every pixel exactly once."""
    res, depth = (96, 64), 8
    covered = np.zeros(res[0] * res[1], bool)
    total = np.zeros((res[0] * res[1], 3), np.float32)
    for rank in range(3):
        s, T = make_pair(gpu_product, O, "cornellObj.txt", res, depth, tile_rows=8, tile_rank=rank, tile_world=3)
        O.set_tile(8, rank, 3); O.pt_init()
        assert T.owned_pixels() == O.pixelcount()
        for it in (1, 2):
            O.iterate(it); T.pathtrace(it)
        img = T.read_image()
        assert beq(img, O.image())
        rows = (np.arange(res[1]) // 8) % 3 == rank
        mask = np.repeat(rows, res[0])
        assert not img[~mask].any()
        assert not (covered & mask).any()
        covered |= mask; total += img
        T.close()
    assert covered.all()
    O.set_tile(0, 0, 1)
    # the assembled frame is a different (equally valid) random sequence than the 1-GPU frame: compare statistically
    s, T = make_pair(gpu_product, O, "cornellObj.txt", res, depth)
    T.render(1, 2)
    one = T.read_image()
    assert abs(float(total.mean()) - float(one.mean())) < 0.05 * float(one.mean()) + 0.02
    T.close()


def test_scripted_camera_orbit_on_a_live_tracer(gpu_product, O):
    """runCuda after a mouse drag (src/main.cpp:105-127): new camera, accumulation restarted, no re-init needed here
    (buffer sizes do not change).  The frame equals the oracle's for the moved camera -- and a cached first bounce from
    the old camera must not survive the move."""
    for scene, opt in (("cornellObj.txt", {}), ("cornell.txt", dict(antialiasing=0))):
        s = gpu_product.Scene(os.path.join(ROOT, "scenes", scene), res=(96, 54), depth=6)
        orb = s.orbit_init()
        s.orbit_events(orb, [])
        with gpu_product.Tracer(s, **opt) as T:
            T.render(1, 3)
            s.orbit_events(orb, [("left", 11.0, -4.0), ("right", 20.0), ("middle", 30.0, 10.0)])
            T.set_camera(s)
            T.reset_image()
            T.render(1, 3)
            d = s.dump()
            O.set_libm(1)
            O.create(d, d["textures"])
            O.set_options(aa=opt.get("antialiasing", 1), dof=0, sort=1, cache=1)
            O.pt_init()
            for it in (1, 2, 3):
                O.iterate(it)
            assert beq(T.read_image(), O.image())


def test_camera_tile_masks_are_supersets(gpu_product, O, tmp_path, monkeypatch):
    """Round 4: the camera-ray bounce tests, per tile of 256 pixels, only the geoms whose conservative screen rectangle reaches the
    tile (host: update_tile_geoms), and tiles that see no geom skip generation, intersection and ranking altogether.  A rectangle that
    is too small would silently turn hits into misses, so: cameras outside and INSIDE the room, looking along and across walls, eye
    a hair's breadth from a wall, geoms behind the eye, very narrow and very wide fields of view, frames whose rows are shorter and
    longer than a tile, a tile split -- every one gives the oracle's image and ray counts, and the image it gives with the masks
    switched off (PTX_DEBUG_NO_TILE_GEOMS).  The moved camera of a live tracer (ptx_set_camera) gets new masks."""
    pt = gpu_product
    import re
    stock = open(os.path.join(ROOT, "scenes", "cornellObj.txt")).read()
    assert re.search(r"CAMERA\n(?:.*\n)*?UP[^\n]*\n", stock)

    def with_camera(eye, look, fovy, res):
        block = "CAMERA\nRES %d %d\nFOVY %g\nITERATIONS 3\nDEPTH 5\nFILE t\nEYE %g %g %g\nLOOKAT %g %g %g\nUP 0 1 0\n" % (tuple(res) + (fovy,) + tuple(eye) + tuple(look))
        return re.sub(r"CAMERA\n(?:.*\n)*?UP[^\n]*\n", block, stock, count=1)
    cams = [((0, 5, 10.5), (0, 5, 0), 45, (200, 120)),          # the stock view
            ((0, 5, 4.0), (0, 5, 0), 70, (300, 37)),            # inside the room, rows longer than a tile
            ((0.5, 9.6, 0.3), (0.2, 0.0, -0.4), 60, (96, 80)),  # under the light, looking down: geoms behind the eye
            ((-4.95, 2.0, 0.0), (4.0, 6.0, -1.0), 50, (130, 64)),      # a hair from the left wall, looking across
            ((0, 5, 30.0), (0, 5, 0), 4, (257, 40)),            # far away, narrow: the room fills the frame
            ((12.0, 14.0, 14.0), (0, 4, 0), 100, (64, 300)),    # outside, above, very wide; tall frame (rows shorter than a tile)
            ((0.0, 5.0, -4.9), (0.0, 5.0, 5.0), 65, (150, 90))]        # in front of the back wall, looking OUT of the room
    for k, (eye, look, fovy, res) in enumerate(cams):
        s = _scene_from_text(pt, with_camera(eye, look, fovy, res), tmp_path)
        img = _vs_oracle(pt, O, s, iters=2)
        # depth of field: rays leave a lens of radius 0.8 in world x / y towards focus points 11 away in z -- the rectangles are then
        # those of the geoms' focus-plane images widened by the lens (update_tile_geoms); cameras that do not look along z keep all geoms
        img_dof = _vs_oracle(pt, O, s, iters=2, depth_of_field=1)
        with pt.Tracer(s, tile_rows=8, tile_rank=1, tile_world=3) as T:
            T.render(1, 2)
            tile_img, tile_rays = T.read_image(), T.stats()["rays_per_bounce"]
        monkeypatch.setenv("PTX_DEBUG_NO_TILE_GEOMS", "1")
        try:
            with pt.Tracer(s) as T:
                T.render(1, 2)
                assert beq(T.read_image(), img), k
            with pt.Tracer(s, depth_of_field=1) as T:
                T.render(1, 2)
                assert beq(T.read_image(), img_dof), k
            with pt.Tracer(s, tile_rows=8, tile_rank=1, tile_world=3) as T:
                T.render(1, 2)
                assert beq(T.read_image(), tile_img) and T.stats()["rays_per_bounce"] == tile_rays, k
        finally:
            monkeypatch.delenv("PTX_DEBUG_NO_TILE_GEOMS")
    # the spaceship scene (split mesh search: tiles without a geom skip generation and intersection in pass 1) with depth of field
    ship = pt.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship.txt"), res=(320, 180), depth=5)
    ship.apply_runcuda_camera()
    _vs_oracle(pt, O, ship, iters=2, depth_of_field=1)
    _vs_oracle(pt, O, ship, iters=2)
    # one live tracer walked through all the cameras of one frame size: the masks follow ptx_set_camera
    s0 = _scene_from_text(pt, with_camera(cams[0][0], cams[0][1], cams[0][2], (160, 100)), tmp_path)
    with pt.Tracer(s0) as T:
        for eye, look, fovy, _ in cams:
            s = _scene_from_text(pt, with_camera(eye, look, fovy, (160, 100)), tmp_path)
            T.set_camera(s)
            T.reset_image()
            T.render(1, 2)
            got = T.read_image()
            with pt.Tracer(s) as T2:
                T2.render(1, 2)
                assert beq(got, T2.read_image())


def test_error_paths(gpu_product):
    pt = gpu_product
    s = pt.Scene(os.path.join(ROOT, "scenes", "sphere.txt"), res=(32, 32), depth=0)
    with pytest.raises(pt.PathTracerError):
        pt.Tracer(s)
    s.set_trace_depth(4)
    with pytest.raises(pt.PathTracerError):
        pt.Tracer(s, bounding_box=1)
    with pytest.raises(pt.PathTracerError):
        pt.Tracer(s, tile_rows=0, tile_rank=0, tile_world=2)
    T = pt.Tracer(s)
    s2 = pt.Scene(os.path.join(ROOT, "scenes", "sphere.txt"), res=(64, 32), depth=4)
    with pytest.raises(pt.PathTracerError):
        T.set_camera(s2)                       # resolution is fixed at init, as in the reference
    T.close(); T.close()                       # pathtraceFree is idempotent


@pytest.mark.parametrize("tile", [None, (16, 1, 4)])
@pytest.mark.parametrize("batch", [2, 5, 8])
def test_batched_iterations_identical(gpu_product, batch, tile):
    """ptx_render traces several iterations per launch set as independent segments (what keeps small multi-GPU tiles
    efficient): the accumulated image and the ray totals are bit-identical to one iteration at a time."""
    pt = gpu_product
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellGlass.txt"), res=(160, 120), depth=10)
    s.apply_runcuda_camera()
    kw = {}
    if tile:
        kw = dict(tile_rows=tile[0], tile_rank=tile[1], tile_world=tile[2])
    with pt.Tracer(s, batch=1, **kw) as A, pt.Tracer(s, batch=batch, **kw) as B:
        A.render(1, 11); B.render(1, 11)
        a, b = A.read_image(), B.read_image()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert A.stats()["rays_total"] == B.stats()["rays_total"]
        A.render(12, 3); B.render(12, 3)                               # a second call continues the accumulation
        assert np.array_equal(A.read_image().view(np.uint32), B.read_image().view(np.uint32))


@pytest.mark.parametrize("scene,opt", [("cornell.txt", dict(antialiasing=0)), ("cornellObj.txt", {}), ("cornellObj.txt", dict(antialiasing=0, apps_variant=1))])
def test_two_launch_sets_in_flight_identical(gpu_product, scene, opt):
    """Default tracer (batches of iterations taking turns over three streams, gathers chained in iteration order;
    with AA off also the batched first-bounce cache: every iteration of a batch starts from the one cached bounce-0
    stream) against one iteration at a time on one stream: same image bits, same ray totals, over several calls."""
    pt = gpu_product
    s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=(200, 120), depth=7)
    s.apply_runcuda_camera()
    os.environ["PTX_DEBUG_SPLIT_MIN"] = "1"       # D: runs shorter than lanes x batch are cut into one launch set per lane
    try:
        D = pt.Tracer(s, **opt)
    finally:
        del os.environ["PTX_DEBUG_SPLIT_MIN"]
    with pt.Tracer(s, batch=1, lanes=1, **opt) as A, pt.Tracer(s, **opt) as B, pt.Tracer(s, lanes=1, **opt) as C, D:
        for first, count in ((1, 37), (38, 5), (43, 20)):
            A.render(first, count); B.render(first, count); C.render(first, count); D.render(first, count)
            a = A.read_image()
            assert beq(a, B.read_image()) and beq(a, C.read_image()) and beq(a, D.read_image())
            assert A.stats()["rays_total"] == B.stats()["rays_total"] == C.stats()["rays_total"] == D.stats()["rays_total"]
        B.reset_image(); A.reset_image()
        A.render(5, 19); B.render(5, 19)                               # cache refilled by an iteration other than 1
        assert beq(A.read_image(), B.read_image())


@pytest.mark.parametrize("scene,opt,res", [("cornellObj.txt", {}, (200, 120)), ("cornell.txt", dict(antialiasing=0), (160, 160)),
                                           ("cornellSpaceship.txt", dict(depth_of_field=1), (96, 54)),
                                           ("cornellObj.txt", dict(tile_rows=8, tile_rank=1, tile_world=3), (160, 96)),
                                           ("cornellGlass.txt", dict(apps_variant=1, sort_by_material=0), (128, 72))])
def test_render_ahead_is_invisible(gpu_product, scene, opt, res):
    """ptx_set_render_ahead: one ptx_iterate per call (the reference's pathtrace(iter) loop) served from batches traced in
    the background gives, after EVERY call, the bits, ray counts and iteration count of the plain call-by-call tracer --
    across batch boundaries, jumps in the iteration number, a repeated iteration, setting the same camera again (no-op),
    a real camera change, a bulk render in between, an image reset, and switching the feature off mid-way."""
    pt = gpu_product
    s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=res, depth=6)
    s.apply_runcuda_camera()
    with pt.Tracer(s, **opt) as A, pt.Tracer(s, **opt) as B:
        B.set_render_ahead(True)

        def both(fn):
            fn(A); fn(B)

        def same():
            assert beq(A.read_image(), B.read_image())
            sa, sb = A.stats(), B.stats()
            assert sa["rays_total"] == sb["rays_total"] and sa["rays_per_bounce"] == sb["rays_per_bounce"] and sa["iterations"] == sb["iterations"]

        for it in range(1, 71):                                  # two and a bit batches of 32
            both(lambda T: T.pathtrace(it))
            if it in (1, 2, 31, 32, 33, 34, 64, 65, 70):
                same()
        assert B.stats()["loop_ms_total"] > 0.0 and B.last_loop_ms() > 0.0
        for it in (200, 201, 202, 202, 203, 150, 151):           # jumps and a repeat
            both(lambda T: T.pathtrace(it))
            same()
        both(lambda T: T.set_camera(s))                          # the same camera again: nothing is dropped, nothing changes
        both(lambda T: T.pathtrace(152))
        same()
        o = s.orbit_init()
        s.orbit_events(o, [("left", 25, -10), ("right", 15)])    # a real camera change
        both(lambda T: T.set_camera(s))
        for it in range(153, 160):
            both(lambda T: T.pathtrace(it))
        same()
        both(lambda T: T.render(160, 9))                         # a bulk call in between
        both(lambda T: T.pathtrace(169))
        both(lambda T: T.pathtrace(170))
        same()
        both(lambda T: T.reset_image())
        for it in range(1, 6):
            both(lambda T: T.pathtrace(it))
            same()
        B.set_render_ahead(False)
        for it in range(6, 10):
            both(lambda T: T.pathtrace(it))
        same()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_render_ahead_random_call_sequences(gpu_product, seed):
    """Random walks over the call surface -- next iteration, a jump, the same camera again, a new camera, a bulk render, a
    strided render, an image reset, a preview, switching render-ahead off and on -- on a plain tracer and on one with
    render-ahead, in step: same image bits and statistics wherever the walk looks."""
    pt = gpu_product
    rng = np.random.default_rng(seed)
    scene = ["cornellObj.txt", "cornell.txt", "cornellGlass.txt"][seed % 3]
    opt = dict(antialiasing=0) if scene == "cornell.txt" else {}
    s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=(96, 72), depth=5)
    s.apply_runcuda_camera()
    orbit = s.orbit_init()
    with pt.Tracer(s, **opt) as A, pt.Tracer(s, **opt) as B:
        B.set_render_ahead(True)
        it = 0
        for step in range(260):
            r = rng.random()
            if r < 0.62:
                it += 1
                A.pathtrace(it); B.pathtrace(it)
            elif r < 0.68:
                it += int(rng.integers(2, 50))
                A.pathtrace(it); B.pathtrace(it)
            elif r < 0.73:
                A.set_camera(s); B.set_camera(s)
            elif r < 0.78:
                s.orbit_events(orbit, [("left", float(rng.integers(-30, 30)), float(rng.integers(-10, 10)))])
                A.set_camera(s); B.set_camera(s)
            elif r < 0.83:
                n = int(rng.integers(1, 70))
                A.render(it + 1, n); B.render(it + 1, n)
                it += n
            elif r < 0.86:
                n = int(rng.integers(1, 9))
                A.render(it + 1, n, stride=3); B.render(it + 1, n, stride=3)
                it += 3 * n
            elif r < 0.89:
                A.reset_image(); B.reset_image()
            elif r < 0.93:
                assert np.array_equal(A.pbo(max(it, 1)), B.pbo(max(it, 1)))
            elif r < 0.96:
                B.set_render_ahead(bool(rng.integers(0, 2)))
            else:
                sa, sb = A.stats(), B.stats()
                assert sa["rays_total"] == sb["rays_total"] and sa["rays_per_bounce"] == sb["rays_per_bounce"] and sa["iterations"] == sb["iterations"]
                assert beq(A.read_image(), B.read_image())
        assert beq(A.read_image(), B.read_image())
        assert A.stats()["rays_total"] == B.stats()["rays_total"]


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_call_sequences_across_launch_plans(gpu_product, seed):
    """The same random walk of render / strided render / single iterations / resets on three execution plans -- one
    iteration at a time on one stream, the default (batches on three streams, short runs cut per lane), and an odd one
    (batches of 5 on four streams) -- ends in the same bits and counts at every look."""
    pt = gpu_product
    rng = np.random.default_rng(100 + seed)
    scene, opt = [("cornellObj.txt", {}), ("cornell.txt", dict(antialiasing=0)), ("cornellSpaceship.txt", dict(depth_of_field=1)),
                  ("cornellGlass.txt", dict(sort_by_material=0))][seed]
    s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=(120, 80), depth=6)
    s.apply_runcuda_camera()
    with pt.Tracer(s, batch=1, lanes=1, **opt) as A, pt.Tracer(s, **opt) as B, pt.Tracer(s, batch=5, lanes=4, **opt) as C:
        it = 0
        for step in range(60):
            r = rng.random()
            if r < 0.45:
                n = int(rng.integers(1, 120))
                for T in (A, B, C):
                    T.render(it + 1, n)
                it += n
            elif r < 0.6:
                n, stride = int(rng.integers(1, 30)), int(rng.integers(2, 5))
                for T in (A, B, C):
                    T.render(it + 1, n, stride=stride)
                it += n * stride
            elif r < 0.8:
                it += 1
                for T in (A, B, C):
                    T.pathtrace(it)
            elif r < 0.85:
                for T in (A, B, C):
                    T.reset_image()
            else:
                a = A.read_image()
                assert beq(a, B.read_image()) and beq(a, C.read_image())
                assert A.stats()["rays_total"] == B.stats()["rays_total"] == C.stats()["rays_total"]
        a = A.read_image()
        assert beq(a, B.read_image()) and beq(a, C.read_image())
        assert A.stats()["rays_per_bounce"] == B.stats()["rays_per_bounce"] == C.stats()["rays_per_bounce"]


def test_strided_render_and_checkpoint_resume(gpu_product, tmp_path):
    """ptx_render_strided traces exactly the iterations it names (each equal to that iteration traced alone), with and
    without batching; a checkpoint written mid-way and resumed in a fresh tracer ends bit-identical to the straight run."""
    s = gpu_product.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(96, 54), depth=6)
    s.apply_runcuda_camera()
    with gpu_product.Tracer(s) as A, gpu_product.Tracer(s, batch=1) as B:
        A.render(2, 5, stride=3)                                       # 2, 5, 8, 11, 14 as one batched launch set
        for it in (2, 5, 8, 11, 14):
            B.render(it, 1)
        assert beq(A.read_image(), B.read_image())
        assert A.stats()["rays_total"] == B.stats()["rays_total"]
    with gpu_product.Tracer(s) as A:
        A.render(1, 12)
        straight = A.read_image()
    ck = str(tmp_path / "half.ckpt")
    with gpu_product.Tracer(s) as A:
        A.render(1, 7)
        A.save_checkpoint(ck, 7)
    with gpu_product.Tracer(s) as A:
        done = A.load_checkpoint(ck)
        assert done == 7
        A.render(done + 1, 12 - done)
        assert beq(A.read_image(), straight)
    # the headless driver writes and resumes the same file format
    import subprocess
    exe = os.path.join(ROOT, "mygpuraytracer_amd", "mi355x_pathtrace")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "mygpuraytracer_amd", "csrc"), "headless"])
    base = [exe, os.path.join(ROOT, "scenes", "cornellObj.txt"), "--res", "96", "54", "--depth", "6"]
    ck2 = str(tmp_path / "drv.ckpt")
    subprocess.check_call(base + ["--iterations", "7", "--out", str(tmp_path / "a"), "--checkpoint", ck2, "--checkpoint-every", "3"])
    out = subprocess.check_output(base + ["--iterations", "12", "--out", str(tmp_path / "b"), "--resume", ck2, "--checkpoint", ck2], text=True)
    assert "Resumed" in out and "at 7 of 12" in out
    with gpu_product.Tracer(s) as A:
        assert A.load_checkpoint(ck2) == 12
        assert beq(A.read_image(), straight)


def test_headless_driver(gpu_product, tmp_path):
    """mi355x_pathtrace = the reference's main.cpp without the window: same scene file, prints the timer sum, writes
    <prefix>.<utc>.<n>samp.png mirrored in x like saveImage; its fp32 frame equals the library's."""
    import glob
    import subprocess
    exe = os.path.join(ROOT, "mygpuraytracer_amd", "mi355x_pathtrace")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "mygpuraytracer_amd", "csrc"), "headless"])
    out = subprocess.check_output([exe, os.path.join(ROOT, "scenes", "cornellObj.txt"), "--res", "64", "48", "--depth", "5",
                                   "--iterations", "3", "--out", str(tmp_path / "img"), "--pfm", "--hdr"], text=True)
    assert "time: " in out and "Saved" in out
    pfm = glob.glob(str(tmp_path / "img.*.3samp.pfm"))
    png = glob.glob(str(tmp_path / "img.*.3samp.png"))
    assert len(pfm) == 1 and len(png) == 1
    raw = open(pfm[0], "rb").read()
    header_end = raw.index(b"-1.0\n") + 5
    frame = np.frombuffer(raw[header_end:], np.float32).reshape(48, 64, 3)[::-1]
    s = gpu_product.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(64, 48), depth=5)
    s.apply_runcuda_camera()
    with gpu_product.Tracer(s) as T:
        T.render(1, 3)
        want = T.read_image().reshape(48, 64, 3) / np.float32(3)
    assert np.array_equal(frame, want)
    # the C++ veneer's own loop -- pathtrace(pbo, frame, iter) once per iteration, as runCuda does -- with and without
    # render-ahead writes the same frame as the bulk run (40 iterations: past one batch of 32)
    frames = {}
    for tag, extra in (("bulk", []), ("percall", ["--per-call"]), ("percall_plain", ["--per-call", "--no-render-ahead"])):
        o2 = subprocess.check_output([exe, os.path.join(ROOT, "scenes", "cornellObj.txt"), "--res", "64", "48", "--depth", "5",
                                      "--iterations", "40", "--out", str(tmp_path / tag), "--pfm"] + extra, text=True)
        assert "time: " in o2 and float(o2.split("time: ")[1].split()[0]) > 0.0
        frames[tag] = open(glob.glob(str(tmp_path / (tag + ".*.40samp.pfm")))[0], "rb").read()
    assert frames["bulk"] == frames["percall"] == frames["percall_plain"]
    # --hdr = saveHDR: the mirrored mean frame as RGBE (byte parity of the encoder itself: test_abi.py, CPU)
    hdr = open(glob.glob(str(tmp_path / "img.*.3samp.hdr"))[0], "rb").read()
    head = b"#?RADIANCE\n# Written by stb_image_write.h\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=          1.0000000000000\n\n-Y 48 +X 64\n"
    assert hdr.startswith(head)
    pos, rows = len(head), []
    for _ in range(48):
        assert hdr[pos:pos + 4] == bytes([2, 2, 0, 64])
        pos += 4
        planes = []
        for _c in range(4):
            line = b""
            while len(line) < 64:
                n = hdr[pos]
                if n > 128:
                    line += bytes([hdr[pos + 1]]) * (n - 128); pos += 2
                else:
                    line += hdr[pos + 1:pos + 1 + n]; pos += 1 + n
            assert len(line) == 64
            planes.append(np.frombuffer(line, np.uint8))
        rows.append(np.stack(planes, 1))
    assert pos == len(hdr)
    rgbe = np.stack(rows).astype(np.float64)
    dec = rgbe[..., :3] * np.exp2(rgbe[..., 3:] - 136.0)
    mirrored = want[:, ::-1].astype(np.float64)
    assert np.all(np.abs(dec - mirrored) <= mirrored.max(axis=2, keepdims=True) / 128 + 1e-30)
    # the same with main.cpp's mouse scripted (--orbit): equals the Python host side driving the same events
    subprocess.check_call([exe, os.path.join(ROOT, "scenes", "cornellObj.txt"), "--res", "64", "48", "--depth", "5", "--iterations", "2",
                           "--out", str(tmp_path / "orb"), "--pfm", "--orbit", "left:11,-4; right:20;middle:30,10"])
    raw = open(glob.glob(str(tmp_path / "orb.*.2samp.pfm"))[0], "rb").read()
    frame = np.frombuffer(raw[raw.index(b"-1.0\n") + 5:], np.float32).reshape(48, 64, 3)[::-1]
    s = gpu_product.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(64, 48), depth=5)
    orb = s.orbit_init()
    s.orbit_events(orb, [("left", 11.0, -4.0), ("right", 20.0), ("middle", 30.0, 10.0)])
    with gpu_product.Tracer(s) as T:
        T.render(1, 2)
        assert np.array_equal(frame, T.read_image().reshape(48, 64, 3) / np.float32(2))


@pytest.mark.parametrize("tag,scene,res", [("apps_ship", "cornellSpaceship.txt", (96, 54)), ("apps_glass", "cornellGlass.txt", (64, 64))])
@pytest.mark.parametrize("batch", [1, 3])
def test_apps_variant_on_device(gpu_product, O, tag, scene, res, batch):
    """apps_variant = 1 reproduces the apps/src copy of the reference: image (x PI) and albedo AOV equal the golden
    vectors from the reference build and the oracle; the denoised-frame preview path clamps like sendDenosiedImageToPBO."""
    r = golden("render_%s.npz" % tag)
    s, T = make_pair(gpu_product, O, scene, res, 8, apps_variant=1, batch=batch)
    O.set_apps_variant(1); O.pt_init()
    for it in (1, 2, 3):
        O.iterate(it)
    T.render(1, 3)
    img, alb = T.read_image(), T.read_albedo()
    assert beq(img, O.image()) and beq(alb, O.albedo())
    assert np.array_equal(img, r["image_spp3"]) and np.array_equal(alb, r["albedo"])
    frame = (img / np.float32(3)).astype(np.float32)
    pbo = T.denoised_pbo(frame)
    want = np.clip((frame.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    assert np.array_equal(pbo[:, :3], want) and not pbo[:, 3].any()
    # sendToGPU's own shape: the pbo is device memory (apps/src/pathtrace.cu:673-685); same bytes
    import torch
    dpbo = torch.full((len(frame), 4), 7, dtype=torch.uint8, device="cuda:0")
    T.denoised_pbo_device(frame, dpbo.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(dpbo.cpu().numpy(), pbo)
    T.close()
    O.set_apps_variant(0)


@pytest.mark.parametrize("apps", [False, True])
def test_cpp_veneer_like_main_cpp(gpu_product, tmp_path, apps):
    """tests/veneer_check.cpp: the C++ veneer with the reference's names driven the way src/main.cpp (and apps/src/main.cpp)
    drive pathtrace.h -- Free before Init, one pathtrace(device pbo, frame, iter) per iteration (render-ahead on, 40 > one
    batch), timer(), Free twice -- leaves in state.image / the pbo / state.albedo / sendToGPU's pbo exactly what the C ABI
    gives when driven from here."""
    import subprocess
    pt = gpu_product
    exe = tmp_path / "veneer_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", str(exe),
                           os.path.join(ROOT, "tests", "veneer_check.cpp"),
                           "-L" + os.path.join(ROOT, "mygpuraytracer_amd"), "-lmi355x_pathtracer", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(ROOT, "mygpuraytracer_amd") + ",-rpath,/opt/rocm/lib"])
    W, H, D, N = 64, 48, 5, 40
    scene = os.path.join(ROOT, "scenes", "cornellObj.txt")
    out = subprocess.check_output([str(exe), scene, str(W), str(H), str(D), str(N), str(tmp_path / "v")] + (["apps"] if apps else []), text=True)
    assert float(out.split("time: ")[1].split()[0]) > 0.0
    # src/timer.h:17-100 in full: the caller's own PerformanceTimer around one pathtrace() call -- both misuse exceptions thrown, a GPU
    # and a CPU interval > 0 (events on the tracer's stream: with render-ahead the call's own stream work is the gather, the preview and
    # the read-back, the batch traced ahead runs beside it) -- and timer() still answering with pathtrace's bounce loop (> 0)
    tm = out.split("timers: ")[1].split()
    caught, mine_gpu, mine_cpu, module = int(tm[1]), float(tm[3]), float(tm[5]), float(tm[7])
    assert caught == 2 and mine_gpu > 0.0 and mine_cpu > 0.0 and module > 0.0
    rd = lambda ext, dt: np.frombuffer(open(str(tmp_path / "v") + ext, "rb").read(), dt)
    s = pt.Scene(scene, res=(W, H), depth=D)
    s.apply_runcuda_camera()
    with pt.Tracer(s, apps_variant=1 if apps else 0) as T:
        T.render(1, N)
        img = T.read_image()
        assert np.array_equal(rd(".image", np.float32).reshape(-1, 3), img)
        if not apps:
            assert np.array_equal(rd(".pbo", np.uint8).reshape(-1, 4), T.pbo(N))
        else:
            assert (rd(".pbo", np.uint8) == 0x5a).all()                      # AI_DENOISE: pathtrace leaves the pbo alone
            assert np.array_equal(rd(".albedo", np.float32).reshape(-1, 3), T.read_albedo())
            fake = np.stack([img[:, 0] / np.float32(N) * np.float32(1.5) - np.float32(0.1), img[:, 1] / np.float32(N),
                             img[:, 2] / np.float32(N) * np.float32(3.0)], 1).astype(np.float32)
            assert np.array_equal(rd(".pbo2", np.uint8).reshape(-1, 4), T.denoised_pbo(fake))


def _scene_from_text(pt, text, tmp_path, res=None, depth=None):
    f = tmp_path / "scene.txt"
    f.write_text(text)
    s = pt.Scene(str(f), base_dir=os.path.join(ROOT, "scenes"), res=res, depth=depth)
    s.apply_runcuda_camera()
    return s


def _vs_oracle(pt, O, s, iters=3, **opt):
    d = s.dump()
    O.set_libm(1); O.create(d, d["textures"])
    O.set_options(aa=opt.get("antialiasing", 1), dof=opt.get("depth_of_field", 0), sort=opt.get("sort_by_material", 1), cache=opt.get("cache_first_bounce", 1))
    O.pt_init()
    with pt.Tracer(s, **opt) as T:
        for it in range(1, iters + 1):
            O.iterate(it)
        T.render(1, iters)
        assert beq(T.read_image(), O.image())
        assert T.stats()["rays_per_bounce"][: len(O.live_counts())] == O.live_counts().tolist()
        return T.read_image()


CAMERA_BLOCK = "CAMERA\nRES 64 64\nFOVY 45\nITERATIONS 10\nDEPTH 8\nFILE edge\nEYE 0.0 5 10.5\nLOOKAT 0 5 0\nUP 0 1 0\n\n"
MAT = "MATERIAL %d\nRGB %g %g %g\nSPECEX 0\nSPECRGB %g %g %g\nREFL %g\nREFR %g\nREFRIOR %g\nEMITTANCE %g\n\n"


def test_edge_empty_scene(gpu_product, O, tmp_path):
    """No geometry at all: every ray misses at bounce 0, the image stays black, nothing is stored."""
    s = _scene_from_text(gpu_product, MAT % (0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 5) + CAMERA_BLOCK, tmp_path)
    img = _vs_oracle(gpu_product, O, s)
    assert not img.any()


@pytest.mark.parametrize("res,depth", [((1, 1), 8), ((37, 23), 8), ((257, 3), 3), ((64, 64), 1), ((300, 200), 2)])
def test_edge_ragged_sizes_and_depths(gpu_product, O, res, depth):
    """Frames that are not a multiple of the 256-path tile, a single pixel, and the shortest legal path (depth 1:
    only lights seen directly contribute)."""
    s = gpu_product.Scene(os.path.join(ROOT, "scenes", "cornellGlass.txt"), res=res, depth=depth)
    s.apply_runcuda_camera()
    _vs_oracle(gpu_product, O, s)
    _vs_oracle(gpu_product, O, s, batch=1, sort_by_material=0)


def test_capture_after_the_camera_bounce_was_cached(gpu_product, O):
    """A debug capture switches the record masks off for its own launches; the cached camera bounce it replays was written with them on.
    The bounce that reads the cache must read it the way it was written: iterations 1-3 fill and use the cache, iteration 4 is captured
    after bounce 1 -- the stream equals the oracle's, field by field."""
    s, T = make_pair(gpu_product, O, "cornell.txt", (96, 72), 8, antialiasing=0)
    d = s.dump()
    T.render(1, 3)
    T.debug_capture(1)
    T.pathtrace(4)
    g = T.debug_stream()
    n, paths, isects = oracle_pending_stream(O, 4, 1)
    pend = (isects["t"] > 0) & (d["materials"][isects["materialId"], 10] <= 0)
    assert len(g["pix"]) == int(pend.sum()) > 100
    assert beq(g["pix"], paths["pixelIndex"][pend]) and beq(g["mat"], isects["materialId"][pend])
    for k, nm in enumerate(("dx", "dy", "dz")):
        assert beq(g[nm], paths["direction"][pend][:, k])
    for k, nm in enumerate(("nx", "ny", "nz")):
        assert beq(g[nm], isects["normal"][pend][:, k])
    for k, nm in enumerate(("cr", "cg", "cb")):
        assert beq(g[nm], paths["color"][pend][:, k])
    T.close()


def test_record_masks_with_more_runs_than_the_reader_tells_apart(gpu_product, O, tmp_path, monkeypatch):
    """The next bounce knows by sorted position which records carry a direction and which a normal code: two ranges of positions per
    mask, so ptx_create keeps at most two runs of set bits in either (dir_bins: gaps filled, ntab_bins: runs dropped).  Materials laid
    out so that both rules have three and more runs before that -- diffuse cubes, specular spheres, glass, a specular cube and a diffuse
    sphere alternating -- and some bins empty at some bounces: frames and ray counts equal the oracle's, with the masks, without them,
    one iteration per launch set, and with the sort off (one bin: every record complete)."""
    kinds = [("cube", 0, 0, 0, 5), ("cube", 0, 0, 0, 0), ("sphere", 1, 0, 0, 0), ("cube", 0, 0, 0, 0), ("sphere", 0, 1, 1.5, 0),
             ("cube", 0, 0, 0, 0), ("cube", 1, 0, 0, 0), ("sphere", 0, 0, 0, 0), ("cube", 0, 0, 0, 0), ("sphere", 0, 1, 1.3, 0), ("cube", 0, 0, 0, 0)]
    rng = np.random.default_rng(21)
    text = ""
    for m, (_, refl, refr, ior, emit) in enumerate(kinds):
        rgb = rng.uniform(0.3, 0.95, 3)
        text += MAT % ((m,) + tuple(rgb) + tuple(rgb[::-1]) + (refl, refr, ior, emit))
    text += CAMERA_BLOCK
    text += "OBJECT 0\ncube\nmaterial 0\nTRANS 0 10 0\nROTAT 0 0 0\nSCALE 8 .3 8\n\n"
    text += "OBJECT 1\ncube\nmaterial 1\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 12 .01 12\n\n"
    text += "OBJECT 2\ncube\nmaterial 3\nTRANS 0 5 -5\nROTAT 0 90 0\nSCALE .01 10 10\n\n"
    k = 3
    for m, (typ, *_r) in enumerate(kinds):
        if m in (0, 1, 3):
            continue
        for _ in range(2):
            pos = rng.uniform([-4, 0.8, -4], [4, 8, 2]); rot = rng.uniform(0, 90, 3); sc = rng.uniform(0.8, 2.2, 3)
            text += "OBJECT %d\n%s\nmaterial %d\nTRANS %g %g %g\nROTAT %g %g %g\nSCALE %g %g %g\n\n" % ((k, typ, m) + tuple(pos) + tuple(rot) + tuple(sc))
            k += 1
    s = _scene_from_text(gpu_product, text, tmp_path, res=(160, 120), depth=7)
    d = s.dump()
    mats = np.arange(len(kinds))
    need, coded = records_with_direction(d, mats), records_with_normal_code(d, mats)
    runs = lambda b: int(np.sum(b & ~np.concatenate([[False], b[:-1]])))
    assert runs(need) >= 3 and runs(coded) >= 3                       # what the rules ask for, before ptx_create trims them
    img = _vs_oracle(gpu_product, O, s, iters=4)
    _vs_oracle(gpu_product, O, s, iters=3, batch=1)
    _vs_oracle(gpu_product, O, s, iters=2, sort_by_material=0)
    with gpu_product.Tracer(s) as T:
        T.render(1, 4)
        st = T.stats()
        assert 0 < st["stored_with_normal_code"] < st["stored_paths"] and 0 < st["stored_with_direction"] < st["stored_paths"]
    monkeypatch.setenv("PTX_DEBUG_NO_DIR_SKIP", "1")
    with gpu_product.Tracer(s) as T:
        T.render(1, 4)
        assert beq(T.read_image(), img)


def test_edge_many_materials(gpu_product, O, tmp_path):
    """More material bins (70) than lanes in a wave, mirrors / glass / lights / diffuse interleaved, rotated and scaled
    cubes and spheres: the per-bin ranking, the chunk prefixes and the segment bases all get exercised."""
    rng = np.random.default_rng(12)
    text = ""
    nm = 70
    for m in range(nm):
        kind = m % 7
        rgb = rng.uniform(0.2, 0.95, 3)
        refl, refr, ior, emit = (1, 0, 0, 0) if kind == 3 else (0, 1, 1.3 + 0.01 * m, 0) if kind == 5 else (0, 0, 0, 4 if kind == 0 else 0)
        text += MAT % ((m,) + tuple(rgb) + tuple(rgb[::-1]) + (refl, refr, ior, emit))
    text += CAMERA_BLOCK
    text += "OBJECT 0\ncube\nmaterial 1\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 12 .01 12\n\n"
    text += "OBJECT 1\ncube\nmaterial 0\nTRANS 0 10 0\nROTAT 0 0 0\nSCALE 8 .3 8\n\n"
    for k in range(2, 40):
        typ = "sphere" if k % 2 else "cube"
        pos = rng.uniform([-4.5, 0.5, -4.5], [4.5, 8.5, 2.0])
        rot = rng.uniform(0, 90, 3)
        sc = rng.uniform(0.4, 1.6, 3)
        text += "OBJECT %d\n%s\nmaterial %d\nTRANS %g %g %g\nROTAT %g %g %g\nSCALE %g %g %g\n\n" % ((k, typ, int(rng.integers(0, nm))) + tuple(pos) + tuple(rot) + tuple(sc))
    s = _scene_from_text(gpu_product, text, tmp_path, res=(120, 90), depth=6)
    assert s.num_materials == 70 and s.num_geoms == 40
    _vs_oracle(gpu_product, O, s, iters=4)
    _vs_oracle(gpu_product, O, s, iters=4, batch=1)
    _vs_oracle(gpu_product, O, s, iters=2, depth_of_field=1, antialiasing=0)
    # the same scene on a frame large enough for full grids: 70 bins x 32 groups of 64 workgroups = 2240 entries in the table the run
    # search scans first (several 512-entry trips), 1792 workgroups per launch when one iteration is traced at a time, mostly
    # empty runs (windows that move on many times); the stream after two bounces and the frames, against the oracle
    s.set_resolution(960, 540)
    s.apply_runcuda_camera()
    O.set_threads(16)
    try:
        _vs_oracle(gpu_product, O, s, iters=3, batch=1, lanes=1)
        _vs_oracle(gpu_product, O, s, iters=3)
        d = s.dump()
        O.set_libm(1); O.create(d, d["textures"]); O.set_options(aa=1, dof=0, sort=1, cache=1); O.pt_init()
        with gpu_product.Tracer(s) as T:
            check_sorted_streams(T, O, d, 6)
    finally:
        O.set_threads(1)


def _fuzz_seeds():
    # PT_FUZZ_SEEDS=7-80 widens the fuzz for a one-off soak on a GPU box (the suite's own six seeds stay the default)
    spec = os.environ.get("PT_FUZZ_SEEDS", "1-6")
    lo, _, hi = spec.partition("-")
    return list(range(int(lo), int(hi or lo) + 1))


@pytest.mark.parametrize("seed", _fuzz_seeds())
def test_random_scenes_on_the_tile_path(gpu_product, O, tmp_path, seed):
    """Fuzz of the fast path (<= 32 geoms: candidate masks, pooled pairs, tabulated normals, chunked small meshes, BVH +
    split mesh search when the stand-in ship is in): random counts of arbitrarily rotated and non-uniformly scaled cubes,
    spheres and meshes, every material class, inside a lit box.  Image and rays per bounce equal the oracle's."""
    rng = np.random.default_rng(1000 + seed)
    mats = [(1, 1, 1, 0, 0, 0, 0, 0, 0, 5), (.9, .9, .9, 0, 0, 0, 0, 0, 0, 0), (.8, .3, .3, 0, 0, 0, 0, 0, 0, 0),
            (.3, .8, .3, 0, 0, 0, 0, 0, 0, 0), (.95, .95, .95, .95, .95, .95, 1, 0, 0, 0), (.95, .95, .95, .8, .85, .95, 0, 1, 1.5, 0),
            (.3, .4, .9, 0, 0, 0, 0, 0, 0, 0)]
    text = "".join(MAT % ((m,) + mats[m]) for m in range(len(mats))) + CAMERA_BLOCK
    objs = ["cube\nmaterial 0\nTRANS 0 10 0\nROTAT 0 0 0\nSCALE 4 .3 4", "cube\nmaterial 1\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 11 .01 11",
            "cube\nmaterial 1\nTRANS 0 10 0\nROTAT 0 0 90\nSCALE .01 11 11", "cube\nmaterial 1\nTRANS 0 5 -5\nROTAT 0 90 0\nSCALE .01 11 11",
            "cube\nmaterial 2\nTRANS -5 5 0\nROTAT 0 0 0\nSCALE .01 11 11", "cube\nmaterial 3\nTRANS 5 5 0\nROTAT 0 0 0\nSCALE .01 11 11"]
    n_extra = int(rng.integers(3, 20))
    ship_at = int(rng.integers(0, n_extra)) if seed % 2 == 0 else -1
    for k in range(n_extra):
        pos = rng.uniform([-3.5, 1.0, -3.5], [3.5, 8.0, 3.0])
        rot = rng.uniform(-180, 180, 3)
        sc = rng.uniform(0.5, 2.2, 3)
        kind = int(rng.integers(0, 4))
        if k == ship_at:
            head = "obj\n../models/standin_ship.obj"
        elif kind == 0:
            head = "obj\n../models/cube.obj"
        else:
            head = ("sphere" if kind == 1 else "cube") + "\nmaterial %d" % int(rng.integers(1, len(mats)))
        objs.append(head + "\nTRANS %g %g %g\nROTAT %g %g %g\nSCALE %g %g %g" % (tuple(pos) + tuple(rot) + tuple(sc)))
    text += "".join("OBJECT %d\n%s\n\n" % (i, o) for i, o in enumerate(objs))
    s = _scene_from_text(gpu_product, text, tmp_path, res=(112, 80), depth=7)
    assert s.num_geoms == len(objs) <= 32
    _vs_oracle(gpu_product, O, s, iters=3)
    _vs_oracle(gpu_product, O, s, iters=2, depth_of_field=1)
    _vs_oracle(gpu_product, O, s, iters=2, no_cull=1, batch=1)


@pytest.mark.parametrize("seed", _fuzz_seeds())
def test_random_cameras_see_what_the_oracle_sees(gpu_product, O, tmp_path, seed):
    """Fuzz of the camera-dependent host work (round 4: per-tile geom masks from projected boxes, with and without the lens model):
    a random eye -- inside the lit box, outside it, sometimes a few centimetres from a wall or inside an object's box -- looking at a
    random point with a random field of view at a random frame size, over a handful of random objects.  Image and rays per bounce
    equal the oracle's, with and without depth of field."""
    rng = np.random.default_rng(5000 + seed)
    mats = [(1, 1, 1, 0, 0, 0, 0, 0, 0, 5), (.9, .9, .9, 0, 0, 0, 0, 0, 0, 0), (.8, .3, .3, 0, 0, 0, 0, 0, 0, 0),
            (.3, .8, .3, 0, 0, 0, 0, 0, 0, 0), (.95, .95, .95, .95, .95, .95, 1, 0, 0, 0), (.95, .95, .95, .8, .85, .95, 0, 1, 1.5, 0)]
    text = "".join(MAT % ((m,) + mats[m]) for m in range(len(mats)))
    W, H = int(rng.integers(40, 400)), int(rng.integers(24, 160))
    if rng.random() < 0.5:
        eye = rng.uniform([-4.8, 0.2, -4.8], [4.8, 9.6, 4.8])                 # inside the room
    else:
        eye = rng.uniform([-12, -2, 6], [12, 14, 25])                         # outside, in front of the open side
    if rng.random() < 0.25:
        eye[int(rng.integers(0, 3))] = [(-4.97, 4.97), (0.03, 9.8), (-4.97, 4.97)][int(rng.integers(0, 3))][int(rng.integers(0, 2))]
    look = rng.uniform([-5, 0, -5], [5, 10, 5])
    fovy = float(rng.choice([3.0, 20.0, 45.0, 75.0, 110.0]))
    text += ("CAMERA\nRES %d %d\nFOVY %g\nITERATIONS 10\nDEPTH 6\nFILE fuzz\nEYE %g %g %g\nLOOKAT %g %g %g\nUP 0 1 0\n\n"
             % ((W, H, fovy) + tuple(eye) + tuple(look)))
    objs = ["cube\nmaterial 0\nTRANS 0 10 0\nROTAT 0 0 0\nSCALE 4 .3 4", "cube\nmaterial 1\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 11 .01 11",
            "cube\nmaterial 1\nTRANS 0 10 0\nROTAT 0 0 90\nSCALE .01 11 11", "cube\nmaterial 1\nTRANS 0 5 -5\nROTAT 0 90 0\nSCALE .01 11 11",
            "cube\nmaterial 2\nTRANS -5 5 0\nROTAT 0 0 0\nSCALE .01 11 11", "cube\nmaterial 3\nTRANS 5 5 0\nROTAT 0 0 0\nSCALE .01 11 11"]
    for k in range(int(rng.integers(2, 8))):
        pos = rng.uniform([-3.5, 1.0, -3.5], [3.5, 8.0, 3.0])
        rot = rng.uniform(-180, 180, 3)
        sc = rng.uniform(0.4, 2.5, 3)
        kind = int(rng.integers(0, 4))
        head = "obj\n../models/cube.obj" if kind == 0 else ("obj\n../models/standin_ship.obj" if kind == 3 and k == 0
                                                            else ("sphere" if kind == 1 else "cube") + "\nmaterial %d" % int(rng.integers(1, len(mats))))
        objs.append(head + "\nTRANS %g %g %g\nROTAT %g %g %g\nSCALE %g %g %g" % (tuple(pos) + tuple(rot) + tuple(sc)))
    text += "".join("OBJECT %d\n%s\n\n" % (i, o) for i, o in enumerate(objs))
    s = _scene_from_text(gpu_product, text, tmp_path)
    _vs_oracle(gpu_product, O, s, iters=2)
    _vs_oracle(gpu_product, O, s, iters=2, depth_of_field=1)


def test_split_mesh_search_with_many_meshes_per_ray(gpu_product, O, tmp_path):
    """Five BVH meshes (the stand-in ship, overlapping boxes) and three small ones without a tree among 22 geoms: the split mesh
    search parks a ray once, whatever the number of meshes whose boxes it reaches (a bit per geom; rounds 1-2 handled two
    meshes per scene and fell back to the unsplit kernel beyond), and k_mesh searches them one after the other.  Frames, rays
    per bounce and the sorted streams equal the oracle's; the unsplit kernel gives the same frame."""
    rng = np.random.default_rng(77)
    mats = [(1, 1, 1, 0, 0, 0, 0, 0, 0, 5), (.9, .9, .9, 0, 0, 0, 0, 0, 0, 0), (.8, .3, .3, 0, 0, 0, 0, 0, 0, 0),
            (.3, .8, .3, 0, 0, 0, 0, 0, 0, 0), (.95, .95, .95, .95, .95, .95, 1, 0, 0, 0), (.95, .95, .95, .8, .85, .95, 0, 1, 1.5, 0)]
    text = "".join(MAT % ((m,) + mats[m]) for m in range(len(mats))) + CAMERA_BLOCK
    objs = ["cube\nmaterial 0\nTRANS 0 10 0\nROTAT 0 0 0\nSCALE 4 .3 4", "cube\nmaterial 1\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 11 .01 11",
            "cube\nmaterial 1\nTRANS 0 10 0\nROTAT 0 0 90\nSCALE .01 11 11", "cube\nmaterial 1\nTRANS 0 5 -5\nROTAT 0 90 0\nSCALE .01 11 11",
            "cube\nmaterial 2\nTRANS -5 5 0\nROTAT 0 0 0\nSCALE .01 11 11", "cube\nmaterial 3\nTRANS 5 5 0\nROTAT 0 0 0\nSCALE .01 11 11"]
    for k in range(16):
        pos = rng.uniform([-2.5, 2.0, -2.5], [2.5, 7.0, 2.0])
        rot = rng.uniform(-180, 180, 3)
        if k < 5:
            head, sc = "obj\n../models/standin_ship.obj", rng.uniform(0.8, 1.6, 3)
        elif k < 8:
            head, sc = "obj\n../models/cube.obj", rng.uniform(0.5, 1.5, 3)
        else:
            head, sc = ("sphere" if k % 2 else "cube") + "\nmaterial %d" % int(rng.integers(1, len(mats))), rng.uniform(0.4, 1.2, 3)
        objs.append(head + "\nTRANS %g %g %g\nROTAT %g %g %g\nSCALE %g %g %g" % (tuple(pos) + tuple(rot) + tuple(sc)))
    text += "".join("OBJECT %d\n%s\n\n" % (i, o) for i, o in enumerate(objs))
    s = _scene_from_text(gpu_product, text, tmp_path, res=(160, 96), depth=6)
    assert s.num_geoms == 22
    a = _vs_oracle(gpu_product, O, s, iters=3)
    b = _vs_oracle(gpu_product, O, s, iters=3, no_mesh_split=1, batch=1)
    assert beq(a, b)
    _vs_oracle(gpu_product, O, s, iters=2, depth_of_field=1, lanes=1)
    with gpu_product.Tracer(s, batch=1, lanes=1) as T:
        T.set_kernel_timing(True)
        T.render(1, 2)
        assert T.kernel_times()["k_mesh"][1] > 0, "the scene must take the split mesh search"
    d = s.dump()
    O.set_libm(1); O.create(d, d["textures"]); O.set_options(aa=1, dof=0, sort=1, cache=1); O.pt_init()
    with gpu_product.Tracer(s) as T:
        check_sorted_streams(T, O, d, 6)


def test_large_scene_beyond_the_fast_path_limits(gpu_product, O, tmp_path):
    """A scene past every limit of the fast path -- 70 geoms (> 32: no candidate masks), four BVH meshes (without candidate masks the mesh
    search stays inside the bounce kernel), 45 materials (45 sort bins) -- takes the general code automatically and
    still equals the oracle bit for bit; so does the same scene with the mesh tree switched off."""
    rng = np.random.default_rng(4242)
    kinds = [(1, 1, 1, 0, 0, 0, 0, 0, 0, 5)]
    for m in range(1, 45):
        c = tuple(np.round(rng.uniform(0.2, 0.95, 3), 3))
        r = rng.random()
        kinds.append(c + ((.9, .9, .9, 1, 0, 0, 0) if r < 0.15 else (.9, .9, .9, 0, 1, 1.4, 0) if r < 0.3 else (0, 0, 0, 0, 0, 0, 0)))
    text = "".join(MAT % ((m,) + kinds[m]) for m in range(len(kinds))) + CAMERA_BLOCK
    objs = ["cube\nmaterial 0\nTRANS 0 10 0\nROTAT 0 0 0\nSCALE 4 .3 4", "cube\nmaterial 1\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 11 .01 11",
            "cube\nmaterial 2\nTRANS 0 10 0\nROTAT 0 0 90\nSCALE .01 11 11", "cube\nmaterial 3\nTRANS 0 5 -5\nROTAT 0 90 0\nSCALE .01 11 11",
            "cube\nmaterial 4\nTRANS -5 5 0\nROTAT 0 0 0\nSCALE .01 11 11", "cube\nmaterial 5\nTRANS 5 5 0\nROTAT 0 0 0\nSCALE .01 11 11"]
    ships = {7, 21, 40, 63}
    for k in range(6, 70):
        pos = rng.uniform([-4.0, 0.6, -4.0], [4.0, 8.5, 3.0])
        rot = rng.uniform(-180, 180, 3)
        sc = rng.uniform(0.3, 1.2, 3)
        if k in ships:
            head = "obj\n../models/standin_ship.obj"
        elif k % 9 == 0:
            head = "obj\n../models/cube.obj"
        else:
            head = ("sphere" if k % 2 else "cube") + "\nmaterial %d" % int(rng.integers(1, len(kinds)))
        objs.append(head + "\nTRANS %g %g %g\nROTAT %g %g %g\nSCALE %g %g %g" % (tuple(pos) + tuple(rot) + tuple(sc)))
    text += "".join("OBJECT %d\n%s\n\n" % (i, o) for i, o in enumerate(objs))
    s = _scene_from_text(gpu_product, text, tmp_path, res=(96, 64), depth=6)
    assert s.num_geoms == 70 and s.num_materials >= 45
    a = _vs_oracle(gpu_product, O, s, iters=2)
    b = _vs_oracle(gpu_product, O, s, iters=2, no_bvh=1, batch=1)
    assert beq(a, b)


def test_png_rgba_maps_render_like_the_oracle(gpu_product, O, tmp_path):
    """The stand-in ship with its four maps as 4-channel PNGs (texel stride 4, alpha ignored as in the reference's
    image[(..) * channels + c]): loader -> HIP tracer equals loader -> oracle, and equals the PPM version of the scene
    because the colour bytes are the same."""
    import shutil
    import pngcases
    for d in ("scenes", "models/materials", "textures"):
        os.makedirs(tmp_path / d)
    shutil.copy(os.path.join(ROOT, "models", "standin_ship.obj"), tmp_path / "models")
    mtl = open(os.path.join(ROOT, "models", "materials", "standin_ship.mtl")).read().replace(".ppm", ".png")
    (tmp_path / "models" / "materials" / "standin_ship.mtl").write_text(mtl)
    rng = np.random.default_rng(4)
    for k in ("kd", "ks", "ke", "bump"):
        raw = open(os.path.join(ROOT, "textures", "standin_%s.ppm" % k), "rb").read()
        head, w, h, mx, body = raw.split(b"\n", 3)[0], *raw.split(b"\n", 3)[1].split(), raw.split(b"\n", 3)[2], raw.split(b"\n", 3)[3]
        w, h = int(w), int(h)
        rgb = np.frombuffer(body, np.uint8).reshape(h, w, 3)
        rgba = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)
        (tmp_path / "textures" / ("standin_%s.png" % k)).write_bytes(pngcases.make_png(w, h, 6, 8, rgba.astype(int), rng=rng))
    text = open(os.path.join(ROOT, "scenes", "cornellSpaceship.txt")).read()
    (tmp_path / "scenes" / "ship.txt").write_text(text)
    s = gpu_product.Scene(str(tmp_path / "scenes" / "ship.txt"), res=(96, 54), depth=8)
    s.apply_runcuda_camera()
    assert all(t.shape[2] == 4 for t in s.dump()["textures"].values()) and len(s.dump()["textures"]) == 4
    img = _vs_oracle(gpu_product, O, s, iters=3)
    s3 = gpu_product.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship.txt"), res=(96, 54), depth=8)
    s3.apply_runcuda_camera()
    with gpu_product.Tracer(s3) as T:
        T.render(1, 3)
        assert beq(T.read_image(), img)


def test_long_run_of_batches_on_two_streams_is_still_sequential(gpu_product):
    """900 iterations of a small frame: hundreds of batches alternating between the two streams, gathers chained by
    events.  Any race on the image or on the per-iteration buffers would show as a difference from the same iterations
    traced one at a time on one stream; three batch sizes and uneven calls to vary the interleaving."""
    s = gpu_product.Scene(os.path.join(ROOT, "scenes", "cornellGlass.txt"), res=(80, 56), depth=7)
    s.apply_runcuda_camera()
    with gpu_product.Tracer(s, batch=1, lanes=1) as A:
        A.render(1, 900)
        want, rays = A.read_image(), A.stats()["rays_total"]
    for batch, lanes in ((0, 0), (3, 0), (8, 0), (5, 3), (4, 4)):
        with gpu_product.Tracer(s, batch=batch, lanes=lanes) as B:
            B.render(1, 400); B.render(401, 77); B.render(478, 423)
            assert beq(B.read_image(), want) and B.stats()["rays_total"] == rays, (batch, lanes)


def _boxed_scene_text(nmat, ngeom, seed):
    """`nmat` diffuse materials (0 = light) and `ngeom` small cubes / spheres inside a lit room."""
    rng = np.random.default_rng(seed)
    text = "".join(MAT % ((m,) + ((1, 1, 1, 0, 0, 0, 0, 0, 0, 5) if m == 0 else tuple(np.round(rng.uniform(0.2, 0.95, 3), 3)) + (0, 0, 0, 0, 0, 0, 0)))
                   for m in range(nmat)) + CAMERA_BLOCK
    objs = ["cube\nmaterial 0\nTRANS 0 10 0\nROTAT 0 0 0\nSCALE 4 .3 4", "cube\nmaterial %d\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 11 .01 11" % (1 % nmat),
            "cube\nmaterial %d\nTRANS 0 5 -5\nROTAT 0 90 0\nSCALE .01 11 11" % (2 % nmat)]
    for k in range(3, ngeom):
        pos = rng.uniform([-4.0, 0.6, -4.0], [4.0, 8.5, 3.0])
        objs.append(("sphere" if k % 2 else "cube") + "\nmaterial %d\nTRANS %g %g %g\nROTAT %g %g %g\nSCALE %g %g %g" % (
            (int(rng.integers(0, nmat)),) + tuple(pos) + tuple(rng.uniform(-90, 90, 3)) + tuple(rng.uniform(0.2, 0.8, 3))))
    return text + "".join("OBJECT %d\n%s\n\n" % (i, o) for i, o in enumerate(objs[:ngeom]))


@pytest.mark.parametrize("nmat,ngeom", [(3, 5), (6, 9), (9, 31), (64, 7), (65, 33)])
def test_lds_layout_for_odd_and_even_table_sizes(gpu_product, O, tmp_path, nmat, ngeom):
    """The kernels' dynamic LDS = scene tables + ranking head + record buffer, and the record buffer holds 64-bit LDS
    atomics: whatever the number of materials (bins) and geoms, odd or even, it must start 8-byte aligned (round 1 lost a
    run to a 4-byte-misaligned ds_min_u64 when the head had an odd number of words)."""
    s = _scene_from_text(gpu_product, _boxed_scene_text(nmat, ngeom, 7 * nmat + ngeom), tmp_path, res=(72, 48), depth=5)
    assert s.num_materials == nmat and s.num_geoms == ngeom
    _vs_oracle(gpu_product, O, s, iters=2)
    _vs_oracle(gpu_product, O, s, iters=2, no_cull=1, batch=1)


def test_scene_tables_beyond_the_lds_budget_step_down(gpu_product, O, tmp_path):
    """260 geoms: their tables (58 words each) no longer fit next to the record buffer in a workgroup's LDS, so ptx_create
    leaves them in global memory (plain per-ray loop) instead of failing at the first launch -- same image as the oracle."""
    s = _scene_from_text(gpu_product, _boxed_scene_text(12, 260, 5), tmp_path, res=(48, 32), depth=4)
    assert s.num_geoms == 260
    _vs_oracle(gpu_product, O, s, iters=2)


def test_too_many_material_bins_is_refused_at_create(gpu_product, O, tmp_path):
    """6000 materials with the material sort on would need more LDS for the ranking histogram (48 B per bin) than a workgroup
    can have (160 KB on MI355X): ptx_create says so (PTX_ERR_UNSUPPORTED, with the numbers) instead of a launch failure
    later; with the sort off (one bin) the same scene renders.  2000 materials (113 KB of LDS per workgroup, past the 64 KB
    a launch gets without asking) still run and equal the oracle."""
    s = _scene_from_text(gpu_product, _boxed_scene_text(6000, 6, 9), tmp_path, res=(32, 24), depth=3)
    with pytest.raises(gpu_product.PathTracerError) as e:
        gpu_product.Tracer(s)
    assert "LDS" in str(e.value) and "6000" in str(e.value)
    with gpu_product.Tracer(s, sort_by_material=0) as T:
        T.render(1, 2)
        assert np.isfinite(T.read_image()).all()
    s2 = _scene_from_text(gpu_product, _boxed_scene_text(2000, 6, 9), tmp_path, res=(32, 24), depth=3)
    _vs_oracle(gpu_product, O, s2, iters=2)


def _oracle_tiles(O, s, world, tile_rows, iters, **opt):
    """the frame the oracle assembles from `world` row tiles, each traced as a stream of its own; + total rays"""
    d = s.dump()
    O.set_libm(1); O.create(d, d["textures"])
    O.set_options(aa=opt.get("antialiasing", 1), dof=opt.get("depth_of_field", 0), sort=opt.get("sort_by_material", 1), cache=opt.get("cache_first_bounce", 1))
    total, rays = None, 0
    try:
        for rank in range(world):
            O.set_tile(tile_rows, rank, world); O.pt_init()
            for it in range(1, iters + 1):
                O.iterate(it)
                rays += int(O.live_counts().sum())
            total = O.image().copy() if total is None else total + O.image()      # disjoint rows: exact
    finally:
        O.set_tile(0, 0, 1)
    return total, rays


@pytest.mark.parametrize("no_peer", [False, True])
@pytest.mark.parametrize("res,devices,tile_rows", [((96, 64), [0, 0], 8), ((100, 60), [0, 0, 0], 8), ((64, 37), [0, 0, 0, 0], 4)])
def test_several_devices_behind_the_c_abi(gpu_product, O, monkeypatch, res, devices, tile_rows, no_peer):
    """ptx_multi_*: one process, one tracer per listed device (here the same GPU several times -- what a one-GPU box can
    check), each tracing its interleaved row blocks; ptx_multi_read_image copies the blocks into device[0]'s frame.  Equals
    the oracle's tiles assembled, bit for bit, including frames whose height is not a multiple of the block (cut-off last
    block) and ranks that own one block less; rays summed over the devices.  no_peer: the gather a device takes when peer access to
    device[0] could not be enabled (block-wise linear peer copies instead of one strided copy; PTX_DEBUG_NO_PEER forces it here).
    The calling thread's current device is what it was before every call."""
    pt = gpu_product
    if no_peer:
        monkeypatch.setenv("PTX_DEBUG_NO_PEER", "1")
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=res, depth=6)
    s.apply_runcuda_camera()
    want, rays = _oracle_tiles(O, s, len(devices), tile_rows, 3)
    with pt.MultiTracer(s, devices, tile_rows=tile_rows) as M:
        M.render(1, 2)
        M.pathtrace(3)
        img = M.read_image()
        assert beq(img, want)
        assert M.stats()["rays_total"] == rays and M.stats()["iterations"] == 3
        assert beq(M.read_image(), want)                   # assembling twice changes nothing
    with pt.MultiTracer(s, [0]) as M1, pt.Tracer(s) as T:  # one device: the plain tracer
        M1.render(1, 3); T.render(1, 3)
        assert beq(M1.read_image(), T.read_image())


def test_cpp_veneer_on_several_devices(gpu_product, O, tmp_path):
    """The C++ veneer with pathtraceDevices() = {0, 0, 0}: pathtraceInit / pathtrace(pbo, frame, iter) per iteration / Free as
    in src/main.cpp; state.image after the last call holds the assembled frame = the oracle's three tiles, the device pbo its
    8-bit preview."""
    import subprocess
    pt = gpu_product
    exe = tmp_path / "veneer_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", str(exe),
                           os.path.join(ROOT, "tests", "veneer_check.cpp"),
                           "-L" + os.path.join(ROOT, "mygpuraytracer_amd"), "-lmi355x_pathtracer", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(ROOT, "mygpuraytracer_amd") + ",-rpath,/opt/rocm/lib"])
    W, H, D, N = 64, 44, 5, 30
    scene = os.path.join(ROOT, "scenes", "cornellObj.txt")
    subprocess.check_output([str(exe), scene, str(W), str(H), str(D), str(N), str(tmp_path / "v"), "devices=0,0,0"], text=True)
    s = pt.Scene(scene, res=(W, H), depth=D)
    s.apply_runcuda_camera()
    want, _ = _oracle_tiles(O, s, 3, 8, N)
    got = np.frombuffer(open(str(tmp_path / "v") + ".image", "rb").read(), np.float32).reshape(-1, 3)
    assert beq(got, want)
    with pt.MultiTracer(s, [0, 0, 0]) as M:
        M.render(1, N)
        M.assemble()
        T0 = pt.Tracer.__new__(pt.Tracer)                 # device 0's tracer of the set, for its preview
        T0.lib, T0.h, T0.width, T0.height = M.lib, M.lib.ptx_multi_tracer(M.h, 0), W, H
        pbo = T0.pbo(N)
        T0.h = None
    assert np.array_equal(np.frombuffer(open(str(tmp_path / "v") + ".pbo", "rb").read(), np.uint8).reshape(-1, 4), pbo)


@pytest.mark.parametrize("scene,res,depth,opt", [("cornellObj.txt", (1920, 1080), 8, {}), ("cornellGlass.txt", (640, 360), 12, {}),
                                                 ("cornell.txt", (400, 400), 8, dict(antialiasing=0)), ("cornellObj.txt", (320, 200), 8, dict(tile_rows=8, tile_rank=1, tile_world=3)),
                                                 ("cornellObj.txt", (480, 270), 8, dict(apps_variant=1)), ("cornellObj.txt", (480, 270), 8, dict(depth_of_field=1))])
def test_specialised_and_general_bounce_kernels_agree(gpu_product, monkeypatch, scene, res, depth, opt):
    """k_bounce<.., FAST> (options as compile-time constants, chosen per launch where its assumptions hold) against the general
    kernel on the same iterations: same image, same ray counts -- at the bench's full size, with the first-bounce cache (whose
    filling pass is general, whose later passes are specialised), on a row tile, with the apps variant (general only for the launch
    set that holds iteration 1) and with depth of field (general only for the kernel that generates the camera rays)."""
    s = gpu_product.Scene(os.path.join(ROOT, "scenes", scene), res=res, depth=depth)
    s.apply_runcuda_camera()
    with gpu_product.Tracer(s, **opt) as A:
        A.render(1, 5)
        img, st = A.read_image(), A.stats()
    monkeypatch.setenv("PTX_DEBUG_NO_FAST", "1")
    with gpu_product.Tracer(s, **opt) as B:
        B.render(1, 5)
        assert beq(B.read_image(), img) and B.stats()["rays_total"] == st["rays_total"]


@pytest.mark.parametrize("scene,opt,why", [("cornellObj.txt", dict(sort_by_material=0), "sort_by_material"),
                                             ("cornell.txt", dict(antialiasing=0), "cache-filling"),
                                             ("cornellObj.txt", dict(depth_of_field=1), "depth of field"),
                                             ("cornellObj.txt", dict(batch=1, lanes=1), "radiance buffers"),
                                             ("cornellSpaceship.txt", dict(no_mesh_split=1), "split mesh|textured|BVH")])
def test_specialised_kernel_is_refused_outside_its_preconditions(gpu_product, monkeypatch, scene, opt, why):
    """Round 2's GPU fault came from a build that launched k_bounce<.., FAST> where the values it hard-wires did not
    hold (sort_by_material = 0: one bin on the host, nmats bins in the kernel).  The preconditions now live in ONE host predicate
    (fast_violation) that every launch goes through; asked for the variant regardless (PTX_DEBUG_FORCE_FAST), ptx_render answers
    PTX_ERR_INVALID and names the assumption -- nothing is launched -- and without the request the same tracer renders what the
    general kernel renders."""
    import re
    s = gpu_product.Scene(os.path.join(ROOT, "scenes", scene), res=(96, 64), depth=4)
    s.apply_runcuda_camera()
    with gpu_product.Tracer(s, **opt) as A:
        A.render(1, 3)
        img = A.read_image()
    monkeypatch.setenv("PTX_DEBUG_FORCE_FAST", "1")
    with gpu_product.Tracer(s, **opt) as B:
        with pytest.raises(gpu_product.PathTracerError) as e:
            B.render(1, 3)
        assert re.search("outside its preconditions.*(%s)" % why, str(e.value)), str(e.value)
    monkeypatch.delenv("PTX_DEBUG_FORCE_FAST")
    monkeypatch.setenv("PTX_DEBUG_NO_FAST", "1")
    with gpu_product.Tracer(s, **opt) as C:
        C.render(1, 3)
        assert beq(C.read_image(), img)


def test_forced_specialised_kernel_where_it_applies(gpu_product, monkeypatch):
    """... and where every precondition holds the request changes nothing (C4's option set, any frame size)."""
    s = gpu_product.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(96, 64), depth=4)
    s.apply_runcuda_camera()
    with gpu_product.Tracer(s) as A:
        A.render(1, 6)
        img = A.read_image()
    monkeypatch.setenv("PTX_DEBUG_FORCE_FAST", "1")
    with gpu_product.Tracer(s) as B:
        B.render(1, 6)
        assert beq(B.read_image(), img)


def test_several_devices_apps_variant(gpu_product, O):
    """The apps/src variant (x PI gather, albedo AOV of iteration 1) through ptx_multi_*: every device holds its own rows of the AOV,
    ptx_multi_read_albedo merges them; frame and AOV equal the oracle's tiles assembled."""
    pt = gpu_product
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellGlass.txt"), res=(80, 52), depth=6)
    s.apply_runcuda_camera()
    d = s.dump()
    O.set_libm(1); O.create(d, d["textures"]); O.set_apps_variant(1)
    img = alb = None
    try:
        for rank in range(3):
            O.set_tile(8, rank, 3); O.pt_init()
            for it in (1, 2, 3):
                O.iterate(it)
            img = O.image().copy() if img is None else img + O.image()
            alb = O.albedo().copy() if alb is None else alb + O.albedo()          # disjoint rows, zeros elsewhere
    finally:
        O.set_tile(0, 0, 1); O.set_apps_variant(0)
    with pt.MultiTracer(s, [0, 0, 0], apps_variant=1) as M:
        M.render(1, 3)
        assert beq(M.read_image(), img)
        assert beq(M.read_albedo(), alb)


@pytest.mark.parametrize("scene,opt", [("cornellObj.txt", {}), ("cornell.txt", dict(antialiasing=0)), ("cornellSpaceship.txt", dict(depth_of_field=1))])
def test_small_memory_budget_changes_nothing_but_the_batch(gpu_product, monkeypatch, scene, opt):
    """ptx_create sizes its launch sets to a quarter of what the device has (hipMemGetInfo; round 5: a CPX / NPS partition or a shared GPU
    must not be refused where a smaller batch runs) -- PTX_DEBUG_MEM_BUDGET_MB stands in for a small device.  Iterations per set only
    decide how much is in flight: the frame, the ray counts and the fences are those of the default."""
    pt = gpu_product
    s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=(192, 108), depth=6)
    s.apply_runcuda_camera()
    with pt.Tracer(s, **opt) as T:
        T.render(1, 30)
        want, st = T.read_image(), T.stats()
    monkeypatch.setenv("PTX_DEBUG_MEM_BUDGET_MB", "8")           # 8 MB for all streams in flight: one or two iterations per set
    with pt.Tracer(s, **opt) as T:
        T.render(1, 30)
        assert beq(T.read_image(), want)
        st2 = T.stats()
        assert st2["rays_total"] == st["rays_total"] and st2["fenced"] == 0


def test_sized_stats_never_write_past_what_the_caller_has(gpu_product):
    """ptx_get_stats_sized (round 5: ptx_stats grew in round 4 and will again): a caller compiled against a shorter struct passes ITS
    size and gets that prefix; a longer one gets zeros behind the library's struct; ptx_abi_version / ptx_sizeof_* say what the library has."""
    import ctypes as C
    pt = gpu_product
    s = pt.Scene(os.path.join(ROOT, "scenes", "sphere.txt"), res=(32, 32), depth=3)
    s.apply_runcuda_camera()
    with pt.Tracer(s) as T:
        T.render(1, 3)
        full = T.stats()
        n = C.sizeof(pt.api.Stats)
        assert T.lib.ptx_sizeof_stats() == n and T.lib.ptx_abi_version() == pt.api.ABI_VERSION
        for size in (8 + 64 * 8 + 8, n, n + 64):                 # up to rays_total; exactly; a caller from the future
            buf = (C.c_uint8 * (n + 128))(*([0xA5] * (n + 128)))
            assert T.lib.ptx_get_stats_sized(T.h, buf, size) == 0
            raw = bytes(buf)
            assert raw[size:] == bytes([0xA5]) * (n + 128 - size)                # nothing behind what the caller said it has
            got = pt.api.Stats.from_buffer_copy(raw[:n] if size >= n else raw[:size] + bytes(n - size))
            assert got.bounces == full["bounces"] and int(got.rays_total) == full["rays_total"]
            if size > n:
                assert raw[n:size] == bytes(size - n)


@pytest.mark.parametrize("scene,opt", [("cornellObj.txt", {}), ("cornell.txt", dict(antialiasing=0)), ("cornellSpaceship20k.txt", dict(depth_of_field=1))])
def test_both_forms_of_the_local_index(gpu_product, O, monkeypatch, scene, opt):
    """The chunk-local sorted index is one word per stored path (16-bit slot distance | 16-bit rank) where no workgroup's chunk passes 128
    tiles, two words otherwise (8K frames with few iterations per set); PTX_DEBUG_NO_IDX16 forces the two-word form.  Same order either
    way: frames equal the oracle's in both, with the first-bounce cache (whose stage keeps the form it was written in) and the split mesh search."""
    pt = gpu_product
    s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=(128, 72), depth=6)
    s.apply_runcuda_camera()
    d = s.dump()
    O.set_libm(1); O.create(d, d["textures"])
    O.set_options(aa=opt.get("antialiasing", 1), dof=opt.get("depth_of_field", 0), sort=1, cache=1)
    O.pt_init()
    for it in range(1, 5):
        O.iterate(it)
    frames = []
    for no16 in (False, True):
        if no16:
            monkeypatch.setenv("PTX_DEBUG_NO_IDX16", "1")
        with pt.Tracer(s, **opt) as T:
            T.render(1, 4)
            frames.append(T.read_image())
            assert T.stats()["fenced"] == 0
    assert beq(frames[0], O.image()) and beq(frames[1], O.image())
