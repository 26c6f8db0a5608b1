"""CPU: the plain-C oracle (glibc libm mode = the reference's host semantics) against the golden vectors that
tests/golden/make_golden.py produced with the reference's own code.  Bit-exact everywhere."""
import numpy as np
import pytest

from conftest import beq, golden, dump_from_golden
from cpulibs import scene_text_with

SCENES = ["sphere", "cornell", "cornellGlass", "cornellObj"]


@pytest.fixture()
def O(oracle_lib):
    oracle_lib.set_libm(0)
    yield oracle_lib
    oracle_lib.set_libm(0)


def test_hash_and_rng(O):
    g = golden("rng_kat.npz")
    assert beq(np.array([O.utilhash(int(v)) for v in g["hash_in"]], np.uint32), g["hash_out"])
    for k, t in enumerate(g["triples"]):
        t = list(map(int, t))
        assert beq(O.rng_raw(*t, 4), g["raw"][k])
        assert beq(O.rng_uniform(*t, 0.0, 1.0, 4), g["u01"][k])
        assert beq(O.rng_uniform(*t, -0.5, 0.5, 4), g["uaa"][k])
    # SURVEY 8(c) anchor: utilhash((1<<31)|1) ^ utilhash(12345) -> first two u01
    assert np.allclose(O.rng_uniform(1, 12345, 0, 0.0, 1.0, 2), [0.601141393, 0.696248889], rtol=0, atol=1e-9)


@pytest.mark.parametrize("scene", SCENES)
def test_loader_arithmetic(O, scene):
    """TRS -> transform / inverse / inverseTranspose, loadCamera and the runCuda recompute."""
    g = golden("loader_%s.npz" % scene)
    for gi in range(len(g["geom_ints"])):
        assert beq(O.build_transforms(g["geom_trs"][gi]), g["geom_mats"][gi])
    ci, cf = g["cam_ints"], g["cam_floats"]
    assert beq(O.camera_from_loader(int(ci[0]), int(ci[1]), float(cf[16]), cf[0:3], cf[3:6], cf[9:12]), cf)
    assert np.isnan(cf[12:15]).all()          # the reference loader leaves camera.right = NaN (scene.cpp:370)
    cf1080 = g["cam_floats_1080p"]
    assert beq(O.camera_from_loader(1920, 1080, float(cf[16]), cf[0:3], cf[3:6], cf[9:12]), cf1080)
    O.create(dump_from_golden(g, cam="cam_floats"))
    O.apply_runcuda_camera()
    assert beq(O.cam_floats, g["cam_floats_runcuda"])
    # SURVEY 3.1 probe
    assert np.allclose(O.cam_floats[:3], [0, 4.99999952, 10.5], atol=1e-6)


@pytest.mark.parametrize("scene", ["cornellGlass", "cornellObj", "cornellSpaceship", "cottage"])
def test_intersection_kats(O, scene):
    """boxIntersectionTest / sphereIntersectionTest / meshIntersectionTest on 1024 rays per geom (cottage: the reference's own
    486-triangle models/cottage_obj.obj as the reference's loader returned it, 2048 rays around it)."""
    g = golden("loader_%s.npz" % scene)
    k = golden("isect_kat_%s.npz" % scene)
    O.create(dump_from_golden(g))
    for gi in range(len(g["geom_ints"])):
        if "rays_%d" % gi not in k.files:
            continue                          # the spaceship fixture covers its bump-mapped mesh only
        out = O.geom_test(gi, k["rays_%d" % gi])
        ref = k["out_%d" % gi]
        hit = ref[:, 0] > 0
        assert hit.sum() > 50 and (~hit).sum() > 5, "fixture should exercise hits and misses"
        assert beq(out[:, 0], ref[:, 0])
        assert beq(out[hit], ref[hit])


@pytest.mark.parametrize("which,loader", [("obj", "loader_cornellObj.npz"), ("cottage", "loader_cottage.npz")])
def test_dead_triangle_functions(O, which, loader):
    """objTriIntersectionTest -> triangleIntersectionLocalTest (src/intersections.h:175-205, 284-315), dead code in the
    reference (the call is commented out, src/pathtrace.cu:313) and on no path here: the oracle's restatement against what the
    reference's own functions returned for 4096 rays around cube.obj and around cottage_obj.obj ([direct] fixture dead_tri_kat.npz)."""
    k = golden("dead_tri_kat.npz")
    O.create(dump_from_golden(golden(loader)))
    out = O.obj_tri_test(int(k[which + "_geom"]), k[which + "_rays"])
    ref = k[which + "_out"]
    hit = ref[:, 0] > 0
    assert hit.sum() > 500 and (~hit).sum() > 500, "fixture should exercise hits and misses"
    assert beq(out[:, 0], ref[:, 0])
    assert beq(out[hit], ref[hit])


def test_dead_jittered_sampler(O):
    """calculateJitteredDirectionHemisphere (src/interactions.h:46-85), dead code in the reference (JITTERED_SAMPLING 0; the
    block that would call it does not compile) and on no path here: the oracle's restatement against what the reference's own function
    returned for 4096 (normal, iteration, stream index, depth) samples, max_iter 5000 and 64 ([direct] fixture jitter_kat.npz).  glibc
    mode, as the reference's host pass; the portable sin / cos the kernels run stay within 1 ulp of it."""
    k = golden("jitter_kat.npz")
    for mi in (5000, 64):
        O.set_libm(0)
        assert beq(O.jittered_test(k["normals"], k["seeds"], mi), k["dir_%d" % mi])
        O.set_libm(1)
        assert np.abs(O.jittered_test(k["normals"], k["seeds"], mi) - k["dir_%d" % mi]).max() <= 2 * np.finfo(np.float32).eps
        O.set_libm(0)


def _scene_for_shade(O, tag):
    from test_loader import product_dump_from_text      # mirror scenes exist only as text: load with the product loader
    k = golden("shade_kat_%s.npz" % tag)
    if tag == "glass":
        O.create(dump_from_golden(golden("loader_cornellGlass.npz")))
    elif tag == "obj":
        O.create(dump_from_golden(golden("loader_cornellObj.npz")))
    elif tag == "ship":
        O.create(dump_from_golden(golden("loader_cornellSpaceship.npz")))
    else:
        O.create(product_dump_from_text(bytes(k["scene_text"]).decode()))
    return k


@pytest.mark.parametrize("tag", ["glass", "obj", "mirror0", "mirror20", "ship"])
def test_shade_kats(O, tag):
    """shadeFakeMaterial + scatterRay on (path, intersection) pairs captured from reference renders: diffuse,
    refractive (enter / exit / total internal reflection), mirror with exponent 0 and 20.5, OBJ spec/diffuse,
    textured OBJ (Kd / Ks / emissive Ke texels, bump-mapped normals), light, miss."""
    k = _scene_for_shade(O, tag)
    keys = sorted(x[:-6] for x in k.files if x.endswith("_paths"))
    assert keys
    total = 0
    for key in keys:
        out = O.shade(int(key[2]), 1, k[key + "_idx"], k[key + "_isects"], k[key + "_paths"])
        assert beq(out, k[key + "_shaded"]), key
        total += len(out)
    assert total > 2000


RENDERS = ["c1_sphere", "c2_cornell_cache", "c3_glass", "c4_obj", "c5_dof", "nosort_obj", "mirror20", "c5_ship"]


def test_cottage_render(O):
    """The cottage scene from its vector fixture (loader_cottage.npz: the reference loader's geoms, faces, materials, camera at
    96x54 depth 6): sorted streams of iteration 1, image and counts after 1 and 4 iterations as oracle/_ref gave them (reference functions, restated loop)."""
    g, r = golden("loader_cottage.npz"), golden("render_cottage.npz")
    O.create(dump_from_golden(g, cam="cam_floats"))
    O.apply_runcuda_camera()
    assert beq(O.cam_floats, g["cam_floats_runcuda"])
    O.set_options(aa=1, dof=0, sort=1, cache=1)
    O.pt_init()
    O.pt_generate(1)
    b = 0
    while True:
        n = O.num_paths()
        O.pt_bounce(1, 3)
        assert beq(O.paths()["pixelIndex"][:n], r["stream_pix_b%d" % b])
        assert beq(O.isects()["materialId"][:n], r["stream_mat_b%d" % b])
        assert beq(O.isects()["t"][:n], r["stream_t_b%d" % b])
        if O.pt_bounce(1, 12) == 0:
            break
        b += 1
    O.pt_final_gather()
    assert beq(O.image(), r["image_spp1"]) and beq(O.live_counts(), r["counts_it1"])
    for it in (2, 3, 4):
        O.iterate(it)
    assert beq(O.image(), r["image_spp4"]) and beq(O.live_counts(), r["counts_it4"])
RENDER_SCENE = dict(c1_sphere=("sphere", (64, 64), 4), c2_cornell_cache=("cornell", (64, 64), 8), c3_glass=("cornellGlass", (96, 54), 12),
                    c4_obj=("cornellObj", (96, 54), 8), c5_dof=("cornellGlass", (96, 54), 8), nosort_obj=("cornellObj", (96, 54), 8),
                    c5_ship=("cornellSpaceship", (96, 54), 8))


def oracle_for_render(O, tag):
    r = golden("render_%s.npz" % tag)
    if tag == "mirror20":
        from test_loader import product_dump_from_text
        k = golden("shade_kat_mirror20.npz")
        d = product_dump_from_text(bytes(k["scene_text"]).decode(), runcuda=True)
    else:
        scene, res, depth = RENDER_SCENE[tag]
        g = golden("loader_%s.npz" % scene)
        d = dump_from_golden(g, cam="cam_floats")
        ci, cf = d["cam_ints"].copy(), d["cam_floats"]
        d["cam_floats"] = O.camera_from_loader(res[0], res[1], float(cf[16]), cf[0:3], cf[3:6], cf[9:12])
        ci[0], ci[1], ci[3] = res[0], res[1], depth
        d["cam_ints"] = ci
    O.create(d)
    if tag != "mirror20":
        O.apply_runcuda_camera()
    aa, dof, sort, cache = map(int, r["options"])
    O.set_options(aa=aa, dof=dof, sort=sort, cache=cache)
    O.pt_init()
    return r


@pytest.mark.parametrize("tag", RENDERS)
def test_full_renders(O, tag):
    """Whole iterations: accumulated fp32 image after 1, 2 and 16 spp, live counts per bounce, the 8-bit preview,
    and the RNG-visible permutation (pixel order + material ids after every sort of iteration 1)."""
    r = oracle_for_render(O, tag)
    for it in range(1, 17):
        if it == 1:
            O.pt_generate(1)
            b = 0
            while True:
                n = O.num_paths()
                O.pt_bounce(1, 3)
                assert beq(O.paths()["pixelIndex"][:n], r["stream_pix_b%d" % b])
                assert beq(O.isects()["materialId"][:n], r["stream_mat_b%d" % b])
                assert beq(O.isects()["t"][:n], r["stream_t_b%d" % b])
                if O.pt_bounce(1, 12) == 0:
                    break
                b += 1
            O.pt_final_gather()
        else:
            O.iterate(it)
        if it in (1, 2, 16):
            assert beq(O.live_counts(), r["counts_it%d" % it])
            assert beq(O.image(), r["image_spp%d" % it])
    assert beq(O.pbo(16), r["pbo_spp16"])


@pytest.mark.parametrize("tag", ["c4_obj", "c5_ship"])
def test_threads_do_not_change_the_oracle(O, tag):
    """o_set_threads(4) (bench.py's all-cores baseline: the intersect and shade loops under OpenMP) against the golden
    image and counts of the single-thread run: the paths of a stage are independent, so the bits are the same."""
    r = oracle_for_render(O, tag)
    O.set_threads(4)
    try:
        for it in range(1, 3):
            O.iterate(it)
        assert beq(O.live_counts(), r["counts_it2"])
        assert beq(O.image(), r["image_spp2"])
    finally:
        O.set_threads(1)


def test_c1_plumbing_cpu_stream_compaction(O):
    """BASELINE config 1: sphere.txt 256x256 depth 4, 1 spp on the CPU path whose dead-ray compaction is the
    StreamCompaction::CPU scan+scatter (oracle partition_paths); counts and radiance as oracle/_ref gives (reference functions, restated loop)."""
    g = golden("loader_sphere.npz")
    d = dump_from_golden(g, cam="cam_floats")
    cf = d["cam_floats"]
    d["cam_floats"] = O.camera_from_loader(256, 256, float(cf[16]), cf[0:3], cf[3:6], cf[9:12])
    ci = d["cam_ints"].copy(); ci[0], ci[1], ci[3] = 256, 256, 4
    d["cam_ints"] = ci
    O.create(d); O.apply_runcuda_camera(); O.pt_init()
    O.iterate(1)
    f = golden("fullres_counts.npz")
    assert beq(O.live_counts(), f["c1_counts"])
    assert np.array_equal(O.image().sum(axis=0, dtype=np.float64), f["c1_image_sum"])
    assert abs(O.image().mean(dtype=np.float64) - 0.0910186768) < 1e-9      # SURVEY 8(c) anchor


def test_stream_compaction_cpu(O):
    rng = np.random.default_rng(5)
    for n in (1, 2, 255, 256, 257, 4099):
        a = (rng.integers(0, 4, n) * rng.integers(0, 2, n)).astype(np.int32)
        assert np.array_equal(O.sc_scan(a), np.concatenate([[0], np.cumsum(a)[:-1]]).astype(np.int32))
        for with_scan in (False, True):
            out, cnt = O.sc_compact(a, with_scan)
            assert cnt == int((a != 0).sum()) and np.array_equal(out, a[a != 0])


@pytest.mark.parametrize("tag,scene,res", [("apps_ship", "cornellSpaceship", (96, 54)), ("apps_glass", "cornellGlass", (64, 64))])
def test_apps_variant(O, tag, scene, res):
    """The apps/src copy of the reference: finalGather * PI and the albedo AOV of iteration 1."""
    r = golden("render_%s.npz" % tag)
    g = golden("loader_%s.npz" % scene)
    d = dump_from_golden(g, cam="cam_floats")
    cf = d["cam_floats"]
    d["cam_floats"] = O.camera_from_loader(res[0], res[1], float(cf[16]), cf[0:3], cf[3:6], cf[9:12])
    ci = d["cam_ints"].copy(); ci[0], ci[1], ci[3] = res[0], res[1], 8
    d["cam_ints"] = ci
    O.create(d); O.apply_runcuda_camera(); O.set_apps_variant(1); O.pt_init()
    for it in (1, 2, 3):
        O.iterate(it)
    assert beq(O.image(), r["image_spp3"]) and beq(O.albedo(), r["albedo"]) and beq(O.live_counts(), r["counts_it3"])
