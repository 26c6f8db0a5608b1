#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE itself (oracle/_ref/libptref.so = the reference's own headers and
loader compiled from /root/reference; see oracle/Makefile).  Run in the dev container only:

    python tests/golden/make_golden.py

The fixtures are data (inputs + expected outputs); no reference source text is stored.  Everything is seeded, so a
re-run reproduces the files bit for bit.

PROVENANCE -- how much of the reference each fixture really pins (see PROVENANCE below, also readable by tests):
  [direct]   the reference's own function, compiled from its own source file, was CALLED to produce the expected values:
             utilhash / thrust minstd_rand + uniform_real (src/intersections.h:12-20, rocThrust standing in for CUDA Thrust),
             the loader (src/scene.cpp, src/utilities.cpp, vendored tinyobj / stb_image), boxIntersectionTest /
             sphereIntersectionTest / meshIntersectionTest (src/intersections.h), scatterRay (src/interactions.h),
             stb_image_write (src/stb.cpp) for the PNG / HDR byte streams.
  [restated] the expected values come out of oracle/ref_driver.cpp's RESTATEMENT of what is CUDA/GL-bound in
             src/pathtrace.cu and src/main.cpp -- the five kernel bodies as host loops, makeSeededRandomEngine,
             thrust::sort_by_key -> std::stable_sort, thrust::stable_partition -> std::stable_partition, runCuda's camera
             recompute -- calling the [direct] functions inside.  Every whole-frame fixture is of this kind: the reference
             holds no golden vectors of its own and its .cu cannot be compiled here, so frame-level parity is
             "unpinned by the reference" in the task's sense, pinned only against this restatement.
Neither kind reproduces the reference's real CUDA build bit for bit (FMA contraction, CUDA libm): the tolerance against
such an arithmetic is stated and tested in tests/test_fp_tolerance.py.  Consumers: tests/test_oracle_golden.py (CPU, pins the plain-C oracle),
tests/test_loader.py (CPU, pins the product's scene loader), tests/test_gpu_parity.py (GPU).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from cpulibs import RefLib, build_ref, scene_text_with, REFERENCE_ROOT, PATH_DTYPE, ISECT_DTYPE  # noqa: E402

SCENES = ["sphere.txt", "cornell.txt", "cornellGlass.txt", "cornellObj.txt"]

# fixture (file name or prefix) -> "direct" | "restated" | "direct+restated", as defined in the module docstring
PROVENANCE = {
    "rng_kat": "direct",                 # utilhash, minstd_rand, uniform_real called as the reference calls them; the seed formula
                                         # (makeSeededRandomEngine, src/pathtrace.cu:62-66) is restated: 3 lines
    "loader_": "direct",                 # Scene::Scene / loadGeom / loadObj / loadCamera of src/scene.cpp; the post-runCuda camera
                                         # inside the same files is restated (main.cpp needs GL)
    "isect_kat_": "direct",              # boxIntersectionTest / sphereIntersectionTest / meshIntersectionTest themselves
    "jitter_kat": "direct",              # calculateJitteredDirectionHemisphere (src/interactions.h:46-85), dead code in the reference, called
                                         # as it stands; the engine's seed formula is the restated three lines of rng_kat
    "dead_tri_kat": "direct",            # objTriIntersectionTest -> triangleIntersectionLocalTest (src/intersections.h:175-205, 284-315), dead
                                         # code in the reference, called as it stands
    "shade_kat_": "direct+restated",     # scatterRay itself, inside the restated body of shadeFakeMaterial; inputs captured mid-render
    "render_": "restated",               # whole frames, per-bounce counts, sorted streams, 8-bit previews
    "render_apps_": "restated",          # the same with the apps/src deltas (x PI gather, albedo AOV)
    "fullres_counts": "restated",        # per-bounce live counts at BASELINE's sizes
    "png_textures": "direct", "jpeg_textures": "direct", "loader_ngons": "direct",     # vendored stb_image / tinyobj through the loader
    "png_files": "direct", "hdr_files": "direct",                                       # vendored stb_image_write through src/image.cpp's calls
}

MIRROR_SCENE = """MATERIAL 0
RGB         1 1 1
SPECEX      0
SPECRGB     0 0 0
REFL        0
REFR        0
REFRIOR     0
EMITTANCE   5

MATERIAL 1
RGB         .98 .98 .98
SPECEX      0
SPECRGB     0 0 0
REFL        0
REFR        0
REFRIOR     0
EMITTANCE   0

MATERIAL 2
RGB         .9 .9 .9
SPECEX      %s
SPECRGB     .95 .9 .85
REFL        1
REFR        0
REFRIOR     0
EMITTANCE   0

CAMERA
RES         64 64
FOVY        45
ITERATIONS  10
DEPTH       8
FILE        mirror
EYE         0.0 5 10.5
LOOKAT      0 5 0
UP          0 1 0

OBJECT 0
cube
material 0
TRANS       0 10 0
ROTAT       0 0 0
SCALE       6 .3 6

OBJECT 1
cube
material 1
TRANS       0 0 0
ROTAT       0 0 0
SCALE       10 .01 10

OBJECT 2
sphere
material 2
TRANS       -1 4 -1
ROTAT       0 0 0
SCALE       5 5 5

OBJECT 3
cube
material 2
TRANS       3 2 1
ROTAT       10 30 0
SCALE       2 4 2
"""


REPO_ROOT = os.path.dirname(os.path.dirname(HERE))
REPO_SCENES = os.path.join(REPO_ROOT, "scenes")


def ship_text(res=None, depth=None):
    """The stand-in for the reference's (missing) spaceship scene lives in this repo; it is loaded BY THE REFERENCE
    LOADER all the same (textures through stb_image)."""
    sys.path.insert(0, os.path.join(REPO_ROOT, "tools"))
    import make_standin_mesh
    make_standin_mesh.main(128)
    return scene_text_with(open(os.path.join(REPO_SCENES, "cornellSpaceship.txt")).read(), res, depth)


def ref_text(name, res=None, depth=None):
    return scene_text_with(open(os.path.join(REFERENCE_ROOT, "scenes", name)).read(), res, depth)


def random_rays(rng, n, centre, radius):
    """Rays aimed around a geom: random origins outside/inside, random + axis-parallel + degenerate directions."""
    o = centre + rng.normal(size=(n, 3)) * radius * 2.0
    o[: n // 8] = centre + rng.normal(size=(n // 8, 3)) * radius * 0.2          # origins inside
    target = centre + rng.normal(size=(n, 3)) * radius * 0.6
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = n // 16
    axes = np.eye(3)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1))
    d[n // 8: n // 8 + k] = axes                                                 # axis-parallel (zero components)
    d[-8:] *= rng.uniform(0.1, 10.0, (8, 1))                                     # unnormalised directions
    return np.concatenate([o, d], 1).astype(np.float32)


def dead_tri(R):
    """objTriIntersectionTest (src/intersections.h:284-315, with triangleIntersectionLocalTest :175-205) of the reference, called directly
    on the two meshes the reference ships and can load: cube.obj inside cornellObj.txt (12 triangles, rotated and scaled) and
    cottage_obj.obj (486).  Its acceptance test -- the three sub-triangle areas must sum to the triangle's within FLT_EPSILON -- rejects
    most true hits to rounding, so the fixture carries many rays per mesh to hold a useful number of accepted ones."""
    rng = np.random.default_rng(20261005)
    out = {}
    R.load(os.path.join(REFERENCE_ROOT, "scenes", "cornellObj.txt"))
    d = R.dump()
    gi = [i for i in range(len(d["geom_ints"])) if d["geom_ints"][i][0] == 3][0]
    centre = d["geom_trs"][gi][:3].astype(np.float64)
    rays = random_rays(rng, 4096, centre, float(np.max(np.abs(d["geom_trs"][gi][6:9]))) * 0.6 + 0.2)
    out["obj_geom"] = np.int32(gi); out["obj_rays"] = rays; out["obj_out"] = R.obj_tri_test(gi, rays)
    R.load_text(cottage_text())                    # the reference's own cottage_obj.obj through the reference's loader, as `cottage` does
    dc = R.dump()
    gi = [k for k in range(len(dc["geom_ints"])) if len(dc["faces"][k])][0]
    rays = random_rays(rng, 4096, dc["geom_trs"][gi][:3].astype(np.float64) + [0, 1.0, 0], 2.5)
    out["cottage_geom"] = np.int32(gi); out["cottage_rays"] = rays; out["cottage_out"] = R.obj_tri_test(gi, rays)
    np.savez_compressed(os.path.join(HERE, "dead_tri_kat.npz"), **out)
    for k in ("obj", "cottage"):
        print(k, "accepted", int((out[k + "_out"][:, 0] > 0).sum()), "of", len(out[k + "_rays"]))


def jitter(R):
    """calculateJitteredDirectionHemisphere (src/interactions.h:46-85) of the reference, called directly: 4096 random unit normals (all
    three branches of the not-normal choice; a few axis-aligned), iterations across and beyond the 5000 of the scene files, both a
    square and a non-square max_iter."""
    rng = np.random.default_rng(20261006)
    n = 4096
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[:6] = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]]
    seeds = np.stack([rng.integers(1, 6000, n), rng.integers(0, 1920 * 1080, n), rng.integers(0, 9, n)], 1).astype(np.int32)
    seeds[:4, 0] = [1, 70, 4999, 5000]
    out = dict(normals=nrm.astype(np.float32), seeds=seeds)
    for mi in (5000, 64):
        out["dir_%d" % mi] = R.jittered_test(out["normals"], seeds, mi)
    np.savez_compressed(os.path.join(HERE, "jitter_kat.npz"), **out)
    print("jitter_kat.npz:", n, "samples; |d| in", float(np.linalg.norm(out["dir_5000"], axis=1).min()), float(np.linalg.norm(out["dir_5000"], axis=1).max()))


def png_textures(R):
    """PNG maps through the reference's loader (stb_image, vertical flip): the files of tests/pngcases.py and the texels
    the reference hands to pathtraceInit for each of them -> png_textures.npz (file bytes + expected texels)."""
    import tempfile
    import pngcases
    out = {}
    with tempfile.TemporaryDirectory() as root:
        for d in ("scenes", "models/materials", "textures"):
            os.makedirs(os.path.join(root, d))
        with open(os.path.join(root, "models", "q.obj"), "w") as f:
            f.write("mtllib q.mtl\nv 0 0 0\nv 3 0 0\nv 3 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nf 1/1 2/2 3/3\n")
        with open(os.path.join(root, "models", "materials", "q.mtl"), "w") as f:      # all four maps: the reference indexes
            f.write("newmtl a\nKd .1 .2 .3\nKs .4 .5 .6\nNi 1.5\n" +                    # its texture vectors by geom
                    "".join("map_%s ../textures/t.png\n" % k for k in ("Kd", "Ks", "Ke", "Bump")))
        text = open(os.path.join(REPO_SCENES, "sphere.txt")).read() + "\nOBJECT 1\nobj\n../models/q.obj\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n"
        with open(os.path.join(root, "scenes", "s.txt"), "w") as f:
            f.write(text)
        for name, png in pngcases.cases():
            with open(os.path.join(root, "textures", "t.png"), "wb") as f:
                f.write(png)
            R.load(os.path.join(root, "scenes", "s.txt"), cwd=os.path.join(root, "scenes"))
            tex = R.dump()["textures"]
            out["file_" + name] = np.frombuffer(png, np.uint8)
            out["texels_" + name] = tex[(1, 0)]
            assert all(np.array_equal(tex[(1, 0)], tex[(1, k)]) for k in (1, 2, 3))
    np.savez_compressed(os.path.join(HERE, "png_textures.npz"), **out)
    print("png_textures.npz:", len(out) // 2, "files")


def jpeg_textures(R):
    """JPEG maps through the reference's loader (stb_image v2.27, vertical flip): the files of tests/jpegcases.py and the texels
    the reference hands to pathtraceInit for each of them -> jpeg_textures.npz (file bytes + expected texels; None = failed load)."""
    import tempfile
    import jpegcases
    out = {}
    with tempfile.TemporaryDirectory() as root:
        for d in ("scenes", "models/materials", "textures"):
            os.makedirs(os.path.join(root, d))
        with open(os.path.join(root, "models", "q.obj"), "w") as f:
            f.write("mtllib q.mtl\nv 0 0 0\nv 3 0 0\nv 3 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nf 1/1 2/2 3/3\n")
        with open(os.path.join(root, "models", "materials", "q.mtl"), "w") as f:
            f.write("newmtl a\nKd .1 .2 .3\nKs .4 .5 .6\nNi 1.5\n" + "".join("map_%s ../textures/t.jpg\n" % k for k in ("Kd", "Ks", "Ke", "Bump")))
        text = open(os.path.join(REPO_SCENES, "sphere.txt")).read() + "\nOBJECT 1\nobj\n../models/q.obj\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n"
        with open(os.path.join(root, "scenes", "s.txt"), "w") as f:
            f.write(text)
        for name, jpg in jpegcases.cases():
            with open(os.path.join(root, "textures", "t.jpg"), "wb") as f:
                f.write(jpg)
            R.load(os.path.join(root, "scenes", "s.txt"), cwd=os.path.join(root, "scenes"))
            tex = R.dump()["textures"].get((1, 0))
            out["file_" + name] = np.frombuffer(jpg, np.uint8)
            out["texels_" + name] = tex if tex is not None else np.zeros((0, 0, 0), np.uint8)
    np.savez_compressed(os.path.join(HERE, "jpeg_textures.npz"), **out)
    import PIL
    print("jpeg_textures.npz:", len(out) // 2, "files (written with PIL %s)" % PIL.__version__)


def ngon_faces(R):
    """tests/ngoncases.py through the reference's loader (tinyobjloader's ear clipping for polygons with > 4 corners)."""
    import tempfile
    import ngoncases
    with tempfile.TemporaryDirectory() as root:
        for d in ("scenes", "models/materials"):
            os.makedirs(os.path.join(root, d))
        text = ngoncases.obj_text()
        with open(os.path.join(root, "models", "n.obj"), "w") as f:
            f.write(text)
        with open(os.path.join(root, "models", "materials", "cube.mtl"), "w") as f:
            f.write(open(os.path.join(REPO_ROOT, "models", "materials", "cube.mtl")).read())
        with open(os.path.join(root, "scenes", "s.txt"), "w") as f:
            f.write(open(os.path.join(REPO_SCENES, "sphere.txt")).read() + "\nOBJECT 1\nobj\n../models/n.obj\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n")
        R.load(os.path.join(root, "scenes", "s.txt"), cwd=os.path.join(root, "scenes"))
        faces = np.asarray(R.dump()["faces"][1], np.float32)
    np.savez_compressed(os.path.join(HERE, "loader_ngons.npz"), faces=faces, obj=np.frombuffer(text.encode(), np.uint8))
    print("loader_ngons.npz:", len(faces), "triangles")


def hdr_files(so):
    """tests/hdrcases.py through the stbi_write_hdr the reference vendors (what image::saveHDR calls, src/image.cpp:41-45)."""
    import ctypes
    import tempfile
    import hdrcases
    lib = ctypes.CDLL(so)
    lib.stbi_write_hdr.restype = ctypes.c_int
    lib.stbi_write_hdr.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    out = {}
    with tempfile.TemporaryDirectory() as root:
        for name, img in hdrcases.cases().items():
            path = os.path.join(root, name + ".hdr")
            assert lib.stbi_write_hdr(path.encode(), img.shape[1], img.shape[0], 3, img.ctypes.data) == 1
            out[name] = np.frombuffer(open(path, "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "hdr_files.npz"), **out)
    print("hdr_files.npz:", len(out), "files")


def png_files(so):
    """tests/pngwritecases.py through the stbi_write_png the reference vendors (what image::savePNG calls, src/image.cpp:33)."""
    import ctypes
    import tempfile
    import pngwritecases
    lib = ctypes.CDLL(so)
    lib.stbi_write_png.restype = ctypes.c_int
    lib.stbi_write_png.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    out = {}
    with tempfile.TemporaryDirectory() as root:
        for name, img in pngwritecases.cases().items():
            path = os.path.join(root, name + ".png")
            assert lib.stbi_write_png(path.encode(), img.shape[1], img.shape[0], 3, img.ctypes.data, img.shape[1] * 3) == 1
            out[name] = np.frombuffer(open(path, "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "png_files.npz"), **out)
    print("png_files.npz:", len(out), "files")


def cottage_text(res=(96, 54), depth=6):
    """the reference's cornellObj.txt with its cube swapped for its models/cottage_obj.obj (32 triangles + 227 quads = 486
    triangles, the one real mesh the reference ships besides the cube), scaled into the box"""
    text = open(os.path.join(REFERENCE_ROOT, "scenes", "cornellObj.txt")).read()
    assert "../models/cube.obj" in text
    text = text.replace("../models/cube.obj", "../models/cottage_obj.obj")
    head, _ = text.rsplit("TRANS", 1)
    return scene_text_with(head + "TRANS       0.5 1.2 0\nROTAT       0 30 0\nSCALE       .02 .02 .02\n", res, depth)


def cottage(R):
    """The cottage scene as VECTORS: what the reference's loader made of it (geoms, 486 x 15 face floats, materials, camera), the
    reference's meshIntersectionTest on rays around it, and a small render by the restated loop -- so that the GPU tier, which
    cannot read /root/reference, builds the scene from arrays (ptx_create) and is checked on a second real mesh."""
    rng = np.random.default_rng(20261005)
    R.load_text(cottage_text())
    d = R.dump()
    R.apply_runcuda_camera()
    out = dict(geom_ints=d["geom_ints"], geom_trs=d["geom_trs"], geom_mats=d["geom_mats"], materials=d["materials"],
               cam_ints=d["cam_ints"], cam_floats=d["cam_floats"], cam_floats_runcuda=R.dump()["cam_floats"],
               cam_floats_1080p=d["cam_floats"], texture_vector_sizes=d["texture_vector_sizes"])
    for gi, f in enumerate(d["faces"]):
        out["faces_%d" % gi] = f
    np.savez_compressed(os.path.join(HERE, "loader_cottage.npz"), **out)
    gi = [k for k in range(len(d["geom_ints"])) if len(d["faces"][k])][0]
    rays = random_rays(rng, 2048, d["geom_trs"][gi][:3].astype(np.float64) + [0, 1.0, 0], 2.5)
    np.savez_compressed(os.path.join(HERE, "isect_kat_cottage.npz"), **{"rays_%d" % gi: rays, "out_%d" % gi: R.geom_test(gi, rays)})
    R.set_options(aa=1, dof=0, sort=1, cache=1)
    R.pt_init()
    out = dict(options=np.array([1, 0, 1, 1], np.int32))
    for it in (1, 2, 3, 4):
        if it == 1:
            R.pt_generate(1)
            b = 0
            while True:
                n = R.num_paths()
                R.pt_bounce(1, 3)
                out["stream_pix_b%d" % b] = R.paths()["pixelIndex"][:n].copy()
                out["stream_mat_b%d" % b] = R.isects()["materialId"][:n].copy()
                out["stream_t_b%d" % b] = R.isects()["t"][:n].copy()
                if R.pt_bounce(1, 12) == 0:
                    break
                b += 1
            R.pt_final_gather()
        else:
            R.iterate(it)
        if it in (1, 4):
            out["image_spp%d" % it] = R.image()
            out["counts_it%d" % it] = R.live_counts()
    np.savez_compressed(os.path.join(HERE, "render_cottage.npz"), **out)
    print("cottage: geom", gi, "faces", len(d["faces"][gi]), "hits", int((R.geom_test(gi, rays)[:, 0] > 0).sum()), "of", len(rays),
          "counts", out["counts_it1"].tolist())


def main():
    so = build_ref()
    if not so:
        sys.exit("oracle/_ref/libptref.so cannot be built here (no /root/reference)")
    if sys.argv[1:] in (["hdr"], ["png"]):           # only this fixture (the others are unchanged by it)
        (hdr_files if sys.argv[1] == "hdr" else png_files)(so)
        return
    if sys.argv[1:] == ["cottage"]:
        cottage(RefLib(so))
        return
    if sys.argv[1:] == ["dead_tri"]:
        dead_tri(RefLib(so))
        return
    if sys.argv[1:] == ["jitter"]:
        jitter(RefLib(so))
        return
    R = RefLib(so)
    rng = np.random.default_rng(20261004)

    # ---- hash + RNG -------------------------------------------------------------------------------------------
    hin = np.concatenate([np.arange(64), rng.integers(0, 2**32, 192, dtype=np.uint64)]).astype(np.uint32)
    hout = np.array([R.utilhash(int(v)) for v in hin], np.uint32)
    triples = np.stack([rng.integers(1, 5001, 256), rng.integers(0, 3840 * 2160, 256), rng.integers(0, 13, 256)], 1).astype(np.int32)
    triples[:4] = [[1, 0, 0], [1, 12345, 0], [5000, 2073599, 8], [2, 1, 12]]
    raw = np.stack([R.rng_raw(*map(int, t), 4) for t in triples])
    u01 = np.stack([R.rng_uniform(*map(int, t), 0.0, 1.0, 4) for t in triples])
    uaa = np.stack([R.rng_uniform(*map(int, t), -0.5, 0.5, 4) for t in triples])
    np.savez_compressed(os.path.join(HERE, "rng_kat.npz"), hash_in=hin, hash_out=hout, triples=triples, raw=raw, u01=u01, uaa=uaa)

    # ---- loader dumps -----------------------------------------------------------------------------------------
    for name in SCENES:
        R.load(os.path.join(REFERENCE_ROOT, "scenes", name))
        d = R.dump()
        R.apply_runcuda_camera()
        d2 = R.dump()
        R.load_text(ref_text(name, (1920, 1080)))
        d3 = R.dump()
        out = dict(geom_ints=d["geom_ints"], geom_trs=d["geom_trs"], geom_mats=d["geom_mats"], materials=d["materials"],
                   cam_ints=d["cam_ints"], cam_floats=d["cam_floats"], cam_floats_runcuda=d2["cam_floats"],
                   cam_floats_1080p=d3["cam_floats"], texture_vector_sizes=d["texture_vector_sizes"])
        for gi, f in enumerate(d["faces"]):
            out["faces_%d" % gi] = f
        np.savez_compressed(os.path.join(HERE, "loader_%s.npz" % name[:-4]), **out)

    R.load_text(ship_text(), cwd=REPO_SCENES)
    d = R.dump()
    R.apply_runcuda_camera()
    out = dict(geom_ints=d["geom_ints"], geom_trs=d["geom_trs"], geom_mats=d["geom_mats"], materials=d["materials"],
               cam_ints=d["cam_ints"], cam_floats=d["cam_floats"], cam_floats_runcuda=R.dump()["cam_floats"],
               cam_floats_1080p=d["cam_floats"], texture_vector_sizes=d["texture_vector_sizes"])
    for gi, f in enumerate(d["faces"]):
        out["faces_%d" % gi] = f
    for (gi, which), img in d["textures"].items():
        out["tex_%d_%d" % (gi, which)] = img
    np.savez_compressed(os.path.join(HERE, "loader_cornellSpaceship.npz"), **out)
    rays = random_rays(rng, 2048, d["geom_trs"][8][:3].astype(np.float64), 1.6)
    np.savez_compressed(os.path.join(HERE, "isect_kat_cornellSpaceship.npz"), rays_8=rays, out_8=R.geom_test(8, rays))

    # ---- per-geom intersection KATs -----------------------------------------------------------------------------
    for name in ("cornellGlass.txt", "cornellObj.txt"):
        R.load(os.path.join(REFERENCE_ROOT, "scenes", name))
        d = R.dump()
        out = {}
        for gi in range(len(d["geom_ints"])):
            centre = d["geom_trs"][gi][:3].astype(np.float64)
            radius = float(np.max(np.abs(d["geom_trs"][gi][6:9]))) * 0.6 + 0.2
            rays = random_rays(rng, 1024, centre, radius)
            out["rays_%d" % gi] = rays
            out["out_%d" % gi] = R.geom_test(gi, rays)
        np.savez_compressed(os.path.join(HERE, "isect_kat_%s.npz" % name[:-4]), **out)

    # ---- shade / scatterRay KATs: real (path, intersection) pairs captured mid-render -----------------------------
    cases = [("glass", ref_text("cornellGlass.txt", (48, 48), 8)), ("obj", ref_text("cornellObj.txt", (48, 48), 8)),
             ("mirror0", MIRROR_SCENE % "0"), ("mirror20", MIRROR_SCENE % "20.5")]
    cases.append(("ship", ship_text((48, 48), 8)))
    for tag, text in cases:
        R.load_text(text, cwd=REPO_SCENES if tag == "ship" else os.path.join(REFERENCE_ROOT, "scenes"))
        R.apply_runcuda_camera()
        R.pt_init()
        out = {}
        for it in (1, 3):
            R.pt_generate(it)
            for b in range(3):
                n = R.num_paths()
                R.pt_bounce(it, 3)                                     # intersect + sort
                paths, isects = R.paths()[:n], R.isects()[:n]
                R.pt_bounce(it, 12)                                    # shade + partition
                idx = rng.integers(0, 2_000_000, n).astype(np.int32)   # arbitrary stream indices for the KAT itself
                shaded = R.shade(it, b + 1, idx, isects, paths)
                key = "it%d_b%d" % (it, b)
                out[key + "_paths"], out[key + "_isects"], out[key + "_idx"], out[key + "_shaded"] = paths, isects, idx, shaded
                if R.num_paths() == 0:
                    break
        out["scene_text"] = np.frombuffer(text.encode(), np.uint8) if tag.startswith("mirror") else np.zeros(0, np.uint8)
        np.savez_compressed(os.path.join(HERE, "shade_kat_%s.npz" % tag), **out)

    # ---- small full renders: images, per-bounce live counts, per-bounce sorted streams ----------------------------
    configs = [
        ("c1_sphere", ref_text("sphere.txt", (64, 64), 4), dict(aa=1, dof=0, sort=1, cache=1)),
        ("c2_cornell_cache", ref_text("cornell.txt", (64, 64), 8), dict(aa=0, dof=0, sort=1, cache=1)),
        ("c3_glass", ref_text("cornellGlass.txt", (96, 54), 12), dict(aa=1, dof=0, sort=1, cache=1)),
        ("c4_obj", ref_text("cornellObj.txt", (96, 54), 8), dict(aa=1, dof=0, sort=1, cache=1)),
        ("c5_dof", ref_text("cornellGlass.txt", (96, 54), 8), dict(aa=1, dof=1, sort=1, cache=1)),
        ("nosort_obj", ref_text("cornellObj.txt", (96, 54), 8), dict(aa=1, dof=0, sort=0, cache=0)),
        ("mirror20", MIRROR_SCENE % "20.5", dict(aa=1, dof=0, sort=1, cache=1)),
    ]
    configs.append(("c5_ship", ship_text((96, 54), 8), dict(aa=1, dof=1, sort=1, cache=1)))
    for tag, text, opt in configs:
        R.load_text(text, cwd=REPO_SCENES if tag == "c5_ship" else os.path.join(REFERENCE_ROOT, "scenes"))
        R.apply_runcuda_camera()
        R.set_options(**opt)
        R.pt_init()
        out = dict(options=np.array([opt["aa"], opt["dof"], opt["sort"], opt["cache"]], np.int32))
        for it in range(1, 17):
            if it == 1:
                # stream after intersect+sort of every bounce: pixel order, material ids (the RNG-visible permutation)
                R.pt_generate(1)
                b = 0
                while True:
                    n = R.num_paths()
                    R.pt_bounce(1, 3)
                    out["stream_pix_b%d" % b] = R.paths()["pixelIndex"][:n].copy()
                    out["stream_mat_b%d" % b] = R.isects()["materialId"][:n].copy()
                    out["stream_t_b%d" % b] = R.isects()["t"][:n].copy()
                    if R.pt_bounce(1, 12) == 0:
                        break
                    b += 1
                R.pt_final_gather()
            else:
                R.iterate(it)
            if it in (1, 2, 16):
                out["image_spp%d" % it] = R.image()
                out["counts_it%d" % it] = R.live_counts()
        out["pbo_spp16"] = R.pbo(16)
        np.savez_compressed(os.path.join(HERE, "render_%s.npz" % tag), **out)

    # ---- apps/src variant (x PI gather + albedo AOV): textured stand-in scene and the glass scene ------------------
    for tag, text, cwd in [("apps_ship", ship_text((96, 54), 8), REPO_SCENES), ("apps_glass", ref_text("cornellGlass.txt", (64, 64), 8), os.path.join(REFERENCE_ROOT, "scenes"))]:
        R.load_text(text, cwd=cwd)
        R.apply_runcuda_camera()
        R.set_apps_variant(1)
        R.pt_init()
        for it in (1, 2, 3):
            R.iterate(it)
        np.savez_compressed(os.path.join(HERE, "render_%s.npz" % tag), image_spp3=R.image(), albedo=R.albedo(), counts_it3=R.live_counts())

    # ---- full-resolution live counts of iteration 1 (SURVEY 8(c) anchors, regenerated) ---------------------------
    full = {}
    for tag, name, res, depth, opt in [("c2", "cornell.txt", (800, 800), 8, dict(aa=0, dof=0, sort=1, cache=1)),
                                       ("c3", "cornellGlass.txt", (1920, 1080), 12, dict(aa=1, dof=0, sort=1, cache=1)),
                                       ("c4", "cornellObj.txt", (1920, 1080), 8, dict(aa=1, dof=0, sort=1, cache=1)),
                                       ("c1", "sphere.txt", (256, 256), 4, dict(aa=1, dof=0, sort=1, cache=1))]:
        R.load_text(ref_text(name, res, depth))
        R.apply_runcuda_camera()
        R.set_options(**opt)
        R.pt_init()
        R.iterate(1)
        img = R.image()
        full[tag + "_counts"] = R.live_counts()
        full[tag + "_image_sum"] = img.sum(axis=0, dtype=np.float64)
        full[tag + "_image_rowsum"] = img.reshape(res[1], res[0], 3).sum(axis=(1, 2), dtype=np.float64).astype(np.float64)
        print(tag, R.live_counts().tolist(), img.mean(dtype=np.float64))
    np.savez_compressed(os.path.join(HERE, "fullres_counts.npz"), **full)
    png_textures(R)
    jpeg_textures(R)
    ngon_faces(R)
    cottage(R)
    hdr_files(so)
    png_files(so)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
