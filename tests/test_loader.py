"""CPU: the product's scene loader (C ABI ptx_scene_*, no GPU needed) against loader dumps produced by the
reference's own Scene class (tests/golden/loader_*.npz)."""
import os
import tempfile

import numpy as np
import pytest

from conftest import ROOT, beq, golden

SCENES = ["sphere", "cornell", "cornellGlass", "cornellObj"]


def product_dump_from_text(text, runcuda=False, base_dir=None):
    import mygpuraytracer_amd as pt
    pt.build_library()
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(text)
    try:
        s = pt.Scene(f.name, base_dir=base_dir or os.path.join(ROOT, "scenes"))
    finally:
        os.unlink(f.name)
    if runcuda:
        s.apply_runcuda_camera()
    return s.dump()


@pytest.mark.parametrize("scene", SCENES)
def test_loader_matches_reference_dump(product, scene):
    g = golden("loader_%s.npz" % scene)
    s = product.Scene(os.path.join(ROOT, "scenes", scene + ".txt"))
    d = s.dump()
    for k in ("geom_ints", "geom_trs", "geom_mats", "materials", "cam_ints", "cam_floats"):
        assert beq(d[k], g[k]), k
    for gi, f in enumerate(d["faces"]):
        assert beq(f, g["faces_%d" % gi])
    assert s.image_name == ("sphere" if scene == "sphere" else "cornell")
    s.apply_runcuda_camera()
    assert beq(s.dump()["cam_floats"], g["cam_floats_runcuda"])
    s2 = product.Scene(os.path.join(ROOT, "scenes", scene + ".txt"), res=(1920, 1080))
    assert beq(s2.dump()["cam_floats"], g["cam_floats_1080p"])


def test_textured_standin_scene_matches_reference_loader(product):
    """config 5 stand-in: OBJ with uv, Ni 2.0 and four PPM maps -- geoms, triangles (quad split), material and the
    texels (stb_image's vertical flip) equal what the reference loader produced from the same files."""
    g = golden("loader_cornellSpaceship.npz")
    d = product.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship.txt")).dump()
    for k in ("geom_ints", "geom_trs", "geom_mats", "materials", "cam_floats"):
        assert beq(d[k], g[k]), k
    assert d["geom_ints"][8].tolist() == [3, 6, 288] and list(g["texture_vector_sizes"]) == [9, 9, 9, 9]
    for gi, f in enumerate(d["faces"]):
        assert beq(f, g["faces_%d" % gi])
    assert sorted(d["textures"]) == [(8, 0), (8, 1), (8, 2), (8, 3)]
    for (gi, which), img in d["textures"].items():
        assert np.array_equal(img, g["tex_%d_%d" % (gi, which)])


def test_obj_has_own_material_and_empty_textures(product):
    """cornellObj: the OBJ geom gets a material appended from the first .mtl entry (scene.cpp:221-231) and four
    empty texture slots (the reference's scene-wide texture vectors are one short here: fixture says 6 < 7)."""
    g = golden("loader_cornellObj.npz")
    assert list(g["texture_vector_sizes"]) == [6, 6, 6, 6] and len(g["geom_ints"]) == 7
    d = product.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt")).dump()
    assert d["geom_ints"][6].tolist() == [3, 6, 12] and len(d["materials"]) == 7 and not d["textures"]


def test_line_endings_and_missing_final_newline(product):
    text = open(os.path.join(ROOT, "scenes", "cornell.txt")).read().rstrip("\n")
    base = product_dump_from_text(text)
    for variant in (text.replace("\n", "\r\n"), text.replace("\n", "\r")):
        d = product_dump_from_text(variant)
        assert beq(d["geom_mats"], base["geom_mats"]) and beq(d["materials"], base["materials"]) and beq(d["cam_floats"], base["cam_floats"])


@pytest.mark.parametrize("mutate, needle", [
    (lambda t: t.replace("MATERIAL 1", "MATERIAL 7", 1), "MATERIAL id"),
    (lambda t: t.replace("OBJECT 1", "OBJECT 9", 1), "OBJECT id"),
    (lambda t: t.replace("cube\n", "torus\n", 1), "unknown object type"),
    (lambda t: t.replace("SPECEX      0\n", "\n", 1), "MATERIAL block"),
    (lambda t: t.replace("material 3", "material 42"), "material that does not exist"),
    (lambda t: t[: t.index("// camera")], "no CAMERA"),
])
def test_malformed_scenes_are_rejected(product, mutate, needle):
    text = mutate(open(os.path.join(ROOT, "scenes", "cornell.txt")).read())
    with pytest.raises(product.PathTracerError) as e:
        product_dump_from_text(text)
    assert needle in str(e.value)


def test_missing_files(product):
    with pytest.raises(product.PathTracerError):
        product.Scene("/nonexistent/scene.txt")
    text = open(os.path.join(ROOT, "scenes", "cornellObj.txt")).read().replace("cube.obj", "nope.obj")
    with pytest.raises(product.PathTracerError) as e:
        product_dump_from_text(text)
    assert "nope.obj" in str(e.value)


def test_ppm_textures_and_quad_split(product, tmp_path):
    """OBJ with uv + map_Kd/map_Ke PPM textures: texels arrive flipped vertically (stbi flip), quads split along the
    shorter diagonal."""
    (tmp_path / "scenes").mkdir(); (tmp_path / "models" / "materials").mkdir(parents=True); (tmp_path / "textures").mkdir()
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)           # h=2, w=3
    with open(tmp_path / "textures" / "t.ppm", "wb") as f:
        f.write(b"P6\n3 2\n255\n" + img.tobytes())
    (tmp_path / "models" / "materials" / "q.mtl").write_text("newmtl a\nKd .1 .2 .3\nKs .4 .5 .6\nNi 1.5\nmap_Kd ..\\\\textures\\\\t.ppm\nmap_Ke ../textures/t.ppm\n")
    (tmp_path / "models" / "q.obj").write_text("mtllib q.mtl\nv 0 0 0\nv 3 0 0\nv 3 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\nf 1/1 2/2 3/3\n")
    text = open(os.path.join(ROOT, "scenes", "sphere.txt")).read() + "\nOBJECT 1\nobj\n../models/q.obj\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n"
    (tmp_path / "scenes" / "s.txt").write_text(text)
    d = product.Scene(str(tmp_path / "scenes" / "s.txt")).dump()
    assert d["geom_ints"][1].tolist() == [3, 1, 3]
    f = d["faces"][1]
    # |v2-v0|^2 = 10 == |v3-v1|^2 = 10 -> tie -> [0,1,3],[1,2,3]
    assert f[0].reshape(3, 5)[:, :3].tolist() == [[0, 0, 0], [3, 0, 0], [0, 1, 0]]
    assert f[1].reshape(3, 5)[:, :3].tolist() == [[3, 0, 0], [3, 1, 0], [0, 1, 0]]
    assert f[0].reshape(3, 5)[:, 3:].tolist() == [[0, 0], [1, 0], [0, 1]]
    assert np.allclose(d["materials"][1], [.1, .2, .3, 0, .4, .5, .6, 0, 0, 1.5, 0])
    assert set(d["textures"]) == {(1, 0), (1, 2)}
    assert np.array_equal(d["textures"][(1, 0)], img[::-1])


def _png_scene(tmp_path, png_bytes):
    for d in ("scenes", "models/materials", "textures"):
        os.makedirs(tmp_path / d, exist_ok=True)
    (tmp_path / "textures" / "t.png").write_bytes(png_bytes)
    (tmp_path / "models" / "materials" / "q.mtl").write_text("newmtl a\nKd .1 .2 .3\nKs .4 .5 .6\nNi 1.5\nmap_Kd ../textures/t.png\nmap_Bump ..\\textures\\t.png\n")
    (tmp_path / "models" / "q.obj").write_text("mtllib q.mtl\nv 0 0 0\nv 3 0 0\nv 3 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nf 1/1 2/2 3/3\n")
    text = open(os.path.join(ROOT, "scenes", "sphere.txt")).read() + "\nOBJECT 1\nobj\n../models/q.obj\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n"
    (tmp_path / "scenes" / "s.txt").write_text(text)
    return str(tmp_path / "scenes" / "s.txt")


def test_png_textures_match_the_reference_loader(product, tmp_path):
    """PNG maps (csrc/pt_png.h) against what the reference's loader -- stb_image with the vertical flip -- made of the same
    files (tests/golden/png_textures.npz, generated by make_golden.py from tests/pngcases.py): every colour type and bit
    depth, all five filters, Adam7, palette and colour-key transparency, stored and dynamic-Huffman streams."""
    import pngcases
    g = golden("png_textures.npz")
    names = [n for n, _ in pngcases.cases()]
    assert len(names) == 17 and all("file_" + n in g.files for n in names)
    for name, png in pngcases.cases():
        assert png == bytes(g["file_" + name])                       # the committed bytes are what the generator writes
        d = product.Scene(_png_scene(tmp_path, png)).dump()
        want = g["texels_" + name]
        got = d["textures"][(1, 0)]
        assert got.shape == want.shape and np.array_equal(got, want), name
        assert np.array_equal(d["textures"][(1, 3)], want)             # map_Bump with Windows separators


def test_broken_png_is_a_failed_load_not_a_crash(product, tmp_path):
    """Truncated, corrupted and non-PNG files give an empty texture, the reference's own outcome for a failed stbi_load."""
    import pngcases
    png = dict(pngcases.cases())["rgb8"]
    rng = np.random.default_rng(3)
    variants = [png[:40], png[:-20], b"\x89PNG\r\n\x1a\n" + b"\0" * 64, b"JFIF" * 10, b""]
    for k in range(40):                                                # random byte flips anywhere in the file
        b = bytearray(png)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(8, len(b)))] = int(rng.integers(0, 256))
        variants.append(bytes(b))
    for v in variants:
        d = product.Scene(_png_scene(tmp_path, v)).dump()             # must not crash; a texture, if any, has sane dimensions
        t = d["textures"].get((1, 0))
        assert t is None or (t.ndim == 3 and t.shape[2] in (1, 2, 3, 4) and t.size <= 1 << 20)


def _jpeg_scene(tmp_path, jpg_bytes):
    for d in ("scenes", "models/materials", "textures"):
        os.makedirs(tmp_path / d, exist_ok=True)
    (tmp_path / "textures" / "t.jpg").write_bytes(jpg_bytes)
    (tmp_path / "models" / "materials" / "q.mtl").write_text("newmtl a\nKd .1 .2 .3\nKs .4 .5 .6\nNi 1.5\nmap_Kd ../textures/t.jpg\n")
    (tmp_path / "models" / "q.obj").write_text("mtllib q.mtl\nv 0 0 0\nv 3 0 0\nv 3 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nf 1/1 2/2 3/3\n")
    text = open(os.path.join(ROOT, "scenes", "sphere.txt")).read() + "\nOBJECT 1\nobj\n../models/q.obj\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n"
    (tmp_path / "scenes" / "s.txt").write_text(text)
    return str(tmp_path / "scenes" / "s.txt")


def test_jpeg_textures_match_the_reference_loader(product, tmp_path):
    """JPEG maps (csrc/pt_jpeg.h) against what the reference's loader -- stb_image v2.27 with the vertical flip -- made of
    the same bytes (tests/golden/jpeg_textures.npz, from tests/jpegcases.py through make_golden.py): JPEG is lossy, so the
    decoder's integer IDCT, chroma upsampling and fixed-point colour conversion have to be the reference's for the texels
    to agree bit for bit.  44 files: baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1 / 4:4:0, optimised tables,
    restart intervals, 16-bit quantisation tables, greyscale, CMYK, untransformed RGB, sizes down to one pixel wide,
    entropy data that ends early."""
    g = golden("jpeg_textures.npz")
    names = [f[5:] for f in g.files if f.startswith("file_")]
    assert len(names) == 44
    kinds = set()
    for name in names:
        d = product.Scene(_jpeg_scene(tmp_path, bytes(g["file_" + name]))).dump()
        want = g["texels_" + name]
        got = d["textures"].get((1, 0))
        if want.size == 0:
            assert got is None, name                                  # the reference failed to load it
            kinds.add("fail")
        else:
            assert got is not None and got.shape == want.shape and np.array_equal(got, want), name
            kinds.add(want.shape[2])
    assert {1, 3} <= kinds


def test_broken_jpeg_is_a_failed_load_not_a_crash(product, tmp_path):
    g = golden("jpeg_textures.npz")
    jpg = bytes(g["file_blobs_prog420"])
    rng = np.random.default_rng(8)
    variants = [jpg[:30], jpg[:-2], b"\xff\xd8\xff", b"\xff\xd8" + b"\0" * 50]
    for k in range(60):
        b = bytearray(jpg)
        for _ in range(int(rng.integers(1, 5))):
            b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
        variants.append(bytes(b))
    for v in variants:
        d = product.Scene(_jpeg_scene(tmp_path, v)).dump()
        t = d["textures"].get((1, 0))
        assert t is None or (t.ndim == 3 and t.shape[2] in (1, 3) and t.size <= 1 << 22)


def test_polygons_with_more_than_four_corners(product, tmp_path):
    """65 pentagons ... dodecagons (convex, concave, star-shaped, non-planar, in all three coordinate planes, with collinear
    and repeated vertices) come out as the same 332 triangles, in the same order, as from the reference's loader
    (tinyobjloader's built-in ear clipping; tests/golden/loader_ngons.npz)."""
    import ngoncases
    g = golden("loader_ngons.npz")
    text = ngoncases.obj_text()
    assert text.encode() == bytes(g["obj"])
    for d in ("scenes", "models/materials"):
        os.makedirs(tmp_path / d)
    (tmp_path / "models" / "n.obj").write_text(text)
    (tmp_path / "models" / "materials" / "cube.mtl").write_text(open(os.path.join(ROOT, "models", "materials", "cube.mtl")).read())
    (tmp_path / "scenes" / "s.txt").write_text(open(os.path.join(ROOT, "scenes", "sphere.txt")).read() +
                                               "\nOBJECT 1\nobj\n../models/n.obj\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n")
    d = product.Scene(str(tmp_path / "scenes" / "s.txt")).dump()
    got = np.asarray(d["faces"][1], np.float32)
    assert got.shape == g["faces"].shape == (332, 15) and beq(got, g["faces"])


def test_sanitizer_fuzz_of_the_file_code():
    """tools/fuzz: the scene / OBJ / MTL loader and the PPM / PNG / JPEG decoders under AddressSanitizer + UBSan (CPU
    build) on the repository's scenes and the image fixtures, each also truncated, bit-flipped, overwritten and stretched
    (deterministic): no memory error, no undefined behaviour, every decoded image consistent with its header.  A short
    pass here; `bash tools/fuzz/run.sh 1000` is the long one."""
    import subprocess
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "fuzz", "run.sh"), "12"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "decoders:" in r.stdout and "scenes:" in r.stdout and "assets:" in r.stdout
