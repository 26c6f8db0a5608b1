#!/usr/bin/env python3
"""Writes the scene / model data files this repo ships (scenes/*.txt, models/*.obj, models/materials/*.mtl).

They are inputs in the reference's own text formats describing the same five scenes the
reference ships -- the GPU box only receives this repo, so the data has to live here.  The files are generated
from the tables below rather than typed by hand; tests/test_loader_parity.py checks (in the dev container, where
/root/reference exists) that loading them gives bit-identical geoms/materials/camera to the reference loader on
the reference's own files.

    python tools/make_scenes.py            # (re)writes scenes/, models/
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MATERIALS = {
    "light":     dict(note="emissive white (ceiling light)", RGB="1 1 1", SPECEX="0", SPECRGB="0 0 0", REFL="0", REFR="0", REFRIOR="0", EMITTANCE="5"),
    "white":     dict(note="diffuse white", RGB=".98 .98 .98", SPECEX="0", SPECRGB="0 0 0", REFL="0", REFR="0", REFRIOR="0", EMITTANCE="0"),
    "red":       dict(note="diffuse red", RGB=".85 .35 .35", SPECEX="0", SPECRGB="0 0 0", REFL="0", REFR="0", REFRIOR="0", EMITTANCE="0"),
    "green":     dict(note="diffuse green", RGB=".35 .85 .35", SPECEX="0", SPECRGB="0 0 0", REFL="0", REFR="0", REFRIOR="0", EMITTANCE="0"),
    "mirror":    dict(note="specular white", RGB=".98 .98 .98", SPECEX="0", SPECRGB=".98 .98 .98", REFL="1", REFR="0", REFRIOR="0", EMITTANCE="0"),
    "glass":     dict(note="refractive, bluish", RGB=".98 .98 .98", SPECEX="0", SPECRGB=".85 .85 .98", REFL="0", REFR="1", REFRIOR="1.65", EMITTANCE="0"),
}
MATERIAL_KEYS = ("RGB", "SPECEX", "SPECRGB", "REFL", "REFR", "REFRIOR", "EMITTANCE")

CAMERA = dict(RES="800 800", FOVY="45", ITERATIONS="5000", DEPTH="8", EYE="0.0 5 10.5", LOOKAT="0 5 0", UP="0 1 0")

# the open Cornell box: (comment, type, material index, TRANS, ROTAT, SCALE)
BOX = [
    ("ceiling light", "cube", 0, "0 10 0", "0 0 0", "3 .3 3"),
    ("floor", "cube", 1, "0 0 0", "0 0 0", "10 .01 10"),
    ("ceiling", "cube", 1, "0 10 0", "0 0 90", ".01 10 10"),
    ("back wall", "cube", 1, "0 5 -5", "0 90 0", ".01 10 10"),
    ("left wall", "cube", 2, "-5 5 0", "0 0 0", ".01 10 10"),
    ("right wall", "cube", 3, "5 5 0", "0 0 0", ".01 10 10"),
]

SCENES = {
    "sphere.txt": dict(file="sphere", materials=["light"],
                       objects=[("emissive sphere", "sphere", 0, "0 0 0", "0 0 0", "3 3 3")]),
    "cornell.txt": dict(file="cornell", materials=["light", "white", "red", "green", "mirror"],
                        objects=BOX + [("diffuse sphere", "sphere", 1, "-1 4 -1", "0 0 0", "3 3 3")]),
    "cornellGlass.txt": dict(file="cornell", materials=["light", "white", "red", "green", "mirror", "glass"],
                             objects=BOX + [("glass sphere", "sphere", 5, "-1 4 -1", "0 0 0", "3 3 3")]),
    "cornellObj.txt": dict(file="cornell", materials=["light", "white", "red", "green", "mirror", "glass"],
                           objects=BOX + [("triangle mesh", "obj", "../models/cube.obj", "-2 4 -3", "0 45 0", "2 2 2")]),
    # The reference's cornellSpaceship.txt points at a mesh that is not in the reference checkout
    # (.MISSING_LARGE_BLOBS); this repo substitutes a deterministic procedural stand-in (tools/make_standin_mesh.py)
    # under the same object block and the same .mtl keys.
    "cornellSpaceship.txt": dict(file="cornell", materials=["light", "white", "red", "green", "mirror", "glass"],
                                 objects=BOX + [("diffuse sphere", "sphere", 1, "-2 7 -1", "0 0 0", "2 2 2"),
                                                ("glass sphere", "sphere", 5, "1 6 0", "0 0 0", "2 2 2"),
                                                ("textured mesh (procedural stand-in)", "obj", "../models/standin_ship.obj",
                                                 "1 3 3", "0 20 180", "1 1 1")]),
}


def scene_text(spec):
    out = []
    for i, name in enumerate(spec["materials"]):
        m = MATERIALS[name]
        out.append("// material %d: %s" % (i, m["note"]))
        out.append("MATERIAL %d" % i)
        for k in MATERIAL_KEYS:
            out.append("%-11s %s" % (k, m[k]))
        out.append("")
    out.append("// camera")
    out.append("CAMERA")
    for k in ("RES", "FOVY", "ITERATIONS", "DEPTH"):
        out.append("%-11s %s" % (k, CAMERA[k]))
    out.append("%-11s %s" % ("FILE", spec["file"]))
    for k in ("EYE", "LOOKAT", "UP"):
        out.append("%-11s %s" % (k, CAMERA[k]))
    out.append("")
    for i, (note, typ, mat, tr, ro, sc) in enumerate(spec["objects"]):
        out.append("")
        out.append("// object %d: %s" % (i, note))
        out.append("OBJECT %d" % i)
        out.append(typ)
        if typ == "obj":
            out.append(mat)
        else:
            out.append("material %d" % mat)
        out.append("%-11s %s" % ("TRANS", tr))
        out.append("%-11s %s" % ("ROTAT", ro))
        out.append("%-11s %s" % ("SCALE", sc))
    return "\n".join(out) + "\n"


def cube_obj():
    """Axis-aligned cube [0,2]^3 as six quads, winding and order as in the reference's models/cube.obj."""
    v = [(0, 2, 2), (0, 0, 2), (2, 0, 2), (2, 2, 2), (0, 2, 0), (0, 0, 0), (2, 0, 0), (2, 2, 0)]
    faces = [("front", (1, 2, 3, 4)), ("back", (8, 7, 6, 5)), ("right", (4, 3, 7, 8)), ("top", (5, 1, 4, 8)),
             ("left", (5, 6, 2, 1)), ("bottom", (2, 6, 7, 3))]
    out = ["# unit-test cube, 8 vertices, 6 quads", "mtllib cube.mtl", ""]
    for p in v:
        out.append("v %.6f %.6f %.6f" % p)
    out.append("")
    for name, f in faces:
        out.append("g %s" % name)
        out.append("f %d %d %d %d" % f)
    return "\n".join(out) + "\n"


CUBE_MTL = """# material of models/cube.obj (values as in the reference's models/materials/cube.mtl)
newmtl Material
Ns 96.078431
Ka 1.000000 1.000000 1.000000
Kd 0.640000 0.640000 0.640000
Ks 0.500000 0.500000 0.500000
Ke 0.000000 0.000000 0.000000
Ni 1.000000
d 1.000000
illum 2
"""


def main():
    os.makedirs(os.path.join(ROOT, "scenes"), exist_ok=True)
    os.makedirs(os.path.join(ROOT, "models", "materials"), exist_ok=True)
    for name, spec in SCENES.items():
        with open(os.path.join(ROOT, "scenes", name), "w") as f:
            f.write(scene_text(spec))
    with open(os.path.join(ROOT, "models", "cube.obj"), "w") as f:
        f.write(cube_obj())
    with open(os.path.join(ROOT, "models", "materials", "cube.mtl"), "w") as f:
        f.write(CUBE_MTL)
    print("wrote %d scenes, models/cube.obj, models/materials/cube.mtl" % len(SCENES))


if __name__ == "__main__":
    sys.exit(main())
