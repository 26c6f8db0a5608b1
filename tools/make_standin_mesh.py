#!/usr/bin/env python3
"""Deterministic stand-in for the reference's missing spaceship mesh.

The reference's scenes/cornellSpaceship.txt points at models/Intergalactic_Spaceship-(Wavefront).obj, which is not in
the reference checkout (.MISSING_LARGE_BLOBS), with four 4096x4096 JPEG maps.  This writes a closed, UV-mapped,
outward-CCW triangle/quad mesh of the same role (models/standin_ship.obj, 320 triangles after triangulation), its
.mtl with the same keys (Ni 2.0, map_Kd / map_Ks / map_Ke / map_Bump) and four procedural textures as binary PPM
(stb_image, which the reference loads textures with, reads PPM too, so the same files drive the reference oracle):

    python tools/make_standin_mesh.py [--size 256]       # writes models/standin_ship.obj, models/materials/standin_ship.mtl,
                                                          # textures/standin_{kd,ks,ke,bump}.ppm

Everything derives from integer hashes, so re-running reproduces the files bit for bit.
"""
import argparse
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RINGS, SEGS = 10, 16


def mesh_text(RINGS=RINGS, SEGS=SEGS):
    out = ["# procedural stand-in mesh: a pinched ellipsoid hull, %d x %d patches" % (RINGS, SEGS), "mtllib standin_ship.mtl", ""]
    verts, uvs = [], []
    for r in range(RINGS + 1):
        th = math.pi * r / RINGS
        for s in range(SEGS + 1):
            ph = 2 * math.pi * s / SEGS
            bulge = 1.0 + 0.25 * math.sin(3 * th) * math.cos(2 * ph)
            x = 1.6 * math.sin(th) * math.cos(ph) * bulge
            y = 0.7 * math.cos(th)
            z = 1.0 * math.sin(th) * math.sin(ph) * bulge
            verts.append((x, y, z))
            uvs.append((0.02 + 0.96 * s / SEGS, 0.02 + 0.96 * r / RINGS))      # strictly inside (0,1): no texel wrap
    for v in verts:
        out.append("v %.6f %.6f %.6f" % v)
    for t in uvs:
        out.append("vt %.6f %.6f" % t)
    out.append("g hull")
    idx = lambda r, s: r * (SEGS + 1) + s + 1
    for r in range(RINGS):
        for s in range(SEGS):
            a, b, c, d = idx(r, s), idx(r, s + 1), idx(r + 1, s + 1), idx(r + 1, s)
            if r == 0:
                out.append("f %d/%d %d/%d %d/%d" % (a, a, c, c, d, d))           # pole cap: triangle (outward CCW)
            elif r == RINGS - 1:
                out.append("f %d/%d %d/%d %d/%d" % (a, a, b, b, d, d))
            else:
                out.append("f %d/%d %d/%d %d/%d %d/%d" % (a, a, b, b, c, c, d, d))   # quad, split by the loader
    return "\n".join(out) + "\n"


MTL = """# material of the stand-in mesh: same keys as the reference's Intergalactic_Spaceship-(Wavefront).mtl
newmtl Material
Ns 96.078431
Ka 1.000000 1.000000 1.000000
Kd 0.640000 0.640000 0.640000
Ks 0.500000 0.500000 0.500000
Ke 0.000000 0.000000 0.000000
Ni 2.000000
d 1.000000
illum 2
map_Bump ../textures/standin_bump.ppm
map_Kd ../textures/standin_kd.ppm
map_Ks ../textures/standin_ks.ppm
map_Ke ../textures/standin_ke.ppm
"""


def ihash(a):
    a = np.asarray(a, np.uint64) & 0xFFFFFFFF
    a = ((a ^ 61) ^ (a >> 16)) & 0xFFFFFFFF
    a = (a * 9) & 0xFFFFFFFF
    a = (a ^ (a >> 4)) & 0xFFFFFFFF
    a = (a * 0x27d4eb2d) & 0xFFFFFFFF
    a = (a ^ (a >> 15)) & 0xFFFFFFFF
    return a


def textures(n):
    y, x = np.mgrid[0:n, 0:n]
    cell = ihash((x // max(n // 16, 1)) * 73856093 ^ (y // max(n // 16, 1)) * 19349663)
    fine = ihash(x * 83492791 ^ y * 2971215073 % (1 << 32))
    kd = np.stack([96 + (cell >> 0) % 128, 96 + (cell >> 8) % 128, 96 + (cell >> 16) % 128], -1).astype(np.uint8)
    ks = np.stack([128 + (fine >> 3) % 96] * 3, -1).astype(np.uint8)
    lit = ((cell >> 24) % 16) == 0                                    # about 6 % of the panels glow
    ke = np.where(lit[..., None], np.stack([200 + (cell % 40), 180 + (cell >> 5) % 60, 90 + (cell >> 9) % 60], -1), 0).astype(np.uint8)
    nx = 128 + ((fine >> 2) % 41).astype(np.int64) - 20
    ny = 128 + ((fine >> 9) % 41).astype(np.int64) - 20
    bump = np.stack([nx, ny, np.full_like(nx, 240)], -1).astype(np.uint8)
    return dict(kd=kd, ks=ks, ke=ke, bump=bump)


def write_ppm(path, img):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img).tobytes())


def main(size=256, root=ROOT, big=True):
    os.makedirs(os.path.join(root, "models", "materials"), exist_ok=True)
    os.makedirs(os.path.join(root, "textures"), exist_ok=True)
    with open(os.path.join(root, "models", "standin_ship.obj"), "w") as f:
        f.write(mesh_text())
    if big:     # the same hull at 72 x 144 patches = 20448 triangles (generated, not committed): the BVH-sized mesh
        with open(os.path.join(root, "models", "standin_ship_20k.obj"), "w") as f:
            f.write(mesh_text(72, 144))
    with open(os.path.join(root, "models", "materials", "standin_ship.mtl"), "w") as f:
        f.write(MTL)
    for k, img in textures(size).items():
        write_ppm(os.path.join(root, "textures", "standin_%s.ppm" % k), img)
    return root


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    a = ap.parse_args()
    main(a.size)
    print("wrote models/standin_ship.obj, models/materials/standin_ship.mtl, textures/standin_*.ppm (%d x %d)" % (a.size, a.size))
