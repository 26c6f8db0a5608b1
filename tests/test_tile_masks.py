"""CPU: the camera-ray bounce's per-tile geom masks (pt_engine.hip: tile_geom_masks, through the device-free entry point
ptx_debug_tile_geoms) are SUPERSETS.  A bit cleared for a tile makes k_bounce<first> skip that geom for every ray of the tile -- silently,
a hit would become a miss -- so the property is checked ray by ray: the oracle generates the camera rays of whole frames
(generateRayFromCamera with its antialiasing jitter and, optionally, its lens: src/pathtrace.cu:206-255), every ray is intersected with
every box in float64, and a ray that reaches a box must find that box's bit in the mask of its tile of 256 owned pixels.  Random eyes
(in front of, beside, above, inside and behind the boxes), fields of view from 2 to 130 degrees, frames whose rows are shorter and
longer than a tile, tile splits, depth of field on and off; boxes from pin-sized to room-sized, some containing the eye."""
import os

import numpy as np
import pytest

from conftest import ROOT

MAT = "MATERIAL 0\nRGB 1 1 1\nSPECEX 0\nSPECRGB 0 0 0\nREFL 0\nREFR 0\nREFRIOR 0\nEMITTANCE 5\n\n"


def _scene(pt, tmp_path, eye, look, fovy, res):
    text = MAT + ("CAMERA\nRES %d %d\nFOVY %g\nITERATIONS 3\nDEPTH 2\nFILE t\nEYE %g %g %g\nLOOKAT %g %g %g\nUP 0 1 0\n\n"
                  % (tuple(res) + (fovy,) + tuple(eye) + tuple(look))) + "OBJECT 0\ncube\nmaterial 0\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n\n"
    f = tmp_path / "cam.txt"
    f.write_text(text)
    s = pt.Scene(str(f), base_dir=os.path.join(ROOT, "scenes"))
    s.apply_runcuda_camera()
    return s


def _reaches(o, d, lo, hi):
    """[ray, box] -> does the ray (t >= 0) touch the closed box; float64 slabs"""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d[:, None, :]
        t0, t1 = (lo[None] - o[:, None, :]) * inv, (hi[None] - o[:, None, :]) * inv
        par = d[:, None, :] == 0.0                           # parallel to a slab: inside it or not
        inside = (o[:, None, :] >= lo[None]) & (o[:, None, :] <= hi[None])
        tn = np.where(par, np.where(inside, -np.inf, np.inf), np.minimum(t0, t1)).max(2)
        tf = np.where(par, np.where(inside, np.inf, -np.inf), np.maximum(t0, t1)).min(2)
    return (tf >= tn) & (tf >= 0.0)


@pytest.mark.parametrize("seed", range(40))
def test_tile_masks_hold_every_geom_a_ray_of_the_tile_reaches(product, oracle_lib, tmp_path, seed):
    pt, O = product, oracle_lib
    rng = np.random.default_rng(9000 + seed)
    W, H = int(rng.integers(30, 420)), int(rng.integers(20, 140))
    eye = rng.uniform([-14, -3, -14], [14, 16, 26])
    look = rng.uniform([-5, 0, -5], [5, 10, 5])
    if np.linalg.norm(look - eye) < 0.5:
        look = eye + np.array([0.3, 0.2, -1.0])
    fovy = float(rng.choice([2.0, 10.0, 30.0, 45.0, 70.0, 100.0, 130.0]))
    s = _scene(pt, tmp_path, eye, look, fovy, (W, H))
    # boxes: pin-sized to room-sized, anywhere around the look-at point; one around the eye, one far behind it
    n = int(rng.integers(3, 14))
    ctr = rng.uniform([-7, -1, -7], [7, 11, 7], (n, 3))
    half = np.exp(rng.uniform(np.log(0.01), np.log(6.0), (n, 3)))
    ctr[0], half[0] = eye, np.array([0.7, 0.7, 0.7])
    view = (look - eye) / np.linalg.norm(look - eye)
    ctr[1] = eye - 5.0 * view
    lo, hi = ctr - half, ctr + half
    boxes = np.concatenate([lo, hi], 1).astype(np.float32)
    lo, hi = boxes[:, :3].astype(np.float64), boxes[:, 3:].astype(np.float64)
    d = s.dump()
    O.set_libm(1)
    O.create(d, d["textures"])
    split = (int(rng.choice([4, 8])), int(rng.integers(0, 3)), 3) if seed % 3 == 0 else None
    if split:
        owned_rows = np.array([y for y in range(H) if (y // split[0]) % split[2] == split[1]])
    else:
        owned_rows = np.arange(H)
    slot_of_row = -np.ones(H, np.int64)
    slot_of_row[owned_rows] = np.arange(len(owned_rows))
    for dof in (0, 1):
        masks = pt.api.debug_tile_geoms(s.camera, boxes, depth_of_field=dof, tile=split)
        assert len(masks) == (max(len(owned_rows) * W, 1) + 255) // 256
        O.set_options(aa=1, dof=dof, sort=1, cache=0)
        O.pt_init()
        for it in (1, 2, 7):
            O.pt_generate(it)
            p = O.paths()[:W * H]
            pix = p["pixelIndex"].astype(np.int64)
            y, x = pix // W, pix % W
            keep = slot_of_row[y] >= 0
            tile = (slot_of_row[y] * W + x) // 256
            hit = _reaches(p["origin"].astype(np.float64), p["direction"].astype(np.float64), lo, hi)      # [ray, box]
            have = (masks[np.where(keep, tile, 0)][:, None] >> np.arange(n)[None]) & 1
            bad = keep[:, None] & hit & (have == 0)
            assert not bad.any(), (seed, dof, it, np.argwhere(bad)[:5].tolist(), (W, H), eye.tolist(), look.tolist(), fovy, split)
    O.set_libm(0)


def test_tile_masks_do_exclude_something(product, tmp_path):
    """... and they are not vacuous: the stock Cornell view at 1920x1080 leaves a third of the tiles without any geom, the others see
    three of the seven on average, and no tile sees all of them."""
    pt = product
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(1920, 1080), depth=8)
    s.apply_runcuda_camera()
    d = s.dump()
    # conservative world boxes of the geoms, as make_world_aabb forms them (transformed unit cube / mesh vertices)
    boxes = []
    for gi in range(len(d["geom_ints"])):
        xf = d["geom_mats"][gi][:16].reshape(4, 4).T.astype(np.float64)
        f = d["faces"][gi]
        if len(f):
            v = np.concatenate([f[:, 0:3], f[:, 5:8], f[:, 10:13]]).astype(np.float64)
        else:
            v = np.array([[sx, sy, sz] for sx in (-.5, .5) for sy in (-.5, .5) for sz in (-.5, .5)], np.float64)
        w = v @ xf[:3, :3].T + xf[:3, 3]
        boxes.append(np.concatenate([w.min(0) - 2e-3, w.max(0) + 2e-3]))
    masks = pt.api.debug_tile_geoms(s.camera, np.array(boxes, np.float32))
    ngeoms = len(boxes)
    counts = np.array([bin(int(m) & ((1 << ngeoms) - 1)).count("1") for m in masks])
    assert (counts == 0).mean() > 0.25 and counts[counts > 0].mean() < 4.5 and counts.max() < ngeoms, ((counts == 0).mean(), counts[counts > 0].mean(), counts.max())
