"""CPU: ptx_create's rule for what a stored path's record carries (ptx_debug_record_masks).  A record without its incoming
direction, or with a normal code instead of its normal, is only right if the next bounce can never ask for what is missing -- and the
next bounce tells the kinds apart by sorted position, two ranges per mask.  Checked on random scenes: the direction mask is a SUPERSET of
the bins whose material scatterRay reads the direction for, the code mask a SUBSET of the bins whose material only cubes have, and
neither has more than two runs of set bits."""
import ctypes as C

import numpy as np
import pytest

from conftest import ROOT  # noqa: F401
import mygpuraytracer_amd as pt
from mygpuraytracer_amd import api


def masks(mats, gtype, gmat, sort=1):
    L = api.load_library()
    L.ptx_debug_record_masks.restype = C.c_int
    L.ptx_debug_record_masks.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    m = np.ascontiguousarray(mats, np.float32).reshape(-1, 11)
    gt, gm = np.ascontiguousarray(gtype, np.int32), np.ascontiguousarray(gmat, np.int32)
    out = np.zeros(2, np.uint64)
    rc = L.ptx_debug_record_masks(len(m), m.ctypes.data, len(gt), gt.ctypes.data, gm.ctypes.data, sort, out.ctypes.data)
    assert rc == 0
    return int(out[0]), int(out[1])


def runs(mask, nb):
    bits = [(mask >> b) & 1 for b in range(nb)]
    return sum(1 for b in range(nb) if bits[b] and not (b and bits[b - 1]))


def material(refl=0.0, refr=0.0, emit=0.0):
    return [0.5, 0.5, 0.5, 0.0, 0.9, 0.9, 0.9, refl, refr, 1.5 if refr else 0.0, emit]


def test_cornell_obj_layout():
    # light, three diffuse wall materials, a mirror and a glass nobody wears, and the OBJ's own material (appended by the loader)
    mats = [material(emit=5), material(), material(), material(), material(refl=1), material(refr=1), material()]
    gtype = [1, 1, 1, 1, 1, 1, 3]
    gmat = [0, 1, 1, 1, 2, 3, 6]
    d, n = masks(mats, gtype, gmat)
    bin_of = lambda m: len(mats) - 1 - m
    assert d == (1 << bin_of(4)) | (1 << bin_of(5)) | (1 << bin_of(6))          # mirror, glass, the mesh's material: bins 0-2, one run
    assert n == (1 << bin_of(0)) | (1 << bin_of(1)) | (1 << bin_of(2)) | (1 << bin_of(3))      # the light's and the walls': cubes only
    d, n = masks(mats, gtype, gmat, sort=0)
    assert d == 1 and n == 0                                                     # one bin: every record complete


def test_random_scenes_keep_the_invariants():
    rng = np.random.default_rng(11)
    seen_fill = seen_drop = 0
    for case in range(400):
        nm = int(rng.integers(1, 65))
        kinds = rng.integers(0, 4, nm)                        # 0 diffuse, 1 mirror, 2 glass, 3 light
        mats = [material(refl=float(k == 1), refr=float(k == 2), emit=5.0 * (k == 3)) for k in kinds]
        ng = int(rng.integers(0, 40))
        gtype = rng.choice([0, 1, 3], ng, p=[0.3, 0.55, 0.15]).astype(np.int32)
        gmat = rng.integers(0, nm, ng).astype(np.int32)
        d, n = masks(mats, gtype, gmat)
        need = 0
        cubes_only = 0
        for m in range(nm):
            on = gtype[gmat == m]
            if kinds[m] in (1, 2) or (on == 3).any():
                need |= 1 << (nm - 1 - m)
            if len(on) and (on == 1).all():
                cubes_only |= 1 << (nm - 1 - m)
        assert d & need == need, (case, bin(d), bin(need))                       # never a direction less than scatterRay can read
        assert n & ~cubes_only == 0, (case, bin(n), bin(cubes_only))             # never a code where a normal could be a sphere's or a mesh's
        assert runs(d, nm) <= 2 and runs(n, nm) <= 2
        assert d >> nm == 0 and n >> nm == 0
        if runs(need, nm) <= 2:
            assert d == need
        else:
            seen_fill += 1
        if runs(cubes_only, nm) <= 2:
            assert n == cubes_only
        else:
            seen_drop += 1
            assert bin(n).count("1") >= 1                                        # (trimmed, not thrown away)
    assert seen_fill > 50 and seen_drop > 50


def test_bad_arguments():
    L = api.load_library()
    L.ptx_debug_record_masks.restype = C.c_int
    L.ptx_debug_record_masks.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    out = np.zeros(2, np.uint64)
    m = np.zeros((65, 11), np.float32)
    assert L.ptx_debug_record_masks(65, m.ctypes.data, 0, None, None, 1, out.ctypes.data) == -1
    assert L.ptx_debug_record_masks(0, m.ctypes.data, 0, None, None, 1, out.ctypes.data) == -1
    assert L.ptx_debug_record_masks(3, m.ctypes.data, 0, None, None, 1, None) == -1
