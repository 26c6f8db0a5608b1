"""pytest configuration: the `gpu` marker and shared fixtures.

CPU tier  (`-m "not gpu"`): the plain-C oracle against the golden vectors generated from the reference itself
(tests/golden/, see make_golden.py), against the live reference build when /root/reference is present, the
product's host logic (scene loader, C-ABI surface, CPU stream compaction, multi-rank driver over gloo).
GPU tier  (`-m gpu`): the HIP path against the oracle, through the C ABI.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


def beq(a, b):
    """bitwise equality (NaN == NaN, -0 != +0)"""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype.itemsize == b.dtype.itemsize and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def dump_from_golden(g, cam="cam_floats_runcuda", res=None):
    """Scene POD dict (the layout OracleLib.create takes) from a loader_*.npz fixture."""
    ng = len(g["geom_ints"])
    faces = [g["faces_%d" % i] for i in range(ng)]
    ci = g["cam_ints"].copy()
    cf = g[cam].copy()
    textures = {}
    for k in g.files:
        if k.startswith("tex_"):
            _, gi, which = k.split("_")
            textures[(int(gi), int(which))] = g[k]
    return dict(geom_ints=g["geom_ints"], geom_trs=g["geom_trs"], geom_mats=g["geom_mats"], materials=g["materials"],
                faces=faces, cam_ints=ci, cam_floats=cf, textures=textures)


def ensure_standin_assets():
    """The stand-in mesh's procedural textures are generated (deterministically) rather than committed."""
    if not os.path.exists(os.path.join(ROOT, "textures", "standin_kd.ppm")) or \
            not os.path.exists(os.path.join(ROOT, "models", "standin_ship_20k.obj")):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import make_standin_mesh
        make_standin_mesh.main(128)


@pytest.fixture(scope="session", autouse=True)
def standin_assets():
    ensure_standin_assets()


@pytest.fixture(scope="session")
def oracle_lib():
    from cpulibs import OracleLib
    return OracleLib()


@pytest.fixture(scope="session")
def ref_lib():
    from cpulibs import RefLib, build_ref
    so = build_ref()
    if not so:
        pytest.skip("oracle/_ref/libptref.so is not available (needs /root/reference to build)")
    return RefLib(so)


@pytest.fixture(scope="session")
def product():
    """The product package with its native library built (hipcc cross-compiles for gfx950 without a GPU)."""
    import mygpuraytracer_amd as pt
    pt.build_library()
    pt.load_library()
    return pt


@pytest.fixture(scope="session")
def gpu_product(product):
    if product.load_library().ptx_device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible -- the HIP path has no fallback")
    return product
