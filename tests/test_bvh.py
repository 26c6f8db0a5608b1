"""Mesh BVH (csrc/pt_bvh.h) against the reference's loop over all faces (src/intersections.h:213-233), on the CPU.

The tree only decides WHICH triangles are looked at; the per-triangle arithmetic and the choice of the winner (nearest
distance, lowest face index on ties) are the loop's.  ptx_debug_bvh_check runs both on the host with the library's own
code (the traversal is the function the kernels inline), so face and distance must agree bit for bit -- including
duplicate triangles (ties), axis-parallel rays (zero direction components), grazing rays and far origins."""
import ctypes as C

import numpy as np
import pytest

from conftest import beq


def run_check(product, faces, rays):
    L = product.load_library()
    faces = np.ascontiguousarray(faces, np.float32).reshape(-1, 15)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n = len(rays)
    fl, fb = np.zeros(n, np.int32), np.zeros(n, np.int32)
    tl, tb = np.zeros(n, np.float32), np.zeros(n, np.float32)
    st = np.zeros(4, np.int64)
    vp = C.c_void_p
    L.ptx_debug_bvh_check.restype = C.c_int
    L.ptx_debug_bvh_check.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp]
    rc = L.ptx_debug_bvh_check(faces.ctypes.data, len(faces), rays.ctypes.data, n, fl.ctypes.data, tl.ctypes.data,
                               fb.ctypes.data, tb.ctypes.data, st.ctypes.data)
    assert rc == 0
    return fl, tl, fb, tb, st


def hull(rings, segs, rng=None):
    """closed UV-mapped ellipsoid-like hull, outward CCW, as 15-float faces"""
    v = []
    for r in range(rings + 1):
        th = np.pi * r / rings
        for s in range(segs + 1):
            ph = 2 * np.pi * s / segs
            b = 1.0 + 0.25 * np.sin(3 * th) * np.cos(2 * ph)
            v.append((1.6 * np.sin(th) * np.cos(ph) * b, 0.7 * np.cos(th), np.sin(th) * np.sin(ph) * b, s / segs, r / rings))
    v = np.array(v, np.float32)
    idx = lambda r, s: r * (segs + 1) + s
    f = []
    for r in range(rings):
        for s in range(segs):
            a, b, c, d = idx(r, s), idx(r, s + 1), idx(r + 1, s + 1), idx(r + 1, s)
            f.append(np.concatenate([v[a], v[c], v[d]]))
            f.append(np.concatenate([v[a], v[b], v[c]]))
    return np.array(f, np.float32)


def rays_around(rng, n, radius, target_scale=1.0):
    o = rng.normal(size=(n, 3)).astype(np.float32)
    o *= (radius / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    tgt = (rng.uniform(-1, 1, size=(n, 3)) * target_scale).astype(np.float32)
    return np.concatenate([o, tgt - o], axis=1).astype(np.float32)


def assert_same(res):
    fl, tl, fb, tb, st = res
    assert beq(fl, fb), "faces differ at %s" % np.nonzero(fl != fb)[0][:8]
    assert beq(tl, tb)
    assert st[3] == 0, "the front-to-back walk or the walk over the four-wide quantised nodes disagrees with the skip-link walk on %d rays (>= 1e6: the wide walk overran its stack bound)" % st[3]
    return fl, st


def test_hull_random_rays(product):
    rng = np.random.default_rng(7)
    faces = hull(48, 96)                      # 9216 triangles
    rays = np.concatenate([rays_around(rng, 20000, 6.0, 1.5), rays_around(rng, 5000, 0.2, 2.0), rays_around(rng, 5000, 1000.0, 1.5)])
    fl, st = assert_same(run_check(product, faces, rays))
    assert (fl >= 0).mean() > 0.3
    assert st[1] == len(faces) and st[0] < 2 * len(faces)
    assert st[2] / len(rays) < 400            # the tree is actually pruning (the loop would visit 9216 triangles per ray)


def test_walk_statistics_of_the_last_check(product):
    """ptx_debug_bvh_visits: what the three walks cost on the rays of the last check -- the figures k_mesh's traversal was tuned
    by.  The four-wide walk visits fewer nodes than the binary front-to-back walk, that one fewer than the skip-link walk; its
    stack bound is inside what k_mesh provides; leaves hold at most four triangles, so a walk tests a handful, not hundreds."""
    rng = np.random.default_rng(17)
    faces = hull(32, 64)                      # 4096 triangles
    rays = rays_around(rng, 6400, 5.0, 1.2)
    assert_same(run_check(product, faces, rays))
    L = product.load_library()
    v = np.zeros(8, np.int64)
    L.ptx_debug_bvh_visits.restype = C.c_int
    L.ptx_debug_bvh_visits.argtypes = [C.c_void_p]
    assert L.ptx_debug_bvh_visits(v.ctypes.data) == 0
    skip, ordered, wide, need, group_max, groups, tris = (int(x) for x in v[:7])
    assert 0 < wide < ordered < skip
    assert 1 <= need <= 32
    assert groups == 100 and wide <= 64 * group_max          # (per 64 rays the longest walk bounds the sum)
    assert 0 < tris < 40 * len(rays)


def test_triangle_soup_and_ties(product):
    rng = np.random.default_rng(11)
    n = 3000
    c = rng.uniform(-2, 2, size=(n, 1, 3))
    tri = (c + rng.normal(scale=0.15, size=(n, 3, 3))).astype(np.float32)
    uv = rng.uniform(0, 1, size=(n, 3, 2)).astype(np.float32)
    faces = np.concatenate([tri, uv], axis=2).reshape(n, 15)
    faces = np.concatenate([faces, faces[:500], faces[100:300]])      # exact duplicates: the lower face index must win
    rays = rays_around(rng, 30000, 5.0, 2.0)
    fl, st = assert_same(run_check(product, faces, rays))
    hit = fl[fl >= 0]
    assert len(hit) > 1000 and hit.max() < n                          # a duplicate never wins over its original


def test_axis_parallel_and_grazing(product):
    rng = np.random.default_rng(13)
    # a flat grid in the plane y = 0 (boxes of zero thickness) plus a vertical wall
    g = 40
    xs = np.linspace(-2, 2, g + 1, dtype=np.float32)
    f = []
    for i in range(g):
        for j in range(g):
            a = (xs[i], 0, xs[j], 0, 0); b = (xs[i + 1], 0, xs[j], 1, 0); c = (xs[i + 1], 0, xs[j + 1], 1, 1); d = (xs[i], 0, xs[j + 1], 0, 1)
            f.append(np.array(a + d + c, np.float32)); f.append(np.array(a + c + b, np.float32))     # facing +y
    for i in range(g):
        for j in range(g):
            a = (xs[i], xs[j] + 2, -1, 0, 0); b = (xs[i + 1], xs[j] + 2, -1, 1, 0); c = (xs[i + 1], xs[j + 1] + 2, -1, 1, 1); d = (xs[i], xs[j + 1] + 2, -1, 0, 1)
            f.append(np.array(a + b + c, np.float32)); f.append(np.array(a + c + d, np.float32))     # facing +z
    faces = np.array(f, np.float32)
    n = 8000
    o = rng.uniform(-2, 2, size=(n, 3)).astype(np.float32)
    o[:, 1] = rng.uniform(0.5, 3, size=n)
    rays = []
    down = np.tile(np.array([0, -1, 0], np.float32), (n, 1))                       # two zero components
    rays.append(np.concatenate([o, down], axis=1))
    o2 = o.copy(); o2[:, 2] = 3
    rays.append(np.concatenate([o2, np.tile(np.array([0, 0, -1], np.float32), (n, 1))], axis=1))
    # origins on the grid lines (edges shared by triangles), axis-parallel: ties between neighbours
    og = np.stack([xs[rng.integers(0, g + 1, n)], np.full(n, 1.0, np.float32), xs[rng.integers(0, g + 1, n)]], axis=1).astype(np.float32)
    rays.append(np.concatenate([og, down], axis=1))
    # grazing: almost inside the plane
    dg = rng.normal(size=(n, 3)).astype(np.float32); dg[:, 1] = -np.abs(rng.normal(scale=1e-4, size=n)).astype(np.float32)
    og2 = o.copy(); og2[:, 1] = rng.uniform(1e-4, 1e-2, size=n)
    rays.append(np.concatenate([og2, dg], axis=1))
    rays = np.concatenate(rays).astype(np.float32)
    fl, st = assert_same(run_check(product, faces, rays))
    assert (fl >= 0).sum() > n


def test_needles_tiny_and_far(product):
    rng = np.random.default_rng(17)
    n = 2000
    base = rng.uniform(-1, 1, size=(n, 3))
    dirs = rng.normal(size=(n, 3))
    tri = np.stack([base, base + dirs * rng.uniform(0.5, 2.0, size=(n, 1)), base + rng.normal(scale=1e-3, size=(n, 3))], axis=1)   # needles
    tiny = rng.uniform(-1, 1, size=(n, 1, 3)) + rng.normal(scale=1e-4, size=(n, 3, 3))
    tri = np.concatenate([tri, tiny]).astype(np.float32)
    faces = np.concatenate([tri, np.zeros((len(tri), 3, 2), np.float32)], axis=2).reshape(len(tri), 15)
    rays = np.concatenate([rays_around(rng, 20000, 3.0, 1.0), rays_around(rng, 10000, 2000.0, 1.0)])
    # aim a share of the rays straight at triangle centroids so that tiny ones are hit
    cent = tri.mean(axis=1)[rng.integers(0, len(tri), 10000)]
    o = rays_around(rng, 10000, 4.0)[:, :3]
    rays = np.concatenate([rays, np.concatenate([o, cent - o], axis=1)]).astype(np.float32)
    fl, st = assert_same(run_check(product, faces, rays))
    assert (fl >= 0).sum() > 3000


@pytest.mark.parametrize("ratio", [1e3, 1e4, 1e5, 1e6])
def test_far_origin_sweep(product, ratio):
    """Origins 10^3 ... 10^6 mesh sizes away (binary32 barycentrics are then accurate to 1e-4 ... 1e-1 of an edge: the loop's
    own hits and misses are partly rounding noise).  The traversal's per-ray slack grows with the distance (pt_bvh.h, "Why the
    tree equals the loop"), so the tree still looks at every triangle the loop can accept: zero disagreements, on a
    well-shaped hull and on a soup with needles."""
    rng = np.random.default_rng(int(ratio) % 9973)
    faces = hull(24, 48)                      # 2304 triangles, size ~ 3
    n = 6000
    rays = rays_around(rng, n, 3.0 * ratio, 1.6)
    fl, st = assert_same(run_check(product, faces, rays))
    assert (fl >= 0).sum() > n // 20
    m = 800
    base = rng.uniform(-1, 1, size=(m, 3))
    tri = np.stack([base, base + rng.normal(size=(m, 3)) * rng.uniform(0.3, 1.5, size=(m, 1)), base + rng.normal(scale=2e-2, size=(m, 3))], axis=1).astype(np.float32)
    soup = np.concatenate([tri, np.zeros((m, 3, 2), np.float32)], axis=2).reshape(m, 15)
    assert_same(run_check(product, soup, rays_around(rng, 4000, 2.0 * ratio, 1.0)))


def test_near_rays_still_prune(product):
    """The slack must not cost the tree its point: from ordinary viewing distances (2-4 mesh sizes) a ray still visits a
    small fraction of a 9216-triangle hull."""
    rng = np.random.default_rng(23)
    faces = hull(48, 96)
    rays = rays_around(rng, 8000, 8.0, 1.5)
    fl, st = assert_same(run_check(product, faces, rays))
    assert st[2] / len(rays) < 450
