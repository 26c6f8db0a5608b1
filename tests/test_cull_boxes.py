"""CPU: the candidate pre-test's boxes.  Every ray is tested against a conservative world box of every geom before the exact tests
(cullMask, pt_device.h); a box that rejects a ray the exact test would accept loses a hit silently.  The device reads the boxes as centre +
half extent (ptx_debug_cull_boxes hands out the same table): it must contain the corner box it was made from, and the device's slab
arithmetic on it -- restated here in binary32, one rounding per fused multiply-add -- must accept every ray that reaches the corner box
in exact arithmetic."""
import numpy as np
import pytest

from conftest import ROOT  # noqa: F401
import mygpuraytracer_amd as pt

f32 = np.float32


def _table(boxes):
    return pt.api.debug_cull_boxes(np.asarray(boxes, np.float32))


def _fma(a, b, c):
    # one rounding of a * b + c (binary32 operands: the product is exact in binary64, the sum is off by <= 2^-53 relative before the rounding)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def _device_slab(tab, o, d):
    """cullMask's arithmetic for rays (n, 3) against ONE box (tab: 8 floats): True = candidate."""
    tiny = f32(1e-20)
    dd = np.where(np.abs(d) < tiny, np.copysign(tiny, d), d).astype(f32)
    inv = (f32(1.0) / dd).astype(f32)          # (v_rcp_f32 is within 1 ulp of this; the boxes' inflation is 1e4 times that)
    off = (-(o * inv)).astype(f32)
    a = np.abs(inv)
    m = _fma(np.broadcast_to(tab[0:3], o.shape).astype(f32), inv, off)
    h = np.broadcast_to(tab[4:7], o.shape).astype(f32)
    near = _fma(h, (-a).astype(f32), m)
    far = _fma(h, a, m)
    with np.errstate(invalid="ignore"):
        tn = np.maximum(np.maximum(near[:, 0], near[:, 1]), near[:, 2])
        tf = np.minimum(np.minimum(far[:, 0], far[:, 1]), far[:, 2])
        culled = (tf < tn) | (tf < 0)
    return ~culled


def _exact_reach(lo, hi, o, d):
    """does the ray o + t d, t >= 0, touch the closed box [lo, hi]?  binary64, axis by axis (a zero direction component = inside the slab or not)"""
    o = o.astype(np.float64); d = d.astype(np.float64)
    tn = np.full(len(o), 0.0); tf = np.full(len(o), np.inf); ok = np.ones(len(o), bool)
    for k in range(3):
        z = d[:, k] == 0
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = (lo[k] - o[:, k]) / d[:, k]; t1 = (hi[k] - o[:, k]) / d[:, k]
        a = np.minimum(t0, t1); b = np.maximum(t0, t1)
        ok &= np.where(z, (o[:, k] >= lo[k]) & (o[:, k] <= hi[k]), True)
        tn = np.where(z, tn, np.maximum(tn, a)); tf = np.where(z, tf, np.minimum(tf, b))
    return ok & (tn <= tf)


def test_table_contains_the_corner_box():
    rng = np.random.default_rng(5)
    lo = rng.uniform(-1, 1, (4000, 3)) * 10.0 ** rng.uniform(-3, 5, (4000, 1))
    ext = rng.uniform(0, 1, (4000, 3)) * 10.0 ** rng.uniform(-6, 5, (4000, 1))
    boxes = np.concatenate([lo, lo + ext], 1).astype(f32)
    boxes[:, 3:] = np.maximum(boxes[:, 3:], boxes[:, :3])
    tab = _table(boxes).astype(np.float64)
    c, h = tab[:, 0:3], tab[:, 4:7]
    assert np.all(c - h <= boxes[:, :3]) and np.all(c + h >= boxes[:, 3:])
    assert np.all(tab[:, 3] == 0) and np.all(tab[:, 7] == 0)
    # ... and not absurdly larger: the half extent exceeds the corner box's by rounding only
    assert np.all(h <= 0.5 * (boxes[:, 3:].astype(np.float64) - boxes[:, :3]) * (1 + 1e-5) + 1e-6 * (np.abs(c) + 1e-30) + 1e-38)


def test_unbounded_and_degenerate_boxes():
    inf = np.inf
    tab = _table([[-inf, -inf, -inf, inf, inf, inf], [-inf, 0, 0, 1, 1, inf], [2, 2, 2, 2, 2, 2], [-3e38, 0, 0, 3e38, 1, 1]])
    assert np.all(tab[0, 0:3] == 0) and np.all(np.isinf(tab[0, 4:7]))
    assert tab[1, 0] == 0 and np.isinf(tab[1, 4]) and np.isinf(tab[1, 6]) and np.isfinite(tab[1, 5])
    assert np.all(tab[2, 0:3] == 2) and np.all(tab[2, 4:7] >= 0)
    assert np.isinf(tab[3, 4]) or tab[3, 4] >= 3e38                      # (out of the finite range: unbounded on that axis)
    # an unbounded box is a candidate for every ray, whatever the arithmetic makes of the infinities (NaN = not culled)
    rng = np.random.default_rng(6)
    o = rng.uniform(-50, 50, (2000, 3)).astype(f32)
    d = rng.normal(size=(2000, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
    d[:50, 0] = 0; d[50:100, 1] = -0.0
    assert _device_slab(tab[0], o, d).all()


@pytest.mark.parametrize("scale", [1.0, 1e-2, 300.0])
def test_device_slab_never_rejects_a_ray_that_reaches_the_box(scale):
    """Boxes as make_world_aabb builds them (the exact extent inflated by 1e-3 + 1e-4 |coordinate|), rays aimed at, past and along them --
    grazing rays and rays starting on a face included: whoever reaches the UNinflated box in exact arithmetic is a candidate."""
    rng = np.random.default_rng(int(scale * 10) + 7)
    bad = 0
    for case in range(60):
        lo = rng.uniform(-8, 8, 3) * scale
        hi = lo + rng.uniform(0.01, 6, 3) * scale * (rng.random(3) < 0.8) + 1e-3 * scale      # (some thin walls)
        m = 1e-3 + 1e-4 * np.maximum(np.abs(lo), np.abs(hi))
        corner = np.concatenate([np.nextafter(f32(lo - m), f32(-np.inf)), np.nextafter(f32(hi + m), f32(np.inf))]).astype(f32)
        tab = _table([corner])[0]
        n = 4000
        o = (rng.uniform(-12, 12, (n, 3)) * scale).astype(f32)
        target = lo + (hi - lo) * rng.uniform(-0.02, 1.02, (n, 3))                              # at and just past the faces and edges
        d = target - o
        d[: n // 4] = rng.normal(size=(n // 4, 3))                                              # anywhere
        on = slice(n // 2, n // 2 + n // 8)                                                     # origins ON a face, leaving and entering
        face = lo + (hi - lo) * rng.uniform(0, 1, (n // 8, 3)); k = rng.integers(0, 3, n // 8)
        face[np.arange(n // 8), k] = np.where(rng.random(n // 8) < 0.5, lo[k], hi[k])
        o[on] = face.astype(f32)
        ax = slice(n - n // 8, n)                                                               # axis-parallel rays (zero components)
        d[ax] = 0; d[np.arange(n - n // 8, n), rng.integers(0, 3, n // 8)] = rng.choice([-1.0, 1.0], n // 8)
        d = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)).astype(f32)
        reach = _exact_reach(lo, hi, o, d)
        cand = _device_slab(tab, o, d)
        bad += int(np.sum(reach & ~cand))
        assert reach.sum() > n // 10
    assert bad == 0
