"""CPU: the portable sin/cos/pow routines (oracle copy; the HIP kernels run the same operation sequence, checked
bit for bit on the GPU in test_gpu_parity.py) against glibc, which is what the reference's host pass calls."""
import ctypes as C

import numpy as np


def _ulps(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def test_sincos_matches_glibc(oracle_lib):
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = C.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [C.c_float]
    rng = np.random.default_rng(3)
    # the path tracer's arguments: u01 * TWO_PI (hemisphere sampling) and the lens angle in [-pi/4, 3pi/4]
    x = np.concatenate([(rng.random(200000) * np.float32(6.2831855)).astype(np.float32),
                        rng.uniform(-0.8, 2.4, 50000).astype(np.float32),
                        np.float32([0, 6.2831855, 3.1415927, 1.5707964, 4.712389, 1e-30, -1e-30])])
    s = np.zeros_like(x); c = np.zeros_like(x)
    gs = np.zeros_like(x); gc = np.zeros_like(x)
    for k, v in enumerate(x):
        s[k], c[k] = oracle_lib.own_sincosf(v)
        gs[k], gc[k] = libm.sinf(float(v)), libm.cosf(float(v))
    ds, dc = _ulps(s, gs), _ulps(c, gc)
    assert ds.max() <= 1 and dc.max() <= 1
    # glibc's sinf/cosf are faithful (< 1 ulp) but not correctly rounded: about 1.3 % of these arguments come out
    # one ulp away from the portable routine, which IS correctly rounded (checked against float64 below)
    assert (ds != 0).mean() < 0.03 and (dc != 0).mean() < 0.03
    for got, f in ((s, np.sin), (c, np.cos)):
        exact = f(x.astype(np.float64))
        assert np.all(np.abs(got.astype(np.float64) - exact) <= np.spacing(np.abs(got)).astype(np.float64) * 0.5001 + 1e-45)


def test_pow5_and_powf(oracle_lib):
    libm = C.CDLL("libm.so.6")
    libm.powf.restype = C.c_float; libm.powf.argtypes = [C.c_float, C.c_float]
    libm.pow.restype = C.c_double; libm.pow.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(4)
    for v in rng.uniform(-0.2, 2.0, 20000):
        a, b = oracle_lib.lib.o_own_pow5(float(v)), libm.pow(float(v), 5.0)
        assert a == b or abs(a - b) <= 4 * np.spacing(abs(b))
    bad = 0
    xs = rng.random(20000).astype(np.float32); ys = rng.uniform(0, 200, 20000).astype(np.float32)
    for x, y in zip(xs, ys):
        a, b = np.float32(oracle_lib.lib.o_own_powf(float(x), float(y))), np.float32(libm.powf(float(x), float(y)))
        d = _ulps([a], [b])[0]
        assert d <= 1, (x, y, a, b)
        bad += d != 0
        exact = np.float64(x) ** np.float64(y)
        if a > 1e-37:                                   # normal range: the portable routine is correctly rounded
            assert abs(np.float64(a) - exact) <= np.float64(np.spacing(a)) * 0.5001, (x, y, a, exact)
    assert bad < 100                                    # glibc's powf is faithful, not correctly rounded (~0.1 %)
    for x, y, want in [(0.0, 0.0, 1.0), (float("nan"), 0.0, 1.0), (0.0, 3.0, 0.0), (1.0, 77.0, 1.0), (0.5, 2.0, 0.25), (2.0, 10.0, 1024.0)]:
        assert oracle_lib.lib.o_own_powf(x, y) == want
