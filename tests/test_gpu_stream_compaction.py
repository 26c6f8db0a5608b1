"""GPU: StreamCompaction::{Naive,Efficient,Thrust}::scan and Efficient::compact equivalents vs numpy / the CPU
entry points, integer-exact, on the sizes of the fixture plan and on path-tracer flag arrays."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, (1 << 20) - 3, 1 << 20, 1920 * 1080, (1 << 23) + 5]


@pytest.mark.parametrize("n", SIZES)
def test_scan_and_compact(gpu_product, n):
    sc = gpu_product.StreamCompaction()
    rng = np.random.default_rng(n)
    a = (rng.integers(0, 9, n) * rng.integers(0, 2, n)).astype(np.int32)
    want = np.concatenate([[0], np.cumsum(a, dtype=np.int64)[:-1]]).astype(np.int32)
    for fn in (sc.naive_scan, sc.efficient_scan, sc.thrust_scan):
        assert np.array_equal(fn(a), want)
    assert np.array_equal(sc.efficient_compact(a), a[a != 0])
    assert np.array_equal(sc.efficient_compact(a), sc.cpu_compact_with_scan(a))
    assert sc.last_gpu_ms() > 0.0


def test_degenerate_inputs(gpu_product):
    sc = gpu_product.StreamCompaction()
    assert len(sc.efficient_scan(np.zeros(0, np.int32))) == 0
    assert len(sc.efficient_compact(np.zeros(0, np.int32))) == 0
    z = np.zeros(5000, np.int32)
    assert len(sc.efficient_compact(z)) == 0 and not sc.efficient_scan(z).any()
    o = np.full(5000, -3, np.int32)
    assert np.array_equal(sc.efficient_compact(o), o)
    big = np.full(3000, 2**20, np.int32)                     # wraps around like the reference's int arithmetic
    assert np.array_equal(sc.efficient_scan(big), np.concatenate([[0], np.cumsum(big, dtype=np.int64)[:-1]]).astype(np.int32))


def test_live_flag_arrays_of_config4(gpu_product):
    """Compaction of remainingBounces>0 flags shaped like config 4's bounces: survivors/total from the golden counts."""
    sc = gpu_product.StreamCompaction()
    counts = golden("fullres_counts.npz")["c4_counts"]
    rng = np.random.default_rng(0)
    for n, live in zip(counts[:-1], counts[1:]):
        flags = np.zeros(n, np.int32)
        flags[rng.choice(n, live, replace=False)] = rng.integers(1, 8, live)
        out = sc.efficient_compact(flags)
        assert len(out) == live and np.array_equal(out, flags[flags != 0])
