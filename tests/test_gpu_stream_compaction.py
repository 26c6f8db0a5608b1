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


@pytest.mark.parametrize("n", [1, 16383, 16384, 16385, 64 * 16384 + 1, 65 * 16384, 1920 * 1080, 3840 * 2160, (1 << 26) + 3])
def test_device_pointer_forms(gpu_product, n):
    """sc_scan_device / sc_compact_device on buffers already in HBM (torch only holds the memory): every tile count around
    the look-back window of 64, unaligned views (pointer + 4 bytes), in-place scan, a side stream, repeated use of one
    workspace; integer-exact against numpy."""
    import torch
    sc = gpu_product.StreamCompaction()
    rng = np.random.default_rng(n)
    a = (rng.integers(-4, 9, n) * (rng.random(n) < 0.4)).astype(np.int32)
    want_scan = np.concatenate([[0], np.cumsum(a, dtype=np.int64)[:-1]]).astype(np.int32)
    want_keep = a[a != 0]
    dev = torch.device("cuda", 0)
    buf_in = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    buf_out = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    ws = torch.zeros((sc.workspace_bytes(n) + 7) // 8, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev)
    host = torch.from_numpy(a)
    for shift in (0, 1):                                    # shift 1: 4-byte-aligned only -> the scalar load/store path
        d_in, d_out = buf_in[shift:shift + n], buf_out[shift:shift + n]
        d_in.copy_(host)
        torch.cuda.synchronize()
        for stream in (0, side.cuda_stream):
            d_out.zero_(); torch.cuda.synchronize()
            sc.scan_device(n, d_out.data_ptr(), d_in.data_ptr(), ws.data_ptr(), stream)
            torch.cuda.synchronize()
            assert np.array_equal(d_out.cpu().numpy(), want_scan)
            d_out.fill_(-77); torch.cuda.synchronize()
            sc.compact_device(n, d_out.data_ptr(), d_in.data_ptr(), count.data_ptr(), ws.data_ptr(), stream)
            torch.cuda.synchronize()
            k = int(count.item())
            assert k == len(want_keep)
            got = d_out.cpu().numpy()
            assert np.array_equal(got[:k], want_keep) and np.all(got[k:] == -77)      # nothing written past the survivors
        sc.scan_device(n, d_in.data_ptr(), d_in.data_ptr(), ws.data_ptr(), 0)         # in place
        torch.cuda.synchronize()
        assert np.array_equal(d_in.cpu().numpy(), want_scan)


@pytest.mark.parametrize("n", [1, 5, 4096, 16385, 1920 * 1080 + 3])
def test_common_kernels_build_the_reference_pipeline(gpu_product, n):
    """StreamCompaction::Common::kernMapToBoolean / kernScatter (stream_compaction/common.cu:25-49) as sc_map_to_boolean_device /
    sc_scatter_device: the reference's own compaction, map -> exclusive scan -> scatter (efficient.cu:100-125), put together from the
    three device entry points equals sc_compact_device and numpy, integer for integer -- aligned and 4-byte-aligned-only views."""
    import torch
    sc = gpu_product.StreamCompaction()
    rng = np.random.default_rng(n)
    a = (rng.integers(-4, 9, n) * (rng.random(n) < 0.4)).astype(np.int32)
    dev = torch.device("cuda", 0)
    ws = torch.zeros((sc.workspace_bytes(n) + 7) // 8, dtype=torch.int64, device=dev)
    for shift in (0, 1):
        store = [torch.zeros(n + 1, dtype=torch.int32, device=dev) for _ in range(4)]
        d_in, d_bools, d_idx, d_out = [b[shift:shift + n] for b in store]
        d_in.copy_(torch.from_numpy(a))
        d_bools.fill_(-5); d_out.fill_(-77)
        torch.cuda.synchronize()
        sc.map_to_boolean_device(n, d_bools.data_ptr(), d_in.data_ptr())
        sc.scan_device(n, d_idx.data_ptr(), d_bools.data_ptr(), ws.data_ptr())
        sc.scatter_device(n, d_out.data_ptr(), d_in.data_ptr(), d_bools.data_ptr(), d_idx.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_bools.cpu().numpy(), (a != 0).astype(np.int32))
        keep = a[a != 0]
        got = d_out.cpu().numpy()
        assert np.array_equal(got[:len(keep)], keep) and np.all(got[len(keep):] == -77)
        assert store[1][n if shift == 0 else 0].item() == 0                    # nothing written outside the n elements
    sc.map_to_boolean_device(0, 0, 0)                                          # n = 0: nothing happens
    with pytest.raises(gpu_product.PathTracerError):
        sc.scatter_device(4, 0, 0, 0, 0)


def test_device_forms_degenerate(gpu_product):
    import torch
    sc = gpu_product.StreamCompaction()
    dev = torch.device("cuda", 0)
    count = torch.full((1,), 5, dtype=torch.int32, device=dev)
    ws = torch.zeros(16, dtype=torch.int64, device=dev)
    sc.compact_device(0, 0, 0, count.data_ptr(), ws.data_ptr())          # n = 0: count := 0, nothing else touched
    torch.cuda.synchronize()
    assert int(count.item()) == 0
    sc.scan_device(0, 0, 0, 0)
    with pytest.raises(gpu_product.PathTracerError):
        sc.scan_device(8, 0, 0, 0)
    with pytest.raises(gpu_product.PathTracerError):
        sc.scan_device(8, ws.data_ptr(), ws.data_ptr(), ws.data_ptr() + 4)   # misaligned workspace
    ones = torch.ones(1 << 22, dtype=torch.int32, device=dev)               # every element survives, prefix = index
    out = torch.zeros_like(ones)
    big_ws = torch.zeros((sc.workspace_bytes(1 << 22) + 7) // 8, dtype=torch.int64, device=dev)
    sc.scan_device(1 << 22, out.data_ptr(), ones.data_ptr(), big_ws.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out, torch.arange(1 << 22, dtype=torch.int32, device=dev))
    sc.compact_device(1 << 22, out.data_ptr(), ones.data_ptr(), count.data_ptr(), big_ws.data_ptr())
    torch.cuda.synchronize()
    assert int(count.item()) == 1 << 22 and bool((out == 1).all())


def test_reference_namespaces_through_the_cpp_veneer(gpu_product, tmp_path):
    """tests/sc_veneer_check.cpp: StreamCompaction::{Naive,Efficient,Thrust}::scan and Efficient::compact with the
    reference's spelling and host pointers (csrc/stream_compaction_api.h) agree with StreamCompaction::CPU on sizes around
    the tile size; the per-namespace timers report the previous operation."""
    import os
    import subprocess
    from conftest import ROOT
    lib_dir = os.path.join(ROOT, "mygpuraytracer_amd")
    exe = tmp_path / "sc_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", str(exe), os.path.join(ROOT, "tests", "sc_veneer_check.cpp"),
                           "-L" + lib_dir, "-lmi355x_pathtracer", "-Wl,-rpath," + lib_dir])
    out = subprocess.check_output([str(exe)], text=True)
    assert "all: 0 mismatches" in out
