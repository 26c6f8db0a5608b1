"""CPU: the N>1 host path (mygpuraytracer_amd/multigpu.py) with world_size 2 over gloo.  The HIP renderer is replaced
by the CPU oracle restricted to the rank's row tiles, so what is tested is the driver: the ownership rule, the
full-frame accumulation buffers with foreign rows left zero, the reduce to rank 0 and the ray total."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import HERE, ROOT, golden, dump_from_golden

RES, DEPTH, ROWS, ITERS = (48, 40), 6, 4, 2


def _scene_dump(O):
    g = golden("loader_cornellObj.npz")
    d = dump_from_golden(g, cam="cam_floats")
    cf = d["cam_floats"]
    d["cam_floats"] = O.camera_from_loader(RES[0], RES[1], float(cf[16]), cf[0:3], cf[3:6], cf[9:12])
    ci = d["cam_ints"].copy(); ci[0], ci[1], ci[3] = RES[0], RES[1], DEPTH
    d["cam_ints"] = ci
    return d


def _oracle_tile(rank, world):
    sys.path.insert(0, HERE)
    from cpulibs import OracleLib
    O = OracleLib()
    O.set_libm(1)
    O.create(_scene_dump(O)); O.apply_runcuda_camera()
    O.set_tile(ROWS, rank, world)
    O.pt_init()
    rays = 0
    for it in range(1, ITERS + 1):
        O.iterate(it)
        rays += int(O.live_counts().sum())
    return O.image().reshape(-1).copy(), rays, O.pixelcount()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mygpuraytracer_amd import multigpu

    def renderer(image, iter_first, count):
        assert (iter_first, count) == (1, ITERS)
        img, rays, owned = _oracle_tile(rank, world)
        assert owned == len(multigpu.owned_rows(RES[1], ROWS, rank, world)) * RES[0]
        image += torch.from_numpy(img)
        return rays

    image, rays = multigpu.render_distributed(renderer, RES[0], RES[1], 1, ITERS, torch.device("cpu"))
    q.put((rank, None if image is None else image.numpy().copy(), rays))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_render_and_reduce():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        rank, img, rays = q.get(timeout=180)
        got[rank] = (img, rays)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1][0] is None and got[0][0] is not None
    tiles = [_oracle_tile(r, 2) for r in range(2)]
    want = tiles[0][0] + tiles[1][0]
    assert np.array_equal(got[0][0], want)
    assert got[0][1] == got[1][1] == tiles[0][1] + tiles[1][1]
    # the two tiles are disjoint and cover the frame
    a, b = tiles[0][0].reshape(RES[1], RES[0] * 3), tiles[1][0].reshape(RES[1], RES[0] * 3)
    from mygpuraytracer_amd import multigpu
    r0, r1 = multigpu.owned_rows(RES[1], ROWS, 0, 2), multigpu.owned_rows(RES[1], ROWS, 1, 2)
    assert sorted(r0 + r1) == list(range(RES[1])) and not set(r0) & set(r1)
    assert not a[r1].any() and not b[r0].any()


def test_owned_rows_rule():
    from mygpuraytracer_amd import multigpu
    for H, rows, world in ((1080, 16, 8), (1080, 16, 3), (7, 4, 2), (2160, 8, 8)):
        seen = []
        for r in range(world):
            seen += multigpu.owned_rows(H, rows, r, world)
        assert sorted(seen) == list(range(H))
    # balance at the bench geometry: 1080 rows, TILE_ROWS-row blocks, 8 ranks -> 128..136 rows each
    sizes = [len(multigpu.owned_rows(1080, multigpu.TILE_ROWS, r, 8)) for r in range(8)]
    assert max(sizes) - min(sizes) <= multigpu.TILE_ROWS


# ---- iteration sharding: rank r traces iterations r+1, r+1+N, ... of the full frame ---------------------------------
ITERS_TURNS = 5


def _oracle_iterations(iters):
    """sum of the given iterations of the full frame, each traced alone (image reset in between) -> (sum, rays)"""
    sys.path.insert(0, HERE)
    from cpulibs import OracleLib
    O = OracleLib()
    O.set_libm(1)
    O.create(_scene_dump(O)); O.apply_runcuda_camera()
    O.pt_init()
    rays = 0
    for it in iters:
        O.iterate(it)
        rays += int(O.live_counts().sum())
    return O.image().reshape(-1).copy(), rays


def _worker_turns(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mygpuraytracer_amd import multigpu

    def renderer(image, iter_first, count):
        first, n = multigpu.iteration_share(iter_first, count, rank, world)
        img, rays = _oracle_iterations([first + k * world for k in range(n)])
        image += torch.from_numpy(img)
        return rays

    image, rays = multigpu.render_distributed(renderer, RES[0], RES[1], 1, ITERS_TURNS, torch.device("cpu"))
    q.put((rank, None if image is None else image.numpy().copy(), rays))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_taking_turns_reproduce_the_single_run():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_turns, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        rank, img, rays = q.get(timeout=180)
        got[rank] = (img, rays)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single, single_rays = _oracle_iterations(range(1, ITERS_TURNS + 1))
    odd, _ = _oracle_iterations([1, 3, 5])
    even, _ = _oracle_iterations([2, 4])
    assert np.array_equal(got[0][0], odd + even)                      # exactly the two partial sums, added once
    assert got[0][1] == got[1][1] == single_rays                      # the same streams were traced, ray for ray
    # against the single run only the order of the fp32 additions differs
    assert np.allclose(got[0][0], single, rtol=ITERS_TURNS * 2.0 ** -23, atol=0)


def test_iteration_share_rule():
    from mygpuraytracer_amd import multigpu
    for first, count, world in ((1, 200, 8), (21, 7, 8), (1, 1, 2), (5, 16, 3)):
        seen = []
        for r in range(world):
            f, n = multigpu.iteration_share(first, count, r, world)
            seen += [f + k * world for k in range(n)]
        assert sorted(seen) == list(range(first, first + count))


# ---- the gather of owned rows as the exchange (instead of the reduce) ------------------------------------------------
def _worker_gather(rank, world, port, q, W, H, rows):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mygpuraytracer_amd import multigpu
    rng = np.random.default_rng(100 + rank)
    img = np.zeros((H, W * 3), np.float32)
    own = multigpu.owned_rows(H, rows, rank, world)
    img[own] = rng.random((len(own), W * 3), dtype=np.float32) + np.float32(rank)
    a, b = torch.from_numpy(img.reshape(-1).copy()), torch.from_numpy(img.reshape(-1).copy())
    gathered = multigpu.assemble_tiles(a, W, H, rows, dst=0)
    # the same through a padded frame buffer (strided-view pack and unpack, what bench.py uses)
    c = multigpu.frame_buffer(W, H, world, "cpu", tile_rows=rows)
    c[:W * H * 3] = torch.from_numpy(img.reshape(-1).copy())
    fast = multigpu.assemble_tiles(c, W, H, rows, dst=0)
    assert (fast is None) == (gathered is None)
    if fast is not None:
        assert torch.equal(fast[:W * H * 3], gathered) and not fast[W * H * 3:].any()
    dist.reduce(b, dst=0, op=dist.ReduceOp.SUM)
    q.put((rank, None if gathered is None else gathered.numpy().copy(), b.numpy().copy() if rank == 0 else None, img.reshape(-1)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,rows", [(3, 16, 41, 4), (2, 8, 7, 8), (8, 16, 1080, 8), (4, 12, 2160, 8)])      # (the last two: the row layout of the 8- and 4-GPU runs of C4 and C5)
def test_gather_of_owned_rows_equals_the_reduce(world, W, H, rows):
    """multigpu.assemble_tiles: ranks own different numbers of rows (41 rows in blocks of 4 over 3 ranks; a rank that owns
    nothing: 7 rows in one block of 8 over 2 ranks) -- the packs are padded, the frame assembled on rank 0 equals what
    reduce(SUM) of the full buffers gives, bit for bit, and the other ranks keep their buffers."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_gather, args=(r, world, port, q, W, H, rows)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, gathered, reduced, own = q.get(timeout=180)
        got[rank] = (gathered, reduced, own)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(got[r][0] is None for r in range(1, world))
    want = sum(got[r][2] for r in range(world))
    assert np.array_equal(got[0][0], got[0][1]) and np.array_equal(got[0][0], want)


def _worker_agree(rank, world, port, q, failing_rank):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mygpuraytracer_amd import multigpu
    log = []

    def phase(name, fails):
        def fn():
            if fails and rank == failing_rank:
                raise RuntimeError("%s broke here" % name)
        ok, msg = multigpu.agreed_phase(fn)
        log.append((name, ok, msg))
        return ok

    # the shape of bench.py's N-rank C5 leg: local phases chained by `and`, a collective only behind an agreed success, and a
    # collective AFTER the leg that every rank must still reach whether or not the leg ran
    if phase("assets", False) and phase("create", True) and phase("warm", False):
        dist.barrier()
        log.append(("collective", True, None))
    t = torch.tensor([rank + 1])
    dist.all_reduce(t)                       # the roofline leg's stand-in: hangs if a rank left the protocol early
    q.put((rank, log, int(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("failing_rank", [0, 1])
def test_a_rank_local_failure_is_every_ranks_failure(failing_rank):
    """multigpu.agreed_phase (bench.py's N-rank legs): one rank raising inside a local phase makes EVERY rank skip the collectives
    behind it and meet again at the next common point -- nobody waits in a barrier for a rank that has moved on."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_agree, args=(r, world, port, q, failing_rank)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, log, total = q.get(timeout=120)
        got[rank] = (log, total)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        log, total = got[rank]
        assert total == 3                                                   # both ranks reached the common collective
        assert [(n, ok) for n, ok, _ in log] == [("assets", True), ("create", False)]      # same verdicts everywhere, nothing behind the failure ran
        msg = log[1][2]
        assert ("rank %d: create broke here" % failing_rank) == msg if rank == failing_rank else msg == "another rank failed"
