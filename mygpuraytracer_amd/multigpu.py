"""Row-tile sharding of one frame over the ranks of a torch.distributed job (one process per GPU, RCCL over xGMI).

The reference has no multi-GPU code; this is the path north_star prescribes: every rank traces the
row blocks it owns (interleaved blocks of TILE_ROWS rows, round-robin: the Cornell frame's cost per row is uneven,
interleaving balances it), accumulating into a full-frame fp32 buffer in which rows it does not own stay zero; one
``reduce(SUM)`` of that buffer to rank 0 per batch of iterations assembles the frame.  There is no collective on the
data path itself -- tiles are independent streams.

A second mode shards ITERATIONS instead of pixels (SURVEY 8(e) "next", item 1): rank r traces iterations r+1, r+1+N,
... of the full frame.  Every iteration then sees exactly the streams the single-GPU run sees, so the summed frame
equals the single-GPU frame up to the order of the fp32 additions -- which pixel tiles cannot offer, because the
shading RNG is seeded by a path's position in the sorted stream.  Same single reduce at the end.

The renderer is injected so that the same driver runs on GPUs (``hip_tile_renderer``: the HIP tracer writing straight
into the torch tensor that RCCL reduces) and, in the CPU tests, over gloo with a stand-in renderer.
"""
import torch
import torch.distributed as dist

TILE_ROWS = 8


class RankFailed(RuntimeError):
    """Raised by EVERY rank at the same point when some rank's local phase failed (see all_ok)."""


def all_ok(ok, device=None):
    """Every rank's verdict on a phase it ran LOCALLY, agreed on by all (all_reduce MIN): one rank's failure becomes every rank's
    failure at the same point, so no rank goes on to a collective the failed one will never reach.  `device`: where the flag lives
    (the rank's GPU for RCCL, None / cpu for gloo).  Itself a synchronisation point, like a barrier."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def agreed_phase(fn, device=None):
    """Runs fn() on this rank under a try, then agrees with the other ranks on the outcome.  Returns (ok, message): ok is the same
    on every rank; message names this rank's own exception, or says that another rank failed."""
    err = None
    try:
        fn()
    except Exception as e:          # noqa: BLE001 -- whatever it is, the other ranks must hear of it
        err = e
    ok = all_ok(err is None, device)
    if ok:
        return True, None
    return False, ("rank %d: %s" % (dist.get_rank(), str(err)[:260])) if err is not None else "another rank failed"


def owned_rows(height, tile_rows, rank, world):
    """Rows y with (y // tile_rows) % world == rank -- the same rule the kernels apply (pt_engine.hip owned_pixel)."""
    return [y for y in range(height) if (y // tile_rows) % world == rank]


def padded_rows(height, tile_rows, world):
    """Rows of a frame buffer padded to whole rounds of `world` blocks of `tile_rows` rows: in such a buffer the blocks of
    rank r are the strided view blocks[r::world], so packing and unpacking the owned rows need no index tensors."""
    per_round = tile_rows * world
    return (height + per_round - 1) // per_round * per_round


def frame_buffer(width, height, world, device, tile_rows=TILE_ROWS):
    """Zeroed accumulation buffer for a row-tile run: `padded_rows` x width x 3 floats.  The tracer gets its data_ptr() as
    external_image and only ever touches the first height*width*3 floats (the frame); the padding rows stay zero."""
    return torch.zeros(padded_rows(height, tile_rows, max(world, 1)) * width * 3, dtype=torch.float32, device=device)


def assemble_tiles(image, width, height, tile_rows=TILE_ROWS, dst=0, via_host=False):
    """The alternative to the reduce for pixel-row tiles ("gather of disjoint ranges, 7/8 of the bytes, no
    adds"): every rank packs the rows it owns into one contiguous buffer, ``gather`` (RCCL: point-to-point sends over xGMI,
    all seven links into `dst` at once) brings the world's packs to `dst`, which copies each into its rows.  1/world of the
    reduce's bytes per rank.  The frame on `dst` is bit-identical to what reduce(SUM) gives, since foreign rows hold zeros
    there.  `image` is the flat accumulation buffer: W*H*3 floats, or -- faster -- a `frame_buffer` (padded to whole rounds
    of blocks: pack = one strided copy, unpack = one strided copy, no index tensors, three launches in all on `dst`).
    Ranks other than `dst` keep theirs unchanged.  via_host: stage through host memory (gloo rehearsal of a GPU run)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    comm_dev = torch.device("cpu") if via_host else image.device
    if image.numel() == padded_rows(height, tile_rows, world) * width * 3:
        blk = tile_rows * width * 3
        v = image.view(-1, world, blk)                    # [round][rank][block]
        pack = v[:, rank, :].contiguous().to(comm_dev)
        buf = torch.empty((world,) + tuple(pack.shape), dtype=image.dtype, device=comm_dev) if rank == dst else None
        dist.gather(pack, list(buf.unbind(0)) if rank == dst else None, dst=dst)
        if rank == dst:
            v.copy_(buf.permute(1, 0, 2))
        return image if rank == dst else None
    frame = image.view(height, width * 3)
    rows = [torch.tensor(owned_rows(height, tile_rows, r, world), dtype=torch.long) for r in range(world)]
    most = max(len(r) for r in rows)                      # the last ranks may own one block less: pad the pack
    pack = torch.zeros(most, width * 3, dtype=image.dtype, device=comm_dev)
    mine = rows[rank].to(image.device)
    if len(mine):
        pack[:len(mine)] = frame.index_select(0, mine).to(comm_dev)
    packs = [torch.empty_like(pack) for _ in range(world)] if rank == dst else None
    dist.gather(pack, packs, dst=dst)
    if rank == dst:
        for r in range(world):
            if r != dst and len(rows[r]):
                frame.index_copy_(0, rows[r].to(image.device), packs[r][:len(rows[r])].to(image.device))
    return image if rank == dst else None


def render_distributed(renderer, width, height, iter_first, count, device, reduce_to=0):
    """Runs `count` iterations of this rank's tile and reduces the accumulation buffers.

    renderer(image_tensor, iter_first, count) must add this rank's radiance into image_tensor (flat W*H*3 fp32 on
    `device`) and return the number of ray-bounces it traced.  Returns (image on reduce_to or None, total rays).
    """
    image = torch.zeros(width * height * 3, dtype=torch.float32, device=device)
    if image.is_cuda:       # the tracer works on a stream of its own: the fill above (torch's current stream) must have landed
        torch.cuda.current_stream(image.device).synchronize()
    rays = renderer(image, iter_first, count)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(image, dst=reduce_to, op=dist.ReduceOp.SUM)
        total = torch.tensor([rays], dtype=torch.int64, device=device)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        rays = int(total.item())
        if dist.get_rank() != reduce_to:
            image = None
    return image, rays


def iteration_share(iter_first, count, rank, world):
    """(first, n): the iterations first, first + world, ... (n of them) that `rank` takes out of
    iter_first .. iter_first + count - 1 when the ranks take turns."""
    n = (count - rank + world - 1) // world if count > rank else 0
    return iter_first + rank, n


def hip_iteration_renderer(scene, rank, world, **opt_kw):
    """Renderer for render_distributed that shards iterations: full frame on every rank, every world-th iteration."""
    from . import api

    state = {}

    def renderer(image, iter_first, count):
        if "tracer" not in state or state["ptr"] != image.data_ptr():
            if "tracer" in state:
                state["tracer"].close()
            kw = dict(opt_kw)
            kw.setdefault("device", image.device.index if image.device.index is not None else 0)
            state["tracer"] = api.Tracer(scene, external_image_ptr=image.data_ptr(), **kw)
            state["ptr"] = image.data_ptr()
        t = state["tracer"]
        before = t.stats()["rays_total"]
        first, n = iteration_share(iter_first, count, rank, world)
        if n:
            t.render(first, n, stride=world)
        t.synchronize()
        return t.stats()["rays_total"] - before

    renderer.state = state
    return renderer


def hip_tile_renderer(scene, rank, world, tile_rows=TILE_ROWS, **opt_kw):
    """Builds the HIP tracer for this rank's tile; returns (renderer, make_tracer_for(image_tensor))."""
    from . import api

    state = {}

    def renderer(image, iter_first, count):
        if "tracer" not in state or state["ptr"] != image.data_ptr():
            if "tracer" in state:
                state["tracer"].close()
            kw = dict(opt_kw)
            if world > 1:
                kw.update(tile_rows=tile_rows, tile_rank=rank, tile_world=world)
            kw.setdefault("device", image.device.index if image.device.index is not None else 0)
            state["tracer"] = api.Tracer(scene, external_image_ptr=image.data_ptr(), **kw)
            state["ptr"] = image.data_ptr()
        t = state["tracer"]
        before = t.stats()["rays_total"]
        t.render(iter_first, count)
        t.synchronize()
        return t.stats()["rays_total"] - before

    renderer.state = state
    return renderer
