#!/usr/bin/env python3
"""Predicted strong scaling of the driver's scaling run (`bench.py --gpus N --steps K --warmup W`, default K 20, W 5) from ONE GPU:
every rank's tile of an N-way split is traced in turn with bench.py's own sequence (warm-up, clock warm-up, K timed steps), the
slowest rank taken; the exchange is the part one GPU cannot show in full, so it is split into
  * measured here with ONE RCCL rank (this script under `python -m torch.distributed.run --nproc-per-node 1`): pack + RCCL gather
    (or reduce) + unpack + the closing barrier -- the software cost of the collective calls on this box;
  * modelled: the bytes that have to cross xGMI into rank 0, at LINK_GBPS per link (gather: (N-1) packs over N-1 links in
    parallel = one pack time; reduce: ring, 2(N-1)/N of the buffer over one link).
Prints one JSON line per N and a markdown table, for C4 (bench.py's timed region) and for C5 at 3840x2160 (its `c5` leg: 72 steps after 36;
PREDICT_ONLY=C4 or C5 picks one).   gpu_scale_predict.py [K] [W]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import mygpuraytracer_amd as pt
from mygpuraytracer_amd import multigpu
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Wm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
LINK_GBPS = 48.0        # one xGMI link, one direction, effective (64 GB/s peak)
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
have_dist = "RANK" in os.environ
if have_dist:
    dist.init_process_group("nccl", device_id=dev)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ensure_standin_assets
ensure_standin_assets()
# the two workloads bench.py --gpus N reports: C4 (the timed region: K steps after W) and C5 (its `c5` leg: 72 steps after 36, at 4K)
WORKLOADS = [("C4", "cornellObj.txt", (1920, 1080), {}, K, Wm, True),
             ("C5", "cornellSpaceship20k.txt", (3840, 2160), dict(depth_of_field=1), 72, 36, False)]
if os.environ.get("PREDICT_ONLY"):
    WORKLOADS = [w for w in WORKLOADS if w[0] in os.environ["PREDICT_ONLY"].split(",")]


def timed_render(s, W, H, opt, world, rank, steps, warm, clock_warmup):
    img = multigpu.frame_buffer(W, H, world, dev); torch.cuda.current_stream().synchronize()
    kw = dict(opt)
    if world > 1:
        kw.update(tile_rows=multigpu.TILE_ROWS, tile_rank=rank, tile_world=world)
    with pt.Tracer(s, external_image_ptr=img.data_ptr(), **kw) as T:
        t0 = time.perf_counter(); T.render(1, warm); T.synchronize()
        while clock_warmup and time.perf_counter() - t0 < 0.15:
            T.render(10_000_000, 36); T.synchronize()
        ts = []
        for rep in range(5 if clock_warmup else 3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); T.render(warm + 1 + rep * steps, steps); T.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


def exchange_software_ms(W, H, world, mode):
    """the collective's calls with one rank: same tensors, same launches as a rank of `world` would issue (its pack is 1/world of the frame)"""
    if not have_dist:
        return None
    img = multigpu.frame_buffer(W, H, world, dev)
    blk = multigpu.TILE_ROWS * W * 3
    v = img.view(-1, world, blk)
    ts = []
    for rep in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == "gather":
            pack = v[:, 0, :].contiguous()
            buf = torch.empty((1,) + tuple(pack.shape), dtype=img.dtype, device=dev)
            dist.gather(pack, list(buf.unbind(0)), dst=0)
            v[:, 0, :].copy_(buf[0])
        else:
            dist.reduce(img[:W * H * 3], dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts[2:])[len(ts[2:]) // 2] * 1e3


for tag, scene, (W, H), opt, steps, warm, clock_warmup in WORKLOADS:
    s = pt.Scene(os.path.join(ROOT, "scenes", scene), res=(W, H), depth=8); s.apply_runcuda_camera()
    rows, base = [], None
    for world in (1, 2, 4, 8):
        # (C5: the slowest rank of C4's interleaved 8-row blocks is within 5 % of the others -- three ranks of eight are enough there)
        ranks = range(world) if tag == "C4" or world <= 2 else (0, world // 2, world - 1)
        per_rank = [timed_render(s, W, H, opt, world, r, steps, warm, clock_warmup) for r in ranks]
        slow = max(per_rank)
        if world == 1:
            base = slow
            rows.append(dict(workload=tag, n_gpus=1, render_ms=round(slow, 3), total_ms=round(slow, 3), speedup=1.0))
            print(json.dumps(rows[-1]), flush=True)
            continue
        out = dict(workload=tag, n_gpus=world, render_ms_slowest_rank=round(slow, 3), render_ms_by_rank=[round(x, 3) for x in per_rank], ranks_traced=list(ranks),
                   speedup_render_only=round(base / slow, 2))
        for mode in ("gather", "reduce"):
            sw = exchange_software_ms(W, H, world, mode)
            frame_bytes = W * H * 12
            wire = (frame_bytes / world if mode == "gather" else 2.0 * (world - 1) / world * frame_bytes) / (LINK_GBPS * 1e9) * 1e3
            out[mode] = dict(software_ms_one_rank=None if sw is None else round(sw, 3), wire_ms_modelled=round(wire, 3),
                             total_ms=None if sw is None else round(slow + sw + wire, 3), speedup=None if sw is None else round(base / (slow + sw + wire), 2))
        rows.append(out)
        print(json.dumps(out), flush=True)
    print("\n%s: %s %dx%d, %d steps after %d\n| GPUs | slowest rank's %d steps (ms) | render only | + gather (software, 1 rank measured + wire modelled) | + reduce |" % (tag, scene, W, H, steps, warm, steps))
    print("|---|---|---|---|---|")
    for r in rows:
        if r["n_gpus"] == 1:
            print("| 1 | %.3f | 1.00x | | |" % r["render_ms"])
        else:
            g, d = r["gather"], r["reduce"]
            f = lambda x: "n/a" if x["total_ms"] is None else "%.3f ms = %.2fx (%.3f + %.3f)" % (x["total_ms"], x["speedup"], x["software_ms_one_rank"], x["wire_ms_modelled"])
            print("| %d | %.3f | %.2fx | %s | %s |" % (r["n_gpus"], r["render_ms_slowest_rank"], r["speedup_render_only"], f(g), f(d)))
    print(flush=True)
if have_dist:
    dist.destroy_process_group()
