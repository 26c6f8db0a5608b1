#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mrays/s and ms/iteration of the path-tracing bounce loop.

Workload (BASELINE.json metric "Mrays/s ... at 1920x1080 depth-8", configs[3], SURVEY 8(d) "C4"): scenes/cornellObj.txt
(6 boxes + a 12-triangle mesh) at 1920x1080, trace depth 8, antialiasing on, material sort on, 1 sample per pixel per
step.  A step = one pathtrace(iter) over the whole frame.  Rays = paths entering the intersect stage, summed over
bounces (the reference's own unit; about 5.37 M per step).  Inputs are resident in HBM before the timed region.

    python bench.py                        # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W      # N ranks: interleaved row tiles + one RCCL reduce per run

Prints ONE JSON line on rank 0.  Extra objects: "roofline" (dominant kernel, algorithmic bytes / measured launch time)
and "cpu_baseline" (the oracle, single thread, on this host), see DESIGN.md "Measurement".  With N > 1 the same line also carries,
all OUTSIDE the timed region: "long_run" (200 steps of the same tile split: the regime where eight GPUs pay), "exchange_alt_ms"
(the gather and the reduce spelling of the one exchange, each timed alone on the same buffers) and "c5" (BASELINE configs[4]: the
3840x2160 textured-mesh scene traced as N row-tile ranks with its exchange).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = "cornellObj.txt 1920x1080 depth 8, AA on, material sort on, 1 spp/step (BASELINE configs[3] / C4)"
SCENE, RES, DEPTH = "cornellObj.txt", (1920, 1080), 8
HBM_PEAK = 8.0e12                 # MI355X HBM3E spec peak, B/s (MI355X_MICROARCH.md)
VALU_PEAK = 256 * 4 * 16 * 2.4e9  # lane-slots/s of non-packed, non-FMA vector issue: CUs x SIMDs x 16 lanes x 2.4 GHz (same guide)
# The CONTRACT's algorithmic bytes per ray-bounce, from the reference's 44-B PathSegment / 32-B ShadeableIntersection records
# (SURVEY 8(d)): intersect 76 + shade 120 for what k_bounce does, + material sort 152 + compaction 88 = 436 for the loop.  This design
# moves 60-B SoA records once per bounce and no dead paths, so those figures are reported under explicit `contract_*` names only.
CONTRACT_BYTES_BOUNCE_KERNEL = 76 + 120
CONTRACT_BYTES_LOOP = 436


LIT_SHARE_OF_ENDED = 0.05      # share of the paths that END at a bounce >= 1 whose end carries radiance (a light hit): 0.047 on C4, 0.059 on C5
                                # (counted with the CPU oracle at 480x270, three iterations: DESIGN.md 6); the others end black and move nothing


def own_layout_bytes(rpb, with_direction=1.0, with_normal=1.0, index_bytes=4):
    """Algorithmic bytes per ray of k_bounce (bounces >= 1) by THIS design's data layout, from the rays per bounce and the
    run's own record mix (ptx_stats: the share of stored paths whose record carries a direction / a normal rather than a 3-bit code):
    a record is two 16-byte quads (shading point + pixel slot, throughput colour + material|geom), a third with the normal (+ texcoord u)
    and a fourth with the incoming direction (+ texcoord v) where the next bounce can need them.  Round 5: a ray entering bounce b reads
    its local-index entry (ONE word where chunks are <= 128 tiles: every configuration the bench runs; 8 B otherwise) and its record; if it
    is stored for bounce b + 1 it writes its record, its 4-B key and, in the kernel's tail, its 4-B local-index entry (re-reading the key:
    4 B); a path that ends writes 12 B of radiance + 1 flag byte if it ends on a light, and NOTHING if it ends black (rounds 3-4: 12 B of
    zeros and a 4-B "no record" key for every ended path, 8-B index entries)."""
    n = [float(x) for x in rpb] + [0.0]
    tot = sum(n[1:-1])
    if tot <= 0:
        return 0.0
    rec = 32.0 + 16.0 * with_normal + 16.0 * with_direction
    b = 0.0
    for k in range(1, len(n) - 1):
        stored = n[k + 1]
        b += n[k] * (index_bytes + rec) + stored * (rec + 4 + index_bytes + 4) + (n[k] - stored) * LIT_SHARE_OF_ENDED * 13.0
    return b / tot


def usable_cores():
    """Cores this process may really use: the affinity mask, cut down to the cgroup CPU quota when there is one (a GPU box
    shows all 256 host cores but grants a share of them), and to 16 -- the share of a one-GPU box -- when the quota cannot
    be read."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    return max(1, min(n, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    return max(1, min(n, int(quota / period + 0.5)))
        except (OSError, ValueError, IndexError):
            continue
    return min(n, 16)


def cpu_baseline(scene, iters):
    """Single-thread CPU oracle on a bounded sample of the same workload (checker code, timed here as a baseline)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpulibs import OracleLib
    d = scene.dump()
    O = OracleLib()
    O.set_libm(0)
    O.create(d, d["textures"])
    O.pt_init()
    t0 = time.time()
    rays = 0
    for it in range(1, iters + 1):
        O.iterate(it)
        rays += int(O.live_counts().sum())
    dt = time.time() - t0
    sec = O.stage_seconds()
    # the same port with its two per-path loops (intersect, shade) on every core the process may use;
    # sort, partition and gather stay serial.  Same results (tests/test_oracle_golden.py), clearly not the 1-thread figure.
    ncores = usable_cores()
    all_cores = None
    if ncores > 1:
        O.set_threads(ncores)
        t1 = time.time()
        mrays = 0
        for it in range(iters + 1, 3 * iters + 1):
            O.iterate(it)
            mrays += int(O.live_counts().sum())
        mdt = time.time() - t1
        O.set_threads(1)
        all_cores = dict(value=mrays / mdt / 1e6, unit="Mrays/s", cores=ncores, sample="%d iterations, %.1f s, OpenMP over the paths of the intersect and shade stages" % (2 * iters, mdt))
    # StreamCompaction::CPU (stream_compaction/cpu.cu:20-95) on the per-bounce "still alive" flag arrays
    # of one iteration, timed alone, next to the library's GPU compaction of the same arrays (host pointers in and out)
    import numpy as np
    import mygpuraytracer_amd as pt
    counts = [int(c) for c in O.live_counts()]
    SC = pt.StreamCompaction()
    rng = np.random.default_rng(1)
    import torch
    ms = dict(cpu_without_scan=0.0, cpu_with_scan=0.0, gpu_efficient_compact=0.0, gpu_compact_device=0.0)
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    for b, n in enumerate(counts):
        alive = counts[b + 1] if b + 1 < len(counts) else 0
        flags = np.zeros(n, np.int32)
        flags[rng.permutation(n)[:alive]] = 1
        a = SC.cpu_compact_without_scan(flags); ms["cpu_without_scan"] += SC.last_cpu_ms()
        c = SC.cpu_compact_with_scan(flags); ms["cpu_with_scan"] += SC.last_cpu_ms()
        # the reference-shaped entry point: host pointers, allocates and copies inside (efficient.cu:79-136)
        g = SC.efficient_compact(flags); ms["gpu_efficient_compact"] += SC.last_gpu_ms()
        assert len(a) == len(c) == len(g) == alive
        # the same array already in HBM, buffers reused (sc_compact_device), mean of 20 calls
        d_in, d_out = torch.from_numpy(flags).to(dev), torch.empty(n, dtype=torch.int32, device=dev)
        ws = torch.zeros((SC.workspace_bytes(n) + 7) // 8, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        SC.compact_device(n, d_out.data_ptr(), d_in.data_ptr(), cnt.data_ptr(), ws.data_ptr(), stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            SC.compact_device(n, d_out.data_ptr(), d_in.data_ptr(), cnt.data_ptr(), ws.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        ms["gpu_compact_device"] += e0.elapsed_time(e1) / 20
        assert int(cnt.item()) == alive
    # the reference's own code (oracle/_ref: its headers and loader compiled for the host, where it has been built) on
    # three iterations of the same frame, next to the port
    reference = None
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libptref.so")
    if os.path.exists(ref_so):
        try:
            from cpulibs import RefLib, scene_text_with
            R = RefLib(ref_so)
            text = scene_text_with(open(os.path.join(ROOT, "scenes", SCENE)).read(), RES, DEPTH)
            R.load_text(text, cwd=os.path.join(ROOT, "scenes"))
            R.apply_runcuda_camera()
            R.pt_init()
            t1 = time.time()
            rrays = 0
            for it in range(1, 4):
                R.iterate(it)
                rrays += int(R.live_counts().sum())
            rdt = time.time() - t1
            reference = dict(value=rrays / rdt / 1e6, unit="Mrays/s", cores=1,
                             sample="3 iterations, %.1f s, the reference's intersections.h / interactions.h built host-only (oracle/_ref)" % rdt)
        except Exception as e:                      # the baseline is optional, the bench line is not
            reference = dict(error=str(e)[:200])
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=rays / dt / 1e6, unit="Mrays/s", cores=1, cpu_model=cpu_model, host_cores_usable=ncores, kind="port", reference=reference, all_cores=all_cores,
                stream_compaction_ms_per_iteration=dict(elements=sum(counts), **ms),
                sample="%d iteration(s) of the same 1920x1080 depth-8 frame, %.1f s, single thread (oracle/pt_oracle.c, gcc -O2)" % (iters, dt),
                stage_seconds=dict(intersect=sec[0], sort=sec[1], shade=sec[2], compact=sec[3], generate=sec[4], gather=sec[5]))


def dropin_per_call(torch, pt, scene, dev_index, calls=96):
    """The reference's own call shape, driver-visible: per call what the C++ veneer's pathtrace(pbo, frame, iter) does
    (csrc/pathtrace_api.cpp; src/pathtrace.cu:434-436, :552-556) -- camera refresh, ONE iteration, the 8-bit preview into a device
    pbo, the fp32 frame copied back into a page-locked host buffer -- so that scene->state.image is valid after every call.
    Render-ahead on, as in the veneer.  ms per call over `calls` calls; the PCIe copy alone is the floor."""
    import ctypes as C
    import numpy as np
    W, H = RES
    vp = C.c_void_p
    with pt.Tracer(scene, device=dev_index) as T:
        lib, h = T.lib, T.h
        T.set_render_ahead(True)
        img = np.zeros((W * H, 3), np.float32)
        pinned = lib.ptx_pin_host_buffer(img.ctypes.data_as(vp), img.nbytes) == 0
        pbo = torch.zeros(W * H * 4, dtype=torch.uint8, device=torch.device("cuda", dev_index))
        torch.cuda.synchronize()

        def call(it):
            T.set_camera(scene)
            T.pathtrace(it)
            lib.ptx_write_pbo_device(h, it, vp(pbo.data_ptr()))
            lib.ptx_read_image(h, img.ctypes.data_as(vp))
        for it in range(1, 37):
            call(it)
        t0 = time.perf_counter()
        for it in range(37, 37 + calls):
            call(it)
        per_call = (time.perf_counter() - t0) / calls * 1e3
        t0 = time.perf_counter()
        for _ in range(16):
            lib.ptx_read_image(h, img.ctypes.data_as(vp))
        copy_ms = (time.perf_counter() - t0) / 16 * 1e3
        if pinned:
            lib.ptx_unpin_host_buffer(img.ctypes.data_as(vp))
    return per_call, copy_ms


def c5_per_iteration(pt, dev_index, iters=72):
    """BASELINE configs[4] on ONE GPU: the cornellSpaceship layout at 3840x2160, depth 8, antialiasing + depth of field, textured
    BVH mesh (the 20448-triangle procedural stand-in: the reference's .obj is missing), split mesh search.  ms per iteration over
    `iters` iterations = six launch sets of 12 on the three streams (rounds 1-3 timed 24 = five sets of 5; with 12 iterations per set,
    round 4's default at 4K, 24 would be two sets and no steady state: old / new library on one box, 24 iterations 1.215 / 1.20, 72
    iterations 1.20 / 1.13 -- tools/gpu_c5_leg.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import ensure_standin_assets
    ensure_standin_assets()
    s = pt.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship20k.txt"), res=(3840, 2160), depth=8)
    s.apply_runcuda_camera()
    with pt.Tracer(s, depth_of_field=1, device=dev_index) as T:
        T.render(1, 36)
        T.synchronize()
        st0 = T.stats()
        r0 = st0["rays_total"]
        t0 = time.perf_counter()
        T.render(100, iters)
        T.synchronize()
        dt = time.perf_counter() - t0
        st1 = T.stats()
        rays = st1["rays_total"] - r0
        sp = max(st1.get("stored_paths", 0) - st0.get("stored_paths", 0), 1)
        mix = dict(with_direction=(st1.get("stored_with_direction", 0) - st0.get("stored_with_direction", 0)) / sp,
                   with_normal=1.0 - (st1.get("stored_with_normal_code", 0) - st0.get("stored_with_normal_code", 0)) / sp)
        mix["mean"] = 32.0 + 16.0 * mix["with_normal"] + 16.0 * mix["with_direction"]
        # bounces >= 1 by the layout's own bytes (own_layout_bytes; the texcoords ride in the normal's and the direction's quads)
        mix["algorithmic_bytes_per_ray_later_bounces"] = own_layout_bytes(st1["rays_per_bounce"], mix["with_direction"], mix["with_normal"])
    return dt / iters * 1e3, rays / iters, mix


def arith_valu_counts():
    """SQ_INSTS_VALU per k_bounce launch at each arithmetic level, from the committed counter passes (profiles/sq_latest.json = level 0,
    profiles/sq_arith1.json / sq_arith2.json = `bench.py --arith N` under the same passes); None where a profile is missing."""
    out = {}
    for lv, f in ((0, "sq_latest.json"), (1, "sq_arith1.json"), (2, "sq_arith2.json")):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f)))
            out[str(lv)] = dict(SQ_INSTS_VALU=(d.get("k_bounce") or {}).get("SQ_INSTS_VALU"), rays_per_launch=(d.get("_units_per_launch") or {}).get("k_bounce"), source="profiles/" + f)
        except Exception:
            out[str(lv)] = None
    return out


def arith_levels(pt, scene, dev_index, steps, warmup, lanes):
    """The timed workload once more at the opt-in arithmetic levels (ptx_options.arith 1 = CONTRACTED: the same kernels as a second code
    object with fused multiply-adds, the arithmetic the reference's real nvcc build has; 2 = FAST: + hardware reciprocal / square root /
    sine).  Same steps, same bracketing, outside the timed region; `value` itself is always the exact level.  Results at these levels
    agree with the exact one within the stated statistical tolerance (tests/test_gpu_arith.py), not bit for bit."""
    out = {}
    for lv, name in ((1, "contracted"), (2, "fast")):
        with pt.Tracer(scene, device=dev_index, lanes=lanes, arith=lv) as T:
            T.render(1, warmup)
            T.synchronize()
            t_warm, extra = time.perf_counter(), 0
            while time.perf_counter() - t_warm < 0.15:
                T.render(10_000_000 + extra, 36)
                T.synchronize()
                extra += 36
            r0 = T.stats()["rays_total"]
            t0 = time.perf_counter()
            T.render(warmup + 1, steps)
            T.synchronize()
            dt = time.perf_counter() - t0
            rays = T.stats()["rays_total"] - r0
            r1 = T.stats()["rays_total"]
            t0 = time.perf_counter()
            T.render(20_000_000, 200)
            T.synchronize()
            lt = time.perf_counter() - t0
            out[name] = dict(arith=lv, value=rays / dt / 1e6, unit="Mrays/s", steps=steps, ms_per_step=dt / steps * 1e3,
                             long_run=dict(steps=200, ms_per_step=lt / 200 * 1e3, value=(T.stats()["rays_total"] - r1) / lt / 1e6), fenced=T.stats()["fenced"])
    return out


def stream_compaction_device(torch, pt, device):
    """the scan / compaction library on arrays already in HBM (sc_scan_device, sc_compact_device): achieved
    rate on algorithmic bytes (scan 4 B read + 4 B written per element; compaction 4 B read + 4 B per survivor) next to a
    device-to-device copy of the same array.  n = 2^28 ints (1 GiB, past the 256 MB Infinity Cache) and the C4 frame."""
    sc = pt.StreamCompaction()
    stream = torch.cuda.current_stream().cuda_stream
    rows = []
    for n in (1 << 28, RES[0] * RES[1]):
        g = torch.Generator(device=device)
        g.manual_seed(n)
        a = (torch.rand(n, device=device, generator=g) < 0.46).to(torch.int32) * 3     # 46 % survive, as in C4's first bounce
        o = torch.empty_like(a)
        ws = torch.zeros((sc.workspace_bytes(n) + 7) // 8, dtype=torch.int64, device=device)
        count = torch.zeros(1, dtype=torch.int32, device=device)
        reps = 5 if n > (1 << 24) else 50

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e-3
        t_scan = timed(lambda: sc.scan_device(n, o.data_ptr(), a.data_ptr(), ws.data_ptr(), stream))
        t_comp = timed(lambda: sc.compact_device(n, o.data_ptr(), a.data_ptr(), count.data_ptr(), ws.data_ptr(), stream))
        kept = int(count.item())
        t_copy = timed(lambda: o.copy_(a))
        rows.append(dict(n=n, scan_GBps=8 * n / t_scan / 1e9, scan_frac_of_hbm_peak=8 * n / t_scan / HBM_PEAK,
                         compact_GBps=(4 * n + 4 * kept) / t_comp / 1e9, survivors=kept, copy_GBps=8 * n / t_copy / 1e9,
                         scan_us=t_scan * 1e6, compact_us=t_comp * 1e6, copy_us=t_copy * 1e6))
        del a, o, ws
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cpu-iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip dropin_per_call_ms and c5_ms_per_iteration (profiling runs)")
    ap.add_argument("--arith", type=int, default=0, choices=[0, 1, 2],
                    help="arithmetic level of the timed tracer (ptx_options.arith): 0 = exact, the default and the only level `value` is ever quoted "
                         "at by the driver's command; 1 / 2 are for profiling the contracted / fast code objects (the line then says so in config.arith)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                    "multi-rank path on a box with fewer GPUs than ranks: the frame is then reduced through host memory)")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "iterations"],
                    help="N > 1: 'tiles' = interleaved pixel-row blocks per rank (what north_star prescribes, the default); "
                    "'iterations' = every rank traces the full frame for every N-th iteration (sums to the single-GPU frame)")
    ap.add_argument("--exchange", default="gather", choices=["reduce", "gather"],
                    help="N > 1, tiles: how rank 0 gets the frame, one RCCL collective per run either way: 'gather' (default) = every rank "
                    "sends only the row blocks it owns (multigpu.assemble_tiles: 1/N of the bytes per rank, point-to-point over the "
                    "xGMI links into rank 0, no adds; SURVEY 8(e)); 'reduce' = reduce(SUM) of the full accumulation buffers, in which "
                    "foreign rows are zero -- the same frame bit for bit, N times the bytes")
    ap.add_argument("--lanes", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6, 7, 8],
                    help="launch sets in flight (ptx_options.lanes): 0 = library default (3: kernels of different batches of iterations "
                    "fill each other's tails); 1 = one at a time, kernels back to back (what the roofline leg always uses, "
                    "because a kernel's duration is only meaningful when it has the GPU to itself)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher (no RANK in the environment): start the N ranks ourselves, as a CHILD
    # process (never an exec) and before anything here has touched the GPU, and hand on rank 0's JSON line and the exit code.
    if args.gpus > 1 and "RANK" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus and os.environ.get("PTX_BENCH_DIST") != "1":
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s: launch with --nproc-per-node %d (or run `python bench.py --gpus %d`, which "
                 "starts the ranks itself)" % (args.gpus, os.environ.get("WORLD_SIZE", "1 (unset)"), args.gpus, args.gpus))

    import torch
    import torch.distributed as dist
    import mygpuraytracer_amd as pt
    from mygpuraytracer_amd import multigpu

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
    # collectives also with one rank when launched under torch.distributed.run with PTX_BENCH_DIST=1: a rehearsal of the
    # RCCL calls (init, reduce, barrier, all_reduce) on a one-GPU box
    dist_on = world > 1 or (os.environ.get("PTX_BENCH_DIST") == "1" and "RANK" in os.environ)
    if dist_on:
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    n_gpus = world
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)

    def reduce_frame(img, res=RES, how=None):
        """One reduce(SUM) of the accumulation buffer to rank 0: RCCL on the device buffer, or (gloo rehearsal) via host.
        --exchange gather: the owned rows only (pixel tiles; iteration sharding needs the sum)."""
        how = how or args.exchange
        if how == "gather" and not (world > 1 and args.shard == "iterations"):
            multigpu.assemble_tiles(img, res[0], res[1], multigpu.TILE_ROWS, dst=0, via_host=args.backend != "nccl")
        elif args.backend == "nccl":
            dist.reduce(img[:res[0] * res[1] * 3], dst=0, op=dist.ReduceOp.SUM)
        else:
            h = img[:res[0] * res[1] * 3].cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                img[:res[0] * res[1] * 3].copy_(h)

    def all_reduce_scalar(value, dtype, op):
        tt = torch.tensor([value], dtype=dtype, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=op)
        return tt.item()

    scene = pt.Scene(os.path.join(ROOT, "scenes", SCENE), res=RES, depth=DEPTH)
    scene.apply_runcuda_camera()
    W, H = RES
    image = multigpu.frame_buffer(W, H, world, device)      # W*H*3 floats (+ padding rows when tiled, for the strided-view gather)
    torch.cuda.current_stream(device).synchronize()      # the tracer uses a stream of its own: the fill must have landed first
    kw = dict(device=dev_index, lanes=args.lanes, arith=args.arith)
    by_iter = world > 1 and args.shard == "iterations"
    if world > 1 and not by_iter:
        kw.update(tile_rows=multigpu.TILE_ROWS, tile_rank=rank, tile_world=world)
    T = pt.Tracer(scene, external_image_ptr=image.data_ptr(), **kw)

    def render_steps(first, count):
        """this rank's part of iterations first .. first+count-1"""
        if by_iter:
            f, n = multigpu.iteration_share(first, count, rank, world)
            if n:
                T.render(f, n, stride=world)
        else:
            T.render(first, count)

    def barrier():
        # barrier, then synchronize.  (The RCCL barrier is ordered behind what this rank has enqueued on its current stream -- the
        # exchange -- so no rank passes it before every rank's part is done; a host synchronize in FRONT of it would only add a
        # round trip to the timed region.)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # W untimed warm-up steps -- and, when those are over in less than 150 ms, further untimed steps until 150 ms have passed:
    # the GPU needs about 100 ms of work to reach its clocks (50 timed steps after 10 warm-up steps: 0.32 ms per step; after
    # the ramp: 0.285).  Untimed either way; how many were added is reported as config.clock_warmup_steps.
    t_warm = time.perf_counter()
    T.render(1, args.warmup)
    T.synchronize()
    clock_warmup_steps = 0
    while time.perf_counter() - t_warm < 0.15:
        T.render(10_000_000 + clock_warmup_steps, 36)
        T.synchronize()
        clock_warmup_steps += 36
    if dist_on:                                     # warm the collective too
        reduce_frame(image.clone())
    rays0 = T.stats()["rays_total"]
    barrier()
    t0 = time.perf_counter()
    render_steps(args.warmup + 1, args.steps)       # EXACTLY K steps, enqueued back to back on the tracer's stream
    T.synchronize()
    t_render = time.perf_counter() - t0
    if dist_on:                                     # one RCCL collective on the accumulation buffer per run
        reduce_frame(image)
    barrier()
    dt = time.perf_counter() - t0
    st = T.stats()
    rays = st["rays_total"] - rays0
    loop_ms = T.last_loop_ms()
    t_render_max = t_render
    if dist_on:
        t_render_max = float(all_reduce_scalar(t_render, torch.float64, dist.ReduceOp.MAX))
        dt = float(all_reduce_scalar(dt, torch.float64, dist.ReduceOp.MAX))
        rays = int(all_reduce_scalar(rays, torch.int64, dist.ReduceOp.SUM))

    # ---- N > 1 only, OUTSIDE the timed region, same JSON line: what a single short scaling run cannot say by itself -------------
    multi = None
    if dist_on and not by_iter:
        # (multigpu.all_ok / agreed_phase / RankFailed: every rank's verdict on a phase it ran LOCALLY, agreed on by all -- a failure of
        # one rank is then every rank's failure at the same point; tests/test_multigpu_gloo.py runs the protocol with 2 gloo ranks)
        flag_dev = device if args.backend == "nccl" else None
        all_ok = lambda ok: multigpu.all_ok(ok, flag_dev)
        RankFailed = multigpu.RankFailed

        def timed_tile_run(tracer, img, res, first, count, how=None):
            """`count` steps of this rank's tile + the exchange, bracketed like the timed region; max over ranks.  The rank-local part
            (the render) may fail on one rank: that rank still walks through every collective below, then all ranks raise together."""
            err = None
            r0 = n = 0
            barrier()
            a = time.perf_counter()
            try:
                r0 = tracer.stats()["rays_total"]
                tracer.render(first, count)
                tracer.synchronize()
            except Exception as e:
                err = e
            b = time.perf_counter() - a
            reduce_frame(img, res, how)
            barrier()
            c = time.perf_counter() - a
            try:
                n = tracer.stats()["rays_total"] - r0
            except Exception as e:
                err = err or e
            out = (float(all_reduce_scalar(c, torch.float64, dist.ReduceOp.MAX)), float(all_reduce_scalar(b, torch.float64, dist.ReduceOp.MAX)),
                   int(all_reduce_scalar(n, torch.int64, dist.ReduceOp.SUM)))
            if not all_ok(err is None):
                raise RankFailed("rank %d: %s" % (rank, err) if err else "another rank failed in the render of this leg")
            return out

        def exchange_alone(img, res, how, reps=5):
            """the exchange by itself on the buffers of the run (median of `reps`; max over ranks): barrier, exchange, barrier"""
            ts = []
            for _ in range(reps + 1):
                barrier()
                a = time.perf_counter()
                reduce_frame(img, res, how)
                barrier()
                ts.append(time.perf_counter() - a)
            return float(all_reduce_scalar(sorted(ts[1:])[len(ts[1:]) // 2], torch.float64, dist.ReduceOp.MAX)) * 1e3

        multi = {}
        # (a) the long run: 200 steps of the same split.  The driver's 20 steps are a 3.4-ms problem on one GPU -- every rank's eight
        # dependent bounce launches per launch set are then latency, not throughput; this is the regime the tile split
        # is for.  Speed-up against one GPU = the N = 1 line's long_run (printed there too) / this.
        LONG = 200
        lt, lr, lrays = timed_tile_run(T, image, RES, 20_000_000, LONG)
        multi["long_run"] = dict(steps=LONG, ms_per_step=lt / LONG * 1e3, value=lrays / lt / 1e6, unit="Mrays/s", slowest_rank_render_ms=lr * 1e3,
                                 exchange_and_barrier_ms=(lt - lr) * 1e3, exchange=args.exchange)
        # (b) both spellings of the one exchange on the same buffers (frames are complete: the long run just ended), each alone
        scratch = image.clone()
        multi["exchange_alt_ms"] = dict(gather=exchange_alone(scratch, RES, "gather"), reduce=exchange_alone(scratch, RES, "reduce"),
                                        timed_with=args.exchange, frame_MB=RES[0] * RES[1] * 12 / 1e6, backend=args.backend,
                                        what="barrier + exchange + barrier, median of 5, max over ranks; gather = every rank sends the 1/N of the "
                                             "frame it owns to rank 0, reduce = RCCL reduce(SUM) of the zero-padded full frames (north_star's spelling): same frame bit for bit")
        del scratch
        # (c) BASELINE configs[4] as it is named: the 3840x2160 textured-mesh scene with depth of field on N ranks, its exchange included
        # Every phase that can fail on ONE rank (the assets on rank 0, a tracer's hipMalloc at 4K, the warm-up render) runs rank-locally
        # under its own try, and the ranks agree on its outcome (all_ok) BEFORE the next collective: either all ranks go on or all skip
        # the leg and the line carries c5.error -- no rank is left waiting in a barrier / reduce for one that has moved on to the
        # roofline leg (round 4's version ran the collectives inside a rank-local try; its first 8-GPU run could have hung there).
        C5RES, C5STEPS = (3840, 2160), 72
        T5 = img5 = warm5 = None
        c5_err = None

        def c5_phase(fn):
            """run fn() locally, agree on the outcome; returns True if every rank got through"""
            nonlocal c5_err
            ok, msg = multigpu.agreed_phase(fn, flag_dev)
            if not ok and c5_err is None:
                c5_err = msg
            return ok

        def c5_assets():
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from conftest import ensure_standin_assets
            if rank == 0:
                ensure_standin_assets()

        def c5_create():
            nonlocal T5, img5, warm5
            if os.environ.get("PTX_BENCH_FAIL_C5_RANK") == str(rank):      # rehearsal of the failure protocol (tools/runs/r5reh.sh): this rank's phase fails
                raise RuntimeError("PTX_BENCH_FAIL_C5_RANK: rehearsed failure of the tracer's creation on rank %d" % rank)
            s5 = pt.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship20k.txt"), res=C5RES, depth=8)
            s5.apply_runcuda_camera()
            img5 = multigpu.frame_buffer(C5RES[0], C5RES[1], world, device)
            warm5 = img5.clone()
            torch.cuda.current_stream(device).synchronize()
            kw5 = dict(device=dev_index, lanes=args.lanes, depth_of_field=1)
            if world > 1:
                kw5.update(tile_rows=multigpu.TILE_ROWS, tile_rank=rank, tile_world=world)
            T5 = pt.Tracer(s5, external_image_ptr=img5.data_ptr(), **kw5)

        def c5_warm():
            T5.render(1, 36)
            T5.synchronize()

        try:
            # (all_ok is itself a barrier: after c5_assets every rank sees rank 0's files)
            if c5_phase(c5_assets) and c5_phase(c5_create) and c5_phase(c5_warm):
                reduce_frame(warm5, C5RES)          # warm the exchange at this size (every rank is here: agreed above)
                ct, cr, crays = timed_tile_run(T5, img5, C5RES, 100, C5STEPS)      # raises on ALL ranks if one failed
                multi["c5"] = dict(ms_per_iteration=ct / C5STEPS * 1e3, steps=C5STEPS, rays_per_iteration=crays / C5STEPS, Mrays_per_s=crays / ct / 1e6,
                                   slowest_rank_render_ms=cr * 1e3, exchange_and_barrier_ms=(ct - cr) * 1e3, exchange=args.exchange,
                                   frame_MB=C5RES[0] * C5RES[1] * 12 / 1e6, fenced=T5.stats()["fenced"],
                                   workload="cornellSpaceship20k.txt 3840x2160 depth 8, AA + DoF, textured 20448-triangle BVH mesh, %d row-tile ranks + 1 %s/run "
                                            "(BASELINE configs[4] / C5; speed-up = the N = 1 line's c5_ms_per_iteration / this)" % (world, args.exchange))
            else:
                multi["c5"] = dict(error=c5_err)
        except RankFailed as e:                     # (raised by every rank at the same point: no collective is pending)
            multi["c5"] = dict(error=str(e)[:300])
        finally:
            if T5 is not None:
                try:
                    T5.close()
                except Exception:
                    pass
            del img5, warm5

    # roofline leg: the same K steps again with hipEvents around every launch (on the tracer's stream)
    T.set_kernel_timing(True)
    render_steps(args.warmup + args.steps + 1, args.steps)
    kt = T.kernel_times()
    T.set_kernel_timing(False)
    st2 = T.stats()
    rays_leg = st2["rays_total"] - st["rays_total"]
    rpb = st2["rays_per_bounce"]
    names = {k: v for k, v in kt.items() if v[1] > 0}
    dominant = max(names, key=lambda k: names[k][0]) if names else "k_bounce"
    dom_ms, dom_n = kt[dominant]
    if dominant == "k_bounce":
        units = rays_leg * (sum(rpb[1:]) / max(sum(rpb), 1)) / max(dom_n, 1)       # rays per launch, bounces >= 1
    else:
        units = rays_leg * (rpb[0] / max(sum(rpb), 1)) / max(dom_n, 1)
    avg_s = dom_ms / max(dom_n, 1) * 1e-3
    sp = max(st2.get("stored_paths", 0) - st.get("stored_paths", 0), 0)
    f_dir = (st2.get("stored_with_direction", 0) - st.get("stored_with_direction", 0)) / sp if sp else 1.0
    f_nrm = 1.0 - (st2.get("stored_with_normal_code", 0) - st.get("stored_with_normal_code", 0)) / sp if sp else 1.0
    rec_bytes = 32.0 + 16.0 * f_nrm + 16.0 * f_dir
    own_bytes = own_layout_bytes(rpb, f_dir, f_nrm) if dominant == "k_bounce" else rec_bytes + 4 + 4 + 4      # (first bounce: nothing read, mostly stored)
    achieved = own_bytes * units / avg_s if avg_s > 0 else 0.0
    contract = CONTRACT_BYTES_BOUNCE_KERNEL * units / avg_s if avg_s > 0 else 0.0
    # HBM bytes and instruction counts per launch come from PMC counters, which need rocprofv3: they are NOT measured in this run but
    # taken from the committed profile of the same kernels (tools/profile_round.sh, tools/pmc_sq.sh -> profiles/), scaled from that
    # profile's rays per launch to this run's, and labelled as such.
    traffic = traffic_source = tjd = None
    tj = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tj):
        try:
            tjd = json.load(open(tj))
            per_launch = tjd.get(dominant)
            ref_units = (tjd.get("_units_per_launch") or {}).get(dominant)
            if per_launch is not None:
                traffic = per_launch * (units / ref_units) if ref_units else per_launch
                traffic_source = ("profiles/traffic_latest.json = %s, not this run; %s" % (
                    tjd.get("_how", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python3 bench.py --lanes 1 --steps 24 --warmup 12 --no-cpu-baseline`"),
                    "scaled by rays per launch %.3g / %.3g" % (units, ref_units) if ref_units else "per launch of that profile, unscaled"))
        except Exception:
            traffic = None
    # The WALL-level HBM figure (round 5): `frac` / `physical_frac` describe k_bounce alone, one launch set at a time, while `value` is
    # wall time with three launch sets overlapped.  Counter bytes of ALL kernels of a launch set (the committed FETCH_SIZE / WRITE_SIZE
    # profile: depth - 1 launches of k_bounce, the camera bounce, the gather; per launch of `_iterations_per_launch` iterations) per
    # iteration, over this run's wall time per step, over the 8 TB/s peak.
    physical_wall = None
    try:
        if tjd and all(k in tjd for k in ("k_bounce", "k_bounce<first>", "k_gather")) and world == 1:
            ipl = float(tjd.get("_iterations_per_launch", 12))
            set_bytes = (DEPTH - 1) * tjd["k_bounce"] + tjd["k_bounce<first>"] + tjd["k_gather"]
            physical_wall = dict(bytes_per_step=set_bytes / ipl, TBps=set_bytes / ipl / (dt / args.steps) / 1e12,
                                 frac=set_bytes / ipl / (dt / args.steps) / HBM_PEAK,
                                 source="profiles/traffic_latest.json: (%d x k_bounce + k_bounce<first> + k_gather) counter bytes per launch of %d iterations, "
                                        "/ this run's ms_per_step / 8 TB/s" % (DEPTH - 1, int(ipl)))
    except Exception:
        physical_wall = None
    valu_issue = None
    sj = os.path.join(ROOT, "profiles", "sq_latest.json")
    if os.path.exists(sj):
        try:
            sq = json.load(open(sj))
            k = sq.get(dominant) or {}
            ref_units = (sq.get("_units_per_launch") or {}).get(dominant)
            if k.get("SQ_INSTS_VALU") and avg_s > 0:
                insts = k["SQ_INSTS_VALU"] * (units / ref_units if ref_units else 1.0)
                lane_slots = insts * 64 / avg_s
                valu_issue = dict(achieved=lane_slots / 1e12, peak=VALU_PEAK / 1e12, unit="T lane-slots/s", frac=lane_slots / VALU_PEAK,
                                  wave_instructions_per_launch=insts, lanes_active=(k.get("_derived") or {}).get("valu_lane_utilisation"),
                                  # `peak` is the textbook one instruction per 4 cycles and SIMD.  Measured on this chip (tools/micro/valu_rates.hip,
                                  # wall clock, 8 waves per SIMD; profiles/round4_valu_rates.txt): plain fp32 / integer-add / logic instructions
                                  # retire one per 2.5 cycles, min / max / select / compare / shift / binary64 / packed ones one per 4.1, a stream
                                  # mixed like k_bounce's one per 3.06 -- the capacity the second fraction is taken of
                                  cycles_per_instruction_per_simd=avg_s * 2.4e9 * 1024 / insts,
                                  measured_capacity_cycles_per_instruction=dict(plain=2.54, second_class=4.09, k_bounce_like_mix=3.06,
                                                                                source="profiles/round4_valu_rates.txt"),
                                  frac_of_measured_mix_capacity=3.06 / (avg_s * 2.4e9 * 1024 / insts),
                                  source="profiles/sq_latest.json (SQ_INSTS_VALU per launch from %s, not this run%s) x 64 lanes / this run's "
                                         "launch time / (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz)" % (
                                             sq.get("_how", "tools/pmc_sq.sh"), "; scaled by rays per launch" if ref_units else ""))
        except Exception:
            valu_issue = None
    # were the committed counter profiles taken with the kernel sources this run was built from?  (they are scaled into `traffic`,
    # `valu_issue` and `bound` below: a stale profile would silently describe another kernel)
    profiles_stale = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from profile_meta import source_sha16
        now = source_sha16()
        shas = {}
        for nm in ("traffic_latest.json", "sq_latest.json"):
            f = os.path.join(ROOT, "profiles", nm)
            if os.path.exists(f):
                shas[nm] = json.load(open(f)).get("_source_sha16")
        profiles_stale = dict(kernel_sources_now=now, **{k: (v if v else "unrecorded (profile older than round 4)") for k, v in shas.items()},
                              stale=any(v != now for v in shas.values()))
        if rank == 0 and profiles_stale["stale"]:
            print("bench.py: profiles/*_latest.json were collected with other kernel sources than this build's: traffic / valu_issue / bound "
                  "describe THAT profile's kernels", file=sys.stderr)
    except Exception as e:
        profiles_stale = dict(error=str(e)[:120])
    loop_contract = CONTRACT_BYTES_LOOP * rays / (loop_ms * 1e-3) if loop_ms > 0 else 0.0
    # One read tells what bounds the kernel: `bound` names it (vector-instruction issue -- valu_issue.frac of the issue peak), achieved /
    # frac / traffic are the HBM side of the same launches by this design's own bytes; the contract's record sizes are under contract_*.
    roofline = dict(schema="r3+: achieved / frac / algorithmic_bytes_per_unit are THIS design's own layout bytes (r3: ~120 B per ray; end of r4 ~85: records "
                           "of 32-64 B by what the next bounce can need, record_bytes; r5 ~71: one-word index entries, no zeros for paths that end black); rounds 1-2 put the "
                           "contract's 196 B there, which now lives under contract_196B_frac -- BENCH_r02's frac compares with contract_196B_frac, not with frac",
                    measured_in_this_run=["achieved", "frac", "avg_launch_us", "launches", "units_per_launch", "kernels_ms_per_step", "contract_*", "loop_ms_per_step"],
                    from_committed_profiles=["traffic", "traffic_over_algorithmic", "physical_frac", "physical_frac_wall (bytes; the time is this run's)", "valu_issue", "bound"], profiles=profiles_stale,
                    bound="valu" if valu_issue and valu_issue["frac"] > achieved / HBM_PEAK else "hbm", kernel=dominant,
                    achieved=achieved / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=achieved / HBM_PEAK,
                    algorithmic_bytes_per_unit=own_bytes, record_bytes=dict(mean=rec_bytes, with_direction=f_dir, with_normal=f_nrm), traffic=traffic, traffic_source=traffic_source,
                    traffic_over_algorithmic=(traffic / (own_bytes * units)) if traffic and own_bytes else None,
                    physical_frac=(traffic / avg_s / HBM_PEAK) if traffic and avg_s > 0 else None,
                    physical_frac_wall=physical_wall["frac"] if physical_wall else None, physical_wall=physical_wall,
                    valu_issue=valu_issue,
                    contract_196B_frac=contract / HBM_PEAK,            # 196 B/ray of the reference's AoS records / launch time / 8 TB/s
                    contract_436B_loop_ratio=loop_contract / (HBM_PEAK * world),     # a RATIO (may exceed 1): the reference layout's loop bytes
                                                                                     # per ray / wall time of the bounce loop / 8 TB/s
                    avg_launch_us=avg_s * 1e6, launches=dom_n, units_per_launch=units, loop_ms_per_step=loop_ms / args.steps,
                    kernels_ms_per_step={k: v[0] / args.steps for k, v in kt.items()},
                    timing="kernel durations: hipEvents around every launch, one launch set at a time (lanes 1), NOT additive to ms_per_step: "
                           "value, ms_per_step and loop_ms_per_step are wall time with %d launch set(s) in flight" % (args.lanes if args.lanes >= 1 else 3))

    out = dict(metric="Mrays/s", value=rays / dt / 1e6, unit="Mrays/s", n_gpus=n_gpus, steps=args.steps, warmup=args.warmup,
               ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f32",
               data="synthetic",
               primary_rays_per_s=RES[0] * RES[1] * args.steps / dt,      # px * spp / s of the whole job (beside Mrays/s)
               config=dict(workload=WORKLOAD, arith=args.arith, rays_per_step=rays / args.steps, rays_per_bounce=rpb, clock_warmup_steps=clock_warmup_steps,
                           rccl_ranks=(dist.get_world_size() if dist_on else 1), backend=(args.backend if dist_on else None),
                           exchange=(None if not dist_on else "reduce" if by_iter else args.exchange),
                           timed_region_ms=dt * 1e3, slowest_rank_render_ms=t_render_max * 1e3,
                           exchange_and_barrier_ms=(dt - t_render_max) * 1e3 if dist_on else 0.0,
                           parallelism=("1 GPU" if world == 1 else
                                        "%d ranks taking turns over the iterations of the full frame + 1 RCCL reduce/run" % world if by_iter else
                                        "%d row-tile ranks (%d-row interleaved blocks) + 1 RCCL %s/run" % (world, multigpu.TILE_ROWS, args.exchange))),
               roofline=roofline)
    if multi:
        out.update(multi)
    elif rank == 0 and not args.no_extra_legs:
        # one GPU: the same 200-step run the N-rank lines carry as "long_run", so that their speed-up has its denominator on record
        r0 = T.stats()["rays_total"]
        torch.cuda.synchronize()
        a = time.perf_counter()
        T.render(20_000_000, 200)
        T.synchronize()
        lt = time.perf_counter() - a
        out["long_run"] = dict(steps=200, ms_per_step=lt / 200 * 1e3, value=(T.stats()["rays_total"] - r0) / lt / 1e6, unit="Mrays/s")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene, args.cpu_iters)
    T.close()
    if rank == 0 and world == 1 and not args.no_extra_legs:
        # two more driver-visible figures, outside the timed region: the reference's one-iteration-per-call shape with the frame
        # read back every call (PCIe-inclusive: never `value`), and BASELINE configs[4] (4K, textured BVH mesh, DoF) on one GPU
        try:
            per_call, copy_ms = dropin_per_call(torch, pt, scene, dev_index)
            out["dropin_per_call_ms"] = per_call
            out["dropin_per_call"] = dict(ms=per_call, frame_copy_alone_ms=copy_ms, Mrays_per_s=rays / args.steps / per_call / 1e3,
                                          what="C ABI sequence of the veneer's pathtrace(pbo, frame, iter): set_camera + 1 iteration + preview "
                                               "into a device pbo + fp32 frame into pinned host memory, per call, render-ahead on")
        except Exception as e:
            out["dropin_per_call_ms"] = None
            out["dropin_per_call"] = dict(error=str(e)[:200])
        try:
            al = arith_levels(pt, scene, dev_index, args.steps, args.warmup, args.lanes)
            out["value_contracted"] = al["contracted"]["value"]
            out["value_fast"] = al["fast"]["value"]
            out["arith_levels"] = dict(al, exact=dict(arith=0, value=out["value"], ms_per_step=out["ms_per_step"]),
                                       what="the timed workload at ptx_options.arith 1 (fused multiply-adds: the arithmetic of the reference's real build) and 2 "
                                            "(+ hardware rcp / rsq / sqrt / sin): same kernels as further code objects, results within the stated statistical "
                                            "tolerance of the exact level (tests/test_gpu_arith.py); `value` is ALWAYS the exact level",
                                       valu_instructions_per_k_bounce_launch=arith_valu_counts())
        except Exception as e:
            out["value_contracted"] = None
            out["arith_levels"] = dict(error=str(e)[:200])
        try:
            c5_ms, c5_rays, c5_mix = c5_per_iteration(pt, dev_index)
            out["c5_ms_per_iteration"] = c5_ms
            out["c5"] = dict(ms_per_iteration=c5_ms, rays_per_iteration=c5_rays, Mrays_per_s=c5_rays / c5_ms / 1e3, record_bytes=c5_mix,
                             workload="cornellSpaceship20k.txt 3840x2160 depth 8, AA + DoF, textured 20448-triangle BVH mesh, 1 GPU, 72 iterations after 36 (BASELINE configs[4] / C5)")
        except Exception as e:
            out["c5_ms_per_iteration"] = None
            out["c5"] = dict(error=str(e)[:200])
    if rank == 0 and world == 1:
        out["stream_compaction"] = stream_compaction_device(torch, pt, device)
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
