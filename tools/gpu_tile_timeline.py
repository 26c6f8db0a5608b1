"""The kernels of ONE short run (20 steps in one call) of rank 0 of TILE_WORLD (default 8), for rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tile_tl -- python3 tools/gpu_tile_timeline.py
    python tools/gpu_tile_timeline.py --read gpurun_out/tile_tl      (per queue: start, duration, gap to the previous kernel of that queue)"""
import os, sys, time
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    import csv, glob, collections
    f = max(glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True))
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))), key=lambda x: x[0])
    bursts, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - max(x[1] for x in cur[-8:]) > 5_000_000: bursts.append(cur); cur = [r]
        else: cur.append(r)
    bursts.append(cur)
    b = bursts[-1]
    t0 = b[0][0]
    print("last burst: %d kernels, span %.1f us" % (len(b), (max(x[1] for x in b) - t0) / 1e3))
    lastend = {}
    for st, en, name, q in b:
        short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:26]
        gap = (st - lastend[q]) / 1e3 if q in lastend else float("nan")
        print("  queue %-3s %-26s start %7.1f  dur %6.1f  gap %5.1f" % (q, short, (st - t0) / 1e3, (en - st) / 1e3, gap))
        lastend[q] = en
    sys.exit(0)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "."))
import mygpuraytracer_amd as pt
world = int(os.environ.get("TILE_WORLD", "8"))
s = pt.Scene("scenes/cornellObj.txt", res=(1920, 1080), depth=8); s.apply_runcuda_camera()
kw = dict(tile_rows=8, tile_rank=0, tile_world=world) if world > 1 else {}
with pt.Tracer(s, **kw) as T:
    t0 = time.perf_counter(); T.render(1, 5); T.synchronize()
    while time.perf_counter() - t0 < 0.15: T.render(10000, 36); T.synchronize()
    for rep in range(4):
        time.sleep(0.02)
        t0 = time.perf_counter(); T.render(1000, 20); T.synchronize(); print("20 steps: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
