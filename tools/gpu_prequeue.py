"""What would a replayed launch graph buy a SHORT run?  20 steps in one call (the driver's scaling command) of rank 0 of 8, of 4 and of the
full frame, timed by the tracer's own events (loop start -> loop end on the device): as the host issues it, and with the whole run
queued BEFORE its first kernel starts (PTX_DEBUG_PREQUEUE_US: a one-wave kernel holds the main stream meanwhile) -- the device's view of
a graph replay with zero issue latency; by launch sets per call.  Median of 9, ms.
usage: python tools/gpu_prequeue.py"""
import os, sys, time, json
sys.path.insert(0, ".")
import mygpuraytracer_amd as pt

s = pt.Scene("scenes/cornellObj.txt", res=(1920, 1080), depth=8); s.apply_runcuda_camera()
for world in (8, 4, 1):
    for lanes, nsets in ((3, 0), (3, 3), (4, 4), (6, 6)):
        kw = dict(tile_rows=8, tile_rank=0, tile_world=world) if world > 1 else {}
        kw["lanes"] = lanes
        if nsets: os.environ["PTX_DEBUG_NSETS"] = str(nsets)
        else: os.environ.pop("PTX_DEBUG_NSETS", None)
        row = {"world": world, "lanes": lanes, "nsets": nsets or "rule"}
        with pt.Tracer(s, **kw) as T:
            t0 = time.perf_counter(); T.render(1, 5); T.synchronize()
            while time.perf_counter() - t0 < 0.15: T.render(10000, 36); T.synchronize()
            for label, hold in (("as_issued", None), ("prequeued", "1500")):
                if hold: os.environ["PTX_DEBUG_PREQUEUE_US"] = hold
                else: os.environ.pop("PTX_DEBUG_PREQUEUE_US", None)
                dev, host = [], []
                for rep in range(9):
                    t0 = time.perf_counter(); T.render(1000, 20); T.synchronize(); host.append(time.perf_counter() - t0)
                    dev.append(T.last_loop_ms())
                row[label + "_device_ms"] = round(sorted(dev)[4], 3)
                row[label + "_host_ms"] = round(sorted(host)[4] * 1e3, 3)
            os.environ.pop("PTX_DEBUG_PREQUEUE_US", None)
        print(json.dumps(row), flush=True)
