"""Is the C4 run short of HBM bandwidth?  The long run (three launch sets in flight) alone, and with a background stream of device-to-device
copies (torch, its own stream) running beside it: what the render loses against what the copies move meanwhile.  If the chip had bandwidth to
spare the copies would take theirs and the render would hardly notice.
usage: python tools/gpu_hbm_slack.py"""
import json, os, sys, time, threading
sys.path.insert(0, ".")
import torch
import mygpuraytracer_amd as pt

s = pt.Scene("scenes/cornellObj.txt", res=(1920, 1080), depth=8); s.apply_runcuda_camera()
STEPS = 240
with pt.Tracer(s) as T:
    T.render(1, 36); T.synchronize()
    def run():
        t0 = time.perf_counter(); T.render(1000, STEPS); T.synchronize(); return (time.perf_counter() - t0) * 1e3 / STEPS
    base = sorted(run() for _ in range(5))[2]
    print(json.dumps({"render_alone_ms_per_step": round(base, 4), "render_traffic_TBps_by_the_profile": round(0.654e-3 / (base * 1e-3) , 2)}), flush=True)
    side = torch.cuda.Stream()
    for mib, gap in ((256, 0), (64, 0), (16, 0), (256, 1)):
        n = mib << 18
        a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.ones(n, dtype=torch.float32, device="cuda")
        # copies alone
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(20): a.copy_(b)
            e1.record(side)
        side.synchronize()
        alone = 20 * 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12
        stop = False
        moved = [0]
        def pump():
            with torch.cuda.stream(side):
                while not stop:
                    for _ in range(8): a.copy_(b)
                    moved[0] += 8
                    side.synchronize()
                    if gap: time.sleep(gap * 1e-3)
        th = threading.Thread(target=pump); th.start()
        time.sleep(0.05)
        m0 = moved[0]; t0 = time.perf_counter()
        both = sorted(run() for _ in range(5))[2]
        dt = time.perf_counter() - t0; m1 = moved[0]
        stop = True; th.join()
        print(json.dumps({"copy_MiB": mib, "gap_ms": gap, "copies_alone_TBps": round(alone, 2), "render_ms_per_step_beside_copies": round(both, 4),
                          "render_slowdown": round(both / base, 3), "copies_beside_render_TBps": round((m1 - m0) * 2 * n * 4 / dt / 1e12, 2)}), flush=True)
        del a, b
