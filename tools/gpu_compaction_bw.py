#!/usr/bin/env python3
"""HBM rate of the one-pass scan and compaction (sc_scan_device / sc_compact_device) on device-resident int arrays, next
to a device-to-device copy of the same arrays on the same box.  Algorithmic bytes: scan 8 B/element, compaction
4 B/element + 4 B/survivor, copy 8 B/element.  Prints one JSON line per size."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mygpuraytracer_amd as pt
sc = pt.StreamCompaction()
dev = torch.device("cuda", 0)
SIZES = [int(a) for a in sys.argv[1:]] or [1920 * 1080, 3840 * 2160, 1 << 26, 1 << 28]
KEEP = 0.46                                                   # C4's first bounce keeps 46 % of the paths
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e-3
for n in SIZES:
    g = torch.Generator(device=dev); g.manual_seed(n)
    a = (torch.rand(n, device=dev, generator=g) < KEEP).to(torch.int32) * 3
    out = torch.empty_like(a)
    ws = torch.zeros((sc.workspace_bytes(n) + 7) // 8, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    reps = max(3, min(200, (1 << 30) // (8 * n)))
    t_scan = timed(lambda: sc.scan_device(n, out.data_ptr(), a.data_ptr(), ws.data_ptr(), stream), reps)
    t_comp = timed(lambda: sc.compact_device(n, out.data_ptr(), a.data_ptr(), count.data_ptr(), ws.data_ptr(), stream), reps)
    kept = int(count.item())
    t_copy = timed(lambda: out.copy_(a), reps)
    print(json.dumps(dict(n=n, scan_us=round(t_scan * 1e6, 2), scan_GBps=round(8 * n / t_scan / 1e9, 1),
                          compact_us=round(t_comp * 1e6, 2), compact_GBps=round((4 * n + 4 * kept) / t_comp / 1e9, 1), kept=kept,
                          copy_us=round(t_copy * 1e6, 2), copy_GBps=round(8 * n / t_copy / 1e9, 1),
                          scan_frac_of_8TBps=round(8 * n / t_scan / 8e12, 3), scan_vs_copy=round(t_copy / t_scan, 3))), flush=True)
