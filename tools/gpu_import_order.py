"""Import-order check: the tracer first, torch afterwards, both must see the GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mygpuraytracer_amd as pt
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellObj.txt"), res=(320, 180), depth=6); s.apply_runcuda_camera()
with pt.Tracer(s) as T:
    T.render(1, 20); a = T.read_image().sum()
import torch
x = torch.zeros(320 * 180 * 3, device="cuda:0")
with pt.Tracer(s, external_image_ptr=x.data_ptr()) as T:
    T.render(1, 20); T.synchronize()
print("both fine:", float(a), float(x.sum().item()))
