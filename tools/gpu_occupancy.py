"""What the runtime says about k_bounce's residency: workgroups per CU by dynamic LDS size."""
import ctypes as C, sys
sys.path.insert(0, ".")
import mygpuraytracer_amd as pt
s = pt.Scene("scenes/cornellObj.txt", res=(1920, 1080), depth=8); s.apply_runcuda_camera()
with pt.Tracer(s) as T:
    L = T.lib
    L.ptx_debug_bounce_occupancy.restype = C.c_int; L.ptx_debug_bounce_occupancy.argtypes = [C.c_void_p, C.c_int]
    print("as launched:", L.ptx_debug_bounce_occupancy(T.h, 0))
    for b in (16384, 18432, 19456, 19968, 20304, 20480, 20992, 21712, 22528, 23248, 24576):
        print(b, L.ptx_debug_bounce_occupancy(T.h, b))
