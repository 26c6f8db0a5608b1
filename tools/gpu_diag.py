#!/usr/bin/env python3
"""GPU diagnostic: per-field comparison of the per-stage entry points against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mygpuraytracer_amd as pt
from cpulibs import OracleLib

def ulp(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)

def cmp_struct(name, g, o, mask=None):
    for f in g.dtype.names:
        a, b = g[f], o[f]
        if mask is not None: a, b = a[mask], b[mask]
        if a.dtype.kind == 'f':
            d = ulp(a, b); nb = int((d != 0).sum())
            print("   %-10s %-16s differing %d max ulp %d" % (name, f, nb, int(d.max()) if d.size else 0))
            if nb:
                k = np.argwhere(d != 0)[0]
                print("      first at", k, repr(a[tuple(k)]), repr(b[tuple(k)]))
        else:
            nb = int((a != b).sum())
            print("   %-10s %-16s differing %d" % (name, f, nb))

name, res, depth = sys.argv[1], (int(sys.argv[2]), int(sys.argv[3])), int(sys.argv[4])
s = pt.Scene(os.path.join(ROOT, "scenes", name), res=res, depth=depth); s.apply_runcuda_camera()
d = s.dump(); O = OracleLib(); O.set_libm(1); O.create(d, d["textures"]); O.pt_init()
T = pt.Tracer(s)
O.pt_generate(1); op = O.paths(); gp = T.generate(1); gp2 = T.generate(1)
print("generate deterministic:", np.array_equal(gp.view(np.uint8), gp2.view(np.uint8)))
cmp_struct("generate", gp, op)
oi = O.compute_intersections(op); gi = T.compute_intersections(op)
cmp_struct("intersect", gi, oi)
hit = oi["t"] > 0
cmp_struct("isect-hit", gi, oi, hit)
idx = np.arange(len(op), dtype=np.int32)
osd = O.shade(1, 1, idx, oi, op); gsd = T.shade(1, idx, oi, op)
cmp_struct("shade", gsd, osd)
# libm
rng = np.random.default_rng(1)
x = (rng.random(100000) * 6.2831855).astype(np.float32)
pw = rng.random(100000)
pxy = np.stack([rng.random(100000).astype(np.float32), (rng.random(100000) * 50).astype(np.float32)], 1)
s_, c_, p5, po = T.libm(x, pw, pxy)
os_ = np.zeros_like(s_); oc_ = np.zeros_like(c_)
for k in range(len(x)):
    a, b = O.own_sincosf(x[k]); os_[k] = a; oc_[k] = b
print("libm sin diff", int((ulp(s_, os_) != 0).sum()), "cos diff", int((ulp(c_, oc_) != 0).sum()))
op5 = np.array([O.lib.o_own_pow5(float(v)) for v in pw])
print("pow5 diff", int((p5 != op5).sum()))
opo = np.array([O.lib.o_own_powf(float(a), float(b)) for a, b in pxy], np.float32)
print("powf diff", int((ulp(po, opo) != 0).sum()))
