#!/usr/bin/env python3
"""CPU experiment (round 4) behind the ORDER in which k_mesh walks its parked rays.  The oracle traces the C5 scene with the
20 448-triangle mesh at a reduced frame; at each bounce the rays that reach the mesh's world box -- the rays pass 1 parks -- are taken
in stream order (the order k_mesh's queue has them in) and tools/mesh_walk_sim.cpp replays them in waves of 64 through the product's
own four-wide walk, for several orderings: as queued, and sorted by a key inside blocks of B queue entries.  No GPU.

    python tools/mesh_walk_order.py [W H]      (default 960 540)
"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mygpuraytracer_amd as pt
from cpulibs import OracleLib
from conftest import ensure_standin_assets
ensure_standin_assets()
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (960, 540)
SIM = "/tmp/mesh_walk_sim"
if not os.path.exists(SIM):
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-ffp-contract=off", "-o", SIM, os.path.join(ROOT, "tools", "mesh_walk_sim.cpp")])
s = pt.Scene(os.path.join(ROOT, "scenes", "cornellSpaceship20k.txt"), res=(W, H), depth=8); s.apply_runcuda_camera()
d = s.dump()
O = OracleLib(); O.set_libm(1); O.create(d, d["textures"]); O.set_options(aa=1, dof=1, sort=1, cache=1); O.pt_init(); O.set_threads(8)
gi = int(np.argmax(d["geom_ints"][:, 2]))                       # the mesh
faces = np.ascontiguousarray(d["faces"][gi], np.float32)
xf = d["geom_mats"][gi][:16].reshape(4, 4).T.astype(np.float64)       # glm column-major -> row-major
inv = d["geom_mats"][gi][16:32].reshape(4, 4).T.astype(np.float64)
verts = np.concatenate([faces[:, 0:3], faces[:, 5:8], faces[:, 10:13]]).astype(np.float64)
wv = verts @ xf[:3, :3].T + xf[:3, 3]
lo, hi = wv.min(0) - 1e-3, wv.max(0) + 1e-3
olo, ohi = verts.min(0), verts.max(0)
faces.tofile("/tmp/mw_faces.f32")


def slab(o, dd, lo, hi):
    with np.errstate(divide="ignore", invalid="ignore"):
        inv_d = 1.0 / np.where(np.abs(dd) < 1e-20, 1e-20, dd)
        t0, t1 = (lo - o) * inv_d, (hi - o) * inv_d
    tn, tf = np.minimum(t0, t1).max(1), np.maximum(t0, t1).min(1)
    return tn, tf


def sim(rays, order=None, refill=None):
    rays.astype(np.float32).tofile("/tmp/mw_rays.f32")
    args = [SIM, "/tmp/mw_faces.f32", "/tmp/mw_rays.f32"]
    if order is not None:
        np.ascontiguousarray(order, np.int32).tofile("/tmp/mw_order.i32"); args.append("/tmp/mw_order.i32")
    env = dict(os.environ)
    if refill:
        env.update(SIM_BATCH=str(refill[0]), SIM_THRESH=str(refill[1]), SIM_NMIN=str(refill[2]))
    lines = subprocess.check_output(args, text=True, env=env).strip().splitlines()
    return json.loads(lines[0])["refill"] if refill else json.loads(lines[-1])


def blocked_sort(key, B):
    n = len(key); order = np.arange(n)
    for a in range(0, n, B):
        order[a:a + B] = a + np.argsort(key[a:a + B], kind="stable")
    return order


it = 1
O.pt_generate(it)
for bounce in range(0, 5):
    n = O.num_paths()
    p = O.paths()[:n]
    o, dd = p["origin"].astype(np.float64), p["direction"].astype(np.float64)
    tn, tf = slab(o, dd, lo, hi)
    cand = (tf >= tn) & (tf >= 0)
    idx = np.nonzero(cand)[0]
    # object-space rays, as meshTestCore forms them
    qo = o[idx] @ inv[:3, :3].T + inv[:3, 3]
    qd = dd[idx] @ inv[:3, :3].T
    qd /= np.linalg.norm(qd, axis=1, keepdims=True)
    rays = np.concatenate([qo, qd], 1)
    # keys: direction octant; entry point on the object-space box in a G^3 grid; both
    otn, otf = slab(qo, qd, olo, ohi)
    entry = qo + np.maximum(otn, 0)[:, None] * qd
    octant = (qd[:, 0] < 0) * 1 + (qd[:, 1] < 0) * 2 + (qd[:, 2] < 0) * 4
    def cell(G):
        c = np.clip(((entry - olo) / (ohi - olo) * G).astype(int), 0, G - 1)
        return c[:, 0] + G * (c[:, 1] + G * c[:, 2])
    # Morton-ish key of the entry cell at G = 4 (2 bits per axis interleaved) so that nearby cells are nearby keys
    c4 = np.clip(((entry - olo) / (ohi - olo) * 4).astype(int), 0, 3)
    mort = np.zeros(len(idx), int)
    for b in range(2):
        for a in range(3):
            mort |= ((c4[:, a] >> b) & 1) << (3 * b + a)
    res = dict(bounce=bounce, paths=int(n), parked=int(len(idx)), share=round(len(idx) / max(n, 1), 4), as_queued=sim(rays))
    rays.astype(np.float32).tofile("/tmp/mw_rays_b%d.f32" % bounce)
    # a wave that refills its idle lanes (tools/mesh_walk_sim.cpp): rays per wave x idle lanes that trigger a refill x node-round minimum
    for batch in (256, 1024, 1 << 30):
        for thresh in (16, 32):
            for nmin in (16, 32):
                r = sim(rays, None, (batch, thresh, nmin))
                res["refill batch=%s thresh=%d nmin=%d" % ("queue" if batch > 1 << 20 else batch, thresh, nmin)] = dict(instr_per_64_rays=r["instr_per_64_rays"], lane_utilisation=r["lane_utilisation"])
    for name, key in (("octant", octant), ("oct+cell2", octant * 8 + cell(2)), ("oct+cell4", octant * 64 + mort)):
        for B in (1024, 4096):
            r = sim(rays, blocked_sort(key, B))
            res["%s B=%s" % (name, "all" if B > 1 << 20 else B)] = dict(instr_per_wave=r["instr_per_wave"], lane_utilisation=r["lane_utilisation"])
    print(json.dumps(res), flush=True)
    O.pt_bounce(it, 15)
