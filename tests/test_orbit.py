"""The interactive camera of src/main.cpp (mouse handlers :166-212, runCuda's recompute :105-123) as scripted events:
the library's ptx_orbit_* against the oracle's restatement, bit for bit, and against the documented behaviour
(theta clamp, zoom floor, recentre).  No reference test covers this; main.cpp needs a GL window, so the oracle's
restatement is the only checker: parity unpinned by the reference itself."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, beq

EVENTS = [("left", 37.0, -12.0), ("right", 55.0), ("middle", 14.0, -9.0), ("left", -420.5, 300.25), ("right", -9000.0),
          ("middle", -3.0, 2.0), ("space",), ("left", 1.0, -100000.0), ("right", 12.5)]


def oracle_orbit(O, cam19, res, events):
    L = O.lib
    vp, d, i = C.c_void_p, C.c_double, C.c_int
    L.o_orbit_init.argtypes = [vp, vp, vp]
    L.o_orbit_left_drag.argtypes = [vp, d, d, i, i]
    L.o_orbit_right_drag.argtypes = [vp, d, i]
    L.o_orbit_middle_drag.argtypes = [vp, d, d]
    L.o_orbit_apply.argtypes = [vp, vp]
    f = np.ascontiguousarray(cam19, np.float32).copy()
    o3, og = np.zeros(3, np.float32), np.zeros(3, np.float32)
    L.o_orbit_init(f.ctypes.data, o3.ctypes.data, og.ctypes.data)
    L.o_orbit_apply(f.ctypes.data, o3.ctypes.data)
    trail = [f.copy()]
    for ev in events:
        if ev[0] == "left":
            L.o_orbit_left_drag(o3.ctypes.data, ev[1], ev[2], res[0], res[1])
        elif ev[0] == "right":
            L.o_orbit_right_drag(o3.ctypes.data, ev[1], res[1])
        elif ev[0] == "middle":
            L.o_orbit_middle_drag(f.ctypes.data, ev[1], ev[2])
        else:
            f[3:6] = og
        L.o_orbit_apply(f.ctypes.data, o3.ctypes.data)
        trail.append(f.copy())
    return trail, o3


@pytest.mark.parametrize("scene", ["cornell.txt", "cornellObj.txt", "sphere.txt"])
def test_scripted_camera_matches_oracle(product, oracle_lib, scene):
    s = product.Scene(os.path.join(ROOT, "scenes", scene), res=(640, 360))
    cam0 = s.dump()["cam_floats"].copy()
    want, o3 = oracle_orbit(oracle_lib, cam0, (640, 360), EVENTS)
    orb = s.orbit_init()
    s.orbit_events(orb, [])
    s2 = product.Scene(os.path.join(ROOT, "scenes", scene), res=(640, 360))
    s2.apply_runcuda_camera()
    assert beq(s2.dump()["cam_floats"], want[0])                       # init + apply == the first runCuda()
    for k, ev in enumerate(EVENTS):
        s.orbit_events(orb, [ev])
        assert beq(s.dump()["cam_floats"][:15], want[k + 1][:15]), (k, ev)
    assert beq(np.float32([orb.phi, orb.theta, orb.zoom]), o3)
    # documented behaviour: theta stays in [0.001, PI], zoom never below 0.1, SPACE restores the loader's lookAt
    assert 0.001 <= orb.theta <= np.float32(np.pi) and orb.zoom >= np.float32(0.1)
    s.orbit_events(orb, [("space",)])
    assert beq(s.dump()["cam_floats"][3:6], cam0[3:6])
