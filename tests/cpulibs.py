"""ctypes bindings for the two CPU checkers (test infrastructure only).

* ``OracleLib``  -- oracle/libptoracle.so, this repo's plain-C restatement (always buildable, gcc).
* ``RefLib``     -- oracle/_ref/libptref.so, the reference's own headers compiled from /root/reference
                    (only buildable in the dev container; absent => tests that need it are skipped).

Both expose the same per-function and per-stage entry points with identical record layouts, so a test can run
the same inputs through either and compare bit for bit.
"""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REFERENCE_ROOT = "/root/reference"

PATH_DTYPE = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("color", "<f4", 3),
                       ("pixelIndex", "<i4"), ("remainingBounces", "<i4")])
ISECT_DTYPE = np.dtype([("t", "<f4"), ("normal", "<f4", 3), ("materialId", "<i4"),
                        ("texcoord", "<f4", 2), ("geomId", "<i4")])
assert PATH_DTYPE.itemsize == 44 and ISECT_DTYPE.itemsize == 32

vp = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(vp)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    return os.path.join(ORACLE_DIR, "libptoracle.so")


def build_ref():
    """Builds oracle/_ref/libptref.so when /root/reference is present; returns its path or None."""
    so = os.path.join(ORACLE_DIR, "_ref", "libptref.so")
    if os.path.isdir(REFERENCE_ROOT):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "_ref"])
    return so if os.path.exists(so) else None


class _Tracer:
    """Shared wrapper over the o_* / ref_* entry points; ``self.h`` is the scene handle."""
    prefix = ""

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def _proto(self):
        L, p = self.lib, self.prefix
        i, f = C.c_int, C.c_float
        protos = {
            "utilhash": (C.c_uint, [C.c_uint]),
            "rng_raw": (None, [i, i, i, i, vp]),
            "rng_uniform": (None, [i, i, i, f, f, i, vp]),
            "geom_test": (None, [vp, i, i, vp, vp]),
            "obj_tri_test": (None, [vp, i, i, vp, vp]),
            "jittered_test": (None, [i, vp, vp, i, vp]),
            "compute_intersections": (None, [vp, i, vp, vp]),
            "shade": (None, [vp, i, i, i, vp, vp, vp]),
            "pt_init": (None, [vp]),
            "pt_generate": (None, [vp, i]),
            "pt_bounce": (i, [vp, i, i]),
            "pt_final_gather": (None, [vp]),
            "pt_iterate": (i, [vp, i]),
            "pt_live_counts": (i, [vp, vp, i]),
            "pt_paths": (vp, [vp]),
            "pt_isects": (vp, [vp]),
            "pt_image": (vp, [vp]),
            "pt_num_paths": (i, [vp]),
            "pt_pixelcount": (i, [vp]),
            "pt_pbo": (None, [vp, i, vp]),
        }
        for name, (res, args) in protos.items():
            fn = getattr(L, p + name)
            fn.restype, fn.argtypes = res, args

    # --- hash / rng -------------------------------------------------------------------------------------
    def utilhash(self, a):
        return self._f("utilhash")(C.c_uint(int(a) & 0xFFFFFFFF))

    def rng_raw(self, it, index, depth, n):
        out = np.zeros(n, np.uint32)
        self._f("rng_raw")(it, index, depth, n, _ptr(out))
        return out

    def rng_uniform(self, it, index, depth, a, b, n):
        out = np.zeros(n, np.float32)
        self._f("rng_uniform")(it, index, depth, a, b, n, _ptr(out))
        return out

    # --- per function -----------------------------------------------------------------------------------
    def geom_test(self, gi, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((len(rays), 10), np.float32)
        self._f("geom_test")(self.h, gi, len(rays), _ptr(rays), _ptr(out))
        return out

    def obj_tri_test(self, gi, rays):
        """objTriIntersectionTest (src/intersections.h:284-315, dead code of the reference) on an OBJ geom: (n, 8) = t, point, normal, outside"""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((len(rays), 8), np.float32)
        self._f("obj_tri_test")(self.h, gi, len(rays), _ptr(rays), _ptr(out))
        return out

    def jittered_test(self, normals, seeds, max_iter=5000):
        """calculateJitteredDirectionHemisphere (src/interactions.h:46-85, dead code of the reference): (n, 3) normals, (n, 3) int
        (iter, index, depth) -> (n, 3) directions"""
        normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        seeds = np.ascontiguousarray(seeds, np.int32).reshape(-1, 3)
        out = np.zeros((len(normals), 3), np.float32)
        self._f("jittered_test")(len(normals), _ptr(normals), _ptr(seeds), int(max_iter), _ptr(out))
        return out

    def compute_intersections(self, paths):
        paths = np.ascontiguousarray(paths, PATH_DTYPE)
        out = np.zeros(len(paths), ISECT_DTYPE)
        self._f("compute_intersections")(self.h, len(paths), _ptr(paths), _ptr(out))
        return out

    def shade(self, it, depth, idx, isects, paths):
        paths = np.array(paths, PATH_DTYPE, copy=True)
        isects = np.ascontiguousarray(isects, ISECT_DTYPE)
        idx = np.ascontiguousarray(idx, np.int32)
        self._f("shade")(self.h, it, depth, len(paths), _ptr(idx), _ptr(isects), _ptr(paths))
        return paths

    # --- iteration --------------------------------------------------------------------------------------
    def pt_init(self):
        self._f("pt_init")(self.h)

    def pt_generate(self, it):
        self._f("pt_generate")(self.h, it)

    def pt_bounce(self, it, stage_mask=15):
        return self._f("pt_bounce")(self.h, it, stage_mask)

    def pt_final_gather(self):
        self._f("pt_final_gather")(self.h)

    def iterate(self, it):
        return self._f("pt_iterate")(self.h, it)

    def live_counts(self):
        buf = np.zeros(256, np.int32)
        n = self._f("pt_live_counts")(self.h, _ptr(buf), 256)
        return buf[:n].copy()

    def pixelcount(self):
        return self._f("pt_pixelcount")(self.h)

    def num_paths(self):
        return self._f("pt_num_paths")(self.h)

    def framepixels(self):
        return self.pixelcount()

    def paths(self):
        n = self.pixelcount()
        addr = self._f("pt_paths")(self.h)
        return np.ctypeslib.as_array((C.c_char * (44 * n)).from_address(addr)).view(PATH_DTYPE).copy()

    def isects(self):
        n = self.pixelcount()
        addr = self._f("pt_isects")(self.h)
        return np.ctypeslib.as_array((C.c_char * (32 * n)).from_address(addr)).view(ISECT_DTYPE).copy()

    def set_apps_variant(self, on):
        fn = self._f("set_apps_variant") if self.prefix == "ref_" else self.lib.o_scene_set_apps_variant
        fn.argtypes = [vp, C.c_int]
        fn(self.h, int(on))

    def albedo(self):
        n = self.framepixels()
        fn = self._f("pt_albedo")
        fn.restype, fn.argtypes = vp, [vp]
        addr = fn(self.h)
        return np.ctypeslib.as_array((C.c_float * (3 * n)).from_address(addr)).reshape(n, 3).copy()

    def image(self):
        n = self.framepixels()
        addr = self._f("pt_image")(self.h)
        return np.ctypeslib.as_array((C.c_float * (3 * n)).from_address(addr)).reshape(n, 3).copy()

    def pbo(self, it):
        out = np.zeros((self.framepixels(), 4), np.uint8)
        self._f("pt_pbo")(self.h, it, _ptr(out))
        return out


def scene_text_with(text, res=None, depth=None):
    """Returns scene text with RES / DEPTH replaced (the loaders themselves are never modified)."""
    if res is not None:
        text = re.sub(r"RES\s+\d+\s+\d+", "RES         %d %d" % tuple(res), text)
    if depth is not None:
        text = re.sub(r"DEPTH\s+\d+", "DEPTH       %d" % depth, text)
    return text


class RefLib(_Tracer):
    prefix = "ref_"

    def __init__(self, so):
        self.lib = C.CDLL(so)
        self._proto()
        L = self.lib
        L.ref_scene_load.restype, L.ref_scene_load.argtypes = vp, [C.c_char_p, C.c_char_p]
        for n in ("ref_num_geoms", "ref_num_materials"):
            getattr(L, n).restype, getattr(L, n).argtypes = C.c_int, [vp]
        L.ref_num_faces.restype, L.ref_num_faces.argtypes = C.c_int, [vp, C.c_int]
        L.ref_texture_vector_sizes.argtypes = [vp, vp]
        L.ref_get_geom.argtypes = [vp, C.c_int, vp, vp]
        L.ref_get_material.argtypes = [vp, C.c_int, vp]
        L.ref_get_faces.argtypes = [vp, C.c_int, vp]
        L.ref_get_camera.argtypes = [vp, vp, vp]
        L.ref_get_texture.argtypes = [vp, C.c_int, C.c_int, vp, vp]
        L.ref_set_depth.argtypes = [vp, C.c_int]
        L.ref_apply_runcuda_camera.argtypes = [vp]
        L.ref_set_options.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
        self.h = None

    def load(self, scene_path, cwd=os.path.join(REFERENCE_ROOT, "scenes")):
        self.h = self.lib.ref_scene_load(cwd.encode(), os.path.abspath(scene_path).encode())
        assert self.h, "reference loader failed"
        return self

    def load_text(self, text, cwd=os.path.join(REFERENCE_ROOT, "scenes")):
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            f.write(text)
        try:
            return self.load(f.name, cwd)
        finally:
            os.unlink(f.name)

    def apply_runcuda_camera(self):
        self.lib.ref_apply_runcuda_camera(self.h)

    def set_depth(self, d):
        self.lib.ref_set_depth(self.h, d)

    def set_options(self, aa=1, dof=0, sort=1, cache=1):
        self.lib.ref_set_options(self.h, aa, dof, sort, cache)

    def dump(self):
        """Scene as POD arrays (the layout OracleLib.create and the product's C-ABI take)."""
        L, h = self.lib, self.h
        ng, nm = L.ref_num_geoms(h), L.ref_num_materials(h)
        gints = np.zeros((ng, 3), np.int32)
        gfl = np.zeros((ng, 57), np.float32)
        for i in range(ng):
            L.ref_get_geom(h, i, _ptr(gints[i]), _ptr(gfl[i]))
        mats = np.zeros((nm, 11), np.float32)
        for i in range(nm):
            L.ref_get_material(h, i, _ptr(mats[i]))
        faces = []
        for i in range(ng):
            nf = L.ref_num_faces(h, i)
            f = np.zeros((nf, 15), np.float32)
            if nf:
                L.ref_get_faces(h, i, _ptr(f))
            faces.append(f)
        ci = np.zeros(4, np.int32)
        cf = np.zeros(19, np.float32)
        L.ref_get_camera(h, _ptr(ci), _ptr(cf))
        tv = np.zeros(4, np.int32)
        L.ref_texture_vector_sizes(h, _ptr(tv))
        textures = {}
        for i in range(ng):
            for which in range(4):
                whc = np.zeros(3, np.int32)
                L.ref_get_texture(h, i, which, _ptr(whc), None)
                if whc[2] > 0:
                    img = np.zeros((int(whc[1]), int(whc[0]), int(whc[2])), np.uint8)
                    L.ref_get_texture(h, i, which, _ptr(whc), _ptr(img))
                    textures[(i, which)] = img
        return dict(geom_ints=gints, geom_trs=gfl[:, :9].copy(), geom_mats=gfl[:, 9:].copy(), materials=mats,
                    faces=faces, cam_ints=ci, cam_floats=cf, texture_vector_sizes=tv, textures=textures)


class OracleLib(_Tracer):
    prefix = "o_"

    def __init__(self, so=None):
        self.lib = C.CDLL(so or build_oracle())
        self._proto()
        L = self.lib
        i, f = C.c_int, C.c_float
        L.o_set_libm.argtypes = [i]
        L.o_set_threads.argtypes = [i]
        L.o_get_libm.restype = i
        L.o_own_sincosf.argtypes = [f, vp, vp]
        L.o_own_pow5.restype, L.o_own_pow5.argtypes = C.c_double, [C.c_double]
        L.o_own_powf.restype, L.o_own_powf.argtypes = f, [f, f]
        L.o_build_transforms.argtypes = [vp, vp]
        L.o_camera_from_loader.argtypes = [i, i, f, vp, vp, vp, vp]
        L.o_runcuda_camera.argtypes = [vp]
        L.o_scene_create.restype, L.o_scene_create.argtypes = vp, [i, vp, vp, i, vp]
        L.o_scene_free.argtypes = [vp]
        L.o_scene_set_faces.argtypes = [vp, i, i, vp]
        L.o_scene_set_texture.argtypes = [vp, i, i, i, i, i, vp]
        L.o_scene_set_camera.argtypes = [vp, vp, vp, i]
        L.o_scene_set_options.argtypes = [vp, i, i, i, i]
        L.o_scene_set_tile.argtypes = [vp, i, i, i]
        L.o_pt_framepixels.restype, L.o_pt_framepixels.argtypes = i, [vp]
        L.o_pt_stage_seconds.argtypes = [vp, vp]
        for n in ("o_sc_scan",):
            getattr(L, n).argtypes = [i, vp, vp]
        for n in ("o_sc_compact_without_scan", "o_sc_compact_with_scan"):
            getattr(L, n).restype, getattr(L, n).argtypes = i, [i, vp, vp]
        self.h = None

    def set_libm(self, mode):
        self.lib.o_set_libm(mode)

    def set_threads(self, n):
        self.lib.o_set_threads(int(n))

    def create(self, dump, textures=None):
        """dump: dict as produced by RefLib.dump() or the product loader; textures: {(geom, which): HxWxC uint8}."""
        if self.h:
            self.lib.o_scene_free(self.h)
        gints = np.ascontiguousarray(dump["geom_ints"], np.int32)
        gm = np.ascontiguousarray(dump["geom_mats"], np.float32)
        mats = np.ascontiguousarray(dump["materials"], np.float32)
        self.h = self.lib.o_scene_create(len(gints), _ptr(gints), _ptr(gm), len(mats), _ptr(mats))
        for gi, f in enumerate(dump["faces"]):
            if len(f):
                f = np.ascontiguousarray(f, np.float32)
                self.lib.o_scene_set_faces(self.h, gi, len(f), _ptr(f))
        if textures is None:
            textures = dump.get("textures")
        for (gi, which), img in (textures or {}).items():
            img = np.ascontiguousarray(img, np.uint8)
            hh, ww, ch = img.shape
            self.lib.o_scene_set_texture(self.h, gi, which, ww, hh, ch, _ptr(img))
        self.set_camera(dump["cam_ints"], dump["cam_floats"])
        return self

    def set_camera(self, cam_ints, cam_floats):
        ci = np.ascontiguousarray(cam_ints, np.int32)
        cf = np.ascontiguousarray(cam_floats, np.float32)
        self.cam_ints, self.cam_floats = ci.copy(), cf.copy()
        self.lib.o_scene_set_camera(self.h, _ptr(ci[:2].copy()), _ptr(cf), int(ci[3]))

    def apply_runcuda_camera(self):
        cf = self.cam_floats.copy()
        self.lib.o_runcuda_camera(_ptr(cf))
        self.set_camera(self.cam_ints, cf)

    def set_depth(self, d):
        ci = self.cam_ints.copy()
        ci[3] = d
        self.set_camera(ci, self.cam_floats)

    def set_options(self, aa=1, dof=0, sort=1, cache=1):
        self.lib.o_scene_set_options(self.h, aa, dof, sort, cache)

    def set_tile(self, rows, rank, world):
        self.lib.o_scene_set_tile(self.h, rows, rank, world)

    def framepixels(self):
        return self.lib.o_pt_framepixels(self.h)

    def stage_seconds(self):
        out = np.zeros(6, np.float64)
        self.lib.o_pt_stage_seconds(self.h, _ptr(out))
        return out

    # loader-side arithmetic
    def build_transforms(self, trs9):
        trs9 = np.ascontiguousarray(trs9, np.float32)
        out = np.zeros(48, np.float32)
        self.lib.o_build_transforms(_ptr(trs9), _ptr(out))
        return out

    def camera_from_loader(self, resx, resy, fovy, eye, lookat, up):
        out = np.zeros(19, np.float32)
        e, l, u = (np.ascontiguousarray(a, np.float32) for a in (eye, lookat, up))
        self.lib.o_camera_from_loader(resx, resy, fovy, _ptr(e), _ptr(l), _ptr(u), _ptr(out))
        return out

    def own_sincosf(self, x):
        s, c = C.c_float(), C.c_float()
        self.lib.o_own_sincosf(float(x), C.byref(s), C.byref(c))
        return np.float32(s.value), np.float32(c.value)

    # stream compaction
    def sc_scan(self, a):
        a = np.ascontiguousarray(a, np.int32)
        out = np.zeros_like(a)
        self.lib.o_sc_scan(len(a), _ptr(out), _ptr(a))
        return out

    def sc_compact(self, a, with_scan):
        a = np.ascontiguousarray(a, np.int32)
        out = np.zeros_like(a)
        fn = self.lib.o_sc_compact_with_scan if with_scan else self.lib.o_sc_compact_without_scan
        n = fn(len(a), _ptr(out), _ptr(a))
        return out[:n].copy(), n
