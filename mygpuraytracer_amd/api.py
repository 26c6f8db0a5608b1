"""ctypes host side over the C ABI of libmi355x_pathtracer.so (include/mi355x_pathtracer.h,
include/mi355x_stream_compaction.h).

Mirrors the reference's interface for the path (nkkk98/MyGPURaytracer):

* ``Scene(path)``                      <->  ``new Scene(sceneFile)``                (src/scene.cpp:10, src/main.cpp:47)
* ``Scene.apply_runcuda_camera()``     <->  first ``runCuda()`` camera recompute    (src/main.cpp:105-123)
* ``Tracer(scene, options)``           <->  ``pathtraceInit(scene)``                (src/pathtrace.cu:101)
* ``Tracer.pathtrace(iter)``           <->  ``pathtrace(pbo, frame, iter)``         (src/pathtrace.cu:433)
* ``Tracer.close()``                   <->  ``pathtraceFree()``                     (src/pathtrace.cu:159)
* ``Tracer.last_loop_ms()``            <->  ``timer().getGpuElapsedTimeForPreviousOperation()``
* ``StreamCompaction.*``               <->  ``StreamCompaction::{CPU,Naive,Efficient,Thrust}::*``

Nothing here computes: every call goes through the library, and there is no CPU fallback.
"""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libmi355x_pathtracer.so")
# A/B timing of variant builds (tools/ab_*.sh): PTX_AB_LIBRARY names another build of the SAME library to load instead, so that the
# scripts no longer overwrite the in-tree product (an interrupted run used to leave a variant in its place).  Development only, and
# behind a second switch: without PTX_DEV=1 the variable is REFUSED (round 4's one GPU fault came through this door -- a test run
# against a stale variant build), so that neither a test run nor the driver can pick up anything but the in-tree product by accident.
if os.environ.get("PTX_AB_LIBRARY"):
    if os.environ.get("PTX_DEV") != "1":
        raise ImportError("PTX_AB_LIBRARY is set (%s) without PTX_DEV=1: variant builds are loaded for A/B timing only; unset it or "
                          "set PTX_DEV=1" % os.environ["PTX_AB_LIBRARY"])
    LIB_PATH = os.path.abspath(os.environ["PTX_AB_LIBRARY"])

PTX_OK = 0
ABI_VERSION = 5          # include/mi355x_pathtracer.h: PTX_ABI_VERSION
vp = C.c_void_p


class PathTracerError(RuntimeError):
    pass


class Material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("specular_exponent", C.c_float), ("specular_color", C.c_float * 3),
                ("hasReflective", C.c_float), ("hasRefractive", C.c_float), ("indexOfRefraction", C.c_float),
                ("emittance", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32),
                ("image", C.POINTER(C.c_uint8))]


class Geom(C.Structure):
    _fields_ = [("type", C.c_int32), ("materialid", C.c_int32),
                ("translation", C.c_float * 3), ("rotation", C.c_float * 3), ("scale", C.c_float * 3),
                ("transform", C.c_float * 16), ("inverseTransform", C.c_float * 16), ("invTranspose", C.c_float * 16),
                ("faceSize", C.c_int32), ("faces", C.POINTER(C.c_float)),
                ("kd", Texture), ("ks", Texture), ("bump", Texture), ("ke", Texture)]


class Camera(C.Structure):
    _fields_ = [("resolution", C.c_int32 * 2), ("position", C.c_float * 3), ("lookAt", C.c_float * 3),
                ("view", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3), ("fov", C.c_float * 2),
                ("pixelLength", C.c_float * 2)]


class Options(C.Structure):
    """Runtime form of the #defines of src/pathtrace.cu:36-40 (+ the multi-GPU row-tile split)."""
    _fields_ = [("depth_of_field", C.c_int32), ("cache_first_bounce", C.c_int32), ("sort_by_material", C.c_int32),
                ("antialiasing", C.c_int32), ("bounding_box", C.c_int32),
                ("tile_rows", C.c_int32), ("tile_rank", C.c_int32), ("tile_world", C.c_int32),
                ("device", C.c_int32), ("batch", C.c_int32), ("no_lds_triangles", C.c_int32), ("apps_variant", C.c_int32), ("no_cull", C.c_int32), ("no_bvh", C.c_int32), ("lanes", C.c_int32), ("no_mesh_split", C.c_int32), ("arith", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("bounces", C.c_int32), ("rays_per_bounce", C.c_int64 * 64), ("rays_total", C.c_int64),
                ("loop_ms_total", C.c_double), ("iterations", C.c_int64), ("fenced", C.c_int64),
                ("stored_paths", C.c_int64), ("stored_with_direction", C.c_int64), ("stored_with_normal_code", C.c_int64)]


def debug_tile_geoms(camera, boxes6, depth_of_field=False, tile=None):
    """CPU only: the per-tile geom masks of the camera-ray bounce (ptx_debug_tile_geoms) for a ctypes Camera, an (n, 6) array of world
    boxes (lo xyz, hi xyz) and an optional row-tile split (rows, rank, world).  Returns a uint32 array, one word per tile of 256 owned
    pixels."""
    L = load_library()
    b = np.ascontiguousarray(boxes6, np.float32).reshape(-1, 6)
    rows, rank, world = tile if tile else (0, 0, 1)
    cap = (camera.resolution[0] * camera.resolution[1] + 255) // 256 + 1
    out = np.zeros(cap, np.uint32)
    n = L.ptx_debug_tile_geoms(C.byref(camera), len(b), _ptr(b), int(bool(depth_of_field)), rows, rank, world, _ptr(out), cap)
    if n < 0:
        raise PathTracerError("ptx_debug_tile_geoms: bad argument")
    return out[:n]


def debug_cull_boxes(boxes6):
    """CPU only: the device's table of the candidate pre-test (ptx_debug_cull_boxes) for an (n, 6) array of corner boxes: (n, 8) float32,
    centre xyz, 0, half extent xyz, 0."""
    L = load_library()
    b = np.ascontiguousarray(boxes6, np.float32).reshape(-1, 6)
    out = np.zeros((len(b), 8), np.float32)
    if L.ptx_debug_cull_boxes(len(b), _ptr(b), _ptr(out)) < 0:
        raise PathTracerError("ptx_debug_cull_boxes: bad argument")
    return out


def build_library(force=False):
    """Compiles the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(PKG_DIR, "csrc")])
    return LIB_PATH


_lib = None


class Orbit(C.Structure):
    """ptx_orbit: phi, theta, zoom and the original lookAt of src/main.cpp:18-20"""
    _fields_ = [("phi", C.c_float), ("theta", C.c_float), ("zoom", C.c_float), ("og_look_at", C.c_float * 3)]


def _share_hip_runtime_with_torch():
    """PyTorch's ROCm wheels bundle their own HIP runtime under torch/lib with the same sonames as /opt/rocm's.
    Whichever copy a process loads first serves both libraries, and torch does not find its GPUs through the system
    copy -- so a process that may import torch later (the multi-GPU driver does) loads torch's copy first.
    Without torch installed this does nothing and the library uses /opt/rocm's runtime."""
    import importlib.util
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(hip):
        try:
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """Loads the native library; raises loudly when it has not been built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise PathTracerError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or make -C mygpuraytracer_amd/csrc). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    i, f = C.c_int, C.c_float
    # the structs below are allocated HERE and filled by the library: a library of another ABI revision (an older variant build
    # loaded for A/B timing) would overrun them or leave their tail unset -- refused, whatever it is
    if not hasattr(L, "ptx_abi_version"):
        raise PathTracerError("%s predates ptx_abi_version (ABI < %d): rebuild it from this tree" % (LIB_PATH, ABI_VERSION))
    L.ptx_abi_version.restype = i
    L.ptx_sizeof_options.restype = L.ptx_sizeof_stats.restype = C.c_size_t
    if L.ptx_abi_version() != ABI_VERSION or L.ptx_sizeof_options() != C.sizeof(Options) or L.ptx_sizeof_stats() != C.sizeof(Stats):
        raise PathTracerError("%s has ABI %d (options %d B, stats %d B), this module expects %d (%d B, %d B): rebuild the library" % (
            LIB_PATH, L.ptx_abi_version(), L.ptx_sizeof_options(), L.ptx_sizeof_stats(), ABI_VERSION, C.sizeof(Options), C.sizeof(Stats)))
    L.ptx_last_error.restype = C.c_char_p
    L.ptx_device_count.restype = i
    L.ptx_default_options.argtypes = [C.POINTER(Options)]
    L.ptx_scene_load.restype, L.ptx_scene_load.argtypes = i, [C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.ptx_scene_free.argtypes = [vp]
    L.ptx_scene_num_geoms.restype, L.ptx_scene_num_geoms.argtypes = i, [vp]
    L.ptx_scene_num_materials.restype, L.ptx_scene_num_materials.argtypes = i, [vp]
    L.ptx_scene_geoms.restype, L.ptx_scene_geoms.argtypes = C.POINTER(Geom), [vp]
    L.ptx_scene_materials.restype, L.ptx_scene_materials.argtypes = C.POINTER(Material), [vp]
    L.ptx_scene_camera.restype, L.ptx_scene_camera.argtypes = C.POINTER(Camera), [vp]
    L.ptx_scene_iterations.restype, L.ptx_scene_iterations.argtypes = i, [vp]
    L.ptx_scene_trace_depth.restype, L.ptx_scene_trace_depth.argtypes = i, [vp]
    L.ptx_scene_set_trace_depth.argtypes = [vp, i]
    L.ptx_scene_set_resolution.argtypes = [vp, i, i]
    L.ptx_scene_image_name.restype, L.ptx_scene_image_name.argtypes = C.c_char_p, [vp]
    L.ptx_scene_apply_runcuda_camera.argtypes = [vp]
    L.ptx_orbit_init.argtypes = [vp, C.POINTER(Orbit)]
    L.ptx_orbit_left_drag.argtypes = [C.POINTER(Orbit), C.c_double, C.c_double, i, i]
    L.ptx_orbit_right_drag.argtypes = [C.POINTER(Orbit), C.c_double, i]
    L.ptx_orbit_middle_drag.argtypes = [vp, C.c_double, C.c_double]
    L.ptx_orbit_recenter.argtypes = [vp, C.POINTER(Orbit)]
    L.ptx_orbit_apply.argtypes = [vp, C.POINTER(Orbit)]
    L.ptx_create.restype = i
    L.ptx_create.argtypes = [i, C.POINTER(Geom), i, C.POINTER(Material), C.POINTER(Camera), i, C.POINTER(Options), vp,
                             vp, C.POINTER(vp)]
    L.ptx_create_from_scene.restype = i
    L.ptx_create_from_scene.argtypes = [vp, C.POINTER(Options), vp, vp, C.POINTER(vp)]
    L.ptx_destroy.argtypes = [vp]
    L.ptx_set_camera.restype, L.ptx_set_camera.argtypes = i, [vp, C.POINTER(Camera), i]
    L.ptx_reset_image.restype, L.ptx_reset_image.argtypes = i, [vp]
    L.ptx_iterate.restype, L.ptx_iterate.argtypes = i, [vp, i]
    L.ptx_set_render_ahead.restype, L.ptx_set_render_ahead.argtypes = i, [vp, i]
    L.ptx_render.restype, L.ptx_render.argtypes = i, [vp, i, i]
    L.ptx_render_strided.restype, L.ptx_render_strided.argtypes = i, [vp, i, i, i]
    L.ptx_write_image.restype, L.ptx_write_image.argtypes = i, [vp, vp]
    L.ptx_synchronize.restype, L.ptx_synchronize.argtypes = i, [vp]
    L.ptx_read_image.restype, L.ptx_read_image.argtypes = i, [vp, vp]
    L.ptx_device_image.restype, L.ptx_device_image.argtypes = vp, [vp]
    L.ptx_read_albedo.restype, L.ptx_read_albedo.argtypes = i, [vp, vp]
    L.ptx_write_denoised_pbo.restype, L.ptx_write_denoised_pbo.argtypes = i, [vp, vp, vp]
    L.ptx_write_denoised_pbo_device.restype, L.ptx_write_denoised_pbo_device.argtypes = i, [vp, vp, vp]
    L.ptx_write_pbo.restype, L.ptx_write_pbo.argtypes = i, [vp, i, vp]
    L.ptx_write_pbo_device.restype, L.ptx_write_pbo_device.argtypes = i, [vp, i, vp]
    L.ptx_last_loop_ms.restype, L.ptx_last_loop_ms.argtypes = C.c_double, [vp]
    L.ptx_get_stats.restype, L.ptx_get_stats.argtypes = i, [vp, C.POINTER(Stats)]
    L.ptx_get_stats_sized.restype, L.ptx_get_stats_sized.argtypes = i, [vp, vp, C.c_size_t]
    L.ptx_owned_pixels.restype, L.ptx_owned_pixels.argtypes = i, [vp]
    L.ptx_stream.restype, L.ptx_stream.argtypes = vp, [vp]
    L.ptx_set_kernel_timing.restype, L.ptx_set_kernel_timing.argtypes = i, [vp, i]
    L.ptx_get_kernel_times.restype, L.ptx_get_kernel_times.argtypes = i, [vp, vp, vp]
    L.ptx_kat_geom_test.restype, L.ptx_kat_geom_test.argtypes = i, [vp, i, i, vp, vp]
    L.ptx_kat_compute_intersections.restype, L.ptx_kat_compute_intersections.argtypes = i, [vp, i, vp, vp]
    L.ptx_kat_obj_tri_test.restype, L.ptx_kat_obj_tri_test.argtypes = i, [vp, i, i, vp, vp]
    L.ptx_kat_jittered_hemisphere.restype, L.ptx_kat_jittered_hemisphere.argtypes = i, [vp, i, vp, vp, i, vp]
    L.ptx_kat_tile_intersect.restype, L.ptx_kat_tile_intersect.argtypes = i, [vp, i, vp, vp, i]
    L.ptx_kat_shade.restype, L.ptx_kat_shade.argtypes = i, [vp, i, i, vp, vp, vp]
    L.ptx_kat_generate.restype, L.ptx_kat_generate.argtypes = i, [vp, i, vp]
    L.ptx_kat_libm.restype, L.ptx_kat_libm.argtypes = i, [vp, i, vp, vp, vp, vp, vp, vp, vp]
    if hasattr(L, "ptx_debug_tile_geoms"):
        L.ptx_debug_tile_geoms.restype, L.ptx_debug_tile_geoms.argtypes = i, [vp, i, vp, i, i, i, i, vp, i]
    if hasattr(L, "ptx_debug_cull_boxes"):
        L.ptx_debug_cull_boxes.restype, L.ptx_debug_cull_boxes.argtypes = i, [i, vp, vp]
    if hasattr(L, "ptx_kat_fast_exact"):        # (absent from the older builds the A/B scripts load through PTX_AB_LIBRARY)
        L.ptx_kat_fast_exact.restype, L.ptx_kat_fast_exact.argtypes = i, [vp, vp]
    L.ptx_debug_set_capture.restype, L.ptx_debug_set_capture.argtypes = i, [vp, i]
    L.ptx_debug_read_stream.restype, L.ptx_debug_read_stream.argtypes = i, [vp, C.POINTER(i), vp, vp, vp, vp, i]
    # several devices behind one handle (csrc/pt_multi.cpp)
    L.ptx_multi_create.restype, L.ptx_multi_create.argtypes = i, [vp, C.POINTER(Options), C.POINTER(C.c_int), i, i, C.POINTER(vp)]
    L.ptx_multi_destroy.argtypes = [vp]
    L.ptx_multi_device_count.restype, L.ptx_multi_device_count.argtypes = i, [vp]
    L.ptx_multi_tracer.restype, L.ptx_multi_tracer.argtypes = vp, [vp, i]
    L.ptx_multi_set_camera.restype, L.ptx_multi_set_camera.argtypes = i, [vp, C.POINTER(Camera), i]
    for n in ("ptx_multi_reset_image", "ptx_multi_synchronize", "ptx_multi_assemble"):
        getattr(L, n).restype, getattr(L, n).argtypes = i, [vp]
    L.ptx_multi_iterate.restype, L.ptx_multi_iterate.argtypes = i, [vp, i]
    L.ptx_multi_set_render_ahead.restype, L.ptx_multi_set_render_ahead.argtypes = i, [vp, i]
    L.ptx_multi_render.restype, L.ptx_multi_render.argtypes = i, [vp, i, i]
    L.ptx_multi_device_image.restype, L.ptx_multi_device_image.argtypes = vp, [vp]
    L.ptx_multi_read_image.restype, L.ptx_multi_read_image.argtypes = i, [vp, vp]
    L.ptx_multi_read_albedo.restype, L.ptx_multi_read_albedo.argtypes = i, [vp, vp]
    L.ptx_multi_get_stats.restype, L.ptx_multi_get_stats.argtypes = i, [vp, C.POINTER(Stats)]
    L.ptx_pin_host_buffer.restype, L.ptx_pin_host_buffer.argtypes = i, [vp, C.c_size_t]
    L.ptx_unpin_host_buffer.restype, L.ptx_unpin_host_buffer.argtypes = i, [vp]
    # stream compaction
    L.sc_cpu_scan.argtypes = [i, vp, vp]
    for n in ("sc_cpu_compact_without_scan", "sc_cpu_compact_with_scan", "sc_efficient_compact", "sc_naive_scan",
              "sc_efficient_scan", "sc_thrust_scan"):
        getattr(L, n).restype, getattr(L, n).argtypes = i, [i, vp, vp]
    L.sc_scan_workspace_bytes.restype, L.sc_scan_workspace_bytes.argtypes = C.c_ulonglong, [i]
    L.sc_scan_device.restype, L.sc_scan_device.argtypes = i, [i, vp, vp, vp, vp]
    L.sc_compact_device.restype, L.sc_compact_device.argtypes = i, [i, vp, vp, vp, vp, vp]
    L.sc_map_to_boolean_device.restype, L.sc_map_to_boolean_device.argtypes = i, [i, vp, vp, vp]
    L.sc_scatter_device.restype, L.sc_scatter_device.argtypes = i, [i, vp, vp, vp, vp, vp]
    L.sc_last_gpu_ms.restype = f
    L.sc_last_cpu_ms.restype = f
    L.sc_ilog2.restype, L.sc_ilog2.argtypes = i, [i]
    L.sc_ilog2ceil.restype, L.sc_ilog2ceil.argtypes = i, [i]
    _lib = L
    return L


def _check(rc, what):
    if rc != PTX_OK:
        raise PathTracerError("%s failed (code %d): %s" % (what, rc, load_library().ptx_last_error().decode()))


def _ptr(a):
    return a.ctypes.data_as(vp)


def default_options(**kw):
    o = Options()
    load_library().ptx_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class Scene:
    """A loaded scenes/*.txt (class Scene, src/scene.h)."""

    def __init__(self, path, base_dir=None, res=None, depth=None):
        self.lib = load_library()
        h = vp()
        _check(self.lib.ptx_scene_load(os.fspath(path).encode(), base_dir.encode() if base_dir else None, C.byref(h)),
               "ptx_scene_load(%s)" % path)
        self.h = h
        if res is not None:
            self.set_resolution(*res)
        if depth is not None:
            self.set_trace_depth(depth)

    def close(self):
        if self.h:
            self.lib.ptx_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_geoms(self):
        return self.lib.ptx_scene_num_geoms(self.h)

    @property
    def num_materials(self):
        return self.lib.ptx_scene_num_materials(self.h)

    @property
    def camera(self):
        return self.lib.ptx_scene_camera(self.h).contents

    @property
    def resolution(self):
        c = self.camera
        return int(c.resolution[0]), int(c.resolution[1])

    @property
    def iterations(self):
        return self.lib.ptx_scene_iterations(self.h)

    @property
    def trace_depth(self):
        return self.lib.ptx_scene_trace_depth(self.h)

    @property
    def image_name(self):
        return self.lib.ptx_scene_image_name(self.h).decode()

    def set_trace_depth(self, d):
        self.lib.ptx_scene_set_trace_depth(self.h, d)

    def set_resolution(self, w, h):
        self.lib.ptx_scene_set_resolution(self.h, w, h)

    def apply_runcuda_camera(self):
        self.lib.ptx_scene_apply_runcuda_camera(self.h)

    # --- the interactive camera of src/main.cpp, driven by a script instead of a mouse ---------------------
    def orbit_init(self):
        """main.cpp:56-70 -> Orbit (phi, theta, zoom, original lookAt)"""
        o = Orbit()
        self.lib.ptx_orbit_init(self.h, C.byref(o))
        return o

    def orbit_events(self, orbit, events):
        """Applies mouse/key events in order, then runCuda's camera recompute (main.cpp:105-123).  events: tuples
        ("left", dx, dy) | ("right", dy) | ("middle", dx, dy) | ("space",) with deltas in window pixels, as the GLFW
        callbacks of main.cpp:166-212 see them.  Hand the result to a tracer with Tracer.set_camera + reset_image."""
        w, h = self.resolution
        for ev in events:
            kind = ev[0]
            if kind == "left":
                self.lib.ptx_orbit_left_drag(C.byref(orbit), float(ev[1]), float(ev[2]), w, h)
            elif kind == "right":
                self.lib.ptx_orbit_right_drag(C.byref(orbit), float(ev[1]), h)
            elif kind == "middle":
                self.lib.ptx_orbit_middle_drag(self.h, float(ev[1]), float(ev[2]))
            elif kind == "space":
                self.lib.ptx_orbit_recenter(self.h, C.byref(orbit))
            else:
                raise PathTracerError("unknown camera event %r" % (ev,))
            # main.cpp applies the recompute on the next frame, i.e. between any two events that set camchanged;
            # the middle drag reads cam.view / cam.right of the recomputed camera, so do the same here
            self.lib.ptx_orbit_apply(self.h, C.byref(orbit))
        return orbit

    def dump(self):
        """POD view of the scene as numpy arrays (same dict layout the CPU checkers use in tests/cpulibs.py)."""
        ng, nm = self.num_geoms, self.num_materials
        geoms = self.lib.ptx_scene_geoms(self.h)
        mats = self.lib.ptx_scene_materials(self.h)
        gints = np.zeros((ng, 3), np.int32)
        trs = np.zeros((ng, 9), np.float32)
        gm = np.zeros((ng, 48), np.float32)
        faces, textures = [], {}
        for i in range(ng):
            g = geoms[i]
            gints[i] = (g.type, g.materialid, g.faceSize)
            trs[i] = list(g.translation) + list(g.rotation) + list(g.scale)
            gm[i] = list(g.transform) + list(g.inverseTransform) + list(g.invTranspose)
            if g.faceSize:
                faces.append(np.ctypeslib.as_array(g.faces, shape=(g.faceSize, 15)).copy())
            else:
                faces.append(np.zeros((0, 15), np.float32))
            for which, t in enumerate((g.kd, g.ks, g.ke, g.bump)):      # checker order: kd, ks, ke, bump
                if t.channels:
                    textures[(i, which)] = np.ctypeslib.as_array(t.image, shape=(t.height, t.width, t.channels)).copy()
        m = np.zeros((nm, 11), np.float32)
        for i in range(nm):
            m[i] = np.frombuffer(bytes(mats[i]), np.float32)
        c = self.camera
        cf = np.array(list(c.position) + list(c.lookAt) + list(c.view) + list(c.up) + list(c.right) + list(c.fov)
                      + list(c.pixelLength), np.float32)
        ci = np.array([c.resolution[0], c.resolution[1], self.iterations, self.trace_depth], np.int32)
        return dict(geom_ints=gints, geom_trs=trs, geom_mats=gm, materials=m, faces=faces, cam_ints=ci, cam_floats=cf,
                    textures=textures)


class MultiTracer:
    """The frame split into interleaved row blocks over several devices of one node, one process (opaque ptx_multi; the C/C++
    side of the tile split -- include/mi355x_pathtracer.h "N GPUs of one node").  `devices` may repeat an ordinal."""

    def __init__(self, scene, devices, tile_rows=8, options=None, **opt_kw):
        self.lib = load_library()
        if self.lib.ptx_device_count() < 1:
            raise PathTracerError("no HIP device is visible; the path tracer has no CPU path")
        self.options = options if options is not None else default_options(**opt_kw)
        devs = (C.c_int * len(devices))(*devices)
        h = vp()
        _check(self.lib.ptx_multi_create(scene.h, C.byref(self.options), devs, len(devices), tile_rows, C.byref(h)), "ptx_multi_create")
        self.h, self.n = h, len(devices)
        self.width, self.height = scene.resolution

    def close(self):
        if getattr(self, "h", None):
            self.lib.ptx_multi_destroy(self.h)
            self.h = None

    __del__ = lambda self: self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def render(self, iter_first, count):
        _check(self.lib.ptx_multi_render(self.h, iter_first, count), "ptx_multi_render")

    def pathtrace(self, iteration):
        _check(self.lib.ptx_multi_iterate(self.h, iteration), "ptx_multi_iterate")

    def set_render_ahead(self, on=True):
        _check(self.lib.ptx_multi_set_render_ahead(self.h, 1 if on else 0), "ptx_multi_set_render_ahead")

    def synchronize(self):
        _check(self.lib.ptx_multi_synchronize(self.h), "ptx_multi_synchronize")

    def assemble(self):
        _check(self.lib.ptx_multi_assemble(self.h), "ptx_multi_assemble")

    def read_image(self):
        out = np.zeros((self.width * self.height, 3), np.float32)
        _check(self.lib.ptx_multi_read_image(self.h, _ptr(out)), "ptx_multi_read_image")
        return out

    def read_albedo(self):
        out = np.zeros((self.width * self.height, 3), np.float32)
        _check(self.lib.ptx_multi_read_albedo(self.h, _ptr(out)), "ptx_multi_read_albedo")
        return out

    def stats(self):
        s = Stats()
        _check(self.lib.ptx_multi_get_stats(self.h, C.byref(s)), "ptx_multi_get_stats")
        return dict(bounces=s.bounces, rays_per_bounce=[int(s.rays_per_bounce[b]) for b in range(min(s.bounces, 64))],
                    rays_total=int(s.rays_total), loop_ms_total=float(s.loop_ms_total), iterations=int(s.iterations), fenced=int(s.fenced), stored_paths=int(s.stored_paths),
                    stored_with_direction=int(s.stored_with_direction), stored_with_normal_code=int(s.stored_with_normal_code))


class Tracer:
    """pathtraceInit / pathtrace / pathtraceFree on one device (opaque ptx_tracer)."""

    def __init__(self, scene, options=None, external_image_ptr=None, stream_ptr=None, **opt_kw):
        self.lib = load_library()
        if self.lib.ptx_device_count() < 1:
            raise PathTracerError("no HIP device is visible; the path tracer has no CPU path")
        self.options = options if options is not None else default_options(**opt_kw)
        h = vp()
        _check(self.lib.ptx_create_from_scene(scene.h, C.byref(self.options), external_image_ptr, stream_ptr, C.byref(h)),
               "ptx_create")
        self.h = h
        self.width, self.height = scene.resolution
        self.trace_depth = scene.trace_depth
        self.iteration = 0

    @classmethod
    def from_pod(cls, dump, options=None, **opt_kw):
        """pathtraceInit from plain scene arrays instead of a scene file: ptx_create (include/mi355x_pathtracer.h) with the POD
        dict `Scene.dump()` returns (geom_ints [type, material, faces], geom_mats [transform, inverse, invTranspose], materials,
        faces per geom, cam_ints [W, H, iterations, depth], cam_floats, textures {(geom, kd|ks|ke|bump): uint8 HxWxC}) -- the
        way a caller that holds a loaded `Scene` of the reference (src/scene.h:11-32) hands its vectors over."""
        self = cls.__new__(cls)
        self.lib = load_library()
        if self.lib.ptx_device_count() < 1:
            raise PathTracerError("no HIP device is visible; the path tracer has no CPU path")
        self.options = options if options is not None else default_options(**opt_kw)
        ng, nm = len(dump["geom_ints"]), len(dump["materials"])
        geoms = (Geom * max(ng, 1))()
        keep = []                                           # host arrays the structs point into, until ptx_create has uploaded them
        trs = dump.get("geom_trs")
        for i in range(ng):
            g = geoms[i]
            g.type, g.materialid = int(dump["geom_ints"][i][0]), int(dump["geom_ints"][i][1])
            if trs is not None:
                g.translation[:], g.rotation[:], g.scale[:] = [list(map(float, trs[i][k:k + 3])) for k in (0, 3, 6)]
            gm = np.ascontiguousarray(dump["geom_mats"][i], np.float32)
            g.transform[:], g.inverseTransform[:], g.invTranspose[:] = list(gm[:16]), list(gm[16:32]), list(gm[32:48])
            f = np.ascontiguousarray(dump["faces"][i], np.float32).reshape(-1, 15)
            keep.append(f)
            g.faceSize = len(f)
            g.faces = f.ctypes.data_as(C.POINTER(C.c_float)) if len(f) else None
            for which, name in enumerate(("kd", "ks", "ke", "bump")):          # checker order: kd, ks, ke, bump
                img = (dump.get("textures") or {}).get((i, which))
                if img is not None:
                    img = np.ascontiguousarray(img, np.uint8)
                    keep.append(img)
                    t = getattr(g, name)
                    t.height, t.width, t.channels = img.shape
                    t.image = img.ctypes.data_as(C.POINTER(C.c_uint8))
        mats = (Material * max(nm, 1))()
        for i in range(nm):
            C.memmove(C.byref(mats[i]), np.ascontiguousarray(dump["materials"][i], np.float32).ctypes.data, 44)
        cam = Camera()
        ci, cf = dump["cam_ints"], [float(v) for v in dump["cam_floats"]]
        cam.resolution[:] = [int(ci[0]), int(ci[1])]
        cam.position[:], cam.lookAt[:], cam.view[:], cam.up[:], cam.right[:] = cf[0:3], cf[3:6], cf[6:9], cf[9:12], cf[12:15]
        cam.fov[:], cam.pixelLength[:] = cf[15:17], cf[17:19]
        h = vp()
        _check(self.lib.ptx_create(ng, geoms, nm, mats, C.byref(cam), int(ci[3]), C.byref(self.options), None, None, C.byref(h)), "ptx_create")
        self.h = h
        self.width, self.height = int(ci[0]), int(ci[1])
        self.trace_depth = int(ci[3])
        self.iteration = 0
        return self

    def close(self):
        if getattr(self, "h", None):
            self.lib.ptx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # --- the reference's per-frame call -----------------------------------------------------------------
    def pathtrace(self, iteration, read_image=False):
        _check(self.lib.ptx_iterate(self.h, iteration), "ptx_iterate")
        self.iteration = iteration
        return self.read_image() if read_image else None

    def set_render_ahead(self, on=True):
        """ptx_set_render_ahead: iterate() calls that count up are served from batches traced in the background"""
        _check(self.lib.ptx_set_render_ahead(self.h, 1 if on else 0), "ptx_set_render_ahead")

    def render(self, iter_first, count, stride=1):
        """iterations iter_first, iter_first + stride, ... (count of them), no host round trip in between"""
        _check(self.lib.ptx_render_strided(self.h, iter_first, count, stride), "ptx_render_strided")
        self.iteration = iter_first + (count - 1) * stride

    def write_image(self, rgb_sum):
        """uploads an accumulation buffer (what read_image returned earlier): resume from a checkpoint"""
        buf = np.ascontiguousarray(rgb_sum, np.float32).reshape(-1)
        if buf.size != self.width * self.height * 3:
            raise PathTracerError("write_image: expected %d floats, got %d" % (self.width * self.height * 3, buf.size))
        _check(self.lib.ptx_write_image(self.h, _ptr(buf)), "ptx_write_image")

    CKPT_MAGIC = b"PTXCKPT1"

    def save_checkpoint(self, path, iterations_done):
        """accumulation buffer + the number of iterations in it; same file format as mi355x_pathtrace --checkpoint
        (csrc/pt_image.h: magic, int32 W, H, int64 iterations, W*H*3 float32)"""
        img = self.read_image()
        with open(path, "wb") as f:
            f.write(self.CKPT_MAGIC)
            f.write(np.int32([self.width, self.height]).tobytes())
            f.write(np.int64([iterations_done]).tobytes())
            f.write(np.ascontiguousarray(img, "<f4").tobytes())

    def load_checkpoint(self, path):
        """-> iterations already in the buffer; continue with render(iterations + 1, ...)"""
        with open(path, "rb") as f:
            if f.read(8) != self.CKPT_MAGIC:
                raise PathTracerError("%s is not a checkpoint file" % path)
            w, h = (int(v) for v in np.frombuffer(f.read(8), "<i4"))
            iters = int(np.frombuffer(f.read(8), "<i8")[0])
            if (w, h) != (self.width, self.height):
                raise PathTracerError("checkpoint %s is %dx%d, the tracer is %dx%d" % (path, w, h, self.width, self.height))
            data = np.frombuffer(f.read(), "<f4")
        if data.size != w * h * 3:
            raise PathTracerError("checkpoint %s is truncated" % path)
        self.write_image(data)
        return iters

    def synchronize(self):
        _check(self.lib.ptx_synchronize(self.h), "ptx_synchronize")

    def set_camera(self, scene):
        _check(self.lib.ptx_set_camera(self.h, C.byref(scene.camera), scene.trace_depth), "ptx_set_camera")

    def reset_image(self):
        _check(self.lib.ptx_reset_image(self.h), "ptx_reset_image")

    def read_image(self):
        out = np.zeros((self.width * self.height, 3), np.float32)
        _check(self.lib.ptx_read_image(self.h, _ptr(out)), "ptx_read_image")
        return out

    def read_albedo(self):
        out = np.zeros((self.width * self.height, 3), np.float32)
        _check(self.lib.ptx_read_albedo(self.h, _ptr(out)), "ptx_read_albedo")
        return out

    def denoised_pbo(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32).reshape(self.width * self.height, 3)
        out = np.zeros((self.width * self.height, 4), np.uint8)
        _check(self.lib.ptx_write_denoised_pbo(self.h, _ptr(rgb), _ptr(out)), "ptx_write_denoised_pbo")
        return out

    def denoised_pbo_device(self, rgb, device_pbo):
        """sendToGPU (apps/src/pathtrace.cu:673-685): host frame -> 8-bit preview in a device buffer (address as int)"""
        rgb = np.ascontiguousarray(rgb, np.float32).reshape(self.width * self.height, 3)
        _check(self.lib.ptx_write_denoised_pbo_device(self.h, _ptr(rgb), device_pbo), "ptx_write_denoised_pbo_device")

    def pbo(self, iteration):
        out = np.zeros((self.width * self.height, 4), np.uint8)
        _check(self.lib.ptx_write_pbo(self.h, iteration, _ptr(out)), "ptx_write_pbo")
        return out

    def device_image_ptr(self):
        return self.lib.ptx_device_image(self.h)

    def stream_ptr(self):
        return self.lib.ptx_stream(self.h)

    def last_loop_ms(self):
        return float(self.lib.ptx_last_loop_ms(self.h))

    def set_kernel_timing(self, on):
        _check(self.lib.ptx_set_kernel_timing(self.h, int(bool(on))), "ptx_set_kernel_timing")

    def kernel_times(self):
        """{kernel: (total ms, launches)} since the last call (needs set_kernel_timing(True))."""
        ms = np.zeros(4, np.float64)
        n = np.zeros(4, np.int64)
        _check(self.lib.ptx_get_kernel_times(self.h, _ptr(ms), _ptr(n)), "ptx_get_kernel_times")
        names = ("k_bounce<first>", "k_bounce", "k_mesh", "k_bounce pass2")      # (split mesh search: the first two are then its pass 1)
        return {nm: (float(ms[k]), int(n[k])) for k, nm in enumerate(names)}

    def kat_fast_exact(self):
        """mismatches of the device's guarded core sqrt / 1/sqrt / 1/a against the compiler's IEEE expansions over all 2^32 operands"""
        m = np.zeros(3, np.int64)
        _check(self.lib.ptx_kat_fast_exact(self.h, _ptr(m)), "ptx_kat_fast_exact")
        return m.tolist()

    def owned_pixels(self):
        return self.lib.ptx_owned_pixels(self.h)

    def stats(self):
        s = Stats()
        _check(self.lib.ptx_get_stats(self.h, C.byref(s)), "ptx_get_stats")
        return dict(bounces=s.bounces, rays_per_bounce=[int(s.rays_per_bounce[b]) for b in range(min(s.bounces, 64))],
                    rays_total=int(s.rays_total), loop_ms_total=float(s.loop_ms_total), iterations=int(s.iterations), fenced=int(s.fenced), stored_paths=int(s.stored_paths),
                    stored_with_direction=int(s.stored_with_direction), stored_with_normal_code=int(s.stored_with_normal_code))

    # --- per-stage entry points used by the parity tests --------------------------------------------------
    def geom_test(self, gi, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((len(rays), 10), np.float32)
        _check(self.lib.ptx_kat_geom_test(self.h, gi, len(rays), _ptr(rays), _ptr(out)), "ptx_kat_geom_test")
        return out

    def obj_tri_test(self, gi, rays):
        """objTriIntersectionTest of the reference (src/intersections.h:284-315, dead code there) on OBJ geom gi: (n, 8) = t, point, normal, outside"""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((len(rays), 8), np.float32)
        _check(self.lib.ptx_kat_obj_tri_test(self.h, gi, len(rays), _ptr(rays), _ptr(out)), "ptx_kat_obj_tri_test")
        return out

    def jittered_hemisphere(self, normals, seeds, max_iter=5000):
        """calculateJitteredDirectionHemisphere of the reference (src/interactions.h:46-85, dead code there): (n, 3) normals and (n, 3) int
        (iter, index, depth) -> (n, 3) directions"""
        normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        seeds = np.ascontiguousarray(seeds, np.int32).reshape(-1, 3)
        out = np.zeros((len(normals), 3), np.float32)
        _check(self.lib.ptx_kat_jittered_hemisphere(self.h, len(normals), _ptr(normals), _ptr(seeds), int(max_iter), _ptr(out)), "ptx_kat_jittered_hemisphere")
        return out

    def compute_intersections(self, paths):
        paths = np.ascontiguousarray(paths)
        assert paths.dtype.itemsize == 44
        out = np.zeros(len(paths), np.dtype([("t", "<f4"), ("normal", "<f4", 3), ("materialId", "<i4"),
                                             ("texcoord", "<f4", 2), ("geomId", "<i4")]))
        _check(self.lib.ptx_kat_compute_intersections(self.h, len(paths), _ptr(paths), _ptr(out)), "ptx_kat_compute_intersections")
        return out

    def tile_intersect(self, paths, split=False):
        """computeIntersections through the production functions (cullMask, primKey / meshKey, decodeKey): ptx_kat_tile_intersect"""
        paths = np.ascontiguousarray(paths)
        assert paths.dtype.itemsize == 44
        out = np.zeros(len(paths), np.dtype([("t", "<f4"), ("normal", "<f4", 3), ("materialId", "<i4"),
                                             ("texcoord", "<f4", 2), ("geomId", "<i4")]))
        _check(self.lib.ptx_kat_tile_intersect(self.h, len(paths), _ptr(paths), _ptr(out), 1 if split else 0), "ptx_kat_tile_intersect")
        return out

    def shade(self, iteration, idx, isects, paths):
        paths = np.array(paths, copy=True)
        isects = np.ascontiguousarray(isects)
        idx = np.ascontiguousarray(idx, np.int32)
        assert paths.dtype.itemsize == 44 and isects.dtype.itemsize == 32
        _check(self.lib.ptx_kat_shade(self.h, iteration, len(paths), _ptr(idx), _ptr(isects), _ptr(paths)), "ptx_kat_shade")
        return paths

    def generate(self, iteration):
        out = np.zeros(self.width * self.height, np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("color", "<f4", 3),
                                                            ("pixelIndex", "<i4"), ("remainingBounces", "<i4")]))
        _check(self.lib.ptx_kat_generate(self.h, iteration, _ptr(out)), "ptx_kat_generate")
        return out

    def libm(self, x, pw, pxy):
        x = np.ascontiguousarray(x, np.float32)
        pw = np.ascontiguousarray(pw, np.float64)
        pxy = np.ascontiguousarray(pxy, np.float32).reshape(-1, 2)
        n = len(x)
        assert len(pw) == n and len(pxy) == n
        s, c, po = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        p5 = np.zeros(n, np.float64)
        _check(self.lib.ptx_kat_libm(self.h, n, _ptr(x), _ptr(s), _ptr(c), _ptr(pw), _ptr(p5), _ptr(pxy), _ptr(po)), "ptx_kat_libm")
        return s, c, p5, po

    def debug_capture(self, bounce):
        _check(self.lib.ptx_debug_set_capture(self.h, bounce), "ptx_debug_set_capture")

    def debug_stream(self):
        cap = self.width * self.height
        pix, idx, mat = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int(0)
        _check(self.lib.ptx_debug_read_stream(self.h, C.byref(n), _ptr(pix), _ptr(idx), _ptr(mat), None, 0), "ptx_debug_read_stream")
        m = n.value
        fields = np.zeros((14, max(m, 1)), np.float32)
        _check(self.lib.ptx_debug_read_stream(self.h, C.byref(n), _ptr(pix), _ptr(idx), _ptr(mat), _ptr(fields), m), "ptx_debug_read_stream")
        names = ("px", "py", "pz", "dx", "dy", "dz", "cr", "cg", "cb", "nx", "ny", "nz", "u", "v")
        out = dict(pix=pix[:m].copy(), idx=idx[:m].copy(), mat=mat[:m].copy())
        for k, nm in enumerate(names):
            out[nm] = fields[k, :m].copy()
        return out


class StreamCompaction:
    """namespace StreamCompaction (stream_compaction/*.cu): host arrays in, host arrays out."""

    def __init__(self):
        self.lib = load_library()

    def _scan(self, fn, a):
        a = np.ascontiguousarray(a, np.int32)
        out = np.zeros_like(a)
        rc = fn(len(a), _ptr(out), _ptr(a))
        if rc:
            _check(rc, fn.__name__)
        return out

    def cpu_scan(self, a):
        a = np.ascontiguousarray(a, np.int32)
        out = np.zeros_like(a)
        self.lib.sc_cpu_scan(len(a), _ptr(out), _ptr(a))
        return out

    def cpu_compact_without_scan(self, a):
        a = np.ascontiguousarray(a, np.int32)
        out = np.zeros_like(a)
        n = self.lib.sc_cpu_compact_without_scan(len(a), _ptr(out), _ptr(a))
        return out[:n].copy()

    def cpu_compact_with_scan(self, a):
        a = np.ascontiguousarray(a, np.int32)
        out = np.zeros_like(a)
        n = self.lib.sc_cpu_compact_with_scan(len(a), _ptr(out), _ptr(a))
        return out[:n].copy()

    def naive_scan(self, a):
        return self._scan(self.lib.sc_naive_scan, a)

    def efficient_scan(self, a):
        return self._scan(self.lib.sc_efficient_scan, a)

    def thrust_scan(self, a):
        return self._scan(self.lib.sc_thrust_scan, a)

    def efficient_compact(self, a):
        a = np.ascontiguousarray(a, np.int32)
        out = np.zeros_like(a)
        n = self.lib.sc_efficient_compact(len(a), _ptr(out), _ptr(a))
        if n < 0:
            raise PathTracerError("sc_efficient_compact failed: %s" % self.lib.ptx_last_error().decode())
        return out[:n].copy()

    # device-pointer forms (addresses as ints, e.g. torch tensor.data_ptr(); stream = a hipStream_t handle or 0)
    def workspace_bytes(self, n):
        return int(self.lib.sc_scan_workspace_bytes(int(n)))

    def scan_device(self, n, d_out, d_in, d_workspace, stream=0):
        _check(self.lib.sc_scan_device(int(n), d_out, d_in, d_workspace, stream), "sc_scan_device")

    def compact_device(self, n, d_out, d_in, d_count, d_workspace, stream=0):
        _check(self.lib.sc_compact_device(int(n), d_out, d_in, d_count, d_workspace, stream), "sc_compact_device")

    # StreamCompaction::Common::kernMapToBoolean / kernScatter (stream_compaction/common.cu:25-49) on device arrays
    def map_to_boolean_device(self, n, d_bools, d_in, stream=0):
        _check(self.lib.sc_map_to_boolean_device(int(n), d_bools, d_in, stream), "sc_map_to_boolean_device")

    def scatter_device(self, n, d_out, d_in, d_bools, d_indices, stream=0):
        _check(self.lib.sc_scatter_device(int(n), d_out, d_in, d_bools, d_indices, stream), "sc_scatter_device")

    def last_gpu_ms(self):
        return float(self.lib.sc_last_gpu_ms())

    def last_cpu_ms(self):
        return float(self.lib.sc_last_cpu_ms())
