"""CPU: the C-ABI library loads without a GPU and exports every symbol the headers in include/ declare; the
product refuses to compute without a device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:ptx|sc)_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header", ["mi355x_pathtracer.h", "mi355x_stream_compaction.h"])
def test_every_declared_symbol_is_exported(product, header):
    lib = ctypes.CDLL(product.LIB_PATH)
    names = declared_symbols(header)
    assert len(names) > 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_layouts_match_the_reference_records(product):
    from mygpuraytracer_amd import api
    assert ctypes.sizeof(api.Material) == 44            # struct Material, sceneStructs.h:71-81
    assert ctypes.sizeof(api.Camera) == 84              # struct Camera ([probe] SURVEY 8)
    assert ctypes.sizeof(api.Options) == 68             # 17 int32 (round 5: + arith)
    # the library says which revision of the structs it was compiled with, and the module checks it at load (api.load_library)
    lib = product.load_library()
    assert lib.ptx_abi_version() == api.ABI_VERSION == 5
    assert lib.ptx_sizeof_options() == ctypes.sizeof(api.Options) and lib.ptx_sizeof_stats() == ctypes.sizeof(api.Stats)


def test_no_cpu_fallback(product):
    lib = product.load_library()
    if lib.ptx_device_count() > 0:
        pytest.skip("a HIP device is present")
    s = product.Scene(os.path.join(ROOT, "scenes", "sphere.txt"))
    with pytest.raises(product.PathTracerError):
        product.Tracer(s)
    sc = product.StreamCompaction()
    with pytest.raises(product.PathTracerError):
        sc.efficient_scan(np.arange(10, dtype=np.int32))
    with pytest.raises(product.PathTracerError):
        sc.efficient_compact(np.arange(10, dtype=np.int32))


def test_cpu_stream_compaction_entry_points(product, oracle_lib):
    """StreamCompaction::CPU::{scan, compactWithoutScan, compactWithScan} of the library vs numpy and the oracle."""
    sc = product.StreamCompaction()
    rng = np.random.default_rng(11)
    for n in (1, 2, 255, 256, 257, (1 << 20) - 3):
        a = (rng.integers(0, 7, n) * rng.integers(0, 2, n)).astype(np.int32)
        want = np.concatenate([[0], np.cumsum(a, dtype=np.int64)[:-1]]).astype(np.int32)
        assert np.array_equal(sc.cpu_scan(a), want)
        assert np.array_equal(sc.cpu_scan(a), oracle_lib.sc_scan(a))
        assert np.array_equal(sc.cpu_compact_without_scan(a), a[a != 0])
        assert np.array_equal(sc.cpu_compact_with_scan(a), a[a != 0])
        assert sc.last_cpu_ms() >= 0.0
    lib = product.load_library()
    assert [lib.sc_ilog2(v) for v in (1, 2, 3, 4, 1023, 1024)] == [0, 1, 1, 2, 9, 10]
    assert [lib.sc_ilog2ceil(v) for v in (1, 2, 3, 4, 1023, 1025)] == [0, 1, 2, 2, 10, 11]


def _png_rgb8_pixels(data, max_rows=None):
    """Decodes an 8-bit RGB, non-interlaced PNG with zlib + the five row filters -> (w, h, rows as int array [rows, w*3])"""
    import struct
    import zlib
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(typ + body) == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        if typ == b"IDAT":
            idat += body
        pos += 12 + n
    w, h = ihdr[:2]
    assert ihdr[2:] == (8, 2, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 3 + 1)
    nrows = h if max_rows is None else min(h, max_rows)
    dec = np.zeros((nrows, w * 3), np.int64)
    for y in range(nrows):
        f, line = rows[y, 0], rows[y, 1:].astype(np.int64)
        for i in range(w * 3):
            a = dec[y, i - 3] if i >= 3 else 0
            b = dec[y - 1, i] if y else 0
            c = dec[y - 1, i - 3] if (y and i >= 3) else 0
            if f == 4:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            else:
                pred = (0, a, b, (a + b) >> 1)[f]
            dec[y, i] = (line[i] + pred) & 255
    return w, h, dec


def test_png_writer_matches_saveimage_semantics(tmp_path):
    """pt_image.h: saveImage's x-mirror + savePNG's clamp*255 truncation (src/main.cpp:86-92, src/image.cpp:26-31) and a
    decodable PNG (checked with zlib / PIL-free parsing)."""
    import struct
    import subprocess
    import zlib
    src = tmp_path / "t.cpp"
    src.write_text('#include "%s/mygpuraytracer_amd/csrc/pt_image.h"\n'
                   'int main(){ float img[2*2*3]={0.5f,1.f,3.f, 0.f,0.25f,-1.f, 2.f,2.f,2.f, 0.999f,0.5f,0.1f}; std::vector<uint8_t> o;'
                   ' ptimg::to_rgb8_mirrored(2,2,img,2.f,o); return ptimg::write_png_rgb8("%s/o.png",2,2,o.data())?0:1; }\n' % (ROOT, tmp_path))
    subprocess.check_call(["g++", "-std=c++17", "-o", str(tmp_path / "t"), str(src)])
    subprocess.check_call([str(tmp_path / "t")])
    w, h, rows = _png_rgb8_pixels((tmp_path / "o.png").read_bytes())
    assert (w, h) == (2, 2)
    # row 0: pixel x=1 (0, .125, clamp(-.5)=0) then x=0 (.25, .5, clamp(1.5)=1)
    assert list(rows[0]) == [0, 31, 0, 63, 127, 255]
    assert list(rows[1]) == [127, 63, 12, 255, 255, 255]


def test_png_writer_matches_savepng_bytes(tmp_path):
    """pt_image.h write_png_rgb8: byte-identical files to image::savePNG's stbi_write_png (src/image.cpp:33) -- the same row
    filters and the same fixed-Huffman LZ stream as the stb_image_write the reference vendors; tests/golden/png_files.npz
    was written by that code (make_golden.py png_files).  Also against the live library when oracle/_ref is present, and
    every file decodes back to its pixels."""
    import subprocess
    import sys
    import zlib
    sys.path.insert(0, os.path.dirname(__file__))
    import pngwritecases
    src = tmp_path / "t.cpp"
    src.write_text('#include "%s/mygpuraytracer_amd/csrc/pt_image.h"\n'
                   'int main(int argc, char **argv){ int w = atoi(argv[1]), h = atoi(argv[2]); std::vector<uint8_t> px((size_t)w*h*3);\n'
                   ' FILE *f = fopen(argv[3], "rb"); if (!f || fread(px.data(), 1, px.size(), f) != px.size()) return 2; fclose(f);\n'
                   ' return ptimg::write_png_rgb8(argv[4], w, h, px.data()) ? 0 : 1; }\n' % ROOT)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", str(tmp_path / "t"), str(src)])
    gold = np.load(os.path.join(ROOT, "tests", "golden", "png_files.npz"))
    cases = pngwritecases.cases()
    assert sorted(cases) == sorted(gold.files)
    ref = os.path.join(ROOT, "oracle", "_ref", "libptref.so")
    live = ctypes.CDLL(ref) if os.path.exists(ref) else None
    for name, img in cases.items():
        raw, out = tmp_path / "in.raw", tmp_path / "out.png"
        raw.write_bytes(img.tobytes())
        subprocess.check_call([str(tmp_path / "t"), str(img.shape[1]), str(img.shape[0]), str(raw), str(out)])
        got = out.read_bytes()
        assert got == gold[name].tobytes(), name
        if live is not None:
            lp = tmp_path / "live.png"
            live.stbi_write_png.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
            assert live.stbi_write_png(str(lp).encode(), img.shape[1], img.shape[0], 3, img.ctypes.data, img.shape[1] * 3) == 1
            assert got == lp.read_bytes(), name
        # and it is a PNG of those pixels (the big frames: the first rows are enough here)
        h, w = img.shape[:2]
        n = 4 if h * w > 4000 else h
        pw, ph, dec = _png_rgb8_pixels(got, max_rows=n)
        assert (pw, ph) == (w, h) and np.array_equal(dec, img.reshape(h, w * 3)[:n].astype(np.int64)), name


def test_hdr_writer_matches_savehdr_bytes(tmp_path):
    """pt_image.h write_hdr: byte-identical files to image::saveHDR (src/image.cpp:41-45), i.e. to the stb_image_write the
    reference vendors -- tests/golden/hdr_files.npz was written by that code (make_golden.py hdr_files).  Also the live
    library when oracle/_ref is present."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import hdrcases
    src = tmp_path / "t.cpp"
    src.write_text('#include "%s/mygpuraytracer_amd/csrc/pt_image.h"\n'
                   'int main(int argc, char **argv){ int w = atoi(argv[1]), h = atoi(argv[2]); std::vector<float> px((size_t)w*h*3);\n'
                   ' FILE *f = fopen(argv[3], "rb"); if (!f || fread(px.data(), 4, px.size(), f) != px.size()) return 2; fclose(f);\n'
                   ' return ptimg::write_hdr(argv[4], w, h, px.data()) ? 0 : 1; }\n' % ROOT)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", str(tmp_path / "t"), str(src)])
    gold = np.load(os.path.join(ROOT, "tests", "golden", "hdr_files.npz"))
    cases = hdrcases.cases()
    assert sorted(cases) == sorted(gold.files)
    ref = os.path.join(ROOT, "oracle", "_ref", "libptref.so")
    live = ctypes.CDLL(ref) if os.path.exists(ref) else None
    for name, img in cases.items():
        raw, out = tmp_path / "in.raw", tmp_path / "out.hdr"
        raw.write_bytes(img.tobytes())
        subprocess.check_call([str(tmp_path / "t"), str(img.shape[1]), str(img.shape[0]), str(raw), str(out)])
        got = out.read_bytes()
        assert got == gold[name].tobytes(), name
        if live is not None:
            lp = tmp_path / "live.hdr"
            live.stbi_write_hdr.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
            assert live.stbi_write_hdr(str(lp).encode(), img.shape[1], img.shape[0], 3, img.ctypes.data) == 1
            assert got == lp.read_bytes(), name


def test_stream_compaction_namespaces_cpu(product, tmp_path):
    """csrc/stream_compaction_api.h: StreamCompaction::CPU::{scan, compactWithoutScan, compactWithScan}, timer() and
    ilog2 / ilog2ceil spelled as the reference's users spell them (stream_compaction/cpu.h, common.h) compile with g++ and
    give the right answers without a GPU (tests/sc_veneer_check.cpp; the GPU namespaces run in the GPU tier)."""
    import subprocess
    exe = tmp_path / "sc_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", str(exe), os.path.join(ROOT, "tests", "sc_veneer_check.cpp"),
                           "-L" + os.path.dirname(product.LIB_PATH), "-lmi355x_pathtracer", "-Wl,-rpath," + os.path.dirname(product.LIB_PATH)])
    out = subprocess.check_output([str(exe), "cpu"], text=True)
    assert "cpu: 0 mismatches" in out


def test_headers_are_plain_c_and_link(product, tmp_path):
    """The boundary is a C ABI: both headers compile as C99 (-pedantic) and a C program links against the library and
    calls entry points that need no GPU (what a cgo / JNI / ctypes binding would do)."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "mi355x_pathtracer.h"\n#include "mi355x_stream_compaction.h"\n#include <stdio.h>\n'
                   'int main(void) {\n    ptx_options opt; ptx_orbit orb; ptx_stats st; (void)orb; (void)st;\n'
                   '    ptx_default_options(&opt);\n    int in[5] = {1, 0, 2, 0, 3}, out[5];\n'
                   '    int n = sc_cpu_compact_without_scan(5, out, in);\n'
                   '    if (ptx_abi_version() != PTX_ABI_VERSION || ptx_sizeof_options() != sizeof opt || ptx_sizeof_stats() != sizeof st) return 2;\n'
                   '    printf("%d %d %d %d %d\\n", (int)sizeof(ptx_options), opt.antialiasing, opt.cache_first_bounce, n, opt.arith);\n    return 0;\n}\n')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(product.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L", libdir, "-lmi355x_pathtracer", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([str(exe)], text=True).split()
    assert out == ["68", "1", "1", "3", "0"]


def test_veneer_null_pbo_is_not_ambiguous(product, tmp_path):
    """csrc/pathtrace_api.h shows ONE function under the name `pathtrace` (and `sendToGPU`) per translation unit: with HIP's
    uchar4 known it is the reference's own signature (src/pathtrace.h:9, apps/src/pathtrace.h:10), without it the void* one.
    Round 3 declared both in a HIP translation unit, so `pathtrace(nullptr, 0, it)` -- the call the header advertises for "no
    preview" and tools/gpu_dropin_loop_cpp.cpp makes -- no longer compiled.  Compile-only (no GPU): the three spellings of a
    null pbo, the address taken with the reference's type, the programs of the repo that include the veneer, and that both
    flavours link against the library."""
    import subprocess
    body = ('#include "%s"\n'
            'int main(int argc, char **) {\n'
            '    if (argc > 1000) {\n'
            '        pathtrace(NULL, 0, 1); pathtrace(nullptr, 0, 2); pathtrace(0, 0, 3);\n'
            '        sendToGPU(NULL, 1); sendToGPU(nullptr, 2);\n'
            '        %s\n'
            '    }\n'
            '    return pathtraceHandle() ? 1 : 0;\n}\n') % (os.path.join(ROOT, "mygpuraytracer_amd", "csrc", "pathtrace_api.h"), "%s")
    hip = tmp_path / "with_hip.cpp"
    hip.write_text("#include <hip/hip_runtime_api.h>\n#include <cstddef>\n" + body %
                   "void (*f)(uchar4 *, int, int) = &pathtrace; void (*g)(uchar4 *, int) = &sendToGPU; uchar4 *p = nullptr; f(p, 0, 4); g(p, 4); pathtrace(p, 0, 5);")
    plain = tmp_path / "without_hip.cpp"
    plain.write_text("#include <cstddef>\n" + body % "void (*f)(void *, int, int) = &pathtrace; f(nullptr, 0, 4);")
    libdir = os.path.dirname(product.LIB_PATH)
    link = ["-L" + libdir, "-lmi355x_pathtracer", "-Wl,-rpath," + libdir]
    for src, extra in ((hip, ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]), (plain, [])):
        exe = tmp_path / (src.stem + ".exe")
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-o", str(exe), str(src)] + extra + link)
        assert subprocess.call([str(exe)]) == 0          # (no tracer exists: pathtraceHandle() is NULL; nothing touches a GPU)
    for prog in ("tests/veneer_check.cpp", "tools/gpu_dropin_loop_cpp.cpp"):
        subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, prog)])


REF_MAIN = "/root/reference/src/main.cpp"
REF_GLM = "/root/reference/external/include"


@pytest.mark.skipif(not (os.path.exists(REF_MAIN) and os.path.isdir(REF_GLM)), reason="needs the reference's main.cpp and vendored glm (dev container only)")
def test_veneer_compiles_main_cpp_camera_block(tmp_path):
    """The data members of Scene / RenderState / Camera that the reference's caller touches compile against the veneer AS WRITTEN: the
    camera set-up of main() (src/main.cpp:50-70), runCuda()'s recompute (:105-123), the recentering key (:166-168), the middle-drag pan
    (:195-206) and saveImage()'s pixel read (:89-90) are read from the reference's file at test time (text, never stored here), wrapped
    in functions over main.cpp's own globals (:15-28), and compiled with the reference's vendored glm on the include path -- then RUN
    against a stub of the struct (no library, no GPU): the camera block must compute what glm computes on plain glm members."""
    import subprocess
    lines = open(REF_MAIN).read().splitlines()
    blk = lambda a, b: "\n".join(lines[a - 1:b])
    src = tmp_path / "main_camera.cpp"
    src.write_text('''#include <cmath>
#include <cstdio>
#include <glm/glm.hpp>
#include "pathtrace_api.h"
#define PI 3.14159265358979323846f
static bool camchanged = true;
float zoom, theta, phi;
glm::vec3 cameraPosition;
glm::vec3 ogLookAt;
RenderState *renderState;
static RenderState theState;
int iteration;
int width;
int height;
static double lastX = 3.0, lastY = 5.0;
struct SceneStub { RenderState &state; } sceneStub{theState}, *scene = &sceneStub;
void camera_setup() {
%s
}
void runCuda_camera() {
%s
}
void recenter() {
%s
}
void middle_drag(double xpos, double ypos) {
%s
}
glm::vec3 save_pixel(int index, float samples) {
%s
    return glm::vec3(pix) / samples;
}
int main() {
    Camera &c = theState.camera;
    c.resolution = mi355x::ivec2(800, 600);
    c.position = glm::vec3(0.f, 5.f, 10.5f); c.lookAt = glm::vec3(0.f, 5.f, 0.f); c.view = glm::vec3(0.f, 0.f, -1.f); c.up = glm::vec3(0.f, 1.f, 0.f);
    theState.image.assign(4, mi355x::vec3(1.f, 2.f, 3.f));
    camera_setup();
    runCuda_camera();
    middle_drag(4.0, 7.0);
    runCuda_camera();
    // the same on plain glm members
    glm::vec3 view(0.f, 0.f, -1.f), lookAt(0.f, 5.f, 0.f), position(0.f, 5.f, 10.5f);
    float phi2 = glm::acos(glm::dot(glm::normalize(glm::vec3(view.x, 0.f, view.z)), glm::vec3(0, 0, -1)));
    float theta2 = glm::acos(glm::dot(glm::normalize(glm::vec3(0.f, view.y, view.z)), glm::vec3(0, 1, 0)));
    float zoom2 = glm::length(position - lookAt);
    auto recompute = [&](glm::vec3 &v, glm::vec3 &u2, glm::vec3 &r, glm::vec3 &pos) {
        glm::vec3 cp(zoom2 * sin(phi2) * sin(theta2), zoom2 * cos(theta2), zoom2 * cos(phi2) * sin(theta2));
        v = -glm::normalize(cp); r = glm::cross(v, glm::vec3(0, 1, 0)); u2 = glm::cross(r, v); cp += lookAt; pos = cp;
    };
    glm::vec3 v, u2, r, pos;
    recompute(v, u2, r, pos);
    glm::vec3 fw = v; fw.y = 0.f; fw = glm::normalize(fw);
    glm::vec3 rt = r; rt.y = 0.f; rt = glm::normalize(rt);
    lookAt -= (float)(4.0 - 3.0) * rt * 0.01f;
    lookAt += (float)(7.0 - 5.0) * fw * 0.01f;
    recompute(v, u2, r, pos);
    auto same = [](const mi355x::vec3 &a, const glm::vec3 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; };
    const bool ok = same(c.view, v) && same(c.up, u2) && same(c.right, r) && same(c.position, pos) && same(c.lookAt, lookAt) &&
                    width == 800 && height == 600 && phi == phi2 && theta == theta2 && zoom == zoom2;
    recenter();
    const glm::vec3 px = save_pixel(2, 2.f);
    const ptx_camera *abi = c.c_abi();
    const bool ok2 = same(c.lookAt, glm::vec3(0.f, 5.f, 0.f)) && px.x == 0.5f && px.z == 1.5f && abi->resolution[0] == 800 && abi->view[2] == c.view.z &&
                     abi->position[1] == c.position.y;
    printf("%%d %%d\\n", ok ? 1 : 0, ok2 ? 1 : 0);
    return ok && ok2 ? 0 : 1;
}
''' % (blk(50, 70), blk(105, 123), blk(166, 168), blk(195, 206), blk(89, 89)))
    exe = tmp_path / "main_camera"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-variable", "-I", REF_GLM, "-I", os.path.join(ROOT, "mygpuraytracer_amd", "csrc"),
                           "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.check_output([str(exe)], text=True).split() == ["1", "1"]


def test_veneer_performance_timer_is_the_whole_class(product, tmp_path):
    """csrc/pathtrace_api.h: PerformanceTimer has the reference's interface (src/timer.h:17-100) -- start/endCpuTimer measure real
    time, the misuse exceptions are the reference's, it cannot be copied -- on the CPU side; the GPU side runs in the GPU tier."""
    import subprocess
    src = tmp_path / "timer.cpp"
    src.write_text('''#include <chrono>
#include <cstdio>
#include <stdexcept>
#include <thread>
#include <type_traits>
#include "pathtrace_api.h"
int main() {
    static_assert(!std::is_copy_constructible<PerformanceTimer>::value && !std::is_move_assignable<PerformanceTimer>::value, "uncopyable, unmovable");
    PerformanceTimer t;
    int caught = 0;
    try { t.endCpuTimer(); } catch (const std::runtime_error &) { caught++; }
    t.startCpuTimer();
    try { t.startCpuTimer(); } catch (const std::runtime_error &) { caught++; }
    std::this_thread::sleep_for(std::chrono::milliseconds(30));
    t.endCpuTimer();
    const float ms = t.getCpuElapsedTimeForPreviousOperation();
    try { t.endGpuTimer(); } catch (const std::runtime_error &) { caught++; }
    printf("%d %d %d\\n", caught, ms >= 29.f && ms < 2000.f ? 1 : 0, timer().getGpuElapsedTimeForPreviousOperation() == 0.f ? 1 : 0);
    return 0;
}
''')
    exe = tmp_path / "timer"
    libdir = os.path.dirname(product.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "mygpuraytracer_amd", "csrc"), str(src), "-o", str(exe),
                           "-L", libdir, "-lmi355x_pathtracer", "-Wl,-rpath," + libdir])
    assert subprocess.check_output([str(exe)], text=True).split() == ["3", "1", "1"]
