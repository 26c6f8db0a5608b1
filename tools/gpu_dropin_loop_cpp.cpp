// gpu_dropin_loop_cpp.cpp -- cost of the reference's own call loop through the C++ veneer on C4 (1920x1080, depth 8):
// pathtraceInit, then N x pathtrace(pbo, 0, iter), each returning with scene->state.image valid (src/pathtrace.cu:555-556).
//   build: hipcc -O2 -o gpu_dropin_loop_cpp tools/gpu_dropin_loop_cpp.cpp -Lmygpuraytracer_amd -lmi355x_pathtracer -Wl,-rpath,$PWD/mygpuraytracer_amd
//   run:   gpu_dropin_loop_cpp scenes/cornellObj.txt [N] [nopin]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "../mygpuraytracer_amd/csrc/pathtrace_api.h"
int main(int argc, char **argv) {
    const int N = argc > 2 ? atoi(argv[2]) : 120;
    Scene *scene = new Scene(argv[1]);
    scene->setResolution(1920, 1080);
    scene->state.traceDepth = 8;
    scene->applyRunCudaCamera();
    pathtraceFree();
    pathtraceInit(scene);
    if (argc > 3 && std::string(argv[3]) == "nopin") ptx_unpin_host_buffer(scene->state.image.data());
    uchar4 *pbo = nullptr;
    if (hipMalloc((void **)&pbo, (size_t)1920 * 1080 * 4) != hipSuccess) return 1;
    for (int it = 1; it <= 40; it++) pathtrace(nullptr, 0, it);
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 41; it <= 40 + N; it++) pathtrace(nullptr, 0, it);
    auto t1 = std::chrono::steady_clock::now();
    for (int it = 41 + N; it <= 40 + 2 * N; it++) pathtrace(pbo, 0, it);
    auto t2 = std::chrono::steady_clock::now();
    double a = std::chrono::duration<double, std::milli>(t1 - t0).count() / N, b = std::chrono::duration<double, std::milli>(t2 - t1).count() / N;
    printf("{\"ms_per_pathtrace_call_with_frame_readback\": %.3f, \"with_preview_too\": %.3f, \"pinned\": %s, \"checksum\": %.6g}\n", a, b,
           (argc > 3 && std::string(argv[3]) == "nopin") ? "false" : "true", (double)scene->state.image[1000].x + (double)scene->state.image[(size_t)1920 * 540 + 960].y);
    pathtraceFree();
    return 0;
}
