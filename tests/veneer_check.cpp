// veneer_check.cpp -- drives the C++ veneer (mygpuraytracer_amd/csrc/pathtrace_api.h) exactly as the reference's main.cpp
// drives its pathtrace.h: Scene, pathtraceFree/Init, pathtrace(pbo, frame, iter) per iteration with a DEVICE pbo, timer(),
// and, with `apps`, the apps/src extras (state.albedo, sendToGPU).  Writes what it saw to OUT.{image,albedo,pbo,pbo2} for
// tests/test_gpu_parity.py to compare with the C ABI driven from Python.  Built by the test with g++.
//   veneer_check SCENE W H DEPTH ITERS OUT [apps] [devices=0,0,...]
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../mygpuraytracer_amd/csrc/pathtrace_api.h"

static void dump(const std::string &path, const void *p, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, 1, n, f) != n) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(1); }
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 7) return 2;
    const int w = atoi(argv[2]), h = atoi(argv[3]), depth = atoi(argv[4]), iters = atoi(argv[5]);
    const std::string out = argv[6];
    bool apps = false;
    for (int a = 7; a < argc; a++) {
        const std::string arg = argv[a];
        if (arg == "apps") apps = true;
        else if (arg.rfind("devices=", 0) == 0) {          // several devices (an ordinal may repeat): the frame is split into row tiles
            pathtraceDevices().clear();
            for (size_t p = 8; p < arg.size();) {
                size_t e = arg.find(',', p);
                if (e == std::string::npos) e = arg.size();
                pathtraceDevices().push_back(atoi(arg.substr(p, e - p).c_str()));
                p = e + 1;
            }
        }
    }
    Scene *scene = new Scene(argv[1]);
    scene->setResolution(w, h);
    scene->state.traceDepth = depth;
    scene->applyRunCudaCamera();
    pathtraceOptions().apps_variant = apps ? 1 : 0;
    pathtraceFree();                                   // main.cpp:129: Free before the first Init must be harmless
    pathtraceInit(scene);
    const size_t n = (size_t)w * h;
    uchar4 *pbo = nullptr;                             // HIP's own uchar4, as in a maintainer's main.cpp
    if (hipMalloc((void **)&pbo, n * 4) != hipSuccess || hipMemset(pbo, 0x5a, n * 4) != hipSuccess) { fprintf(stderr, "no device pbo\n"); return 1; }
    // the reference's exact signatures exist where uchar4 is known (src/pathtrace.h:9, apps/src/pathtrace.h:10)
    void (*ref_pathtrace)(uchar4 *, int, int) = &pathtrace;
    void (*ref_send)(uchar4 *, int) = &sendToGPU;
    float sum = 0.f;
    // src/timer.h's whole interface: a caller's own PerformanceTimer around one call (GPU and CPU halves), its misuse exceptions, and the
    // module timer still answering with pathtrace's own bounce loop afterwards
    PerformanceTimer mine;
    float mine_gpu = -1.f, mine_cpu = -1.f, after_mine = -1.f;
    int caught = 0;
    for (int it = 1; it <= iters; it++) {
        const bool timed_by_caller = it == iters - 1;
        if (timed_by_caller) {
            try { mine.endGpuTimer(); } catch (const std::runtime_error &) { caught++; }
            mine.startGpuTimer(); mine.startCpuTimer();
            try { mine.startGpuTimer(); } catch (const std::runtime_error &) { caught++; }
        }
        // a literal null pbo in each of its spellings must compile as it does against src/pathtrace.h:9 (ONE function named
        // pathtrace in this translation unit) and means "no preview"; the last call writes the preview the test compares
        if (it == 2 && iters > 2) pathtrace(NULL, 0, it);
        else if (it == 3 && iters > 3) pathtrace(nullptr, 0, it);
        else if (it == 4 && iters > 4) pathtrace(0, 0, it);
        else if (it & 1) ref_pathtrace(pbo, 0, it);
        else pathtrace(pbo, 0, it);
        if (timed_by_caller) {
            mine.endGpuTimer(); mine.endCpuTimer();
            mine_gpu = mine.getGpuElapsedTimeForPreviousOperation(); mine_cpu = mine.getCpuElapsedTimeForPreviousOperation();
            after_mine = timer().getGpuElapsedTimeForPreviousOperation();
        }
        sum += timer().getGpuElapsedTimeForPreviousOperation();
    }
    if (iters >= 2) printf("timers: caught %d mine_gpu %g mine_cpu %g module %g\n", caught, mine_gpu, mine_cpu, after_mine);
    std::vector<unsigned char> host(n * 4);
    if (hipMemcpy(host.data(), pbo, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    dump(out + ".image", scene->state.image.data(), n * 12);
    dump(out + ".pbo", host.data(), n * 4);            // apps: untouched (AI_DENOISE skips sendImageToPBO), else the preview of `iters`
    if (apps) {
        dump(out + ".albedo", scene->state.albedo.data(), n * 12);
        for (size_t i = 0; i < n; i++) {               // stand-in for the denoiser: mean radiance, partly out of range on purpose
            scene->state.output[i].x = scene->state.image[i].x / iters * 1.5f - 0.1f;
            scene->state.output[i].y = scene->state.image[i].y / iters;
            scene->state.output[i].z = scene->state.image[i].z / iters * 3.0f;
        }
        sendToGPU(NULL, iters);                        // NULL pbo: skipped, as pathtrace's is
        ref_send(pbo, iters);
        if (hipMemcpy(host.data(), pbo, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        dump(out + ".pbo2", host.data(), n * 4);
    }
    printf("time: %g\n", sum);
    (void)ref_send;
    (void)hipFree(pbo);
    pathtraceFree();
    pathtraceFree();                                   // idempotent
    delete scene;
    return sum > 0.f ? 0 : 3;
}
