// main_headless.cpp -- the reference's main.cpp/preview.cpp without the window (src/main.cpp:36-152):
// load the scene, recompute the camera as the first runCuda() does, pathtraceFree/pathtraceInit, run
// state.iterations iterations of pathtrace, print "time: <ms>" (sum of the bounce-loop timer, main.cpp:141-146) and
// save <FILE>.<utc>.<n>samp.png with saveImage's mirroring (main.cpp:81-102).
//
//   mi355x_pathtrace SCENEFILE.txt [--res W H] [--depth D] [--iterations N] [--out PREFIX] [--pfm] [--hdr]
//                                  [--no-aa] [--dof] [--no-sort] [--no-cache] [--device K] [--arith 0|1|2]   (ptx_options.arith: 0 exact, the default)
//                                  [--checkpoint FILE [--checkpoint-every N]] [--resume FILE]
//                                  [--orbit "left:DX,DY;right:DY;middle:DX,DY;space"]   (the mouse of main.cpp:166-212, scripted)
//                                  [--per-call [--no-render-ahead]]   (one pathtrace(pbo, frame, iter) per iteration with the frame read
//                                                                      back after each, exactly the reference's runCuda loop)
//
// RES / DEPTH / ITERATIONS overrides and the four switches are what the reference can only change by editing the
// scene file or the #defines of src/pathtrace.cu:36-40.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <sstream>
#include <algorithm>
#include <string>
#include <vector>

#include "pathtrace_api.h"
#include "pt_image.h"

static std::string currentTimeString() {          // src/preview.cpp:13-19
    time_t now;
    time(&now);
    char buf[sizeof "0000-00-00_00-00-00z"];
    strftime(buf, sizeof buf, "%Y-%m-%d_%H-%M-%Sz", gmtime(&now));
    return std::string(buf);
}

int main(int argc, char **argv) {
    const std::string startTimeString = currentTimeString();
    if (argc < 2) {
        printf("Usage: %s SCENEFILE.txt [--res W H] [--depth D] [--iterations N] [--out PREFIX] [--pfm] [--hdr] [--no-aa] [--dof] [--no-sort] [--no-cache] [--device K] [--arith 0|1|2] [--checkpoint FILE [--checkpoint-every N]] [--resume FILE] [--orbit SCRIPT] [--per-call [--no-render-ahead]]\n", argv[0]);
        return 1;
    }
    int resw = 0, resh = 0, depth = 0, iterations = 0;
    bool pfm = false, hdr = false, per_call = false;
    std::string out_prefix, ckpt_path, resume_path, orbit_script;
    int ckpt_every = 0;
    ptx_options &opt = pathtraceOptions();
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { fprintf(stderr, "%s needs %d value(s)\n", a.c_str(), n); exit(1); } };
        if (a == "--res") { need(2); resw = atoi(argv[++i]); resh = atoi(argv[++i]); }
        else if (a == "--depth") { need(1); depth = atoi(argv[++i]); }
        else if (a == "--iterations") { need(1); iterations = atoi(argv[++i]); }
        else if (a == "--out") { need(1); out_prefix = argv[++i]; }
        else if (a == "--device") { need(1); opt.device = atoi(argv[++i]); }
        else if (a == "--arith") { need(1); opt.arith = atoi(argv[++i]); }
        else if (a == "--pfm") pfm = true;
        else if (a == "--hdr") hdr = true;
        else if (a == "--checkpoint") { need(1); ckpt_path = argv[++i]; }
        else if (a == "--checkpoint-every") { need(1); ckpt_every = atoi(argv[++i]); }
        else if (a == "--resume") { need(1); resume_path = argv[++i]; }
        else if (a == "--orbit") { need(1); orbit_script = argv[++i]; }
        else if (a == "--no-aa") opt.antialiasing = 0;
        else if (a == "--dof") opt.depth_of_field = 1;
        else if (a == "--no-sort") opt.sort_by_material = 0;
        else if (a == "--no-cache") opt.cache_first_bounce = 0;
        else if (a == "--per-call") per_call = true;
        else if (a == "--no-render-ahead") pathtraceRenderAhead() = false;
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
    }
    Scene *scene = nullptr;
    try {
        scene = new Scene(argv[1]);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    if (resw > 0 && resh > 0) scene->setResolution(resw, resh);
    if (depth > 0) scene->state.traceDepth = depth;
    if (iterations > 0) scene->state.iterations = (unsigned)iterations;
    if (orbit_script.empty()) scene->applyRunCudaCamera();
    else if (!scene->runOrbitScript(orbit_script)) { fprintf(stderr, "bad --orbit script: %s\n", orbit_script.c_str()); return 1; }
    const int width = scene->state.camera.resolution[0], height = scene->state.camera.resolution[1];

    pathtraceFree();
    pathtraceInit(scene);
    ptx_tracer *t = pathtraceHandle();
    const int n = (int)scene->state.iterations;
    int done = 0;
    if (!resume_path.empty()) {                 // continue a render: the buffer and the iteration count are all the state
        long long it = 0;
        std::string why;
        if (!ptimg::read_checkpoint(resume_path, width, height, it, &scene->state.image[0].x, why)) { fprintf(stderr, "%s\n", why.c_str()); return 1; }
        if (ptx_write_image(t, &scene->state.image[0].x) != PTX_OK) { fprintf(stderr, "resume failed: %s\n", ptx_last_error()); return 1; }
        done = (int)std::min<long long>(it, n);
        printf("Resumed %s at %d of %d samples.\n", resume_path.c_str(), done, n);
    }
    const int rendered = std::max(n - done, 0);
    const int chunk = (!ckpt_path.empty() && ckpt_every > 0) ? ckpt_every : n;
    double per_call_ms = 0.0;
    if (per_call) {
        // the reference's own loop (runCuda, src/main.cpp:128-148): one pathtrace(pbo, frame, iteration) per frame, the fp32 frame
        // in state.image after each, the timer summed per call.  No window, so no PBO.
        for (int it = done + 1; it <= n; it++) {
            pathtrace(nullptr, 0, it);
            per_call_ms += timer().getGpuElapsedTimeForPreviousOperation();
        }
        done = n;
    }
    while (done < n) {
        const int count = std::min(chunk, n - done);
        if (ptx_render(t, done + 1, count) != PTX_OK || ptx_read_image(t, &scene->state.image[0].x) != PTX_OK) {
            fprintf(stderr, "render failed: %s\n", ptx_last_error());
            return 1;
        }
        done += count;
        if (!ckpt_path.empty()) {
            if (!ptimg::write_checkpoint(ckpt_path, width, height, done, &scene->state.image[0].x)) { fprintf(stderr, "cannot write %s\n", ckpt_path.c_str()); return 1; }
        }
    }
    ptx_stats st;
    ptx_get_stats(t, &st);
    printf("time: %g\n", per_call ? per_call_ms : st.loop_ms_total);                    // main.cpp:146
    if (rendered > 0)
        printf("%d x %d, depth %d, %d samples (%d traced now): %.3f ms/iteration, %.1f Mrays/s\n", width, height, scene->state.traceDepth, n,
               rendered, st.loop_ms_total / rendered, st.rays_total / (st.loop_ms_total * 1e-3) / 1e6);

    std::ostringstream ss;                                                              // saveImage, main.cpp:94-97
    ss << (out_prefix.empty() ? scene->state.imageName : out_prefix) << "." << startTimeString << "." << n << "samp";
    std::vector<uint8_t> rgb8;
    ptimg::to_rgb8_mirrored(width, height, &scene->state.image[0].x, (float)n, rgb8);
    if (!ptimg::write_png_rgb8(ss.str() + ".png", width, height, rgb8.data())) { fprintf(stderr, "cannot write %s.png\n", ss.str().c_str()); return 1; }
    printf("Saved %s.png.\n", ss.str().c_str());
    if (pfm) { ptimg::write_pfm(ss.str() + ".pfm", width, height, &scene->state.image[0].x, (float)n); printf("Saved %s.pfm.\n", ss.str().c_str()); }
    if (hdr) {                                                                          // img.saveHDR, main.cpp:101
        std::vector<float> mean;
        ptimg::to_mean_mirrored(width, height, &scene->state.image[0].x, (float)n, mean);
        if (!ptimg::write_hdr(ss.str() + ".hdr", width, height, mean.data())) { fprintf(stderr, "cannot write %s.hdr\n", ss.str().c_str()); return 1; }
        printf("Saved %s.hdr.\n", ss.str().c_str());
    }
    pathtraceFree();
    delete scene;
    return 0;
}
