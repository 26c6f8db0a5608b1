// pathtrace_api.cpp -- the reference's four free functions (src/pathtrace.h:6-9) over the C ABI.
// Keeps the reference's module-static, one-scene-per-process contract (src/pathtrace.cu:91-98) and its
// print-and-exit error policy (checkCUDAError, src/pathtrace.cu:42-60); the C ABI underneath returns codes.
#include "pathtrace_api.h"

#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace {
Scene *hst_scene = nullptr;          // borrowed, must outlive pathtraceFree (src/pathtrace.cu:91,102)
ptx_tracer *g_tracer = nullptr;
ptx_multi *g_multi = nullptr;        // != NULL: several devices (pathtraceDevices()); g_tracer is then device 0's tracer
bool g_albedo_read = false;          // several devices, apps variant: state.albedo holds the merged AOV of iteration 1
void *g_pinned[3] = {nullptr, nullptr, nullptr};      // state.image / state.albedo page-locked for the per-iteration read-back
ptx_options g_options;
bool g_options_init = false;

void check(int rc, const char *what) {
    if (rc == PTX_OK) return;
    fprintf(stderr, "mi355x pathtracer error (%s): %s\n", what, ptx_last_error());
    exit(EXIT_FAILURE);
}
}  // namespace

Scene::Scene(const std::string &filename, const std::string &base_dir) : impl_(nullptr) {
    int rc = ptx_scene_load(filename.c_str(), base_dir.empty() ? nullptr : base_dir.c_str(), &impl_);
    if (rc != PTX_OK) throw std::runtime_error(std::string("Scene: ") + ptx_last_error());
    int ng = ptx_scene_num_geoms(impl_), nm = ptx_scene_num_materials(impl_);
    if (ng) geoms.assign(ptx_scene_geoms(impl_), ptx_scene_geoms(impl_) + ng);
    if (nm) materials.assign(ptx_scene_materials(impl_), ptx_scene_materials(impl_) + nm);
    state.camera = *ptx_scene_camera(impl_);
    state.iterations = (unsigned)ptx_scene_iterations(impl_);
    state.traceDepth = ptx_scene_trace_depth(impl_);
    state.imageName = ptx_scene_image_name(impl_);
    state.image.assign((size_t)state.camera.resolution[0] * state.camera.resolution[1], mi355x::vec3{0.f, 0.f, 0.f});
    state.albedo = state.image;            // apps/src/scene.cpp:380-383
    state.output = state.image;
}

Scene::~Scene() { ptx_scene_free(impl_); }

void Scene::applyRunCudaCamera() {
    *ptx_scene_camera(impl_) = *state.camera.c_abi();
    ptx_scene_apply_runcuda_camera(impl_);
    state.camera = *ptx_scene_camera(impl_);
}

bool Scene::runOrbitScript(const std::string &script) {
    *ptx_scene_camera(impl_) = *state.camera.c_abi();
    ptx_orbit o;
    ptx_orbit_init(impl_, &o);
    ptx_orbit_apply(impl_, &o);
    const int w = state.camera.resolution[0], h = state.camera.resolution[1];
    size_t pos = 0;
    bool ok = true;
    while (pos <= script.size() && ok) {
        size_t end = script.find(';', pos);
        if (end == std::string::npos) end = script.size();
        std::string ev = script.substr(pos, end - pos);
        pos = end + 1;
        while (!ev.empty() && ev.front() == ' ') ev.erase(ev.begin());
        while (!ev.empty() && ev.back() == ' ') ev.pop_back();
        if (ev.empty()) continue;
        double a = 0.0, b = 0.0;
        if (ev == "space") ptx_orbit_recenter(impl_, &o);
        else if (sscanf(ev.c_str(), "left:%lf,%lf", &a, &b) == 2) ptx_orbit_left_drag(&o, a, b, w, h);
        else if (sscanf(ev.c_str(), "middle:%lf,%lf", &a, &b) == 2) ptx_orbit_middle_drag(impl_, a, b);
        else if (sscanf(ev.c_str(), "right:%lf", &a) == 1) ptx_orbit_right_drag(&o, a, h);
        else ok = false;
        if (ok) ptx_orbit_apply(impl_, &o);
    }
    state.camera = *ptx_scene_camera(impl_);
    return ok;
}

void Scene::setResolution(int w, int h) {
    ptx_scene_set_resolution(impl_, w, h);
    state.camera = *ptx_scene_camera(impl_);
    state.image.assign((size_t)w * h, mi355x::vec3{0.f, 0.f, 0.f});
    state.albedo = state.image;
    state.output = state.image;
}

ptx_options &pathtraceOptions() {
    if (!g_options_init) { ptx_default_options(&g_options); g_options_init = true; }
    return g_options;
}

bool &pathtraceRenderAhead() {
    static bool on = true;
    return on;
}

PerformanceTimer &timer() {
    static PerformanceTimer t;
    return t;
}

std::vector<int> &pathtraceDevices() {
    static std::vector<int> devices;
    return devices;
}

// ---- PerformanceTimer, src/timer.h:17-100 ---------------------------------------------------------------------------------
// (events are created on first use: the instance behind timer() is a function-local static that may be constructed before the
// process has picked a device)
PerformanceTimer::PerformanceTimer() {}
PerformanceTimer::~PerformanceTimer() {
    if (event_start) (void)hipEventDestroy((hipEvent_t)event_start);
    if (event_end) (void)hipEventDestroy((hipEvent_t)event_end);
}

void PerformanceTimer::startCpuTimer() {
    if (cpu_timer_started) throw std::runtime_error("CPU timer already started");
    cpu_timer_started = true;
    time_start_cpu_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::high_resolution_clock::now().time_since_epoch()).count();
}

void PerformanceTimer::endCpuTimer() {
    const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::high_resolution_clock::now().time_since_epoch()).count();
    if (!cpu_timer_started) throw std::runtime_error("CPU timer not started");
    prev_elapsed_time_cpu_milliseconds = (float)((double)(now - time_start_cpu_ns) * 1e-6);
    cpu_timer_started = false;
}

// The reference records on the null stream, which every one of ITS launches runs on.  Here the work runs on the tracer's own
// (non-blocking) stream, so that is where the events go while a tracer exists: the interval then covers the pathtrace() calls
// between start and end, as it does in the reference.
static hipStream_t timer_stream() { return g_tracer ? (hipStream_t)ptx_stream(g_tracer) : (hipStream_t) nullptr; }

void PerformanceTimer::startGpuTimer() {
    if (gpu_timer_started) throw std::runtime_error("GPU timer already started");
    gpu_timer_started = true;
    if (!event_start) { hipEvent_t a = nullptr, b = nullptr; (void)hipEventCreate(&a); (void)hipEventCreate(&b); event_start = a; event_end = b; }
    if (event_start) (void)hipEventRecord((hipEvent_t)event_start, timer_stream());
}

void PerformanceTimer::endGpuTimer() {
    if (event_end) { (void)hipEventRecord((hipEvent_t)event_end, timer_stream()); (void)hipEventSynchronize((hipEvent_t)event_end); }
    if (!gpu_timer_started) throw std::runtime_error("GPU timer not started");
    float ms = 0.f;
    if (event_start && event_end && hipEventElapsedTime(&ms, (hipEvent_t)event_start, (hipEvent_t)event_end) != hipSuccess) { ms = 0.f; (void)hipGetLastError(); }
    prev_elapsed_time_gpu_milliseconds = ms;
    gpu_timer_started = false;
    gpu_interval_is_callers = true;
}

// pathtrace() ran: the module's timer answers with ITS bounce loop again, as after the reference's own start/endGpuTimer pair
// inside pathtrace (src/pathtrace.cu:489, 545)
void mi355x_timer_note_pathtrace(PerformanceTimer &t) { t.gpu_interval_is_callers = false; }

float PerformanceTimer::getGpuElapsedTimeForPreviousOperation() {
    if (gpu_interval_is_callers || this != &timer()) return prev_elapsed_time_gpu_milliseconds;
    if (!g_tracer) return 0.f;
    if (!g_multi) return (float)ptx_last_loop_ms(g_tracer);
    double ms = 0.0;                 // several devices work side by side: the slowest one's bounce loop
    for (int i = 0; i < ptx_multi_device_count(g_multi); i++) ms = std::max(ms, ptx_last_loop_ms(ptx_multi_tracer(g_multi, i)));
    return (float)ms;
}

void pathtraceInit(Scene *scene) {
    hst_scene = scene;
    g_albedo_read = false;
    const std::vector<int> &devs = pathtraceDevices();
    if (devs.size() > 1) {
        // the C ABI's multi-device layer takes the loaded scene: hand it the caller's camera and depth first
        *ptx_scene_camera(scene->handle()) = *scene->state.camera.c_abi();
        ptx_scene_set_trace_depth(scene->handle(), scene->state.traceDepth);
        check(ptx_multi_create(scene->handle(), &pathtraceOptions(), devs.data(), (int)devs.size(), 0, &g_multi), "pathtraceInit");
        g_tracer = ptx_multi_tracer(g_multi, 0);
        check(ptx_multi_set_render_ahead(g_multi, pathtraceRenderAhead() ? 1 : 0), "pathtraceInit");
    } else {
        ptx_options o = pathtraceOptions();
        if (devs.size() == 1) o.device = devs[0];
        check(ptx_create((int)scene->geoms.size(), scene->geoms.data(), (int)scene->materials.size(), scene->materials.data(),
                         scene->state.camera.c_abi(), scene->state.traceDepth, &o, nullptr, nullptr, &g_tracer),
              "pathtraceInit");
        check(ptx_set_render_ahead(g_tracer, pathtraceRenderAhead() ? 1 : 0), "pathtraceInit");
    }
    // the reference copies the whole fp32 frame into scene->state.image after every iteration (src/pathtrace.cu:555-556):
    // page-lock the destination once, so that each of those copies is one DMA at PCIe rate (not fatal if it cannot be)
    const size_t bytes = scene->state.image.size() * sizeof(mi355x::vec3);
    if (bytes && ptx_pin_host_buffer(scene->state.image.data(), bytes) == PTX_OK) g_pinned[0] = scene->state.image.data();
    if (pathtraceOptions().apps_variant && bytes && scene->state.albedo.size() == scene->state.image.size() &&
        ptx_pin_host_buffer(scene->state.albedo.data(), bytes) == PTX_OK) g_pinned[1] = scene->state.albedo.data();
}

void pathtraceFree() {          // safe before init and idempotent, as main.cpp:129 relies on
    for (void *&p : g_pinned) { if (p) ptx_unpin_host_buffer(p); p = nullptr; }
    if (g_multi) ptx_multi_destroy(g_multi);
    else ptx_destroy(g_tracer);
    g_multi = nullptr;
    g_tracer = nullptr;
}

void mi355x::pathtrace_raw(void *pbo, int frame, int iter) {
    (void)frame;                 // unused by the reference as well
    if (!g_tracer || !hst_scene) { fprintf(stderr, "pathtrace called before pathtraceInit\n"); exit(EXIT_FAILURE); }
    mi355x_timer_note_pathtrace(timer());
    // the reference re-reads camera and traceDepth on every call (src/pathtrace.cu:434-436)
    const bool apps = pathtraceOptions().apps_variant != 0;
    if (g_multi) {               // several devices: every device its tile, then the row blocks into device 0's frame
        check(ptx_multi_set_camera(g_multi, hst_scene->state.camera.c_abi(), hst_scene->state.traceDepth), "pathtrace camera");
        check(ptx_multi_iterate(g_multi, iter), "pathtrace");
        check(ptx_multi_read_image(g_multi, &hst_scene->state.image[0].x), "image readback");      // assemble + :555-556
        if (!apps) check(ptx_write_pbo_device(g_tracer, iter, pbo), "sendImageToPBO");             // from the assembled frame
        // the AOV is written by iteration 1 alone (apps/src/pathtrace.cu:412-462): merged from the devices once, when it is new,
        // not n full-frame reads and a host merge per call
        if (apps && (iter == 1 || !g_albedo_read)) {
            check(ptx_multi_read_albedo(g_multi, &hst_scene->state.albedo[0].x), "albedo readback");
            g_albedo_read = true;
        }
        check(ptx_synchronize(g_tracer), "pathtrace");
        return;
    }
    check(ptx_set_camera(g_tracer, hst_scene->state.camera.c_abi(), hst_scene->state.traceDepth), "pathtrace camera");
    check(ptx_iterate(g_tracer, iter), "pathtrace");
    // apps/src builds with AI_DENOISE: no preview from here (sendToGPU shows the denoised frame), the albedo AOV comes back
    // with the image (apps/src/pathtrace.cu:658-669)
    if (!apps) check(ptx_write_pbo_device(g_tracer, iter, pbo), "sendImageToPBO");
    check(ptx_read_image(g_tracer, &hst_scene->state.image[0].x), "image readback");     // :555-556
    if (apps) check(ptx_read_albedo(g_tracer, &hst_scene->state.albedo[0].x), "albedo readback");
}

// apps/src/pathtrace.cu:673-685: the denoised frame in state.output -> the pbo, scaled by 255 and clamped, no division by iter
void mi355x::sendToGPU_raw(void *pbo, int iter) {
    (void)iter;                  // passed to the kernel but unused there as well (apps/src/pathtrace.cu:96-116)
    if (!g_tracer || !hst_scene) { fprintf(stderr, "sendToGPU called before pathtraceInit\n"); exit(EXIT_FAILURE); }
    check(ptx_write_denoised_pbo_device(g_tracer, &hst_scene->state.output[0].x, pbo), "sendToGPU");
}

// the same two under the reference's names, for translation units without HIP's vector types (see pathtrace_api.h)
void pathtrace(void *pbo, int frame, int iter) { mi355x::pathtrace_raw(pbo, frame, iter); }
void sendToGPU(void *pbo, int iter) { mi355x::sendToGPU_raw(pbo, iter); }

ptx_tracer *pathtraceHandle() { return g_tracer; }
