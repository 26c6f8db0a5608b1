// pt_multi.cpp -- N GPUs of one node behind one handle, one process: the C/C++ side of the row-tile split.
//
// The reference is single-GPU (src/preview.cpp:107 cudaGLSetGLDevice(0)); a caller shaped like its main loop
// (src/main.cpp:128-148: Free, Init, pathtrace(iter) per frame, Free) gets N devices through this layer without touching
// Python or torch.distributed.  Device i of n traces the interleaved row blocks (y / tile_rows) % n == i of the frame as a
// stream of its own (ptx_options.tile_*), with its own streams and buffers; nothing is exchanged while tracing.  One
// exchange per read: the row blocks a device owns are copied into device[0]'s frame with one strided peer copy per device
// (hipMemcpy2DAsync over xGMI: rows = blocks, pitch = n blocks; block-wise hipMemcpyPeerAsync for a device on which peer
// access to device[0] could not be enabled) -- the gather SURVEY 8(e) names; foreign rows of every
// device's own buffer stay zero, so an RCCL reduce(SUM) of the buffers would give the same frame bit for bit.
// Parity: every tile equals the oracle run on that tile (stream indices are local to a tile).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mi355x_pathtracer.h"

extern "C" void ptx_internal_set_error(const char *msg);

// One host thread per device for the calls that enqueue a lot (render, iterate): a launch set is ~20 launches, i.e. ~0.1 ms of
// host time, and one thread handing eight devices their sets one after the other would have the last device start 0.7 ms late --
// more than a short call traces.  The worker owns nothing but the right to call into ITS tracer; errors come back by value.
struct ptx_multi_worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, finished = true, quit = false;
    int rc = 0;
    std::string err;
    void loop(int device) {
        (void)hipSetDevice(device);
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) return;
            std::function<int()> j = std::move(job);
            has_job = false;
            lk.unlock();
            const int r = j();
            const std::string e = r != PTX_OK ? std::string(ptx_last_error()) : std::string();
            lk.lock();
            rc = r; err = e; finished = true;
            cv.notify_all();
        }
    }
    void submit(std::function<int()> f) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(f); has_job = true; finished = false;
        cv.notify_all();
    }
    int wait(std::string &e) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return finished; });
        e = err;
        return rc;
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; cv.notify_all(); }
        if (th.joinable()) th.join();
    }
};

struct ptx_multi {
    std::vector<ptx_tracer *> tr;
    std::vector<int> dev;
    std::vector<char> peer;                                      // peer[i]: device i can write device 0's memory directly (or is device 0)
    std::vector<std::unique_ptr<ptx_multi_worker>> workers;      // one per device when there are several
    int W = 0, H = 0, tile_rows = 0;
};

namespace {
// The entry points below switch the calling thread's current device (hipSetDevice); a caller shaped like the reference's main.cpp
// owns allocations and GL interop on ITS device and must find it current again after every call, error paths included.
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) { saved = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};
int fail(int code, const std::string &msg) { ptx_internal_set_error(msg.c_str()); return code; }
#define MHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(PTX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
}  // namespace

extern "C" {

int ptx_multi_create(const ptx_scene *s, const ptx_options *options, const int *devices, int ndevices, int tile_rows, ptx_multi **out) {
    if (!out) return fail(PTX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!s || !devices || ndevices < 1 || ndevices > 64) return fail(PTX_ERR_INVALID, "ptx_multi_create: need a scene and 1..64 device ordinals");
    if (tile_rows < 1) tile_rows = 8;
    DeviceGuard guard;
    const int have = ptx_device_count();
    if (have < 1) return fail(PTX_ERR_NODEVICE, "no HIP device available; this library has no CPU path");
    for (int i = 0; i < ndevices; i++)
        if (devices[i] < 0 || devices[i] >= have) return fail(PTX_ERR_INVALID, "ptx_multi_create: device ordinal out of range");
    ptx_multi *m = new ptx_multi;
    const ptx_camera *cam = ptx_scene_camera(const_cast<ptx_scene *>(s));
    m->W = cam->resolution[0]; m->H = cam->resolution[1]; m->tile_rows = tile_rows;
    for (int i = 0; i < ndevices; i++) {
        ptx_options o;
        if (options) o = *options; else ptx_default_options(&o);
        o.device = devices[i];
        if (ndevices > 1) { o.tile_rows = tile_rows; o.tile_rank = i; o.tile_world = ndevices; }
        ptx_tracer *t = nullptr;
        int rc = ptx_create_from_scene(s, &o, nullptr, nullptr, &t);
        if (rc != PTX_OK) {
            for (ptx_tracer *p : m->tr) ptx_destroy(p);
            delete m;
            return rc;                                  // (ptx_last_error() says why)
        }
        m->tr.push_back(t); m->dev.push_back(devices[i]);
    }
    // Peer access device[i] -> device[0], so that the gather is one strided copy written straight over xGMI.  Whether it really
    // is enabled is RECORDED per device: a rectangular device-to-device copy into memory the copying device has no mapping of
    // is not staged by the runtime the way a linear peer copy is -- ptx_multi_assemble takes the linear peer copies (one per
    // row block, hipMemcpyPeerAsync, which the runtime may stage) for such a device instead.
    m->peer.assign((size_t)ndevices, 1);
    if (getenv("PTX_DEBUG_NO_PEER"))          // tests: take the block-wise path on every device but the first, whatever the hardware offers
        for (int i = 1; i < ndevices; i++) m->peer[(size_t)i] = 0;
    for (int i = 1; i < ndevices; i++)
        if (devices[i] != devices[0] && m->peer[(size_t)i]) {
            int can = 0;
            bool ok = false;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can && hipSetDevice(devices[i]) == hipSuccess) {
                const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
                ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            }
            (void)hipGetLastError();
            m->peer[(size_t)i] = ok ? 1 : 0;
        }
    if (ndevices > 1)
        for (int i = 0; i < ndevices; i++) {
            m->workers.emplace_back(new ptx_multi_worker);
            ptx_multi_worker *w = m->workers.back().get();
            const int d = devices[i];
            w->th = std::thread([w, d] { w->loop(d); });
        }
    *out = m;
    return PTX_OK;
}

void ptx_multi_destroy(ptx_multi *m) {
    if (!m) return;
    DeviceGuard guard;
    for (auto &w : m->workers) w->stop();
    for (ptx_tracer *t : m->tr) ptx_destroy(t);
    delete m;
}

int ptx_multi_device_count(const ptx_multi *m) { return m ? (int)m->tr.size() : 0; }
ptx_tracer *ptx_multi_tracer(ptx_multi *m, int i) { return (m && i >= 0 && i < (int)m->tr.size()) ? m->tr[i] : nullptr; }

#define FOR_ALL(call) do { if (!m) return fail(PTX_ERR_INVALID, "null ptx_multi"); DeviceGuard guard_; for (ptx_tracer *t : m->tr) { int rc_ = (call); if (rc_ != PTX_OK) return rc_; } return PTX_OK; } while (0)
int ptx_multi_set_camera(ptx_multi *m, const ptx_camera *camera, int trace_depth) { FOR_ALL(ptx_set_camera(t, camera, trace_depth)); }
int ptx_multi_reset_image(ptx_multi *m) { FOR_ALL(ptx_reset_image(t)); }
// the same call on every device at once, each from that device's own host thread; the first failure (in device order) is reported
#define PAR_ALL(call) do {                                                                                              \
        if (!m) return fail(PTX_ERR_INVALID, "null ptx_multi");                                                        \
        DeviceGuard guard_;                                                                                            \
        if (m->workers.empty()) { for (ptx_tracer *t : m->tr) { int rc_ = (call); if (rc_ != PTX_OK) return rc_; } return PTX_OK; } \
        for (size_t i_ = 0; i_ < m->tr.size(); i_++) { ptx_tracer *t = m->tr[i_]; m->workers[i_]->submit([=]() -> int { return (call); }); } \
        int first_rc = PTX_OK; std::string first_err;                                                                  \
        for (size_t i_ = 0; i_ < m->tr.size(); i_++) {                                                                 \
            std::string e_; const int rc_ = m->workers[i_]->wait(e_);                                                  \
            if (rc_ != PTX_OK && first_rc == PTX_OK) { first_rc = rc_; first_err = e_; }                               \
        }                                                                                                              \
        if (first_rc != PTX_OK) return fail(first_rc, first_err);                                                      \
        return PTX_OK;                                                                                                 \
    } while (0)
int ptx_multi_render(ptx_multi *m, int iter_first, int count) { PAR_ALL(ptx_render(t, iter_first, count)); }     /* enqueues on every device, returns */
int ptx_multi_iterate(ptx_multi *m, int iter) { PAR_ALL(ptx_iterate(t, iter)); }
int ptx_multi_set_render_ahead(ptx_multi *m, int on) { FOR_ALL(ptx_set_render_ahead(t, on)); }
int ptx_multi_synchronize(ptx_multi *m) { FOR_ALL(ptx_synchronize(t)); }

// the row blocks device i owns -> the same rows of device[0]'s frame; returns when they have arrived
int ptx_multi_assemble(ptx_multi *m) {
    if (!m) return fail(PTX_ERR_INVALID, "null ptx_multi");
    const int n = (int)m->tr.size();
    DeviceGuard guard;
    if (n == 1) return ptx_synchronize(m->tr[0]);
    const size_t row_bytes = (size_t)m->W * 3 * sizeof(float), blk_bytes = row_bytes * m->tile_rows;
    const int nblocks = (m->H + m->tile_rows - 1) / m->tile_rows;
    char *dst0 = reinterpret_cast<char *>(ptx_device_image(m->tr[0]));
    for (int i = 1; i < n; i++) {
        const char *src = reinterpret_cast<const char *>(ptx_device_image(m->tr[i]));
        hipStream_t st = (hipStream_t)ptx_stream(m->tr[i]);          // behind that device's tracing
        MHIP(hipSetDevice(m->dev[i]));
        // blocks i, i + n, i + 2n, ...: the whole ones as one strided copy, a cut-off last block (H not a multiple of
        // tile_rows) on its own
        int mine = 0, whole = 0;
        for (int b = i; b < nblocks; b += n) { mine++; if ((b + 1) * m->tile_rows <= m->H) whole++; }
        if (!m->peer[(size_t)i]) {                 // no mapping of device 0's memory on this device: linear peer copies, block by block
            for (int b = i; b < nblocks; b += n) {
                const size_t off = (size_t)b * blk_bytes;
                const size_t bytes = std::min(blk_bytes, (size_t)(m->H - b * m->tile_rows) * row_bytes);
                MHIP(hipMemcpyPeerAsync(dst0 + off, m->dev[0], src + off, m->dev[i], bytes, st));
            }
            continue;
        }
        if (whole > 0)
            MHIP(hipMemcpy2DAsync(dst0 + (size_t)i * blk_bytes, (size_t)n * blk_bytes, src + (size_t)i * blk_bytes, (size_t)n * blk_bytes,
                                  blk_bytes, (size_t)whole, hipMemcpyDeviceToDevice, st));
        if (mine > whole) {
            const int b = i + whole * n;
            const size_t off = (size_t)b * blk_bytes, bytes = (size_t)(m->H - b * m->tile_rows) * row_bytes;
            MHIP(hipMemcpyAsync(dst0 + off, src + off, bytes, hipMemcpyDeviceToDevice, st));
        }
    }
    for (ptx_tracer *t : m->tr) { int rc = ptx_synchronize(t); if (rc != PTX_OK) return rc; }
    return PTX_OK;
}

float *ptx_multi_device_image(ptx_multi *m) { return m ? ptx_device_image(m->tr[0]) : nullptr; }     /* complete after ptx_multi_assemble */

int ptx_multi_read_image(ptx_multi *m, float *host_rgb) {
    if (!m || !host_rgb) return fail(PTX_ERR_INVALID, "null argument");
    DeviceGuard guard;
    int rc = ptx_multi_assemble(m);
    if (rc != PTX_OK) return rc;
    return ptx_read_image(m->tr[0], host_rgb);
}

// apps_variant: every device's albedo AOV holds its own rows of iteration 1; merged on the host (a one-off, 1 frame)
int ptx_multi_read_albedo(ptx_multi *m, float *host_rgb) {
    if (!m || !host_rgb) return fail(PTX_ERR_INVALID, "null argument");
    const int n = (int)m->tr.size();
    DeviceGuard guard;
    if (n == 1) return ptx_read_albedo(m->tr[0], host_rgb);
    std::vector<float> tmp((size_t)m->W * m->H * 3);
    const size_t row = (size_t)m->W * 3;
    for (int i = 0; i < n; i++) {
        int rc = ptx_read_albedo(m->tr[i], tmp.data());
        if (rc != PTX_OK) return rc;
        for (int y = 0; y < m->H; y++)
            if ((y / m->tile_rows) % n == i) memcpy(host_rgb + (size_t)y * row, tmp.data() + (size_t)y * row, row * sizeof(float));
    }
    return PTX_OK;
}

// Page-locks a host buffer the caller owns (e.g. the reference's scene->state.image, the destination of its per-iteration
// frame read-back, src/pathtrace.cu:555-556) so that ptx_read_image copies into it by DMA, without the runtime's staging
// through its own pinned chunks.  The buffer must not move or be freed while pinned.  Failure is not fatal to the caller:
// reads into unpinned memory work, slower.
int ptx_pin_host_buffer(void *p, size_t bytes) {
    if (!p || !bytes) return fail(PTX_ERR_INVALID, "ptx_pin_host_buffer: null buffer");
    MHIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return PTX_OK;
}
int ptx_unpin_host_buffer(void *p) {
    if (!p) return PTX_OK;
    MHIP(hipHostUnregister(p));
    return PTX_OK;
}

// rays and iterations summed over the devices (each traces its tile of every iteration: iterations = device 0's), loop time
// = the slowest device's
int ptx_multi_get_stats(ptx_multi *m, ptx_stats *out) {
    if (!m || !out) return fail(PTX_ERR_INVALID, "null argument");
    memset(out, 0, sizeof *out);
    DeviceGuard guard;
    for (size_t i = 0; i < m->tr.size(); i++) {
        ptx_stats s;
        int rc = ptx_get_stats(m->tr[i], &s);
        if (rc != PTX_OK) return rc;
        out->bounces = s.bounces > out->bounces ? s.bounces : out->bounces;
        for (int b = 0; b < 64; b++) out->rays_per_bounce[b] += s.rays_per_bounce[b];
        out->rays_total += s.rays_total;
        out->fenced += s.fenced;
        out->stored_paths += s.stored_paths; out->stored_with_direction += s.stored_with_direction; out->stored_with_normal_code += s.stored_with_normal_code;
        if (s.loop_ms_total > out->loop_ms_total) out->loop_ms_total = s.loop_ms_total;
        if (i == 0) out->iterations = s.iterations;
    }
    return PTX_OK;
}

}  // extern "C"
