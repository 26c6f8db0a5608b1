// pathtrace_api.h -- C++ veneer with the reference's own names over the C ABI (include/mi355x_pathtracer.h).
//
// A maintainer of the reference replaces `#include "pathtrace.h"` / `#include "scene.h"` by this header and links
// libmi355x_pathtracer.so; main.cpp's calls (src/main.cpp:47, :128-148) compile unchanged in meaning:
//
//     scene = new Scene(sceneFile);                       // src/scene.cpp:10
//     pathtraceFree(); pathtraceInit(scene);              // src/pathtrace.h:7-8
//     pathtrace(pbo_dptr, frame, iteration);              // src/pathtrace.h:9 ; fills scene->state.image
//     timer().getGpuElapsedTimeForPreviousOperation();    // src/timer.h
//
// glm is not required: vec3 members are plain float[3]-compatible structs with the same memory layout.
#pragma once
#include <string>
#include <vector>

#include "../../include/mi355x_pathtracer.h"

// The pbo arguments are device pointers to uchar4 exactly as in the reference (the mapped GL buffer of src/main.cpp:131-135).
// hip_runtime.h makes uchar4 a typedef of a class template, which can neither be forward-declared nor be mangled the same way
// from a file without it, so the EXPORTED entry points take void* (mi355x::pathtrace_raw / sendToGPU_raw, and the same two under
// the reference's names for callers without HIP's headers).  What a translation unit SEES under the names `pathtrace` and
// `sendToGPU` is ONE function each, never an overload set:
//   * uchar4 known (hip_runtime.h / hip_vector_types.h included before this header, as src/main.cpp's includes do through
//     pathtrace.h's <cuda_runtime.h> counterpart): the reference's exact signatures, `pathtrace(uchar4 *, int, int)`
//     (src/pathtrace.h:9) and `sendToGPU(uchar4 *, int)` (apps/src/pathtrace.h:10), inline over the raw entry points --
//     `pathtrace(NULL, 0, it)` / `pathtrace(nullptr, ...)` / `pathtrace(0, ...)` compile as they do against the reference's
//     header, and so does taking the function's address with the reference's type;
//   * otherwise: `pathtrace(void *, int, int)` / `sendToGPU(void *, int)`, the exported symbols themselves.
// (Round 3 declared both at once in a HIP translation unit: a literal null pbo was then ambiguous.)

namespace mi355x {
struct vec3 { float x, y, z; };
}

typedef ptx_geom Geom;            // src/sceneStructs.h:50-69 (POD subset the tracer reads)
typedef ptx_material Material;    // src/sceneStructs.h:71-81
typedef ptx_camera Camera;        // src/sceneStructs.h:83-92

// src/sceneStructs.h:94-100
struct RenderState {
    Camera camera;
    unsigned int iterations;
    int traceDepth;
    std::vector<mi355x::vec3> image;      // sum of per-iteration radiance, row-major x + y*W, y = 0 on top
    std::vector<mi355x::vec3> albedo;     // apps/src/sceneStructs.h:100: first-hit albedo of iteration 1 (apps_variant only)
    std::vector<mi355x::vec3> output;     // apps/src/sceneStructs.h:101: the denoiser's result, input of sendToGPU
    std::string imageName;
};

// src/scene.h:11-32
class Scene {
public:
    // base_dir plays the role of the reference's process CWD for "../models/..." paths; "" = directory of filename
    explicit Scene(const std::string &filename, const std::string &base_dir = "");
    ~Scene();
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;
    std::vector<Geom> geoms;
    std::vector<Material> materials;
    RenderState state;
    ptx_scene *handle() const { return impl_; }
    void applyRunCudaCamera();            // src/main.cpp:56-70 + 105-123 (what the first runCuda() call does)
    // main.cpp's mouse, scripted: "left:DX,DY;right:DY;middle:DX,DY;space" (window pixels).  Starts from the loader's
    // camera like main() does (:56-70), applies runCuda's recompute (:105-123) first and after every event.
    bool runOrbitScript(const std::string &script);
    void setResolution(int w, int h);     // harness override of RES (re-derives fov / pixelLength as loadCamera does)
private:
    ptx_scene *impl_;
};

// src/timer.h:17-100 -- only the GPU half is meaningful here
class PerformanceTimer {
public:
    float getGpuElapsedTimeForPreviousOperation();
    float getCpuElapsedTimeForPreviousOperation() { return 0.f; }
};

// Runtime form of the #defines in src/pathtrace.cu:36-40; read by the next pathtraceInit.
ptx_options &pathtraceOptions();
bool &pathtraceRenderAhead();                           // true (default): pathtrace(iter) calls that count up are served from
                                                        // iterations traced ahead in the background (ptx_set_render_ahead); same results

// Devices the next pathtraceInit uses.  Empty (default) = the current device, as the reference (src/preview.cpp:107).  Two or
// more ordinals: the frame is split into interleaved 8-row blocks, device i traces blocks i, i + n, ... (ptx_multi_*,
// include/mi355x_pathtracer.h); pathtrace() still returns with the whole frame in scene->state.image.
std::vector<int> &pathtraceDevices();

PerformanceTimer &timer();                              // src/pathtrace.h:6
void pathtraceInit(Scene *scene);                       // src/pathtrace.h:7
void pathtraceFree();                                   // src/pathtrace.h:8
ptx_tracer *pathtraceHandle();                          // the C-ABI handle behind the module-static state (device 0's with several devices)

namespace mi355x {
void pathtrace_raw(void *pbo, int frame, int iteration);   // what both spellings below run; pbo = device uchar4*, may be NULL (no preview)
void sendToGPU_raw(void *pbo, int iter);                   // state.output -> 8-bit preview in the device pbo
}
#if defined(HIP_INCLUDE_HIP_HIP_VECTOR_TYPES_H) || defined(HIP_INCLUDE_HIP_AMD_DETAIL_HIP_VECTOR_TYPES_H)
// uchar4 is known here: the reference's own signatures and nothing else under these names
inline void pathtrace(uchar4 *pbo, int frame, int iteration) { mi355x::pathtrace_raw(static_cast<void *>(pbo), frame, iteration); }   // src/pathtrace.h:9
inline void sendToGPU(uchar4 *pbo, int iter) { mi355x::sendToGPU_raw(static_cast<void *>(pbo), iter); }                              // apps/src/pathtrace.h:10
#else
void pathtrace(void *pbo, int frame, int iteration);    // src/pathtrace.h:9 (uchar4 *pbo); pbo may be NULL (no preview)
void sendToGPU(void *pbo, int iter);                    // apps/src/pathtrace.h:10 (uchar4 *pbo)
#endif
