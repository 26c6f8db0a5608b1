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
// glm is not required: the vector members are plain structs of floats with ptx_camera's layout that convert from and to any
// {x, y, z} type -- with glm included first, main.cpp's camera code (src/main.cpp:52-70, 105-123) compiles against them as written.
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

// ---- the vector members of Camera / RenderState ------------------------------------------------------------------------
// src/main.cpp reads and writes them as glm values (`glm::vec3 view = cam.view;`, `cam.view = -glm::normalize(cameraPosition);`,
// `cam.position - ogLookAt`, `cameraPosition += cam.lookAt`, `cam.resolution.x`: src/main.cpp:52-70, 105-123, 86-88).  They stay
// plain structs of floats here -- the veneer needs no glm, and Camera stays layout-identical to the C ABI's ptx_camera, which is what
// crosses the boundary -- but they convert from and to ANY type with public x, y, z members (glm::vec3, float3, a caller's own),
// assign from one, index with [], and mix with one in + - * / (the result takes the other operand's type, so glm:: functions
// accept it).  With glm included BEFORE this header (GLM_VERSION defined, as in main.cpp through utilities.h) the compound
// assignments glm declares as catch-all member templates (`tvec3::operator+=(U)`) get exact glm overloads too, so that
// `cameraPosition += cam.lookAt;` compiles as it does against the reference's sceneStructs.h.
// tests/test_abi.py::test_veneer_compiles_main_cpp_camera_block feeds those very lines of the reference's main.cpp through this header.
#include <type_traits>
#include <utility>
#include <cstddef>

namespace mi355x {
namespace detail {
template <class...> struct voider { typedef void type; };
// V has members x, y, z (and is not one of ours)
template <class V, class = void> struct has_xyz : std::false_type {};
template <class V> struct has_xyz<V, typename voider<decltype(std::declval<const V &>().x), decltype(std::declval<const V &>().y),
                                                    decltype(std::declval<const V &>().z)>::type> : std::true_type {};
template <class V, class = void> struct has_xy : std::false_type {};
template <class V> struct has_xy<V, typename voider<decltype(std::declval<const V &>().x), decltype(std::declval<const V &>().y)>::type> : std::true_type {};
}  // namespace detail

struct vec3 {
    float x, y, z;
    vec3() = default;
    constexpr vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    constexpr explicit vec3(float s) : x(s), y(s), z(s) {}
    template <class V, class = typename std::enable_if<detail::has_xyz<V>::value && !std::is_same<V, vec3>::value>::type>
    vec3(const V &v) : x((float)v.x), y((float)v.y), z((float)v.z) {}
    template <class V, class = typename std::enable_if<detail::has_xyz<V>::value && !std::is_same<V, vec3>::value>::type>
    vec3 &operator=(const V &v) { x = (float)v.x; y = (float)v.y; z = (float)v.z; return *this; }
    template <class V, class = typename std::enable_if<detail::has_xyz<V>::value && !std::is_same<V, vec3>::value &&
                                                       std::is_constructible<V, float, float, float>::value>::type>
    operator V() const { return V(x, y, z); }
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
struct vec2 {
    float x, y;
    vec2() = default;
    constexpr vec2(float x_, float y_) : x(x_), y(y_) {}
    template <class V, class = typename std::enable_if<detail::has_xy<V>::value && !detail::has_xyz<V>::value && !std::is_same<V, vec2>::value>::type>
    vec2(const V &v) : x((float)v.x), y((float)v.y) {}
    template <class V, class = typename std::enable_if<detail::has_xy<V>::value && !detail::has_xyz<V>::value && !std::is_same<V, vec2>::value>::type>
    vec2 &operator=(const V &v) { x = (float)v.x; y = (float)v.y; return *this; }
    template <class V, class = typename std::enable_if<detail::has_xy<V>::value && !detail::has_xyz<V>::value && !std::is_same<V, vec2>::value &&
                                                       std::is_constructible<V, float, float>::value>::type>
    operator V() const { return V(x, y); }
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
struct ivec2 {
    int x, y;
    ivec2() = default;
    constexpr ivec2(int x_, int y_) : x(x_), y(y_) {}
    template <class V, class = typename std::enable_if<detail::has_xy<V>::value && !detail::has_xyz<V>::value && !std::is_same<V, ivec2>::value>::type>
    ivec2(const V &v) : x((int)v.x), y((int)v.y) {}
    template <class V, class = typename std::enable_if<detail::has_xy<V>::value && !detail::has_xyz<V>::value && !std::is_same<V, ivec2>::value>::type>
    ivec2 &operator=(const V &v) { x = (int)v.x; y = (int)v.y; return *this; }
    template <class V, class = typename std::enable_if<detail::has_xy<V>::value && !detail::has_xyz<V>::value && !std::is_same<V, ivec2>::value &&
                                                       std::is_constructible<V, int, int>::value>::type>
    operator V() const { return V(x, y); }
    int &operator[](int i) { return (&x)[i]; }              // (the veneer's own sources index resolution[0] / [1])
    const int &operator[](int i) const { return (&x)[i]; }
};

// vec3 with itself
inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(const vec3 &a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, const vec3 &a) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator/(const vec3 &a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 operator-(const vec3 &a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 &operator+=(vec3 &a, const vec3 &b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline vec3 &operator-=(vec3 &a, const vec3 &b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
inline bool operator==(const vec3 &a, const vec3 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(const vec3 &a, const vec3 &b) { return !(a == b); }
// vec3 with a foreign {x, y, z} type: component-wise, the result has the FOREIGN type (glm::length(cam.position - ogLookAt))
#define MI355X_VEC3_MIXED(OP)                                                                                                              \
    template <class V, class = typename std::enable_if<detail::has_xyz<V>::value && !std::is_same<V, vec3>::value &&                      \
                                                       std::is_constructible<V, float, float, float>::value>::type>                       \
    V operator OP(const vec3 &a, const V &b) { return V((float)(a.x OP b.x), (float)(a.y OP b.y), (float)(a.z OP b.z)); }                  \
    template <class V, class = typename std::enable_if<detail::has_xyz<V>::value && !std::is_same<V, vec3>::value &&                      \
                                                       std::is_constructible<V, float, float, float>::value>::type>                       \
    V operator OP(const V &a, const vec3 &b) { return V((float)(a.x OP b.x), (float)(a.y OP b.y), (float)(a.z OP b.z)); }
MI355X_VEC3_MIXED(+)
MI355X_VEC3_MIXED(-)
MI355X_VEC3_MIXED(*)
MI355X_VEC3_MIXED(/)
#undef MI355X_VEC3_MIXED
// ... and compound assignment FROM one (`cam.lookAt -= (float)(xpos - lastX) * right * 0.01f;`, src/main.cpp:204-205)
#define MI355X_VEC3_COMPOUND(OP)                                                                                                           \
    template <class V, class = typename std::enable_if<detail::has_xyz<V>::value && !std::is_same<V, vec3>::value>::type>                 \
    vec3 &operator OP(vec3 &a, const V &b) { a.x OP (float)b.x; a.y OP (float)b.y; a.z OP (float)b.z; return a; }
MI355X_VEC3_COMPOUND(+=)
MI355X_VEC3_COMPOUND(-=)
MI355X_VEC3_COMPOUND(*=)
MI355X_VEC3_COMPOUND(/=)
#undef MI355X_VEC3_COMPOUND
}  // namespace mi355x

#ifdef GLM_VERSION
// glm's compound assignments are member templates that take ANYTHING by value (`tvec3::operator+=(U s)`, a scalar broadcast): exact
// overloads for the veneer's types, which overload resolution prefers (non-template over template), so `glmvec += cam.lookAt` means
// what it means with the reference's glm members.
namespace glm {
inline vec3 &operator+=(vec3 &a, const mi355x::vec3 &b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline vec3 &operator-=(vec3 &a, const mi355x::vec3 &b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
inline vec3 &operator*=(vec3 &a, const mi355x::vec3 &b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
inline vec3 &operator/=(vec3 &a, const mi355x::vec3 &b) { a.x /= b.x; a.y /= b.y; a.z /= b.z; return a; }
}  // namespace glm
#endif

typedef ptx_geom Geom;            // src/sceneStructs.h:50-69 (POD subset the tracer reads)
typedef ptx_material Material;    // src/sceneStructs.h:71-81

// src/sceneStructs.h:83-92 -- the reference's member names and (glm-like) member types; the bytes are ptx_camera's, so the veneer hands
// `&state.camera` to the C ABI as it is (asserted below)
struct Camera {
    mi355x::ivec2 resolution;
    mi355x::vec3 position;
    mi355x::vec3 lookAt;
    mi355x::vec3 view;
    mi355x::vec3 up;
    mi355x::vec3 right;
    mi355x::vec2 fov;
    mi355x::vec2 pixelLength;
    Camera() = default;
    Camera(const ptx_camera &c) { *this = c; }
    Camera &operator=(const ptx_camera &c) { *reinterpret_cast<ptx_camera *>(this) = c; return *this; }
    ptx_camera *c_abi() { return reinterpret_cast<ptx_camera *>(this); }
    const ptx_camera *c_abi() const { return reinterpret_cast<const ptx_camera *>(this); }
    operator ptx_camera() const { return *c_abi(); }
};
static_assert(std::is_standard_layout<Camera>::value && std::is_trivially_copyable<Camera>::value, "Camera crosses the C ABI as bytes");
static_assert(sizeof(Camera) == sizeof(ptx_camera) && alignof(Camera) == alignof(ptx_camera), "Camera = ptx_camera");
static_assert(offsetof(Camera, resolution) == offsetof(ptx_camera, resolution) && offsetof(Camera, position) == offsetof(ptx_camera, position) &&
              offsetof(Camera, lookAt) == offsetof(ptx_camera, lookAt) && offsetof(Camera, view) == offsetof(ptx_camera, view) &&
              offsetof(Camera, up) == offsetof(ptx_camera, up) && offsetof(Camera, right) == offsetof(ptx_camera, right) &&
              offsetof(Camera, fov) == offsetof(ptx_camera, fov) && offsetof(Camera, pixelLength) == offsetof(ptx_camera, pixelLength),
              "Camera's members lie where ptx_camera's do");
static_assert(sizeof(mi355x::vec3) == 12 && sizeof(mi355x::vec2) == 8 && sizeof(mi355x::ivec2) == 8, "no padding in the vector members");

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

// src/timer.h:17-100, the whole class: start/endCpuTimer (std::chrono, as the reference), start/endGpuTimer (HIP events on the
// tracer's stream of the current pathtraceInit, or on the null stream before it), the same "already started" / "not started"
// exceptions, uncopyable.  The instance behind timer() ALSO answers getGpuElapsedTimeForPreviousOperation() with the device time
// of the last pathtrace() call's bounce loop -- what the reference's pathtrace brackets with timer().start/endGpuTimer()
// (src/pathtrace.cu:489, 545) -- unless the caller has used start/endGpuTimer on it since: then it is the caller's interval.
class PerformanceTimer {
public:
    PerformanceTimer();
    ~PerformanceTimer();
    void startCpuTimer();
    void endCpuTimer();
    void startGpuTimer();
    void endGpuTimer();
    float getCpuElapsedTimeForPreviousOperation() { return prev_elapsed_time_cpu_milliseconds; }
    float getGpuElapsedTimeForPreviousOperation();
    PerformanceTimer(const PerformanceTimer &) = delete;
    PerformanceTimer(PerformanceTimer &&) = delete;
    PerformanceTimer &operator=(const PerformanceTimer &) = delete;
    PerformanceTimer &operator=(PerformanceTimer &&) = delete;
private:
    friend void mi355x_timer_note_pathtrace(PerformanceTimer &);
    void *event_start = nullptr, *event_end = nullptr;       // hipEvent_t (this header does not need HIP's)
    long long time_start_cpu_ns = 0;
    bool cpu_timer_started = false, gpu_timer_started = false;
    bool gpu_interval_is_callers = false;                     // the last GPU figure came from start/endGpuTimer, not from pathtrace()
    float prev_elapsed_time_cpu_milliseconds = 0.f, prev_elapsed_time_gpu_milliseconds = 0.f;
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
