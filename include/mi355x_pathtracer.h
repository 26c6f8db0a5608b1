/* mi355x_pathtracer.h -- C ABI of the MI355X-native path tracer (libmi355x_pathtracer.so).
 *
 * Drop-in boundary for the reference's path-tracing module (nkkk98/MyGPURaytracer, all file:line citations are
 * relative to the reference root):
 *
 *   reference interface                                  this ABI
 *   ---------------------------------------------------  --------------------------------------------------
 *   Scene::Scene(filename)            src/scene.cpp:10    ptx_scene_load / ptx_scene_get_* / ptx_scene_free
 *   runCuda() camera recompute        src/main.cpp:105    ptx_scene_apply_runcuda_camera
 *   pathtraceInit(Scene*)             src/pathtrace.h:7   ptx_create        (scene flattened to POD, options =
 *                                     src/pathtrace.cu:101                   the #defines of pathtrace.cu:36-40)
 *   pathtrace(uchar4* pbo, frame, it) src/pathtrace.h:9   ptx_iterate (+ ptx_write_pbo, ptx_read_image)
 *                                     src/pathtrace.cu:433
 *   pathtraceFree()                   src/pathtrace.h:8   ptx_destroy
 *   timer()                           src/pathtrace.h:6   ptx_last_loop_ms
 *   checkCUDAError -> exit()          src/pathtrace.cu:42 return codes + ptx_last_error()
 *
 * Plain C types only: no C++ classes, no torch types.  Every pointer marked "device" is a HIP device pointer on
 * the device the handle was created on; everything else is host memory.  The C++ veneer with the reference's
 * own names (pathtraceInit / pathtrace / pathtraceFree, class Scene) lives in
 * mygpuraytracer_amd/csrc/pathtrace_api.h and is a few lines over this ABI.
 */
#ifndef MI355X_PATHTRACER_H
#define MI355X_PATHTRACER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI revision of this header: bumped whenever a struct a caller allocates (ptx_options, ptx_stats, ptx_camera ...) grows or an entry
 * point changes meaning.  A caller compiled against one revision and loaded against a library of another must not hand it those
 * structs: check ptx_abi_version() == PTX_ABI_VERSION after loading (mygpuraytracer_amd/api.py does, and refuses), or use the sized
 * entry points (ptx_get_stats_sized), which never write more than the caller says it has.
 * 5 = round 5: ptx_options.arith; ptx_stats as of round 4 (fenced, stored_*).  Libraries before 5 do not export ptx_abi_version. */
#define PTX_ABI_VERSION 5
int ptx_abi_version(void);
size_t ptx_sizeof_options(void);           /* sizeof(ptx_options) / sizeof(ptx_stats) as the LIBRARY was compiled */
size_t ptx_sizeof_stats(void);

#define PTX_OK 0
#define PTX_ERR_INVALID 1      /* bad argument / malformed scene                      */
#define PTX_ERR_IO 2           /* file could not be read                              */
#define PTX_ERR_HIP 3          /* a HIP runtime call failed (message in ptx_last_error) */
#define PTX_ERR_NODEVICE 4     /* no usable HIP device: there is no CPU fallback      */
#define PTX_ERR_UNSUPPORTED 5

/* enum GeomType, src/sceneStructs.h:10-15 */
enum { PTX_SPHERE = 0, PTX_CUBE = 1, PTX_TRIANGLE = 2, PTX_OBJ = 3 };

/* struct Material, src/sceneStructs.h:71-81 -- same 44-byte layout */
typedef struct ptx_material {
    float color[3];
    float specular_exponent;
    float specular_color[3];
    float hasReflective;
    float hasRefractive;
    float indexOfRefraction;
    float emittance;
} ptx_material;

/* struct Texture, src/sceneStructs.h:36-48 (host pixels; uploaded by ptx_create) */
typedef struct ptx_texture {
    int32_t width, height, channels;
    const uint8_t *image;          /* width*height*channels bytes, or NULL when channels == 0 */
} ptx_texture;

/* struct Geom, src/sceneStructs.h:50-69.  Matrices in glm memory order (column major, m[c*4+r]).
 * faces: faceSize x 15 floats = 3 vertices x (position xyz, texcoord uv) -- the only Face fields the path
 * tracer reads (src/intersections.h:216-262). */
typedef struct ptx_geom {
    int32_t type;
    int32_t materialid;
    float translation[3], rotation[3], scale[3];
    float transform[16], inverseTransform[16], invTranspose[16];
    int32_t faceSize;
    const float *faces;
    ptx_texture kd, ks, bump, ke;
} ptx_geom;

/* struct Camera, src/sceneStructs.h:83-92 */
typedef struct ptx_camera {
    int32_t resolution[2];
    float position[3], lookAt[3], view[3], up[3], right[3];
    float fov[2];
    float pixelLength[2];
} ptx_camera;

/* Runtime form of the compile-time switches of src/pathtrace.cu:36-40 (same defaults via ptx_default_options),
 * plus the multi-GPU row-tile split.  A device owns the row blocks b (of tile_rows rows each) with
 * b % tile_world == tile_rank; pixelIndex stays global (x + y*W). */
typedef struct ptx_options {
    int32_t depth_of_field;      /* DEPTH_OF_FIELD 0      */
    int32_t cache_first_bounce;  /* CACHE_FIRST_BOUNCE 1  */
    int32_t sort_by_material;    /* SORT_BY_MATERIAL 1    */
    int32_t antialiasing;        /* ANTIALIASING 1        */
    int32_t bounding_box;        /* BOUNDING_BOX 0 (accepted, must be 0) */
    int32_t tile_rows, tile_rank, tile_world;   /* 0,0,1 = whole frame */
    int32_t device;              /* HIP device ordinal, -1 = current device */
    int32_t batch;               /* iterations traced per launch set by ptx_render (independent streams, results
                                    identical to one at a time); 0 = choose from the tile size */
    int32_t no_lds_triangles;    /* 1 = read the triangle table from global memory even when it would fit in LDS */
    int32_t apps_variant;        /* 1 = behave like the apps/src copy of the reference (the one its CMake builds):
                                    finalGather adds color * PI (apps/src/pathtrace.cu:508) and iteration 1 fills an
                                    albedo AOV (apps/src/pathtrace.cu:412-462, ptx_read_albedo) */
    int32_t no_cull;             /* 1 = every ray tests every geom (the reference's loop) instead of per-lane candidate
                                    lists from conservative world boxes; results are identical either way */
    int32_t no_bvh;              /* 1 = every mesh is searched by the reference's loop over all its faces; 0 = meshes of
                                    24+ faces get a bounding-volume hierarchy (same nearest face, see csrc/pt_bvh.h)      */
    int32_t lanes;               /* launch sets in flight, each on a stream of its own: 0 = default (3), 1 .. 8 explicit  */
    int32_t no_mesh_split;       /* 1 = meshes are searched inside the bounce kernel even when they have a BVH; 0 = scenes
                                    with BVH meshes run the mesh search as a kernel of its own between two halves of it   */
    int32_t arith;               /* 0 = EXACT (default, and what every headline number is measured with): fp32 without contraction,
                                    IEEE division and square root, own correctly rounded sin/cos -- bit-identical to the CPU
                                    oracle.  1 = CONTRACTED: the same kernels compiled as a second code object with fused
                                    multiply-adds and the hardware's reciprocal / square-root / sine instructions, i.e. the kind
                                    of arithmetic the reference's real build runs (nvcc contracts by default); results agree with
                                    EXACT within the statistical fp32 tolerance of DESIGN.md section 3 (tests/test_fp_tolerance.py),
                                    not bit for bit.  PTX_ERR_UNSUPPORTED if the library was built without that code object.  */
} ptx_options;
#define PTX_ARITH_EXACT 0
#define PTX_ARITH_CONTRACTED 1

typedef struct ptx_stats {
    int32_t bounces;                 /* bounce-loop passes of the last iteration                       */
    int64_t rays_per_bounce[64];     /* paths entering the intersect stage, per bounce (last iteration) */
    int64_t rays_total;              /* sum over all iterations since create/reset                      */
    double loop_ms_total;            /* device time of the bounce loops since create/reset              */
    int64_t iterations;
    int64_t fenced;                  /* indices from internal tables (mesh-search queue, sort index) that the kernels found out of
                                        range and skipped or clamped instead of faulting, since create/reset.  Always 0: anything else
                                        means corrupted internal state (and a wrong pixel somewhere) -- report it                      */
    int64_t stored_paths;            /* paths stored for a next bounce since create/reset, and how many of their records carry ...        */
    int64_t stored_with_direction;   /* ... the incoming direction (reflective / refractive materials, materials of OBJ geoms: 16 B more) */
    int64_t stored_with_normal_code; /* ... a 3-bit code instead of the normal (materials only cubes have: 16 B less)                      */
} ptx_stats;

typedef struct ptx_tracer ptx_tracer;     /* opaque: one scene on one device */
typedef struct ptx_scene ptx_scene;       /* opaque: a loaded scenes/<x>.txt  */

const char *ptx_last_error(void);
int ptx_device_count(void);               /* HIP devices visible; 0 means nothing here can run */
void ptx_default_options(ptx_options *o);

/* ---- scene loader (src/scene.cpp, src/utilities.cpp) ------------------------------------------------------ */
/* Paths inside the scene file ("../models/x.obj", mtl search path "../models/materials") resolve relative to
 * base_dir, which plays the role of the reference's process CWD; NULL = directory of scene_path. */
int ptx_scene_load(const char *scene_path, const char *base_dir, ptx_scene **out);
void ptx_scene_free(ptx_scene *s);
int ptx_scene_num_geoms(const ptx_scene *s);
int ptx_scene_num_materials(const ptx_scene *s);
const ptx_geom *ptx_scene_geoms(const ptx_scene *s);
const ptx_material *ptx_scene_materials(const ptx_scene *s);
ptx_camera *ptx_scene_camera(ptx_scene *s);            /* mutable: RenderState.camera      */
int ptx_scene_iterations(const ptx_scene *s);          /* RenderState.iterations           */
int ptx_scene_trace_depth(const ptx_scene *s);         /* RenderState.traceDepth           */
void ptx_scene_set_trace_depth(ptx_scene *s, int depth);
void ptx_scene_set_resolution(ptx_scene *s, int w, int h);   /* re-derives fov/pixelLength as loadCamera does */
const char *ptx_scene_image_name(const ptx_scene *s);  /* RenderState.imageName            */
void ptx_scene_apply_runcuda_camera(ptx_scene *s);     /* src/main.cpp:56-70 + :105-123    */

/* The interactive camera of src/main.cpp without the window: the state its mouse handlers keep (main.cpp:18-20) and
 * one function per handler, so that a script of events moves the camera exactly as the same drags would.
 * After any of them call ptx_orbit_apply (runCuda's recompute) and hand the camera to the tracer with ptx_set_camera +
 * ptx_reset_image (runCuda restarts the accumulation: iteration = 0, :106). */
typedef struct ptx_orbit { float phi, theta, zoom; float og_look_at[3]; } ptx_orbit;
void ptx_orbit_init(const ptx_scene *s, ptx_orbit *o);                                  /* main.cpp:56-70   */
void ptx_orbit_left_drag(ptx_orbit *o, double dx, double dy, int width, int height);    /* main.cpp:184-189 */
void ptx_orbit_right_drag(ptx_orbit *o, double dy, int height);                         /* main.cpp:190-194 */
void ptx_orbit_middle_drag(ptx_scene *s, double dx, double dy);                         /* main.cpp:195-209 */
void ptx_orbit_recenter(ptx_scene *s, const ptx_orbit *o);                              /* SPACE, :166-171  */
void ptx_orbit_apply(ptx_scene *s, const ptx_orbit *o);                                 /* main.cpp:105-123 */

/* ---- tracer ------------------------------------------------------------------------------------------------ */
/* pathtraceInit.  external_image: optional device buffer of W*H*3 floats to accumulate into (caller keeps
 * ownership, e.g. a torch tensor that is later reduced over RCCL); NULL = the tracer allocates and zeroes one.
 * stream: optional hipStream_t to run on; NULL = the tracer creates its own (non-blocking) stream.
 * Ordering is the caller's: whatever initialised external_image (a fill, a checkpoint copy) must have COMPLETED, or have
 * been issued on `stream`, before the first render call -- the tracer's stream does not wait for other streams. */
int ptx_create(int ngeoms, const ptx_geom *geoms, int nmaterials, const ptx_material *materials,
               const ptx_camera *camera, int trace_depth, const ptx_options *options,
               float *external_image, void *stream, ptx_tracer **out);
int ptx_create_from_scene(const ptx_scene *s, const ptx_options *options, float *external_image, void *stream,
                          ptx_tracer **out);
void ptx_destroy(ptx_tracer *t);                        /* pathtraceFree; NULL is a no-op */

int ptx_set_camera(ptx_tracer *t, const ptx_camera *camera, int trace_depth);  /* camera edits without re-init */
int ptx_reset_image(ptx_tracer *t);

/* One iteration of pathtrace() (iter is 1-based and seeds the RNG).  Enqueues on the tracer's stream and
 * returns without waiting; any read entry point below synchronises. */
int ptx_iterate(ptx_tracer *t, int iter);
/* Render-ahead for callers that keep the reference's shape -- one pathtrace(iter) per call, iter counting up
 * (src/main.cpp:128-148).  on != 0: ptx_iterate traces the next batch of iterations in the background (other streams,
 * per-iteration radiance buffers) and each call adds exactly its own iteration to the image, so what every call returns
 * -- image, statistics, preview -- is unchanged bit for bit, at the cost per iteration of ptx_render.  A camera change,
 * a jump in iter or any other render call simply drops what was traced ahead.  Off by default in this ABI; the C++ veneer
 * (pathtrace_api.h) switches it on.  Needs the default launch-set layout (lanes >= 3, batch > 1); otherwise it is a no-op. */
int ptx_set_render_ahead(ptx_tracer *t, int on);
/* iterations iter_first .. iter_first+count-1 back to back, no host round trip in between */
int ptx_render(ptx_tracer *t, int iter_first, int count);
/* iterations iter_first, iter_first+stride, ... (count of them): N ranks that take turns over the iterations of one
 * full frame (rank r: iter_first = r+1, stride = N) and sum their buffers reproduce the single-GPU frame, which
 * pixel-row tiles cannot (the shading RNG is seeded by stream position) */
int ptx_render_strided(ptx_tracer *t, int iter_first, int count, int stride);
int ptx_synchronize(ptx_tracer *t);

int ptx_read_image(ptx_tracer *t, float *host_rgb);     /* W*H*3 floats = sum over iterations (state.image) */
int ptx_write_image(ptx_tracer *t, const float *host_rgb);  /* the reverse: resume from a saved accumulation buffer   */
int ptx_read_albedo(ptx_tracer *t, float *host_rgb);    /* RenderState.albedo of apps/src (apps_variant only)  */
/* sendToGPU, apps/src/pathtrace.h:10: a finished (e.g. denoised) host frame -> 8-bit preview, no division by iter */
int ptx_write_denoised_pbo(ptx_tracer *t, const float *host_rgb, uint8_t *host_rgba);
int ptx_write_denoised_pbo_device(ptx_tracer *t, const float *host_rgb, void *device_uchar4);   /* pbo in device memory, as the reference's */
float *ptx_device_image(ptx_tracer *t);                 /* device pointer of the accumulation buffer        */
int ptx_write_pbo(ptx_tracer *t, int iter, uint8_t *host_rgba);          /* sendImageToPBO, pathtrace.cu:69 */
int ptx_write_pbo_device(ptx_tracer *t, int iter, void *device_uchar4);
double ptx_last_loop_ms(ptx_tracer *t);                 /* timer(): bounce loop of the last iteration       */
int ptx_get_stats(ptx_tracer *t, ptx_stats *out);
/* the same, writing at most out_bytes of the struct (a caller compiled against an older, shorter ptx_stats passes ITS sizeof;
 * fields past the library's own struct are zeroed) */
int ptx_get_stats_sized(ptx_tracer *t, void *out, size_t out_bytes);
int ptx_owned_pixels(const ptx_tracer *t);              /* pixels this tracer generates (tile split)        */
void *ptx_stream(ptx_tracer *t);
/* ---- N GPUs of one node, one process (csrc/pt_multi.cpp) ---------------------------------------------------------
 * No reference counterpart: the reference drives device 0 only (src/preview.cpp:107).  A main.cpp-shaped caller
 * (src/main.cpp:128-148) uses these instead of ptx_create / ptx_iterate / ptx_read_image: device i of n traces the
 * interleaved row blocks (y / tile_rows) % n == i as a stream of its own; ptx_multi_assemble copies every device's row
 * blocks into device[0]'s frame (one strided peer copy per device over xGMI), ptx_multi_read_image = assemble + read.
 * `devices` may name one ordinal more than once (two tiles on one GPU: how the one-GPU tests exercise it).
 * tile_rows <= 0 = 8.  Each tile equals the reference's algorithm run on that tile. */
typedef struct ptx_multi ptx_multi;
int ptx_multi_create(const ptx_scene *s, const ptx_options *options, const int *devices, int ndevices, int tile_rows, ptx_multi **out);
void ptx_multi_destroy(ptx_multi *m);
int ptx_multi_device_count(const ptx_multi *m);
ptx_tracer *ptx_multi_tracer(ptx_multi *m, int i);            /* device i's tracer (statistics, kernel timing, ...) */
int ptx_multi_set_camera(ptx_multi *m, const ptx_camera *camera, int trace_depth);
int ptx_multi_reset_image(ptx_multi *m);
int ptx_multi_iterate(ptx_multi *m, int iter);                /* pathtrace(iter) on every device's tile; enqueues, returns */
int ptx_multi_set_render_ahead(ptx_multi *m, int on);
int ptx_multi_render(ptx_multi *m, int iter_first, int count);
int ptx_multi_synchronize(ptx_multi *m);
int ptx_multi_assemble(ptx_multi *m);                         /* owned row blocks -> device[0]'s frame; waits for them */
float *ptx_multi_device_image(ptx_multi *m);                  /* device[0]'s frame (complete after ptx_multi_assemble) */
int ptx_multi_read_image(ptx_multi *m, float *host_rgb);      /* assemble + W*H*3 floats to the host */
int ptx_multi_read_albedo(ptx_multi *m, float *host_rgb);     /* apps_variant: the devices' rows of the albedo AOV, merged */
int ptx_multi_get_stats(ptx_multi *m, ptx_stats *out);        /* rays summed over the devices */
/* Page-lock / release a caller-owned host buffer (the reference's scene->state.image: the destination of its per-iteration
 * read-back, src/pathtrace.cu:555-556), so that ptx_read_image / ptx_multi_read_image into it are direct DMA. */
int ptx_pin_host_buffer(void *p, size_t bytes);
int ptx_unpin_host_buffer(void *p);

/* Optional per-kernel device timing (hipEvents on the tracer's stream around every launch while on).
 * kinds: 0 = k_bounce<first> (ray generation + intersect), 1 = k_bounce (shade + intersect), 2 = k_mesh + k_finish (split mesh search only: the search of the parked rays' meshes and their
 * finishing, one bracket around both launches), 3 = pass 2 of the split bounce (the ranking pass after k_mesh, first and later bounces; 0 for scenes without the split: kinds 0 / 1 are then the whole bounce,
 * with the split they are its pass 1).
 * ptx_get_kernel_times returns the sums since it was last called and clears them. */
int ptx_set_kernel_timing(ptx_tracer *t, int on);
int ptx_get_kernel_times(ptx_tracer *t, double ms_by_kind[4], int64_t launches_by_kind[4]);

/* ---- per-stage entry points (parity tests; same record layouts as the reference's PathSegment 44 B and
 *      ShadeableIntersection 32 B, host arrays in/out, the work runs on the device) --------------------------- */
int ptx_kat_geom_test(ptx_tracer *t, int geom, int n, const float *rays6, float *out10);
/* objTriIntersectionTest / triangleIntersectionLocalTest, src/intersections.h:175-205, 284-315: dead code in the reference (its call is
 * commented out, src/pathtrace.cu:313) and on no path here; a known-answer entry point only.  out8 per ray: t (object space), world
 * point, world normal, outside.  Non-OBJ geoms give t = -1. */
int ptx_kat_obj_tri_test(ptx_tracer *t, int geom, int n, const float *rays6, float *out8);
/* calculateJitteredDirectionHemisphere, src/interactions.h:46-85: dead code in the reference (JITTERED_SAMPLING 0, its call site does not
 * compile) and on no path here; a known-answer entry point only.  Per sample: a normal and (iter, index, depth), which seed the engine as
 * makeSeededRandomEngine does (src/pathtrace.cu:62-66); out3 = the direction. */
int ptx_kat_jittered_hemisphere(ptx_tracer *t, int n, const float *normals3, const int32_t *seeds3, int max_iter, float *out3);
int ptx_kat_compute_intersections(ptx_tracer *t, int n, const void *paths44, void *isects32);
/* computeIntersections (src/pathtrace.cu:261-344) through the functions the bounce kernels really run -- candidate masks from the
 * world boxes, the tile's (ray, geom) pairs tested by the key functions, 64-bit minimum, winner decoded (split != 0: the three
 * pieces of the split mesh search, BVH traversal with the stack included) -- where ptx_kat_compute_intersections runs the plain
 * per-ray loop over all geoms.  PTX_ERR_UNSUPPORTED for a scene that does not take that path (more than 32 geoms, no_cull, ...). */
int ptx_kat_tile_intersect(ptx_tracer *t, int n, const void *paths44, void *isects32, int split);
int ptx_kat_shade(ptx_tracer *t, int iter, int n, const int32_t *idx, const void *isects32, void *paths44);
int ptx_kat_generate(ptx_tracer *t, int iter, void *paths44);             /* all W*H camera rays */
int ptx_kat_libm(ptx_tracer *t, int n, const float *x, float *sin_out, float *cos_out,
                 const double *pw_in, double *pow5_out, const float *powf_xy, float *powf_out);
/* The device's guarded core square root / reciprocal (pt_device.h: pt_sqrt, pt_rsqrt_glm, pt_rcp_pos) against the compiler's IEEE
 * expansions of sqrtf(x), 1 / sqrtf(x) and 1 / a, on ALL 2^32 operand bit patterns: mismatches[3] = how many patterns differ
 * bitwise (two NaNs count as equal).  0, 0, 0 is the only acceptable answer. */
int ptx_kat_fast_exact(ptx_tracer *t, int64_t mismatches[3]);
/* CPU-only (no device, no tracer): the per-tile geom masks the camera-ray bounce uses (bit g of masks_out[tile] clear = no camera ray of
 * that tile of 256 owned pixels can reach box g; boxes6 = lo xyz, hi xyz per geom, <= 32 geoms), for a camera, depth of field on / off and a
 * row-tile split.  Returns the number of tiles, -1 on a bad argument.  The CPU tests check the superset property ray by ray. */
int ptx_debug_tile_geoms(const ptx_camera *camera, int ngeoms, const float *boxes6, int depth_of_field, int tile_rows, int tile_rank,
                         int tile_world, uint32_t *masks_out, int max_tiles);
/* CPU-only: which material bins' stored paths carry what: masks[0] bit b = records of bin b carry the incoming direction
 * (reflective / refractive materials, materials of OBJ geoms), masks[1] bit b = they carry a 3-bit code instead of the normal (materials
 * only cubes have); bin = nmaterials - 1 - material when sorting by material, else 0; at most two runs of set bits in either.
 * geom_type: 0 sphere, 1 cube, 3 OBJ.  Returns 0, -1 on a bad argument (nmaterials 1..64). */
int ptx_debug_record_masks(int nmaterials, const ptx_material *materials, int ngeoms, const int32_t *geom_type, const int32_t *geom_material,
                           int sort_by_material, uint64_t masks[2]);
/* CPU-only: the table the candidate pre-test (the conservative world boxes every ray is tested against before the exact tests) reads on
 * the device, for n corner boxes (lo xyz, hi xyz): 8 floats per box = centre xyz, 0, half extent xyz, 0.  The CPU tests check that it
 * contains the corner box and that the device's slab arithmetic on it never rejects a ray that reaches the corner box. */
int ptx_debug_cull_boxes(int n, const float *boxes6, float *centre_half8);
/* Debug: workgroups of the specialised later-bounce kernel that fit a CU with lds_bytes of dynamic LDS each (0: what this tracer launches). */
int ptx_debug_bounce_occupancy(ptx_tracer *t, int lds_bytes);
/* Debug capture: the sorted stream of paths that will be shaded at bounce+1, as it stands after the given bounce
 * of the next iteration(s). */
int ptx_debug_set_capture(ptx_tracer *t, int bounce);   /* -1 = off */
/* CPU-only (no GPU call): the mesh BVH of csrc/pt_bvh.h against the reference's loop over all faces
   (src/intersections.h:213-233) on `nrays` object-space rays (6 floats: origin, direction).  stats4 = nodes, leaf
   triangles, nodes visited in total, rays on which the front-to-back traversal (the one k_mesh uses) disagrees with
   the skip-link traversal in face, distance or barycentrics (must be 0).  */
int ptx_debug_bvh_check(const float *faces15, int nfaces, const float *rays6, int nrays, int32_t *face_loop, float *t_loop,
                        int32_t *face_bvh, float *t_bvh, int64_t *stats4);
/* CPU-only: node visits of the last ptx_debug_bvh_check -- skip-link walk, front-to-back binary walk, four-wide walk (nodes),
   the stack entries the four-wide walk of that tree can need, the sum over groups of 64 consecutive rays of the longest four-wide
   walk in the group and the number of groups (what a wave of one-lane-per-ray walks costs), triangles tested by the four-wide walk;
   [7] unused */
int ptx_debug_bvh_visits(int64_t out8[8]);
/* zeros unless the library was built with -DPT_STAMPS (in-kernel phase timing, never in the shipped build) */
int ptx_debug_read_stamps(ptx_tracer *t, unsigned long long out48[48]);
/* fields14 (optional): 14 rows of min(n, cap) floats: px py pz (= origin + t*direction, the point that will be
 * shaded) dx dy dz cr cg cb nx ny nz u v (u, v only meaningful when the scene has textures) */
int ptx_debug_read_stream(ptx_tracer *t, int *n_out, int32_t *pixel_index, int32_t *stream_idx,
                          int32_t *material, float *fields14, int cap);

#ifdef __cplusplus
}
#endif
#endif
