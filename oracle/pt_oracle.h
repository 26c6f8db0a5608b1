/* oracle/pt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference path tracer's hot path (nkkk98/MyGPURaytracer, top-level src/).
 * It is the checker the HIP path is compared against; nothing in mygpuraytracer_amd/ may include, link or
 * call it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Pinning: oracle/_ref/libptref.so is the reference's own headers compiled from /root/reference (see
 * oracle/Makefile).  With o_set_libm(0) (glibc sinf/cosf/pow/powf, exactly what the reference's host pass
 * calls) this restatement is bit-identical to it on every fixture in tests/golden and on full frames
 * (tests/test_oracle_pin.py).  With o_set_libm(1) the four libm calls are replaced by the portable
 * IEEE-only routines the HIP kernels use (own_sincos / own_pow5 / own_powf below), so that GPU == oracle
 * bit for bit; the two modes differ by at most 1 ulp in those calls (the portable ones are correctly rounded,
 * glibc's sinf/cosf are off by one ulp on ~1.3 % of arguments: tests/test_own_libm.py) and make the same
 * hit/miss/material decisions on whole frames (tests/test_oracle_pin.py).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* geometry types, reference src/sceneStructs.h:10-15 */
enum { O_SPHERE = 0, O_CUBE = 1, O_TRIANGLE = 2, O_OBJ = 3 };

/* same memory layout as the reference's PathSegment (sceneStructs.h:103-108), 44 bytes */
typedef struct {
    float origin[3];
    float direction[3];
    float color[3];
    int pixelIndex;
    int remainingBounces;
} o_path;

/* same memory layout as ShadeableIntersection (sceneStructs.h:113-119), 32 bytes */
typedef struct {
    float t;
    float normal[3];
    int materialId;
    float texcoord[2];
    int geomId;
} o_isect;

void o_set_threads(int n);           /* threads of the per-path loops (results do not depend on it); default 1 */
void o_set_libm(int mode);            /* 0 = glibc (reference host semantics), 1 = portable own routines */
int o_get_libm(void);

/* hash + RNG (intersections.h:12-20, pathtrace.cu:62-66, thrust minstd_rand + uniform_real_distribution) */
unsigned o_utilhash(unsigned a);
void o_rng_raw(int iter, int index, int depth, int n, unsigned *out);
void o_rng_uniform(int iter, int index, int depth, float a, float b, int n, float *out);

/* own libm, exposed for the accuracy tests */
void o_own_sincosf(float x, float *s, float *c);
double o_own_pow5(double x);
float o_own_powf(float x, float y);

/* loader-side arithmetic (utilities.cpp:65-72, scene.cpp:301-304 / :364-374, main.cpp:56-70,105-123) */
void o_build_transforms(const float trs9[9], float out48[48]);
void o_camera_from_loader(int resx, int resy, float fovy, const float eye[3], const float lookat[3],
                          const float up[3], float f19[19]);
void o_runcuda_camera(float f19[19]);
/* camera controls of src/main.cpp:56-70,105-123,166-171,180-212 (orbit3 = phi, theta, zoom) */
void o_orbit_init(const float f19[19], float orbit3[3], float og_look_at[3]);
void o_orbit_left_drag(float orbit3[3], double dx, double dy, int width, int height);
void o_orbit_right_drag(float orbit3[3], double dy, int height);
void o_orbit_middle_drag(float f19[19], double dx, double dy);
void o_orbit_apply(float f19[19], const float orbit3[3]);

/* scene = POD arrays */
void *o_scene_create(int ngeoms, const int *gints3, const float *gmats48, int nmat, const float *mats11);
void o_scene_free(void *s);
void o_scene_set_faces(void *s, int gi, int nfaces, const float *faces15);
void o_scene_set_texture(void *s, int gi, int which, int w, int h, int ch, const unsigned char *data);
void o_scene_set_camera(void *s, const int res2[2], const float f19[19], int traceDepth);
void o_scene_set_options(void *s, int aa, int dof, int sort, int cache);
void o_scene_set_apps_variant(void *s, int on);   /* apps/src deltas: gather * PI, albedo AOV at iter 1 */
float *o_pt_albedo(void *s);
void o_scene_set_tile(void *s, int rows, int rank, int world);   /* multi-GPU row tiles; 0,0,1 = whole frame */

/* per-function entry points; same record layouts as oracle/ref_driver.cpp */
void o_geom_test(void *s, int gi, int n, const float *rays6, float *out10);
/* objTriIntersectionTest / triangleIntersectionLocalTest (src/intersections.h:175-205, 284-315: dead code in the reference) */
void o_obj_tri_test(void *s, int gi, int n, const float *rays6, float *out8);
/* calculateJitteredDirectionHemisphere (src/interactions.h:46-85: dead code in the reference) */
void o_jittered_test(int n, const float *normals3, const int *seeds3, int max_iter, float *out3);
void o_compute_intersections(void *s, int n, const o_path *paths, o_isect *out);
void o_shade(void *s, int iter, int depth, int n, const int *idx, const o_isect *isects, o_path *paths);

/* iteration driver (pathtrace.cu:433-558) */
void o_pt_init(void *s);
void o_pt_generate(void *s, int iter);
int o_pt_bounce(void *s, int iter, int stage_mask);
void o_pt_final_gather(void *s);
int o_pt_iterate(void *s, int iter);
int o_pt_live_counts(void *s, int *out, int cap);
o_path *o_pt_paths(void *s);
o_isect *o_pt_isects(void *s);
float *o_pt_image(void *s);
int o_pt_num_paths(void *s);
int o_pt_pixelcount(void *s);      /* pixels this instance traces (its tile) */
int o_pt_framepixels(void *s);     /* W*H: size of the image buffer */
void o_pt_pbo(void *s, int iter, unsigned char *pbo);
/* seconds spent per stage since o_pt_init: intersect, sort, shade, compact, generate, gather */
void o_pt_stage_seconds(void *s, double out6[6]);

/* StreamCompaction::CPU (stream_compaction/cpu.cu:20-95) */
void o_sc_scan(int n, int *odata, const int *idata);
int o_sc_compact_without_scan(int n, int *odata, const int *idata);
int o_sc_compact_with_scan(int n, int *odata, const int *idata);

#ifdef __cplusplus
}
#endif
#endif
