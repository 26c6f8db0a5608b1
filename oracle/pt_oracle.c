/* oracle/pt_oracle.c -- TEST INFRASTRUCTURE ONLY (see pt_oracle.h).
 *
 * CPU restatement of the reference path tracer's hot path, one C function per reference function, each citing
 * the reference file:line it follows (paths relative to /root/reference).  Arithmetic is written operation by
 * operation in the order glm 0.9.6.3 evaluates it (apps/external/include/glm), compiled with
 * -ffp-contract=off so that every float operation is a single IEEE-754 binary32 operation.
 */
#define _GNU_SOURCE
#include "pt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------------ */
/* small vector helpers (glm semantics: detail/func_geometric.inl, detail/type_vec3.inl)                     */
typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
static inline void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* func_geometric.inl:65-72: tmp = x*y; tmp.x + tmp.y + tmp.z */
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* func_geometric.inl:134-141 */
static inline v3 cross3(v3 x, v3 y) {
    return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* func_geometric.inl:154-159 + func_exponential.inl:62-68: x * (1 / sqrt(dot(x,x))) */
static inline v3 normalize3(v3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline float fminq(float x, float y) { return x < y ? x : y; }   /* func_common.inl:409-414 */
static inline float fmaxq(float x, float y) { return x > y ? x : y; }   /* func_common.inl:428-433 */

/* mat4 is 16 floats in glm memory order: m[c*4 + r] = m[c][r] */
/* intersections.h:34-36 + type_mat4x4.inl:617-628: (m[0]*v0 + m[1]*v1) + (m[2]*v2 + m[3]*v3), xyz only */
static inline v3 multiplyMV(const float *m, v3 v, float w) {
    v3 r;
    r.x = (m[0] * v.x + m[4] * v.y) + (m[8] * v.z + m[12] * w);
    r.y = (m[1] * v.x + m[5] * v.y) + (m[9] * v.z + m[13] * w);
    r.z = (m[2] * v.x + m[6] * v.y) + (m[10] * v.z + m[14] * w);
    return r;
}

/* ------------------------------------------------------------------------------------------------------ */
/* libm switch                                                                                              */
static int g_libm_mode = 0;
void o_set_libm(int mode) { g_libm_mode = mode; }

/* Threads for the two per-path loops (intersect, shade) of o_pt_bounce: the paths of a stage are independent, so the
 * results do not depend on it.  1 (the default) is the single-thread baseline; > 1 is bench.py's clearly labelled
 * all-cores figure.  The sort, the partition and the gather stay serial. */
static int g_threads = 1;
void o_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int o_get_libm(void) { return g_libm_mode; }

/* Portable sin/cos for float arguments: Cody-Waite reduction by pi/2 and Taylor polynomials, all in binary64
 * with one final rounding to binary32.  Only +, -, *, fused multiply-add (an IEEE operation: fma() here, v_fma_f64 there) and int
 * conversion are used, so the HIP kernels can run the same sequence (mygpuraytracer_amd/csrc carries its own copy; the product never
 * includes this file).  Round 4 fused the Horner steps; on every binary32 with |x| <= 2 pi the results equal the unfused form's
 * (tools/sincos_fma_exhaustive.c), so the fixtures made with the unfused form stand.
 * Domain: |x| <= 1e5 (the path tracer calls it on [0, 2*pi] and [-pi/4, pi]); outside, NaN.               */
static void own_sincos_d(double x, double *s, double *c) {
    static const double INVPIO2 = 0x1.45f306dc9c883p-1;
    static const double PIO2_1 = 0x1.921fb54400000p+0;    /* first 33 bits of pi/2 */
    static const double PIO2_1T = 0x1.0b4611a626331p-34;  /* pi/2 - PIO2_1 */
    if (!(x >= -1.0e5 && x <= 1.0e5)) { *s = NAN; *c = NAN; return; }
    double y = x * INVPIO2;
    int k = (int)(y + (y >= 0.0 ? 0.5 : -0.5));
    double kd = (double)k;
    double r = fma(-kd, PIO2_1T, fma(-kd, PIO2_1, x));
    double z = r * r;
    double ps = -0x1.ae7f3e733b81fp-41 + z * 0x1.952c77030ad4ap-49;      /* (the innermost steps stay unfused: see the device's copy) */
    ps = fma(z, ps, 0x1.6124613a86d09p-33);
    ps = fma(z, ps, -0x1.ae64567f544e4p-26);
    ps = fma(z, ps, 0x1.71de3a556c734p-19);
    ps = fma(z, ps, -0x1.a01a01a01a01ap-13);
    ps = fma(z, ps, 0x1.1111111111111p-7);
    ps = fma(z, ps, -0x1.5555555555555p-3);
    double sr = fma(r, z * ps, r);
    double pc = -0x1.93974a8c07c9dp-37 + z * 0x1.ae7f3e733b81fp-45;
    pc = fma(z, pc, 0x1.1eed8eff8d898p-29);
    pc = fma(z, pc, -0x1.27e4fb7789f5cp-22);
    pc = fma(z, pc, 0x1.a01a01a01a01ap-16);
    pc = fma(z, pc, -0x1.6c16c16c16c17p-10);
    pc = fma(z, pc, 0x1.5555555555555p-5);
    pc = fma(z, pc, -0x1.0000000000000p-1);
    double cr = fma(z, pc, 1.0);
    switch (k & 3) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
    }
}

void o_own_sincosf(float x, float *s, float *c) {
    double sd, cd;
    own_sincos_d((double)x, &sd, &cd);
    *s = (float)sd;
    *c = (float)cd;
}

double o_own_pow5(double x) {
    double x2 = x * x;
    double x4 = x2 * x2;
    return x4 * x;
}

static inline double dbl_from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t bits_from_dbl(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

/* Portable powf for x >= 0: exp(y*log(x)) in binary64 (atanh series for log, Taylor for exp), one rounding to
 * binary32 at the end.  powf(x,0)=1 for every x including NaN, as C99 requires.                            */
float o_own_powf(float xf, float yf) {
    if (yf == 0.0f) return 1.0f;
    if (xf != xf || yf != yf) return NAN;
    if (xf == 1.0f) return 1.0f;
    if (xf < 0.0f) return NAN;                       /* never reached by the path tracer (base is max(.,0)) */
    if (xf == 0.0f) return yf > 0.0f ? 0.0f : INFINITY;
    if (isinf(xf)) return yf > 0.0f ? INFINITY : 0.0f;
    if (isinf(yf)) return ((xf > 1.0f) == (yf > 0.0f)) ? INFINITY : 0.0f;
    double x = (double)xf;                            /* normal binary64 even for subnormal floats */
    uint64_t u = bits_from_dbl(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = dbl_from_bits((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);   /* [1,2) */
    if (m > 0x1.6a09e667f3bcdp+0) { m = m * 0.5; e += 1; }                          /* [sqrt(.5), sqrt(2)) */
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 0x1.af286bca1af28p-4 + z * 0x1.8618618618618p-4;
    p = 0x1.e1e1e1e1e1e1ep-4 + z * p;
    p = 0x1.1111111111111p-3 + z * p;
    p = 0x1.3b13b13b13b14p-3 + z * p;
    p = 0x1.745d1745d1746p-3 + z * p;
    p = 0x1.c71c71c71c71cp-3 + z * p;
    p = 0x1.2492492492492p-2 + z * p;
    p = 0x1.999999999999ap-2 + z * p;
    p = 0x1.5555555555555p-1 + z * p;
    double logm = 2.0 * s + s * (z * p);
    static const double LN2_HI = 0x1.62e42fee00000p-1, LN2_LO = 0x1.a39ef35793c76p-33;
    double ed = (double)e;
    double lg = (ed * LN2_HI + logm) + ed * LN2_LO;
    double a = (double)yf * lg;
    if (a > 89.0) return INFINITY;
    if (a < -104.0) return 0.0f;
    double kk = a * 0x1.71547652b82fep+0;
    int k = (int)(kk + (kk >= 0.0 ? 0.5 : -0.5));
    double kd = (double)k;
    double r = (a - kd * LN2_HI) - kd * LN2_LO;
    double q = 0x1.1eed8eff8d898p-29 + r * 0x1.6124613a86d09p-33;
    q = 0x1.ae64567f544e4p-26 + r * q;
    q = 0x1.27e4fb7789f5cp-22 + r * q;
    q = 0x1.71de3a556c734p-19 + r * q;
    q = 0x1.a01a01a01a01ap-16 + r * q;
    q = 0x1.a01a01a01a01ap-13 + r * q;
    q = 0x1.6c16c16c16c17p-10 + r * q;
    q = 0x1.1111111111111p-7 + r * q;
    q = 0x1.5555555555555p-5 + r * q;
    q = 0x1.5555555555555p-3 + r * q;
    q = 0x1.0000000000000p-1 + r * q;
    double er = 1.0 + (r + r * (r * q));
    double two_k = dbl_from_bits((uint64_t)(k + 1023) << 52);   /* |k| <= 151 here */
    return (float)(er * two_k);
}

static inline float lm_cosf(float x) {
    if (g_libm_mode == 0) return cosf(x);
    float s, c; o_own_sincosf(x, &s, &c); return c;
}
static inline float lm_sinf(float x) {
    if (g_libm_mode == 0) return sinf(x);
    float s, c; o_own_sincosf(x, &s, &c); return s;
}
static inline double lm_pow5(double x) { return g_libm_mode == 0 ? pow(x, 5.0) : o_own_pow5(x); }
static inline float lm_powf(float x, float y) { return g_libm_mode == 0 ? powf(x, y) : o_own_powf(x, y); }

/* ------------------------------------------------------------------------------------------------------ */
/* hash + RNG                                                                                               */
/* intersections.h:12-20 */
unsigned o_utilhash(unsigned a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

/* thrust::minstd_rand = linear_congruential_engine<uint32, 48271, 0, 2147483647>
 * (thrust/random/detail/linear_congruential_engine.inl: seed s -> s % m, 0 -> 1; step x -> a*x mod m)      */
typedef struct { uint32_t x; } o_rng;

/* pathtrace.cu:62-66: h = utilhash((1 << 31) | (depth << 22) | iter) ^ utilhash(index) */
static o_rng make_seeded_engine(int iter, int index, int depth) {
    unsigned h = o_utilhash((1u << 31) | ((unsigned)depth << 22) | (unsigned)iter) ^ o_utilhash((unsigned)index);
    o_rng r;
    r.x = h % 2147483647u;
    if (r.x == 0) r.x = 1;
    return r;
}
static inline uint32_t rng_next(o_rng *r) {
    r->x = (uint32_t)(((uint64_t)r->x * 48271u) % 2147483647u);
    return r->x;
}
/* thrust/random/detail/uniform_real_distribution.inl:
 *   result = float(urng() - min); result /= (1.0f + float(max - min)); return result*(b-a) + a
 * with min = 1, max = m-1: float(2147483645) = 2^31, + 1.0f = 2^31.                                       */
static inline float rng_uniform(o_rng *r, float a, float b) {
    float result = (float)(rng_next(r) - 1u);
    result /= (1.0f + (float)(2147483646u - 1u));
    return (result * (b - a)) + a;
}

void o_rng_raw(int iter, int index, int depth, int n, unsigned *out) {
    o_rng r = make_seeded_engine(iter, index, depth);
    for (int i = 0; i < n; i++) out[i] = rng_next(&r);
}
void o_rng_uniform(int iter, int index, int depth, float a, float b, int n, float *out) {
    o_rng r = make_seeded_engine(iter, index, depth);
    for (int i = 0; i < n; i++) out[i] = rng_uniform(&r, a, b);
}

/* ------------------------------------------------------------------------------------------------------ */
/* scene                                                                                                    */
typedef struct { int width, height, channels; unsigned char *image; } o_texture;
typedef struct {
    int type, materialid, faceSize;
    float transform[16], inverseTransform[16], invTranspose[16];
    float *faces;                      /* 15 floats per face: 3 x (position xyz, texcoord uv) */
    o_texture tex[4];                  /* kd, ks, ke, bump */
} o_geom;
typedef struct {                       /* sceneStructs.h:71-81, 44 bytes */
    float color[3];
    float exponent;
    float speccolor[3];
    float hasReflective, hasRefractive, indexOfRefraction, emittance;
} o_material;
typedef struct {
    int resx, resy;
    float position[3], lookAt[3], view[3], up[3], right[3], fov[2], pixelLength[2];
} o_camera;
typedef struct {
    int ngeoms, nmat;
    o_geom *geoms;
    o_material *mats;
    o_camera cam;
    int traceDepth;
    int opt_aa, opt_dof, opt_sort, opt_cache;
    int tile_rows, tile_rank, tile_world;   /* multi-GPU row-tile split (not in the reference): 0,0,1 = whole frame */
    int apps;                               /* 1 = deltas of the apps/src copy: gather * PI, albedo AOV */
    float *albedo;
    /* iteration state (pathtrace.cu:91-98) */
    int pixelcount, num_paths, depth;
    float *image;
    o_path *paths, *paths_tmp;
    o_isect *isects, *isects_tmp, *first_isects;
    int *flags, *scan, *perm;
    int live_counts[256];
    int nlive;
    double secs[6];
} o_scene;

void *o_scene_create(int ngeoms, const int *gints3, const float *gmats48, int nmat, const float *mats11) {
    o_scene *s = (o_scene *)calloc(1, sizeof(o_scene));
    s->ngeoms = ngeoms; s->nmat = nmat;
    s->geoms = (o_geom *)calloc((size_t)(ngeoms > 0 ? ngeoms : 1), sizeof(o_geom));
    s->mats = (o_material *)calloc((size_t)(nmat > 0 ? nmat : 1), sizeof(o_material));
    for (int i = 0; i < ngeoms; i++) {
        o_geom *g = &s->geoms[i];
        g->type = gints3[i * 3 + 0]; g->materialid = gints3[i * 3 + 1]; g->faceSize = 0;
        memcpy(g->transform, gmats48 + i * 48, 64);
        memcpy(g->inverseTransform, gmats48 + i * 48 + 16, 64);
        memcpy(g->invTranspose, gmats48 + i * 48 + 32, 64);
    }
    memcpy(s->mats, mats11, sizeof(o_material) * (size_t)nmat);
    s->opt_aa = 1; s->opt_dof = 0; s->opt_sort = 1; s->opt_cache = 1;   /* pathtrace.cu:36-40 */
    s->tile_rows = 0; s->tile_rank = 0; s->tile_world = 1;
    return s;
}
static void free_iter_state(o_scene *s) {
    free(s->image); free(s->albedo); free(s->paths); free(s->paths_tmp); free(s->isects); free(s->isects_tmp);
    free(s->first_isects); free(s->flags); free(s->scan); free(s->perm);
    s->image = NULL; s->albedo = NULL; s->paths = s->paths_tmp = NULL; s->isects = s->isects_tmp = s->first_isects = NULL;
    s->flags = s->scan = s->perm = NULL;
}
void o_scene_free(void *h) {
    o_scene *s = (o_scene *)h;
    if (!s) return;
    for (int i = 0; i < s->ngeoms; i++) {
        free(s->geoms[i].faces);
        for (int k = 0; k < 4; k++) free(s->geoms[i].tex[k].image);
    }
    free_iter_state(s);
    free(s->geoms); free(s->mats); free(s);
}
void o_scene_set_faces(void *h, int gi, int nfaces, const float *faces15) {
    o_geom *g = &((o_scene *)h)->geoms[gi];
    free(g->faces);
    g->faces = (float *)malloc(sizeof(float) * 15 * (size_t)(nfaces > 0 ? nfaces : 1));
    memcpy(g->faces, faces15, sizeof(float) * 15 * (size_t)nfaces);
    g->faceSize = nfaces;
}
void o_scene_set_texture(void *h, int gi, int which, int w, int hh, int ch, const unsigned char *data) {
    o_texture *t = &((o_scene *)h)->geoms[gi].tex[which];
    free(t->image);
    t->width = w; t->height = hh; t->channels = ch;
    size_t n = (size_t)w * (size_t)hh * (size_t)ch;
    t->image = (unsigned char *)malloc(n > 0 ? n : 1);
    memcpy(t->image, data, n);
}
void o_scene_set_camera(void *h, const int res2[2], const float f19[19], int traceDepth) {
    o_scene *s = (o_scene *)h;
    s->cam.resx = res2[0]; s->cam.resy = res2[1];
    memcpy(s->cam.position, f19, sizeof(float) * 19);
    s->traceDepth = traceDepth;
}
/* apps/src variant of the reference (the copy its CMake builds): finalGather multiplies by its own
 * #define PI 3.14159265358f (apps/src/pathtrace.cu:44,508) and the first shade of iteration 1 writes an albedo
 * AOV (apps/src/pathtrace.cu:412-462). */
void o_scene_set_apps_variant(void *h, int on) { ((o_scene *)h)->apps = on; }
float *o_pt_albedo(void *h) { return ((o_scene *)h)->albedo; }

/* Row-tile split used by the multi-GPU driver: this instance traces only the rows y with
 * (y / rows) % world == rank, as its own stream (local stream indices); pixelIndex stays global. */
void o_scene_set_tile(void *h, int rows, int rank, int world) {
    o_scene *s = (o_scene *)h;
    s->tile_rows = rows; s->tile_rank = rank; s->tile_world = world < 1 ? 1 : world;
}
static int row_owned(const o_scene *s, int y) {
    if (s->tile_world <= 1) return 1;
    return (y / s->tile_rows) % s->tile_world == s->tile_rank;
}

void o_scene_set_options(void *h, int aa, int dof, int sort, int cache) {
    o_scene *s = (o_scene *)h;
    s->opt_aa = aa; s->opt_dof = dof; s->opt_sort = sort; s->opt_cache = cache;
}

/* ------------------------------------------------------------------------------------------------------ */
/* intersections                                                                                            */
typedef struct { v3 origin, direction; } o_ray;

/* intersections.h:27-29: origin + (t - .0001f) * normalize(direction) */
static inline v3 getPointOnRay(o_ray r, float t) {
    v3 nd = normalize3(r.direction);
    float tt = t - .0001f;
    return add3(r.origin, V3(tt * nd.x, tt * nd.y, tt * nd.z));
}

/* intersections.h:48-90 */
static float boxIntersectionTest(const o_geom *box, o_ray r, v3 *intersectionPoint, v3 *normal, int *outside) {
    o_ray q;
    q.origin = multiplyMV(box->inverseTransform, r.origin, 1.0f);
    q.direction = normalize3(multiplyMV(box->inverseTransform, r.direction, 0.0f));
    float tmin = -1e38f;
    float tmax = 1e38f;
    float tmin_n[3] = {0.f, 0.f, 0.f};
    float tmax_n[3] = {0.f, 0.f, 0.f};
    const float qo[3] = {q.origin.x, q.origin.y, q.origin.z};
    const float qd[3] = {q.direction.x, q.direction.y, q.direction.z};
    for (int xyz = 0; xyz < 3; ++xyz) {
        float qdxyz = qd[xyz];
        float t1 = (-0.5f - qo[xyz]) / qdxyz;
        float t2 = (+0.5f - qo[xyz]) / qdxyz;
        float ta = fminq(t1, t2);
        float tb = fmaxq(t1, t2);
        float n[3] = {0.f, 0.f, 0.f};
        n[xyz] = t2 < t1 ? +1.f : -1.f;
        if (ta > 0 && ta > tmin) { tmin = ta; memcpy(tmin_n, n, 12); }
        if (tb < tmax) { tmax = tb; memcpy(tmax_n, n, 12); }
    }
    if (tmax >= tmin && tmax > 0) {
        *outside = 1;
        if (tmin <= 0) { tmin = tmax; memcpy(tmin_n, tmax_n, 12); *outside = 0; }
        *intersectionPoint = multiplyMV(box->transform, getPointOnRay(q, tmin), 1.0f);
        *normal = normalize3(multiplyMV(box->invTranspose, ld3(tmin_n), 0.0f));
        return length3(sub3(r.origin, *intersectionPoint));
    }
    return -1;
}

/* intersections.h:102-144 */
static float sphereIntersectionTest(const o_geom *sphere, o_ray r, v3 *intersectionPoint, v3 *normal, int *outside) {
    float radius = .5f;
    v3 ro = multiplyMV(sphere->inverseTransform, r.origin, 1.0f);
    v3 rd = normalize3(multiplyMV(sphere->inverseTransform, r.direction, 0.0f));
    o_ray rt; rt.origin = ro; rt.direction = rd;
    float vDotDirection = dot3(rt.origin, rt.direction);
    /* powf(radius, 2) with radius = .5f is exactly 0.25f in every libm */
    float radicand = vDotDirection * vDotDirection - (dot3(rt.origin, rt.origin) - radius * radius);
    if (radicand < 0) return -1;
    float squareRoot = sqrtf(radicand);
    float firstTerm = -vDotDirection;
    float t1 = firstTerm + squareRoot;
    float t2 = firstTerm - squareRoot;
    float t = 0;
    if (t1 < 0 && t2 < 0) {
        return -1;
    } else if (t1 > 0 && t2 > 0) {
        t = fminq(t1, t2);
        *outside = 1;
    } else {
        t = fmaxq(t1, t2);
        *outside = 0;
    }
    v3 objspaceIntersection = getPointOnRay(rt, t);
    *intersectionPoint = multiplyMV(sphere->transform, objspaceIntersection, 1.f);
    *normal = normalize3(multiplyMV(sphere->invTranspose, objspaceIntersection, 0.f));
    if (!*outside) *normal = neg3(*normal);
    return length3(sub3(r.origin, *intersectionPoint));
}

/* glm/gtx/intersect.inl:37-74 (single-sided Moeller-Trumbore) */
static int intersectRayTriangle(v3 orig, v3 dir, v3 v0, v3 v1, v3 v2, float bary[3]) {
    v3 e1 = sub3(v1, v0);
    v3 e2 = sub3(v2, v0);
    v3 p = cross3(dir, e2);
    float a = dot3(e1, p);
    if (a < FLT_EPSILON) return 0;
    float f = 1.0f / a;
    v3 s = sub3(orig, v0);
    bary[0] = f * dot3(s, p);
    if (bary[0] < 0.0f) return 0;
    if (bary[0] > 1.0f) return 0;
    v3 q = cross3(s, e1);
    bary[1] = f * dot3(dir, q);
    if (bary[1] < 0.0f) return 0;
    if (bary[1] + bary[0] > 1.0f) return 0;
    bary[2] = f * dot3(e2, q);
    return bary[2] >= 0.0f;
}

/* Texel fetch as the reference indexes it (interactions.h:172-179 etc.): no wrap; out-of-range indices, which
 * are undefined behaviour in the reference, are given defined behaviour here by clamping the byte index.  */
static inline unsigned texel(const o_texture *t, int pixelID, int c) {
    long long idx = (long long)pixelID * t->channels + c;
    long long n = (long long)t->width * t->height * t->channels;
    if (idx < 0) idx = 0;
    if (idx >= n) idx = n - 1;
    return (unsigned)t->image[idx];
}

/* intersections.h:207-282 */
static float meshIntersectionTest(const o_geom *geom, o_ray r, v3 *intersectionPoint, v3 *normal, float texcoord[2],
                                  int *outside) {
    o_ray q;
    q.origin = multiplyMV(geom->inverseTransform, r.origin, 1.0f);
    q.direction = normalize3(multiplyMV(geom->inverseTransform, r.direction, 0.0f));
    float tmin = FLT_MAX;
    int nearest = -1;
    for (int j = 0; j < geom->faceSize; j++) {
        const float *tri = geom->faces + j * 15;
        v3 p0 = ld3(tri), p1 = ld3(tri + 5), p2 = ld3(tri + 10);
        float bary[3];
        if (intersectRayTriangle(q.origin, q.direction, p0, p1, p2, bary)) {
            float w = 1 - bary[0] - bary[1];
            v3 p = add3(add3(scale3(p0, w), scale3(p1, bary[0])), scale3(p2, bary[1]));
            float t = length3(sub3(q.origin, p));     /* glm::distance(p, q.origin) = length(q.origin - p) */
            if (t < tmin) {
                tmin = t;
                nearest = j;
                texcoord[0] = (w * tri[3] + bary[0] * tri[8]) + bary[1] * tri[13];
                texcoord[1] = (w * tri[4] + bary[0] * tri[9]) + bary[1] * tri[14];
            }
        }
    }
    if (nearest == -1) return -1;
    v3 objspaceIntersection = getPointOnRay(q, tmin);
    const float *tri = geom->faces + nearest * 15;
    v3 e1 = sub3(ld3(tri + 5), ld3(tri));
    v3 e2 = sub3(ld3(tri + 10), ld3(tri));
    v3 objspaceNormal = normalize3(cross3(e1, e2));
    *intersectionPoint = multiplyMV(geom->transform, objspaceIntersection, 1.f);
    *normal = normalize3(multiplyMV(geom->invTranspose, objspaceNormal, 0.f));
    *outside = dot3(*normal, r.direction) < 0;
    if (geom->type == O_OBJ && geom->tex[3].channels) {
        const o_texture *bump = &geom->tex[3];
        float dUV1x = tri[8] - tri[3], dUV1y = tri[9] - tri[4];
        float dUV2x = tri[13] - tri[3], dUV2y = tri[14] - tri[4];
        float f = 1.0f / (dUV1x * dUV2y - dUV2x * dUV1y);
        v3 tangent, bitangent;
        tangent.x = f * (dUV2y * e1.x - dUV1y * e2.x);
        tangent.y = f * (dUV2y * e1.y - dUV1y * e2.y);
        tangent.z = f * (dUV2y * e1.z - dUV1y * e2.z);
        tangent = normalize3(tangent);
        bitangent.x = f * (-dUV2x * e1.x + dUV1x * e2.x);
        bitangent.y = f * (-dUV2x * e1.y + dUV1x * e2.y);
        bitangent.z = f * (-dUV2x * e1.z + dUV1x * e2.z);
        bitangent = normalize3(bitangent);
        v3 T = normalize3(multiplyMV(geom->transform, tangent, 0.f));
        v3 B = normalize3(multiplyMV(geom->transform, bitangent, 0.f));
        v3 N = *normal;
        int coordU = (int)(texcoord[0] * bump->width);
        int coordV = (int)(texcoord[1] * bump->height);
        int pixelID = coordV * bump->width + coordU;
        unsigned colR = texel(bump, pixelID, 0), colG = texel(bump, pixelID, 1), colB = texel(bump, pixelID, 2);
        v3 tsn = normalize3(V3(colR / 255.f, colG / 255.f, colB / 255.f));
        tsn = normalize3(V3(tsn.x * 2.0f - 1.0f, tsn.y * 2.0f - 1.0f, tsn.z * 2.0f - 1.0f));
        /* mat3(T,B,N) * v : type_mat3x3.inl:487-493 */
        v3 w3 = V3(T.x * tsn.x + B.x * tsn.y + N.x * tsn.z,
                   T.y * tsn.x + B.y * tsn.y + N.y * tsn.z,
                   T.z * tsn.x + B.z * tsn.y + N.z * tsn.z);
        *normal = normalize3(w3);
    }
    return tmin;
}

/* triangleIntersectionLocalTest, src/intersections.h:175-205 -- DEAD CODE in the reference (its only caller is objTriIntersectionTest,
 * whose call in computeIntersections is commented out, src/pathtrace.cu:313).  Restated for SURVEY 8(a10) as a known-answer function
 * only; nothing on the path calls it.  Plane hit by the ray, then the three sub-triangle areas against the triangle's. */
static float triangleIntersectionLocalTest(v3 ro, v3 rd, v3 v0, v3 v1, v3 v2, v3 *intersectionPoint, v3 *normal) {
    v3 planeNormal = normalize3(cross3(sub3(v1, v0), sub3(v2, v0)));
    float t = dot3(planeNormal, sub3(v0, ro)) / dot3(planeNormal, rd);
    if (t < 0) return -1;
    v3 p = add3(ro, scale3(rd, t));                       /* ro + t * rd */
    float S = 0.5f * length3(cross3(sub3(v0, v1), sub3(v0, v2)));
    float s1 = 0.5f * length3(cross3(sub3(p, v1), sub3(p, v2))) / S;
    float s2 = 0.5f * length3(cross3(sub3(p, v2), sub3(p, v0))) / S;
    float s3 = 0.5f * length3(cross3(sub3(p, v0), sub3(p, v1))) / S;
    float sum = s1 + s2 + s3;
    if (s1 >= 0 && s1 <= 1 && s2 >= 0 && s2 <= 1 && s3 >= 0 && s3 <= 1 && fabsf(sum - 1.0f) < FLT_EPSILON) {
        *intersectionPoint = p;
        *normal = planeNormal;
        return t;
    }
    return -1;
}

/* objTriIntersectionTest, src/intersections.h:284-315 (dead, see above): nearest face by the local test in object space; returns the
 * OBJECT-space distance (unlike meshIntersectionTest), the world-space point and normal of that face. */
static float objTriIntersectionTest(const o_geom *geom, o_ray r, v3 *intersectionPoint, v3 *normal, int *outside) {
    float min_tri_t = FLT_MAX;
    v3 tmp_tri_intersect = V3(0, 0, 0), tmp_tri_normal = V3(0, 0, 0), min_tri_intersect = V3(0, 0, 0), min_tri_normal = V3(0, 0, 0);
    int nearest = -1;
    o_ray q;
    q.origin = multiplyMV(geom->inverseTransform, r.origin, 1.0f);
    q.direction = normalize3(multiplyMV(geom->inverseTransform, r.direction, 0.0f));
    for (int j = 0; j < geom->faceSize; j++) {
        const float *tri = geom->faces + j * 15;
        float tmp_tri_t = triangleIntersectionLocalTest(q.origin, q.direction, ld3(tri), ld3(tri + 5), ld3(tri + 10), &tmp_tri_intersect, &tmp_tri_normal);
        if (tmp_tri_t > 0 && tmp_tri_t < min_tri_t) {
            min_tri_intersect = tmp_tri_intersect;
            min_tri_normal = tmp_tri_normal;
            min_tri_t = tmp_tri_t;
            nearest = j;
        }
    }
    if (nearest == -1) return -1;
    *intersectionPoint = multiplyMV(geom->transform, min_tri_intersect, 1.f);
    *normal = normalize3(multiplyMV(geom->invTranspose, min_tri_normal, 0.f));
    *outside = dot3(*normal, r.direction) < 0;
    return min_tri_t;
}

/* out per ray: t, point(3), normal(3), outside = 8 floats (point / normal / outside as they were passed in when there is no hit: 0, 0, 1) */
void o_obj_tri_test(void *h, int gi, int n, const float *rays6, float *out8) {
    o_scene *s = (o_scene *)h;
    const o_geom *g = &s->geoms[gi];
    for (int i = 0; i < n; i++) {
        o_ray r; r.origin = ld3(rays6 + i * 6); r.direction = ld3(rays6 + i * 6 + 3);
        v3 p = V3(0, 0, 0), nrm = V3(0, 0, 0);
        int outside = 1;
        float t = objTriIntersectionTest(g, r, &p, &nrm, &outside);
        float *o = out8 + i * 8;
        o[0] = t; st3(o + 1, p); st3(o + 4, nrm); o[7] = outside ? 1.f : 0.f;
    }
}

void o_geom_test(void *h, int gi, int n, const float *rays6, float *out10) {
    o_scene *s = (o_scene *)h;
    const o_geom *g = &s->geoms[gi];
    for (int i = 0; i < n; i++) {
        o_ray r; r.origin = ld3(rays6 + i * 6); r.direction = ld3(rays6 + i * 6 + 3);
        v3 p = V3(0, 0, 0), nrm = V3(0, 0, 0);
        float uv[2] = {0.f, 0.f};
        int outside = 1;
        float t = -1.f;
        if (g->type == O_CUBE) t = boxIntersectionTest(g, r, &p, &nrm, &outside);
        else if (g->type == O_SPHERE) t = sphereIntersectionTest(g, r, &p, &nrm, &outside);
        else if (g->type == O_OBJ) t = meshIntersectionTest(g, r, &p, &nrm, uv, &outside);
        float *o = out10 + i * 10;
        o[0] = t; st3(o + 1, p); st3(o + 4, nrm); o[7] = uv[0]; o[8] = uv[1]; o[9] = outside ? 1.f : 0.f;
    }
}

/* pathtrace.cu:270-343, one path */
static void compute_intersection_one(const o_scene *s, const o_path *pathSegment, o_isect *dst) {
    o_ray ray; ray.origin = ld3(pathSegment->origin); ray.direction = ld3(pathSegment->direction);
    float t = 0.f;
    v3 normal = V3(0, 0, 0);
    float t_min = FLT_MAX;
    int hit_geom_index = -1;
    int outside = 1;
    float uv[2] = {0.f, 0.f};
    v3 tmp_intersect = V3(0, 0, 0), tmp_normal = V3(0, 0, 0);
    float tmp_uv[2] = {0.f, 0.f};
    for (int i = 0; i < s->ngeoms; i++) {
        const o_geom *geom = &s->geoms[i];
        if (geom->type == O_CUBE) t = boxIntersectionTest(geom, ray, &tmp_intersect, &tmp_normal, &outside);
        else if (geom->type == O_SPHERE) t = sphereIntersectionTest(geom, ray, &tmp_intersect, &tmp_normal, &outside);
        else if (geom->type == O_OBJ) t = meshIntersectionTest(geom, ray, &tmp_intersect, &tmp_normal, tmp_uv, &outside);
        if (t > 0.0f && t_min > t) {
            t_min = t;
            hit_geom_index = i;
            normal = tmp_normal;
            uv[0] = tmp_uv[0]; uv[1] = tmp_uv[1];
        }
    }
    if (hit_geom_index == -1) {
        dst->t = -1.0f;
    } else {
        dst->t = t_min;
        dst->materialId = s->geoms[hit_geom_index].materialid;
        st3(dst->normal, normal);
        dst->geomId = hit_geom_index;
        dst->texcoord[0] = uv[0]; dst->texcoord[1] = uv[1];
    }
}

void o_compute_intersections(void *h, int n, const o_path *paths, o_isect *out) {
    o_scene *s = (o_scene *)h;
    memset(out, 0, sizeof(o_isect) * (size_t)n);
    for (int i = 0; i < n; i++) compute_intersection_one(s, &paths[i], &out[i]);
}

/* ------------------------------------------------------------------------------------------------------ */
/* BSDF                                                                                                     */
#define O_TWO_PI 6.2831853071795864769252867665590057683943f            /* utilities.h:13 */
#define O_SQRT_OF_ONE_THIRD 0.5773502691896257645091487805019574556476f /* utilities.h:14 */

/* interactions.h:11-43 */
static v3 calculateRandomDirectionInHemisphere(v3 normal, o_rng *rng) {
    float up = sqrtf(rng_uniform(rng, 0, 1));
    float over = sqrtf(1 - up * up);
    float around = rng_uniform(rng, 0, 1) * O_TWO_PI;
    v3 directionNotNormal;
    if (fabsf(normal.x) < O_SQRT_OF_ONE_THIRD) directionNotNormal = V3(1, 0, 0);
    else if (fabsf(normal.y) < O_SQRT_OF_ONE_THIRD) directionNotNormal = V3(0, 1, 0);
    else directionNotNormal = V3(0, 0, 1);
    v3 perpendicularDirection1 = normalize3(cross3(normal, directionNotNormal));
    v3 perpendicularDirection2 = normalize3(cross3(normal, perpendicularDirection1));
    float c = lm_cosf(around), sn = lm_sinf(around);
    v3 a = scale3(normal, up);
    v3 b = scale3(perpendicularDirection1, c * over);
    v3 d = scale3(perpendicularDirection2, sn * over);
    return add3(add3(a, b), d);
}

/* calculateJitteredDirectionHemisphere, src/interactions.h:46-85 -- DEAD CODE in the reference (JITTERED_SAMPLING 0, :5; the block that
 * would call it, :243-251, does not even compile: it names variables scatterRay does not have).  The function itself compiles, so it is
 * restated for SURVEY 8(a13) as a known-answer function only; nothing on the path calls it.  A stratified variant of the sampler above:
 * cell (iter % n, iter / n) of an n x n grid over the unit square, n = round(sqrt(max_iter)), jittered inside the cell. */
static v3 calculateJitteredDirectionHemisphere(v3 normal, o_rng *rng, int iter, int max_iter) {
    int samples = max_iter;
    int sqrtVal = (int)(sqrtf((float)samples) + 0.5f);
    float invSqrtVal = 1.f / (float)sqrtVal;
    int x = iter % sqrtVal;
    int y = (int)((float)(iter) / (float)sqrtVal);
    float x_point = (float)x + rng_uniform(rng, 0, 1);
    x_point = x_point * invSqrtVal;
    x_point = x_point < 0.f ? 0.f : x_point; x_point = 1.f < x_point ? 1.f : x_point;      /* glm::clamp = min(max(x, 0), 1) */
    float y_point = (float)y + rng_uniform(rng, 0, 1);
    y_point = y_point * invSqrtVal;
    y_point = y_point < 0.f ? 0.f : y_point; y_point = 1.f < y_point ? 1.f : y_point;
    float up = sqrtf(y_point);
    float over = sqrtf(1.f - (up * up));
    float around = x_point * O_TWO_PI;
    v3 directionNotNormal;
    if (fabsf(normal.x) < O_SQRT_OF_ONE_THIRD) directionNotNormal = V3(1, 0, 0);
    else if (fabsf(normal.y) < O_SQRT_OF_ONE_THIRD) directionNotNormal = V3(0, 1, 0);
    else directionNotNormal = V3(0, 0, 1);
    v3 perpendicularDirection1 = normalize3(cross3(normal, directionNotNormal));
    v3 perpendicularDirection2 = normalize3(cross3(normal, perpendicularDirection1));
    float c = lm_cosf(around), sn = lm_sinf(around);
    return add3(add3(scale3(normal, up), scale3(perpendicularDirection1, c * over)), scale3(perpendicularDirection2, sn * over));
}
/* per sample: normal(3) in, (iter, index, depth) seed the engine as makeSeededRandomEngine does, direction(3) out */
void o_jittered_test(int n, const float *normals3, const int *seeds3, int max_iter, float *out3) {
    for (int i = 0; i < n; i++) {
        o_rng r = make_seeded_engine(seeds3[i * 3], seeds3[i * 3 + 1], seeds3[i * 3 + 2]);
        st3(out3 + i * 3, calculateJitteredDirectionHemisphere(ld3(normals3 + i * 3), &r, seeds3[i * 3], max_iter));
    }
}

/* glm reflect: I - N * dot(N, I) * 2   (func_geometric.inl:175-178) */
static inline v3 reflect3(v3 I, v3 N) { return sub3(I, scale3(scale3(N, dot3(N, I)), 2.0f)); }
/* glm refract (func_geometric.inl:190-197) */
static inline v3 refract3(v3 I, v3 N, float eta) {
    float dotValue = dot3(N, I);
    float k = 1.0f - eta * eta * (1.0f - dotValue * dotValue);
    float f = eta * dotValue + sqrtf(k);
    v3 r = sub3(scale3(I, eta), scale3(N, f));
    return scale3(r, (float)(k >= 0.0f));
}

/* Schlick term as the reference writes it (interactions.h:151-152, :190-191):
 *   float r0 = ((n1-n2)/(n1+n2))*((n1-n2)/(n1+n2));  float c = r0 + (1.0f - r0) * pow((1.0 - cosTheta), 5);
 * the product and the sum are binary64, rounded to binary32 on assignment.                                 */
static inline float schlick(float IoR1, float IoR2, float cosTheta) {
    float r0 = ((IoR1 - IoR2) / (IoR1 + IoR2)) * ((IoR1 - IoR2) / (IoR1 + IoR2));
    return (float)((double)r0 + (double)(1.0f - r0) * lm_pow5(1.0 - (double)cosTheta));
}

/* interactions.h:111-256 */
static void scatterRay(const o_scene *s, o_path *pathSegment, v3 intersect, o_isect intersection, const o_material *m,
                       o_rng *rng) {
    v3 dir = ld3(pathSegment->direction);
    v3 color = ld3(pathSegment->color);
    v3 n = ld3(intersection.normal);
    if (m->hasReflective > 0) {
        v3 reflectDir = reflect3(dir, n);
        float spec = lm_powf(fmaxq(dot3(neg3(dir), reflectDir), 0.0f), m->exponent);
        color = mul3(color, scale3(ld3(m->speccolor), m->hasReflective * spec));
        st3(pathSegment->color, color);
        st3(pathSegment->origin, add3(intersect, scale3(n, 0.01f)));
        st3(pathSegment->direction, reflectDir);
    } else if (m->hasRefractive > 0) {
        float IoR1 = 1.0f;
        float IoR2 = m->indexOfRefraction;
        float cosTheta = dot3(neg3(dir), n);
        if (cosTheta < 0) {
            n = scale3(n, -1.0f);
            IoR1 = IoR2;
            IoR2 = 1.0f;
            cosTheta = fabsf(cosTheta);
        }
        float sinTheta = (float)sqrt(1.0 - (double)(cosTheta * cosTheta));
        v3 newdir;
        if (IoR1 / IoR2 * sinTheta > 1.0f) {
            newdir = reflect3(dir, n);
        } else {
            float reflect_coeff = schlick(IoR1, IoR2, cosTheta);
            float random = rng_uniform(rng, 0, 1);
            if (random < reflect_coeff) newdir = reflect3(dir, n);
            else newdir = refract3(dir, n, IoR1 / IoR2);
        }
        st3(pathSegment->direction, newdir);
        st3(pathSegment->color, mul3(color, ld3(m->speccolor)));
        st3(pathSegment->origin, add3(intersect, scale3(newdir, 0.01f)));
    } else if (s->geoms[intersection.geomId].type == O_OBJ) {
        const o_geom *geom = &s->geoms[intersection.geomId];
        const o_texture *kd = &geom->tex[0], *ks = &geom->tex[1], *ke = &geom->tex[2];
        v3 emission = V3(0.f, 0.f, 0.f);
        if (ke->channels) {
            int coordU = (int)(intersection.texcoord[0] * ke->width);
            int coordV = (int)(intersection.texcoord[1] * ke->height);
            int pixelID = coordV * ke->width + coordU;
            emission = V3(texel(ke, pixelID, 0) / 255.f, texel(ke, pixelID, 1) / 255.f, texel(ke, pixelID, 2) / 255.f);
        }
        if (emission.x > FLT_EPSILON || emission.y > FLT_EPSILON || emission.z > FLT_EPSILON) {
            st3(pathSegment->color, mul3(color, scale3(emission, 5.0f)));
            pathSegment->remainingBounces = 1;
            return;
        }
        float IoR1 = 1.0f;
        float IoR2 = m->indexOfRefraction;
        float cosTheta = dot3(neg3(dir), n);
        float reflect_coeff = schlick(IoR1, IoR2, cosTheta);
        float random = rng_uniform(rng, 0, 1);
        if (random < reflect_coeff) {
            int coordU = (int)(intersection.texcoord[0] * ks->width);
            int coordV = (int)(intersection.texcoord[1] * ks->height);
            int pixelID = coordV * ks->width + coordU;
            v3 reflectDir = reflect3(dir, n);
            float spec = lm_powf(fmaxq(dot3(neg3(dir), reflectDir), 0.0f), 0.0f);
            v3 specColor;
            if (ks->channels)
                specColor = V3(texel(ks, pixelID, 0) / 255.f, texel(ks, pixelID, 1) / 255.f, texel(ks, pixelID, 2) / 255.f);
            else
                specColor = ld3(m->speccolor);
            specColor = scale3(specColor, spec);
            st3(pathSegment->color, mul3(color, specColor));
            st3(pathSegment->origin, add3(intersect, scale3(n, 0.01f)));
            st3(pathSegment->direction, reflectDir);
        } else {
            int coordU = (int)(intersection.texcoord[0] * kd->width);
            int coordV = (int)(intersection.texcoord[1] * kd->height);
            int pixelID = coordV * kd->width + coordU;
            v3 diffuseColor;
            if (kd->channels)
                diffuseColor = V3(texel(kd, pixelID, 0) / 255.f, texel(kd, pixelID, 1) / 255.f, texel(kd, pixelID, 2) / 255.f);
            else
                diffuseColor = ld3(m->color);
            st3(pathSegment->color, mul3(color, diffuseColor));
            v3 newdir = calculateRandomDirectionInHemisphere(n, rng);
            st3(pathSegment->direction, newdir);
            st3(pathSegment->origin, add3(intersect, scale3(newdir, 0.01f)));
        }
    } else {
        v3 newdir = calculateRandomDirectionInHemisphere(n, rng);
        st3(pathSegment->direction, newdir);
        st3(pathSegment->origin, add3(intersect, scale3(newdir, 0.01f)));
        st3(pathSegment->color, mul3(color, ld3(m->color)));
    }
}

/* pathtrace.cu:365-403, one path; idx seeds the RNG */
static void shade_one(const o_scene *s, int iter, int idx, const o_isect *intersection, o_path *seg) {
    if (intersection->t > 0.0f) {
        o_rng rng = make_seeded_engine(iter, idx, 0);
        const o_material *material = &s->mats[intersection->materialId];
        v3 materialColor = ld3(material->color);
        if (material->emittance > 0.0f) {
            st3(seg->color, mul3(ld3(seg->color), scale3(materialColor, material->emittance)));
            seg->remainingBounces = 0;
        } else if (seg->remainingBounces == 1) {
            st3(seg->color, V3(0.f, 0.f, 0.f));
            seg->remainingBounces = 0;
        } else {
            v3 o = ld3(seg->origin), d = ld3(seg->direction);
            v3 intersect = add3(o, scale3(d, intersection->t));
            scatterRay(s, seg, intersect, *intersection, material, &rng);
            seg->remainingBounces -= 1;
        }
    } else {
        st3(seg->color, V3(0.f, 0.f, 0.f));
        seg->remainingBounces = 0;
    }
}

void o_shade(void *h, int iter, int depth, int n, const int *idx, const o_isect *isects, o_path *paths) {
    (void)depth;
    o_scene *s = (o_scene *)h;
    for (int i = 0; i < n; i++) shade_one(s, iter, idx[i], &isects[i], &paths[i]);
}

/* ------------------------------------------------------------------------------------------------------ */
/* ray generation                                                                                           */
/* pathtrace.cu:183-197 */
static void ConcentricSampleDisk(float px, float py, float *ox, float *oy) {
    float ux = 2.f * px - 1.f, uy = 2.f * py - 1.f;
    if (ux == 0 && uy == 0) { *ox = 0; *oy = 0; return; }
    float theta, r;
    if (fabsf(ux) > fabsf(uy)) {
        r = ux;
        theta = 0.785398f * (uy / ux);
    } else {
        r = uy;
        theta = 1.570796f - 0.785398f * (ux / uy);
    }
    *ox = r * lm_cosf(theta);
    *oy = r * lm_sinf(theta);
}

/* pathtrace.cu:208-254 */
static void generate_one(const o_scene *s, int iter, int traceDepth, int x, int y, o_path *segment) {
    const o_camera *cam = &s->cam;
    int index = x + (y * cam->resx);
    o_rng rng = make_seeded_engine(iter, index, traceDepth);
    v3 origin = ld3(cam->position);
    float antia_x = (float)x;
    float antia_y = (float)y;
    if (s->opt_aa) {
        o_rng rngANTIA = make_seeded_engine(iter, index, traceDepth);
        antia_x += rng_uniform(&rngANTIA, -0.5f, 0.5f);
        antia_y += rng_uniform(&rngANTIA, -0.5f, 0.5f);
    }
    v3 a = scale3(scale3(ld3(cam->right), cam->pixelLength[0]), antia_x - (float)cam->resx * 0.5f);
    v3 b = scale3(scale3(ld3(cam->up), cam->pixelLength[1]), antia_y - (float)cam->resy * 0.5f);
    v3 direction = normalize3(sub3(sub3(ld3(cam->view), a), b));
    if (s->opt_dof) {
        float lensRadius = 0.8f;
        float focalDistance = 11.0f;
        float s0 = rng_uniform(&rng, 0, 1);
        float s1 = rng_uniform(&rng, 0, 1);
        float lx, ly;
        ConcentricSampleDisk(s0, s1, &lx, &ly);
        lx = lensRadius * lx; ly = lensRadius * ly;
        float ft = fabsf(focalDistance / direction.z);
        v3 pFocus = add3(origin, scale3(direction, ft));
        origin = add3(origin, V3(lx, ly, 0.f));
        direction = normalize3(sub3(pFocus, origin));
    }
    st3(segment->origin, origin);
    st3(segment->direction, direction);
    st3(segment->color, V3(1.0f, 1.0f, 1.0f));
    segment->pixelIndex = index;
    segment->remainingBounces = traceDepth;
}

/* ------------------------------------------------------------------------------------------------------ */
/* StreamCompaction::CPU (stream_compaction/cpu.cu)                                                         */
/* cpu.cu:20-32 */
void o_sc_scan(int n, int *odata, const int *idata) {
    if (n <= 0) return;
    odata[0] = idata[0];
    for (int i = 1; i < n; i++) odata[i] = odata[i - 1] + idata[i];
    for (int i = 0; i < n; i++) odata[i] -= idata[i];
}
/* cpu.cu:39-51 */
int o_sc_compact_without_scan(int n, int *odata, const int *idata) {
    int num = 0;
    for (int i = 0; i < n; i++) if (idata[i] != 0) odata[num++] = idata[i];
    return num;
}
/* cpu.cu:58-95 */
int o_sc_compact_with_scan(int n, int *odata, const int *idata) {
    if (n <= 0) return 0;
    int *tmpArray = (int *)malloc(sizeof(int) * (size_t)n);
    int *scanResult = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) tmpArray[i] = idata[i] == 0 ? 0 : 1;
    scanResult[0] = tmpArray[0];
    for (int i = 1; i < n; i++) scanResult[i] = scanResult[i - 1] + tmpArray[i];
    for (int i = 0; i < n; i++) scanResult[i] -= tmpArray[i];
    int num = 0;
    for (int i = 0; i < n; i++) if (tmpArray[i] == 1) { odata[scanResult[i]] = idata[i]; num++; }
    free(tmpArray); free(scanResult);
    return num;
}

/* ------------------------------------------------------------------------------------------------------ */
/* iteration driver                                                                                         */
static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* pathtraceInit, pathtrace.cu:101-157 (buffers only; the scene is already POD) */
void o_pt_init(void *h) {
    o_scene *s = (o_scene *)h;
    free_iter_state(s);
    s->pixelcount = 0;
    for (int y = 0; y < s->cam.resy; y++) if (row_owned(s, y)) s->pixelcount += s->cam.resx;
    size_t n = (size_t)(s->pixelcount > 0 ? s->pixelcount : 1);
    size_t nfull = (size_t)s->cam.resx * (size_t)s->cam.resy;
    s->image = (float *)calloc((nfull > 0 ? nfull : 1) * 3, sizeof(float));
    s->albedo = (float *)calloc((nfull > 0 ? nfull : 1) * 3, sizeof(float));
    s->paths = (o_path *)calloc(n, sizeof(o_path));
    s->paths_tmp = (o_path *)calloc(n, sizeof(o_path));
    s->isects = (o_isect *)calloc(n, sizeof(o_isect));
    s->isects_tmp = (o_isect *)calloc(n, sizeof(o_isect));
    s->first_isects = (o_isect *)calloc(n, sizeof(o_isect));
    s->flags = (int *)calloc(n, sizeof(int));
    s->scan = (int *)calloc(n, sizeof(int));
    s->perm = (int *)calloc(n, sizeof(int));
    memset(s->secs, 0, sizeof s->secs);
    s->nlive = 0;
}

void o_pt_generate(void *h, int iter) {
    o_scene *s = (o_scene *)h;
    double t0 = now_s();
    int k = 0;
    for (int y = 0; y < s->cam.resy; y++) {
        if (!row_owned(s, y)) continue;
        for (int x = 0; x < s->cam.resx; x++) generate_one(s, iter, s->traceDepth, x, y, &s->paths[k++]);
    }
    s->num_paths = s->pixelcount;
    s->depth = 0;
    s->nlive = 0;
    s->secs[4] += now_s() - t0;
}

/* thrust::sort_by_key(isects, isects+n, paths, a.materialId > b.materialId) (pathtrace.cu:418-422,518) is a
 * stable sort: a stable counting sort over the material ids, largest id first, is result-identical.       */
static void sort_by_material(o_scene *s, int n) {
    if (n <= 1) return;
    int lo = s->isects[0].materialId, hi = lo;
    for (int i = 1; i < n; i++) {
        int k = s->isects[i].materialId;
        if (k < lo) lo = k;
        if (k > hi) hi = k;
    }
    int nb = hi - lo + 1;
    int *count = (int *)calloc((size_t)nb + 1, sizeof(int));
    for (int i = 0; i < n; i++) count[hi - s->isects[i].materialId + 1]++;
    for (int b = 0; b < nb; b++) count[b + 1] += count[b];
    for (int i = 0; i < n; i++) {
        int pos = count[hi - s->isects[i].materialId]++;
        s->isects_tmp[pos] = s->isects[i];
        s->paths_tmp[pos] = s->paths[i];
    }
    memcpy(s->isects, s->isects_tmp, sizeof(o_isect) * (size_t)n);
    memcpy(s->paths, s->paths_tmp, sizeof(o_path) * (size_t)n);
    free(count);
}

/* thrust::stable_partition(paths, paths+n, remainingBounces > 0) (pathtrace.cu:424-428,541) done the
 * StreamCompaction::CPU way: map to flags, exclusive scan, scatter; dead paths keep their order behind.  */
static int partition_paths(o_scene *s, int n) {
    if (n <= 0) return 0;
    for (int i = 0; i < n; i++) s->flags[i] = s->paths[i].remainingBounces > 0 ? 1 : 0;
    o_sc_scan(n, s->scan, s->flags);
    int nlive = s->scan[n - 1] + s->flags[n - 1];
    for (int i = 0; i < n; i++) {
        int pos = s->flags[i] ? s->scan[i] : nlive + (i - s->scan[i]);
        s->paths_tmp[pos] = s->paths[i];
    }
    memcpy(s->paths, s->paths_tmp, sizeof(o_path) * (size_t)n);
    return nlive;
}

/* one pass of the while-loop body of pathtrace() (pathtrace.cu:490-544); stage_mask as in ref_driver.cpp */
int o_pt_bounce(void *h, int iter, int stage_mask) {
    o_scene *s = (o_scene *)h;
    int num_paths = s->num_paths;
    int cache_compiled = s->opt_cache && !s->opt_aa && !s->opt_dof;
    if (stage_mask & 1) {
        double t0 = now_s();
        if (s->nlive < 256) s->live_counts[s->nlive++] = num_paths;
        if (cache_compiled && s->depth == 0 && iter != 1) {
            /* :492-499 -- restores and pre-sorts; the memset below then wipes what was restored */
            memcpy(s->isects, s->first_isects, sizeof(o_isect) * (size_t)s->pixelcount);
            if (s->opt_sort) sort_by_material(s, s->pixelcount);
        }
        memset(s->isects, 0, sizeof(o_isect) * (size_t)s->pixelcount);           /* :501 */
        _Pragma("omp parallel for schedule(dynamic, 2048) num_threads(g_threads) if (g_threads > 1)")
        for (int i = 0; i < num_paths; i++) compute_intersection_one(s, &s->paths[i], &s->isects[i]);
        if (cache_compiled && iter == 1 && s->depth == 0)
            memcpy(s->first_isects, s->isects, sizeof(o_isect) * (size_t)s->pixelcount);
        s->secs[0] += now_s() - t0;
    }
    if (stage_mask & 2) {
        double t0 = now_s();
        if (s->opt_sort) sort_by_material(s, num_paths);
        s->depth++;
        s->secs[1] += now_s() - t0;
    }
    if (stage_mask & 4) {
        double t0 = now_s();
        if (s->apps && iter == 1 && s->depth == 1) {                        /* apps/src/pathtrace.cu:412-462 */
            for (int i = 0; i < num_paths; i++) {
                const o_isect *is = &s->isects[i];
                float *dst = s->albedo + (size_t)s->paths[i].pixelIndex * 3;
                if (is->t > 0.0f) {
                    const o_material *m = &s->mats[is->materialId];
                    const o_geom *geom = &s->geoms[is->geomId];
                    st3(dst, ld3(m->color));
                    if (geom->type == O_OBJ) {
                        const o_texture *kd = &geom->tex[0], *ke = &geom->tex[2];
                        v3 emission = V3(0.f, 0.f, 0.f);
                        if (ke->channels) {
                            int coordU = (int)(is->texcoord[0] * ke->width), coordV = (int)(is->texcoord[1] * ke->height);
                            int pixelID = coordV * ke->width + coordU;
                            emission = V3(texel(ke, pixelID, 0) / 255.f, texel(ke, pixelID, 1) / 255.f, texel(ke, pixelID, 2) / 255.f);
                        }
                        if (emission.x > FLT_EPSILON || emission.y > FLT_EPSILON || emission.z > FLT_EPSILON) {
                            st3(dst, scale3(emission, 5.0f));
                        } else if (kd->channels) {
                            int coordU = (int)(is->texcoord[0] * kd->width), coordV = (int)(is->texcoord[1] * kd->height);
                            int pixelID = coordV * kd->width + coordU;
                            st3(dst, V3(texel(kd, pixelID, 0) / 255.f, texel(kd, pixelID, 1) / 255.f, texel(kd, pixelID, 2) / 255.f));
                        }
                    } else if (m->emittance > 0.0f) {
                        st3(dst, scale3(ld3(m->color), m->emittance));
                    } else if (m->hasRefractive > 0.0f) {
                        st3(dst, ld3(m->speccolor));
                    }
                } else {
                    st3(dst, V3(0.f, 0.f, 0.f));
                }
            }
        }
        _Pragma("omp parallel for schedule(dynamic, 2048) num_threads(g_threads) if (g_threads > 1)")
        for (int i = 0; i < num_paths; i++) shade_one(s, iter, i, &s->isects[i], &s->paths[i]);
        s->secs[2] += now_s() - t0;
    }
    if (stage_mask & 8) {
        double t0 = now_s();
        s->num_paths = partition_paths(s, num_paths);
        s->secs[3] += now_s() - t0;
    }
    return s->num_paths;
}

/* pathtrace.cu:407-416 */
void o_pt_final_gather(void *h) {
    o_scene *s = (o_scene *)h;
    double t0 = now_s();
    for (int i = 0; i < s->pixelcount; i++) {
        const o_path *p = &s->paths[i];
        float *px = s->image + (size_t)p->pixelIndex * 3;
        if (s->apps) {
            const float APPS_PI = 3.14159265358f;
            px[0] += p->color[0] * APPS_PI; px[1] += p->color[1] * APPS_PI; px[2] += p->color[2] * APPS_PI;
        } else {
            px[0] += p->color[0]; px[1] += p->color[1]; px[2] += p->color[2];
        }
    }
    s->secs[5] += now_s() - t0;
}

int o_pt_iterate(void *h, int iter) {
    o_scene *s = (o_scene *)h;
    o_pt_generate(h, iter);
    for (;;) {
        int n = o_pt_bounce(h, iter, 15);
        if (n == 0) break;
    }
    o_pt_final_gather(h);
    return s->nlive;
}

int o_pt_live_counts(void *h, int *out, int cap) {
    o_scene *s = (o_scene *)h;
    for (int i = 0; i < s->nlive && i < cap; i++) out[i] = s->live_counts[i];
    return s->nlive;
}
o_path *o_pt_paths(void *h) { return ((o_scene *)h)->paths; }
o_isect *o_pt_isects(void *h) { return ((o_scene *)h)->isects; }
float *o_pt_image(void *h) { return ((o_scene *)h)->image; }
int o_pt_num_paths(void *h) { return ((o_scene *)h)->num_paths; }
int o_pt_pixelcount(void *h) { return ((o_scene *)h)->pixelcount; }
int o_pt_framepixels(void *h) { return ((o_scene *)h)->cam.resx * ((o_scene *)h)->cam.resy; }
void o_pt_stage_seconds(void *h, double out6[6]) { memcpy(out6, ((o_scene *)h)->secs, sizeof(double) * 6); }

/* pathtrace.cu:69-89 */
void o_pt_pbo(void *h, int iter, unsigned char *pbo) {
    o_scene *s = (o_scene *)h;
    const int npix = s->cam.resx * s->cam.resy;
    for (int index = 0; index < npix; index++) {
        const float *pix = s->image + (size_t)index * 3;
        int c[3];
        for (int k = 0; k < 3; k++) {
            int v = (int)(pix[k] / (float)iter * 255.0);
            c[k] = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        pbo[index * 4 + 3] = 0;
        pbo[index * 4 + 0] = (unsigned char)c[0];
        pbo[index * 4 + 1] = (unsigned char)c[1];
        pbo[index * 4 + 2] = (unsigned char)c[2];
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* loader-side arithmetic                                                                                   */
static void mat4_identity(float *m) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.f; }
/* type_mat4x4.inl operator*(mat4, mat4): Result[c] = A0*B[c][0] + A1*B[c][1] + A2*B[c][2] + A3*B[c][3] */
static void mat4_mul(const float *a, const float *b, float *out) {
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int k = 0; k < 4; k++)
            r[c * 4 + k] = ((a[0 * 4 + k] * b[c * 4 + 0] + a[1 * 4 + k] * b[c * 4 + 1]) + a[2 * 4 + k] * b[c * 4 + 2]) +
                           a[3 * 4 + k] * b[c * 4 + 3];
    memcpy(out, r, 64);
}
/* gtc/matrix_transform.inl:40-50: Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3] */
static void mat4_translate(const float *m, const float v[3], float *out) {
    float r[16]; memcpy(r, m, 64);
    for (int k = 0; k < 4; k++) r[12 + k] = ((m[k] * v[0] + m[4 + k] * v[1]) + m[8 + k] * v[2]) + m[12 + k];
    memcpy(out, r, 64);
}
/* gtc/matrix_transform.inl:52-84 */
static void mat4_rotate(const float *m, float angle, const float v[3], float *out) {
    float a = angle;
    float c = cosf(a);
    float s = sinf(a);
    v3 axis = normalize3(ld3(v));
    v3 temp = V3((1.f - c) * axis.x, (1.f - c) * axis.y, (1.f - c) * axis.z);
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = 0 + temp.x * axis.y + s * axis.z;
    R[0][2] = 0 + temp.x * axis.z - s * axis.y;
    R[1][0] = 0 + temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = 0 + temp.y * axis.z + s * axis.x;
    R[2][0] = 0 + temp.z * axis.x + s * axis.y;
    R[2][1] = 0 + temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    float r[16];
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 4; k++)
            r[j * 4 + k] = (m[k] * R[j][0] + m[4 + k] * R[j][1]) + m[8 + k] * R[j][2];
    for (int k = 0; k < 4; k++) r[12 + k] = m[12 + k];
    memcpy(out, r, 64);
}
/* gtc/matrix_transform.inl:120-133 */
static void mat4_scale(const float *m, const float v[3], float *out) {
    float r[16];
    for (int k = 0; k < 4; k++) { r[k] = m[k] * v[0]; r[4 + k] = m[4 + k] * v[1]; r[8 + k] = m[8 + k] * v[2]; r[12 + k] = m[12 + k]; }
    memcpy(out, r, 64);
}
#define M(c, r) m[(c) * 4 + (r)]
/* detail/type_mat4x4.inl:36-89 compute_inverse */
static void mat4_inverse(const float *m, float *out) {
    float Coef00 = M(2,2) * M(3,3) - M(3,2) * M(2,3);
    float Coef02 = M(1,2) * M(3,3) - M(3,2) * M(1,3);
    float Coef03 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
    float Coef04 = M(2,1) * M(3,3) - M(3,1) * M(2,3);
    float Coef06 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
    float Coef07 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
    float Coef08 = M(2,1) * M(3,2) - M(3,1) * M(2,2);
    float Coef10 = M(1,1) * M(3,2) - M(3,1) * M(1,2);
    float Coef11 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
    float Coef12 = M(2,0) * M(3,3) - M(3,0) * M(2,3);
    float Coef14 = M(1,0) * M(3,3) - M(3,0) * M(1,3);
    float Coef15 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
    float Coef16 = M(2,0) * M(3,2) - M(3,0) * M(2,2);
    float Coef18 = M(1,0) * M(3,2) - M(3,0) * M(1,2);
    float Coef19 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
    float Coef20 = M(2,0) * M(3,1) - M(3,0) * M(2,1);
    float Coef22 = M(1,0) * M(3,1) - M(3,0) * M(1,1);
    float Coef23 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
    float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
    float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
    float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
    float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    float Vec0[4] = {M(1,0), M(0,0), M(0,0), M(0,0)};
    float Vec1[4] = {M(1,1), M(0,1), M(0,1), M(0,1)};
    float Vec2[4] = {M(1,2), M(0,2), M(0,2), M(0,2)};
    float Vec3[4] = {M(1,3), M(0,3), M(0,3), M(0,3)};
    static const float SignA[4] = {+1, -1, +1, -1};
    static const float SignB[4] = {-1, +1, -1, +1};
    float Inv[16];
    for (int k = 0; k < 4; k++) {
        float Inv0 = (Vec1[k] * Fac0[k] - Vec2[k] * Fac1[k]) + Vec3[k] * Fac2[k];
        float Inv1 = (Vec0[k] * Fac0[k] - Vec2[k] * Fac3[k]) + Vec3[k] * Fac4[k];
        float Inv2 = (Vec0[k] * Fac1[k] - Vec1[k] * Fac3[k]) + Vec3[k] * Fac5[k];
        float Inv3 = (Vec0[k] * Fac2[k] - Vec1[k] * Fac4[k]) + Vec2[k] * Fac5[k];
        Inv[0 * 4 + k] = Inv0 * SignA[k];
        Inv[1 * 4 + k] = Inv1 * SignB[k];
        Inv[2 * 4 + k] = Inv2 * SignA[k];
        Inv[3 * 4 + k] = Inv3 * SignB[k];
    }
    float Row0[4] = {Inv[0], Inv[4], Inv[8], Inv[12]};
    float Dot0[4] = {M(0,0) * Row0[0], M(0,1) * Row0[1], M(0,2) * Row0[2], M(0,3) * Row0[3]};
    float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int i = 0; i < 16; i++) out[i] = Inv[i] * OneOverDeterminant;
}
/* gtc/matrix_inverse.inl:93-147 inverseTranspose(mat4) -- including its SubFactor11 quirk */
static void mat4_inverse_transpose(const float *m, float *out) {
    float SubFactor00 = M(2,2) * M(3,3) - M(3,2) * M(2,3);
    float SubFactor01 = M(2,1) * M(3,3) - M(3,1) * M(2,3);
    float SubFactor02 = M(2,1) * M(3,2) - M(3,1) * M(2,2);
    float SubFactor03 = M(2,0) * M(3,3) - M(3,0) * M(2,3);
    float SubFactor04 = M(2,0) * M(3,2) - M(3,0) * M(2,2);
    float SubFactor05 = M(2,0) * M(3,1) - M(3,0) * M(2,1);
    float SubFactor06 = M(1,2) * M(3,3) - M(3,2) * M(1,3);
    float SubFactor07 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
    float SubFactor08 = M(1,1) * M(3,2) - M(3,1) * M(1,2);
    float SubFactor09 = M(1,0) * M(3,3) - M(3,0) * M(1,3);
    float SubFactor10 = M(1,0) * M(3,2) - M(3,0) * M(1,2);
    float SubFactor11 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
    float SubFactor12 = M(1,0) * M(3,1) - M(3,0) * M(1,1);
    float SubFactor13 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
    float SubFactor14 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
    float SubFactor15 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
    float SubFactor16 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
    float SubFactor17 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
    float SubFactor18 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
    float I[16];
    I[0]  = +((M(1,1) * SubFactor00 - M(1,2) * SubFactor01) + M(1,3) * SubFactor02);
    I[1]  = -((M(1,0) * SubFactor00 - M(1,2) * SubFactor03) + M(1,3) * SubFactor04);
    I[2]  = +((M(1,0) * SubFactor01 - M(1,1) * SubFactor03) + M(1,3) * SubFactor05);
    I[3]  = -((M(1,0) * SubFactor02 - M(1,1) * SubFactor04) + M(1,2) * SubFactor05);
    I[4]  = -((M(0,1) * SubFactor00 - M(0,2) * SubFactor01) + M(0,3) * SubFactor02);
    I[5]  = +((M(0,0) * SubFactor00 - M(0,2) * SubFactor03) + M(0,3) * SubFactor04);
    I[6]  = -((M(0,0) * SubFactor01 - M(0,1) * SubFactor03) + M(0,3) * SubFactor05);
    I[7]  = +((M(0,0) * SubFactor02 - M(0,1) * SubFactor04) + M(0,2) * SubFactor05);
    I[8]  = +((M(0,1) * SubFactor06 - M(0,2) * SubFactor07) + M(0,3) * SubFactor08);
    I[9]  = -((M(0,0) * SubFactor06 - M(0,2) * SubFactor09) + M(0,3) * SubFactor10);
    I[10] = +((M(0,0) * SubFactor11 - M(0,1) * SubFactor09) + M(0,3) * SubFactor12);
    I[11] = -((M(0,0) * SubFactor08 - M(0,1) * SubFactor10) + M(0,2) * SubFactor12);
    I[12] = -((M(0,1) * SubFactor13 - M(0,2) * SubFactor14) + M(0,3) * SubFactor15);
    I[13] = +((M(0,0) * SubFactor13 - M(0,2) * SubFactor16) + M(0,3) * SubFactor17);
    I[14] = -((M(0,0) * SubFactor14 - M(0,1) * SubFactor16) + M(0,3) * SubFactor18);
    I[15] = +((M(0,0) * SubFactor15 - M(0,1) * SubFactor17) + M(0,2) * SubFactor18);
    float Determinant = ((+M(0,0) * I[0] + M(0,1) * I[1]) + M(0,2) * I[2]) + M(0,3) * I[3];
    for (int i = 0; i < 16; i++) out[i] = I[i] / Determinant;
}
#undef M

#define O_PI 3.1415926535897932384626422832795028841971f   /* utilities.h:12 */

/* utilities.cpp:65-72 buildTransformationMatrix, then scene.cpp:301-304.  out48 = transform, inverse, invTranspose */
void o_build_transforms(const float trs9[9], float out48[48]) {
    float I[16], T[16], R[16], Rt[16], S[16], TR[16];
    static const float X[3] = {1, 0, 0}, Y[3] = {0, 1, 0}, Z[3] = {0, 0, 1};
    mat4_identity(I);
    mat4_translate(I, trs9, T);
    mat4_rotate(I, trs9[3] * O_PI / 180, X, R);
    mat4_rotate(I, trs9[4] * O_PI / 180, Y, Rt); mat4_mul(R, Rt, R);
    mat4_rotate(I, trs9[5] * O_PI / 180, Z, Rt); mat4_mul(R, Rt, R);
    mat4_scale(I, trs9 + 6, S);
    mat4_mul(T, R, TR);
    mat4_mul(TR, S, out48);
    mat4_inverse(out48, out48 + 16);
    mat4_inverse_transpose(out48, out48 + 32);
}

/* scene.cpp:364-374.  f19 = position lookAt view up right fov(2) pixelLength(2).  camera.right is computed from
 * the still-zero view (glm default ctor) => normalize(0) = NaN, exactly as the reference loader leaves it.   */
void o_camera_from_loader(int resx, int resy, float fovy, const float eye[3], const float lookat[3],
                          const float up[3], float f19[19]) {
    float yscaled = tanf(fovy * (O_PI / 180));
    float xscaled = (yscaled * resx) / resy;
    float fovx = (atanf(xscaled) * 180) / O_PI;
    v3 view0 = V3(0.f, 0.f, 0.f);
    v3 right = normalize3(cross3(view0, ld3(up)));
    float plx = 2 * xscaled / (float)resx;
    float ply = 2 * yscaled / (float)resy;
    v3 view = normalize3(sub3(ld3(lookat), ld3(eye)));
    st3(f19 + 0, ld3(eye)); st3(f19 + 3, ld3(lookat)); st3(f19 + 6, view); st3(f19 + 9, ld3(up)); st3(f19 + 12, right);
    f19[15] = fovx; f19[16] = fovy; f19[17] = plx; f19[18] = ply;
}

/* ---- camera controls of src/main.cpp as plain functions (the mouse handlers :180-212, SPACE :166-171) -------------
 * orbit4 = {phi, theta, zoom} of main.cpp:18 and f19 = the camera floats (position 0, lookAt 3, view 6, up 9, right 12).
 * xpos/ypos deltas are doubles as in the GLFW callbacks; phi, theta and zoom are floats, so each update rounds once. */
void o_orbit_init(const float f19[19], float orbit3[3], float og_look_at[3]) {            /* main.cpp:56-70 */
    v3 position = ld3(f19 + 0), lookAt = ld3(f19 + 3), view = ld3(f19 + 6);
    v3 viewXZ = V3(view.x, 0.0f, view.z);
    v3 viewZY = V3(0.0f, view.y, view.z);
    orbit3[0] = acosf(dot3(normalize3(viewXZ), V3(0, 0, -1)));
    orbit3[1] = acosf(dot3(normalize3(viewZY), V3(0, 1, 0)));
    orbit3[2] = length3(sub3(position, lookAt));
    st3(og_look_at, lookAt);
}
void o_orbit_left_drag(float orbit3[3], double dx, double dy, int width, int height) {    /* :184-189 */
    orbit3[0] = (float)((double)orbit3[0] - dx / width);
    orbit3[1] = (float)((double)orbit3[1] - dy / height);
    orbit3[1] = fmaxf(0.001f, fminf(orbit3[1], 3.1415926535897932384626422832795028841971f));
}
void o_orbit_right_drag(float orbit3[3], double dy, int height) {                          /* :190-194 */
    orbit3[2] = (float)((double)orbit3[2] + dy / height);
    orbit3[2] = fmaxf(0.1f, orbit3[2]);
}
void o_orbit_middle_drag(float f19[19], double dx, double dy) {                            /* :195-209 */
    v3 forward = ld3(f19 + 6);
    forward.y = 0.0f;
    forward = normalize3(forward);
    v3 right = ld3(f19 + 12);
    right.y = 0.0f;
    right = normalize3(right);
    v3 lookAt = ld3(f19 + 3);
    lookAt = sub3(lookAt, scale3(scale3(right, (float)dx), 0.01f));       /* (float)(dx) * right * 0.01f, left to right */
    lookAt = add3(lookAt, scale3(scale3(forward, (float)dy), 0.01f));
    st3(f19 + 3, lookAt);
}
void o_orbit_apply(float f19[19], const float orbit3[3]) {                                  /* runCuda, :105-123 */
    const float phi = orbit3[0], theta = orbit3[1], zoom = orbit3[2];
    v3 lookAt = ld3(f19 + 3);
    v3 cameraPosition;
    cameraPosition.x = zoom * sinf(phi) * sinf(theta);
    cameraPosition.y = zoom * cosf(theta);
    cameraPosition.z = zoom * cosf(phi) * sinf(theta);
    v3 v = neg3(normalize3(cameraPosition));
    v3 u = V3(0, 1, 0);
    v3 r = cross3(v, u);
    v3 up = cross3(r, v);
    cameraPosition = add3(cameraPosition, lookAt);
    st3(f19 + 0, cameraPosition); st3(f19 + 6, v); st3(f19 + 9, up); st3(f19 + 12, r);
}

/* main.cpp:56-70 then runCuda's recompute main.cpp:105-123 */
void o_runcuda_camera(float f19[19]) {
    v3 position = ld3(f19 + 0), lookAt = ld3(f19 + 3), view = ld3(f19 + 6);
    v3 viewXZ = V3(view.x, 0.0f, view.z);
    v3 viewZY = V3(0.0f, view.y, view.z);
    float phi = acosf(dot3(normalize3(viewXZ), V3(0, 0, -1)));
    float theta = acosf(dot3(normalize3(viewZY), V3(0, 1, 0)));
    float zoom = length3(sub3(position, lookAt));
    v3 cameraPosition;
    cameraPosition.x = zoom * sinf(phi) * sinf(theta);
    cameraPosition.y = zoom * cosf(theta);
    cameraPosition.z = zoom * cosf(phi) * sinf(theta);
    v3 v = neg3(normalize3(cameraPosition));
    v3 u = V3(0, 1, 0);
    v3 r = cross3(v, u);
    v3 up = cross3(r, v);
    cameraPosition = add3(cameraPosition, lookAt);
    st3(f19 + 0, cameraPosition); st3(f19 + 6, v); st3(f19 + 9, up); st3(f19 + 12, r);
}
