// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Host-side driver around the REFERENCE's own path-tracing headers, compiled from where they lie
// under /root/reference (nothing is copied into this repo; see oracle/Makefile, target _ref).
// What comes from the reference, unmodified and by #include path:
//   src/sceneStructs.h, src/intersections.h, src/interactions.h, src/scene.{h,cpp},
//   src/utilities.{h,cpp}, src/stb.cpp, src/tiny_obj_loader.h, vendored glm 0.9.6.3.
// What this file restates (the parts of src/pathtrace.cu and src/main.cpp that are CUDA/GL only):
//   makeSeededRandomEngine (pathtrace.cu:62-66), the bodies of generateRayFromCamera (:206-255),
//   computeIntersections (:261-344), shadeFakeMaterial (:355-404), finalGather (:407-416),
//   sendImageToPBO (:69-89) as host loops, the bounce loop of pathtrace() (:433-558) with
//   thrust::sort_by_key -> std::stable_sort and thrust::stable_partition -> std::stable_partition,
//   and runCuda()'s camera recompute (main.cpp:105-123).
// Thrust's minstd_rand / uniform_real_distribution come from rocThrust in /opt/rocm (same algorithm
// as CUDA Thrust).  Build: hipcc host-only pass, -ffp-contract=off, glibc libm.
#include <cfloat>
#include <cmath>
#include <math.h>   // global float overloads of cos/sin/abs, as device code and nvcc's host pass see them
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <numeric>
#include <vector>
#include <string>
#include <unistd.h>
#include <thrust/random.h>

#include "sceneStructs.h"
#include "scene.h"
#include "intersections.h"
#include "interactions.h"

// pathtrace.cu:62-66
static thrust::default_random_engine makeSeededRandomEngine(int iter, int index, int depth) {
    int h = utilhash((1 << 31) | (depth << 22) | iter) ^ utilhash(index);
    return thrust::default_random_engine(h);
}

// pathtrace.cu:183-197
static glm::vec2 ConcentricSampleDisk(const glm::vec2 &point) {
    glm::vec2 uOffset = 2.f * point - glm::vec2(1, 1);
    if (uOffset.x == 0 && uOffset.y == 0)
        return glm::vec2(0, 0);
    float theta, r;
    if (std::abs(uOffset.x) > std::abs(uOffset.y)) {
        r = uOffset.x;
        theta = 0.785398f * (uOffset.y / uOffset.x);
    }
    else {
        r = uOffset.y;
        theta = 1.570796f - 0.785398f * (uOffset.x / uOffset.y);
    }
    return r * glm::vec2(std::cos(theta), std::sin(theta));
}

struct RefOptions {
    int antialiasing;    // ANTIALIASING      (pathtrace.cu:39) default 1
    int depth_of_field;  // DEPTH_OF_FIELD    (:36) default 0
    int sort_by_material;// SORT_BY_MATERIAL  (:38) default 1
    int cache_first;     // CACHE_FIRST_BOUNCE(:37) default 1
    int apps;            // 1 = the apps/src copy's deltas: finalGather * PI (apps/src/pathtrace.cu:508) + albedo AOV (:412-462)
};

struct RefState {
    Scene *scene = nullptr;
    std::vector<Geom> geoms;              // with host pointers patched in (pathtraceInit :111-140)
    std::vector<Texture> empty;
    RefOptions opt{1, 0, 1, 1, 0};
    std::vector<glm::vec3> image, albedo;
    std::vector<PathSegment> paths;
    std::vector<ShadeableIntersection> isects, first_isects;
    int num_paths = 0, depth = 0, pixelcount = 0;
    std::vector<int> live_counts;         // paths entering computeIntersections per bounce
};

extern "C" {

unsigned ref_utilhash(unsigned a) { return utilhash(a); }

void ref_rng_uniform(int iter, int index, int depth, float a, float b, int n, float *out) {
    thrust::default_random_engine rng = makeSeededRandomEngine(iter, index, depth);
    thrust::uniform_real_distribution<float> u(a, b);
    for (int i = 0; i < n; i++) out[i] = u(rng);
}

void ref_rng_raw(int iter, int index, int depth, int n, unsigned *out) {
    thrust::default_random_engine rng = makeSeededRandomEngine(iter, index, depth);
    for (int i = 0; i < n; i++) out[i] = rng();
}

// Loads a scene with the reference loader.  The loader resolves "../models/..." relative to the CWD,
// so the caller passes the directory to chdir into first (normally /root/reference/scenes).
void *ref_scene_load(const char *cwd, const char *path) {
    char old[4096];
    if (!getcwd(old, sizeof old)) return nullptr;
    if (cwd && chdir(cwd) != 0) return nullptr;
    FILE *so = stdout;
    fflush(stdout);
    int saved = dup(1);
    FILE *devnull = fopen("/dev/null", "w");
    dup2(fileno(devnull), 1);                 // the loader is chatty on stdout
    RefState *st = new RefState;
    st->scene = new Scene(path);
    fflush(stdout);
    dup2(saved, 1); close(saved); fclose(devnull); (void)so;
    if (chdir(old) != 0) { /* ignore */ }
    Scene *sc = st->scene;
    // pathtraceInit :111-140 indexes kd/ks/ke/bumpTextures[i]; for OBJ geoms whose .mtl names no texture
    // the vectors are shorter than geoms -> treat "no texture" as an empty Texture.
    st->geoms = sc->geoms;
    size_t ng = st->geoms.size();
    bool aligned = sc->kdTextures.size() == ng && sc->ksTextures.size() == ng &&
                   sc->keTextures.size() == ng && sc->bumpTextures.size() == ng;
    for (size_t i = 0; i < ng; i++) {
        Geom &g = st->geoms[i];
        g.dev_faces = sc->allFaces[i].data();
        g.faceSize = (int)sc->allFaces[i].size();
        Texture none;
        // Only when every OBJ names all four maps do the reference's texture vectors line up with geoms.
        g.kd = aligned ? sc->kdTextures[i] : none;
        g.ks = aligned ? sc->ksTextures[i] : none;
        g.ke = aligned ? sc->keTextures[i] : none;
        g.bump = aligned ? sc->bumpTextures[i] : none;
    }
    return st;
}

void ref_scene_free(void *h) {
    RefState *st = (RefState *)h;
    // scene.h declares ~Scene() but scene.cpp never defines it (main.cpp never deletes its Scene): leak it.
    delete st;
}

int ref_num_geoms(void *h) { return (int)((RefState *)h)->geoms.size(); }
int ref_num_materials(void *h) { return (int)((RefState *)h)->scene->materials.size(); }
int ref_num_faces(void *h, int g) { return ((RefState *)h)->geoms[g].faceSize; }
int ref_texture_vector_sizes(void *h, int *out4) {
    Scene *sc = ((RefState *)h)->scene;
    out4[0] = (int)sc->kdTextures.size(); out4[1] = (int)sc->ksTextures.size();
    out4[2] = (int)sc->keTextures.size(); out4[3] = (int)sc->bumpTextures.size();
    return 0;
}

// geom dump: type, materialid, faceSize, then translation(3) rotation(3) scale(3), transform(16),
// inverseTransform(16), invTranspose(16) in glm column-major memory order.
void ref_get_geom(void *h, int i, int *ints3, float *floats57) {
    const Geom &g = ((RefState *)h)->geoms[i];
    ints3[0] = (int)g.type; ints3[1] = g.materialid; ints3[2] = g.faceSize;
    memcpy(floats57 + 0, &g.translation, 12);
    memcpy(floats57 + 3, &g.rotation, 12);
    memcpy(floats57 + 6, &g.scale, 12);
    memcpy(floats57 + 9, &g.transform, 64);
    memcpy(floats57 + 25, &g.inverseTransform, 64);
    memcpy(floats57 + 41, &g.invTranspose, 64);
}

// material dump: color(3) exponent specular.color(3) hasReflective hasRefractive ior emittance = 11 floats
void ref_get_material(void *h, int i, float *f11) {
    const Material &m = ((RefState *)h)->scene->materials[i];
    memcpy(f11, &m, sizeof(Material));
}

// faces: per face 3 x (position(3), texcoord(2)) = 15 floats
void ref_get_faces(void *h, int g, float *out) {
    const Geom &ge = ((RefState *)h)->geoms[g];
    for (int j = 0; j < ge.faceSize; j++) {
        const Face &f = ge.dev_faces[j];
        const Vertex *vs[3] = {&f.v0, &f.v1, &f.v2};
        for (int k = 0; k < 3; k++) {
            memcpy(out + j * 15 + k * 5, &vs[k]->position, 12);
            memcpy(out + j * 15 + k * 5 + 3, &vs[k]->texcoord, 8);
        }
    }
}

// texture of a geom as pathtraceInit would upload it: which = 0 kd, 1 ks, 2 ke, 3 bump; whc = width, height, channels;
// pixels may be NULL to query the size only
void ref_get_texture(void *h, int g, int which, int *whc, unsigned char *pixels) {
    const Geom &ge = ((RefState *)h)->geoms[g];
    const Texture *t = which == 0 ? &ge.kd : which == 1 ? &ge.ks : which == 2 ? &ge.ke : &ge.bump;
    whc[0] = t->width; whc[1] = t->height; whc[2] = t->channels;
    if (pixels && t->image && t->channels) memcpy(pixels, t->image, (size_t)t->width * t->height * t->channels);
}

// camera dump: resolution(2 ints) ; position lookAt view up right (15 floats) fov(2) pixelLength(2) ;
// iterations, traceDepth
void ref_get_camera(void *h, int *ints4, float *f19) {
    RenderState &s = ((RefState *)h)->scene->state;
    const Camera &c = s.camera;
    ints4[0] = c.resolution.x; ints4[1] = c.resolution.y; ints4[2] = (int)s.iterations; ints4[3] = s.traceDepth;
    memcpy(f19 + 0, &c.position, 12); memcpy(f19 + 3, &c.lookAt, 12); memcpy(f19 + 6, &c.view, 12);
    memcpy(f19 + 9, &c.up, 12); memcpy(f19 + 12, &c.right, 12); memcpy(f19 + 15, &c.fov, 8);
    memcpy(f19 + 17, &c.pixelLength, 8);
}

void ref_set_depth(void *h, int depth) { ((RefState *)h)->scene->state.traceDepth = depth; }

// main.cpp:56-70 (spherical coordinates from the loader camera) followed by runCuda's recompute :105-123.
void ref_apply_runcuda_camera(void *h) {
    Camera &cam = ((RefState *)h)->scene->state.camera;
    glm::vec3 view = cam.view;
    glm::vec3 viewXZ = glm::vec3(view.x, 0.0f, view.z);
    glm::vec3 viewZY = glm::vec3(0.0f, view.y, view.z);
    float phi = glm::acos(glm::dot(glm::normalize(viewXZ), glm::vec3(0, 0, -1)));
    float theta = glm::acos(glm::dot(glm::normalize(viewZY), glm::vec3(0, 1, 0)));
    float zoom = glm::length(cam.position - cam.lookAt);
    glm::vec3 cameraPosition;
    cameraPosition.x = zoom * sin(phi) * sin(theta);
    cameraPosition.y = zoom * cos(theta);
    cameraPosition.z = zoom * cos(phi) * sin(theta);
    cam.view = -glm::normalize(cameraPosition);
    glm::vec3 v = cam.view;
    glm::vec3 u = glm::vec3(0, 1, 0);
    glm::vec3 r = glm::cross(v, u);
    cam.up = glm::cross(r, v);
    cam.right = r;
    cam.position = cameraPosition;
    cameraPosition += cam.lookAt;
    cam.position = cameraPosition;
}

void ref_set_options(void *h, int aa, int dof, int sort, int cache) {
    RefState *st = (RefState *)h;
    st->opt = RefOptions{aa, dof, sort, cache, st->opt.apps};
}

void ref_set_apps_variant(void *h, int on) { ((RefState *)h)->opt.apps = on; }
float *ref_pt_albedo(void *h) { return (float *)((RefState *)h)->albedo.data(); }

// ---- per-function known-answer entry points (call the reference functions directly) ----------------
// rays: o(3) d(3) per ray.  out per ray: t, point(3), normal(3), uv(2), outside  = 10 floats
void ref_geom_test(void *h, int gi, int n, const float *rays, float *out) {
    RefState *st = (RefState *)h;
    const Geom &g = st->geoms[gi];
    for (int i = 0; i < n; i++) {
        Ray r;
        memcpy(&r.origin, rays + i * 6, 12);
        memcpy(&r.direction, rays + i * 6 + 3, 12);
        glm::vec3 p(0.f), nrm(0.f);
        glm::vec2 uv(0.f);
        bool outside = true;
        float t = -1.f;
        if (g.type == CUBE) t = boxIntersectionTest(g, r, p, nrm, outside);
        else if (g.type == SPHERE) t = sphereIntersectionTest(g, r, p, nrm, outside);
        else if (g.type == OBJ) t = meshIntersectionTest(g, r, p, nrm, uv, outside);
        float *o = out + i * 10;
        o[0] = t; memcpy(o + 1, &p, 12); memcpy(o + 4, &nrm, 12); memcpy(o + 7, &uv, 8); o[9] = outside ? 1.f : 0.f;
    }
}

// the reference's dead calculateJitteredDirectionHemisphere (src/interactions.h:46-85), called as it stands; the engine is seeded as
// makeSeededRandomEngine does (iter, index, depth per sample) and `iter` is the sample's own
void ref_jittered_test(int n, const float *normals, const int *seeds, int max_iter, float *out) {
    for (int i = 0; i < n; i++) {
        thrust::default_random_engine rng = makeSeededRandomEngine(seeds[i * 3], seeds[i * 3 + 1], seeds[i * 3 + 2]);
        glm::vec3 nrm;
        memcpy(&nrm, normals + i * 3, 12);
        const glm::vec3 d = calculateJitteredDirectionHemisphere(nrm, rng, seeds[i * 3], max_iter);
        memcpy(out + i * 3, &d, 12);
    }
}

// the reference's dead pair (src/intersections.h:175-205, 284-315), called as it stands.  out per ray: t, point(3), normal(3), outside
void ref_obj_tri_test(void *h, int gi, int n, const float *rays, float *out) {
    RefState *st = (RefState *)h;
    const Geom &g = st->geoms[gi];
    for (int i = 0; i < n; i++) {
        Ray r;
        memcpy(&r.origin, rays + i * 6, 12);
        memcpy(&r.direction, rays + i * 6 + 3, 12);
        glm::vec3 p(0.f), nrm(0.f);
        bool outside = true;
        float t = g.type == OBJ ? objTriIntersectionTest(g, r, p, nrm, outside) : -1.f;
        float *o = out + i * 8;
        o[0] = t; memcpy(o + 1, &p, 12); memcpy(o + 4, &nrm, 12); o[7] = outside ? 1.f : 0.f;
    }
}

// body of computeIntersections (pathtrace.cu:270-343) for one path
static void compute_intersection_one(const std::vector<Geom> &geoms_v, const PathSegment &pathSegment,
                                     ShadeableIntersection &dst) {
    Geom *geoms = const_cast<Geom *>(geoms_v.data());
    int geoms_size = (int)geoms_v.size();
    float t;
    glm::vec3 intersect_point;
    glm::vec3 normal;
    float t_min = FLT_MAX;
    int hit_geom_index = -1;
    bool outside = true;
    glm::vec2 uv = glm::vec2(0.0f, 0.0f);
    glm::vec3 tmp_intersect;
    glm::vec3 tmp_normal;
    glm::vec2 tmp_uv;
    t = 0.f;   // the reference leaves t uninitialised; only a TRIANGLE geom (no test routine) would read it
    for (int i = 0; i < geoms_size; i++) {
        Geom &geom = geoms[i];
        if (geom.type == CUBE) {
            t = boxIntersectionTest(geom, pathSegment.ray, tmp_intersect, tmp_normal, outside);
        } else if (geom.type == SPHERE) {
            t = sphereIntersectionTest(geom, pathSegment.ray, tmp_intersect, tmp_normal, outside);
        } else if (geom.type == OBJ) {
            t = meshIntersectionTest(geom, pathSegment.ray, tmp_intersect, tmp_normal, tmp_uv, outside);
        }
        if (t > 0.0f && t_min > t) {
            t_min = t;
            hit_geom_index = i;
            intersect_point = tmp_intersect;
            normal = tmp_normal;
            uv = tmp_uv;
        }
    }
    if (hit_geom_index == -1) {
        dst.t = -1.0f;
    } else {
        dst.t = t_min;
        dst.materialId = geoms[hit_geom_index].materialid;
        dst.surfaceNormal = normal;
        dst.geomId = hit_geom_index;
        dst.texcoord = uv;
    }
}

// paths: n x 11 words (PathSegment layout); out: n x 8 words (ShadeableIntersection layout), pre-zeroed here
void ref_compute_intersections(void *h, int n, const void *paths, void *out) {
    RefState *st = (RefState *)h;
    const PathSegment *p = (const PathSegment *)paths;
    ShadeableIntersection *o = (ShadeableIntersection *)out;
    memset(o, 0, sizeof(ShadeableIntersection) * (size_t)n);
    for (int i = 0; i < n; i++) compute_intersection_one(st->geoms, p[i], o[i]);
}

// body of shadeFakeMaterial (pathtrace.cu:365-403) for one path; idx is the stream index that seeds the RNG
static void shade_one(RefState *st, int iter, int idx, const ShadeableIntersection &intersection,
                      PathSegment &seg, int depth) {
    Material *materials = st->scene->materials.data();
    Geom *geoms = st->geoms.data();
    if (intersection.t > 0.0f) {
        thrust::default_random_engine rng = makeSeededRandomEngine(iter, idx, 0);
        Material material = materials[intersection.materialId];
        glm::vec3 materialColor = material.color;
        if (material.emittance > 0.0f) {
            seg.color *= (materialColor * material.emittance);
            seg.remainingBounces = 0;
        } else if (seg.remainingBounces == 1) {
            seg.color = glm::vec3(0.0);
            seg.remainingBounces = 0;
        } else {
            scatterRay(seg, seg.ray.origin + intersection.t * seg.ray.direction, intersection, material, rng,
                       geoms, iter, depth);
            seg.remainingBounces -= 1;
        }
    } else {
        seg.color = glm::vec3(0.0f);
        seg.remainingBounces = 0;
    }
}

// shade n paths in place; idx[i] is the RNG stream index of path i
void ref_shade(void *h, int iter, int depth, int n, const int *idx, const void *isects, void *paths) {
    RefState *st = (RefState *)h;
    const ShadeableIntersection *is = (const ShadeableIntersection *)isects;
    PathSegment *p = (PathSegment *)paths;
    for (int i = 0; i < n; i++) shade_one(st, iter, idx[i], is[i], p[i], depth);
}

// body of generateRayFromCamera (pathtrace.cu:208-254)
static void generate_one(const RefState *st, const Camera &cam, int iter, int traceDepth, int x, int y,
                         PathSegment &segment) {
    int index = x + (y * cam.resolution.x);
    thrust::default_random_engine rng = makeSeededRandomEngine(iter, index, traceDepth);
    segment.ray.origin = cam.position;
    segment.color = glm::vec3(1.0f, 1.0f, 1.0f);
    float antia_x = x;
    float antia_y = y;
    if (st->opt.antialiasing) {
        thrust::default_random_engine rngANTIA = makeSeededRandomEngine(iter, index, traceDepth);
        thrust::uniform_real_distribution<float> uANTIA(-0.5, 0.5);
        antia_x += uANTIA(rngANTIA);
        antia_y += uANTIA(rngANTIA);
    }
    segment.ray.direction = glm::normalize(cam.view
        - cam.right * cam.pixelLength.x * ((float)antia_x - (float)cam.resolution.x * 0.5f)
        - cam.up * cam.pixelLength.y * ((float)antia_y - (float)cam.resolution.y * 0.5f)
        );
    if (st->opt.depth_of_field) {
        float lensRadius = 0.8f;
        float focalDistance = 11.0f;
        thrust::uniform_real_distribution<float> uDOF(0, 1);
        if (lensRadius > 0) {
            // C++ leaves the evaluation order of the two uDOF(rng) arguments unspecified; nvcc and clang
            // evaluate left to right, which is what is restated here explicitly.
            float s0 = uDOF(rng);
            float s1 = uDOF(rng);
            glm::vec2 pLens = lensRadius * ConcentricSampleDisk(glm::vec2(s0, s1));
            float ft = glm::abs(focalDistance / segment.ray.direction.z);
            glm::vec3 pFocus = segment.ray.origin + segment.ray.direction * ft;
            segment.ray.origin += glm::vec3(pLens.x, pLens.y, 0);
            segment.ray.direction = normalize(pFocus - segment.ray.origin);
        }
    }
    segment.pixelIndex = index;
    segment.remainingBounces = traceDepth;
}

void ref_pt_init(void *h) {
    RefState *st = (RefState *)h;
    const Camera &cam = st->scene->state.camera;
    st->pixelcount = cam.resolution.x * cam.resolution.y;
    st->image.assign(st->pixelcount, glm::vec3(0.f));
    st->albedo.assign(st->pixelcount, glm::vec3(0.f));
    st->paths.assign(st->pixelcount, PathSegment());
    ShadeableIntersection z; memset(&z, 0, sizeof z);
    st->isects.assign(st->pixelcount, z);
    st->first_isects.assign(st->pixelcount, z);
}

void ref_pt_generate(void *h, int iter) {
    RefState *st = (RefState *)h;
    const Camera &cam = st->scene->state.camera;
    int traceDepth = st->scene->state.traceDepth;
    for (int y = 0; y < cam.resolution.y; y++)
        for (int x = 0; x < cam.resolution.x; x++)
            generate_one(st, cam, iter, traceDepth, x, y, st->paths[x + y * cam.resolution.x]);
    st->num_paths = st->pixelcount;
    st->depth = 0;
    st->live_counts.clear();
}

static void sort_by_material(RefState *st, int n) {
    // thrust::sort_by_key(keys = isects, values = paths, cmp a.materialId > b.materialId) is stable
    std::vector<int> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
        return st->isects[a].materialId > st->isects[b].materialId; });
    std::vector<ShadeableIntersection> is2(n);
    std::vector<PathSegment> p2(n);
    for (int i = 0; i < n; i++) { is2[i] = st->isects[perm[i]]; p2[i] = st->paths[perm[i]]; }
    std::copy(is2.begin(), is2.end(), st->isects.begin());
    std::copy(p2.begin(), p2.end(), st->paths.begin());
}

// One pass of the while-loop body of pathtrace() (:490-544).  stage_mask selects how far to go so tests can
// look at intermediate state: 1 = intersect(+cache), 2 = +sort, 4 = +shade, 8 = +partition.  Returns num_paths.
int ref_pt_bounce(void *h, int iter, int stage_mask) {
    RefState *st = (RefState *)h;
    int num_paths = st->num_paths;
    bool cache_compiled = st->opt.cache_first && !st->opt.antialiasing && !st->opt.depth_of_field;
    if (stage_mask & 1) {
        st->live_counts.push_back(num_paths);
        if (cache_compiled && st->depth == 0 && iter != 1) {
            std::copy(st->first_isects.begin(), st->first_isects.end(), st->isects.begin());
            if (st->opt.sort_by_material) sort_by_material(st, st->pixelcount);
        }
        ShadeableIntersection z; memset(&z, 0, sizeof z);
        std::fill(st->isects.begin(), st->isects.end(), z);                 // cudaMemset :501
        for (int i = 0; i < num_paths; i++) compute_intersection_one(st->geoms, st->paths[i], st->isects[i]);
        if (cache_compiled && iter == 1 && st->depth == 0)
            std::copy(st->isects.begin(), st->isects.end(), st->first_isects.begin());
    }
    if (stage_mask & 2) {
        if (st->opt.sort_by_material) sort_by_material(st, num_paths);
        st->depth++;
    }
    if (stage_mask & 4) {
        if (st->opt.apps && iter == 1 && st->depth == 1) {
            // albedo AOV of the apps/src copy (apps/src/pathtrace.cu:412-462), first shade of the first iteration
            for (int i = 0; i < num_paths; i++) {
                const ShadeableIntersection &intersection = st->isects[i];
                glm::vec3 &dst = st->albedo[st->paths[i].pixelIndex];
                if (intersection.t > 0.0f) {
                    Material material = st->scene->materials[intersection.materialId];
                    glm::vec3 materialColor = material.color;
                    dst = materialColor;
                    Geom geom = st->geoms[intersection.geomId];
                    if (geom.type == OBJ) {
                        glm::vec3 emission(0.0f);
                        if (geom.ke.channels) {
                            int coordU = (int)(intersection.texcoord.x * geom.ke.width);
                            int coordV = (int)(intersection.texcoord.y * geom.ke.height);
                            int pixelID = coordV * geom.ke.width + coordU;
                            unsigned int colR = (unsigned int)geom.ke.image[pixelID * geom.ke.channels];
                            unsigned int colG = (unsigned int)geom.ke.image[pixelID * geom.ke.channels + 1];
                            unsigned int colB = (unsigned int)geom.ke.image[pixelID * geom.ke.channels + 2];
                            emission = glm::vec3(colR / 255.f, colG / 255.f, colB / 255.f);
                        }
                        if (emission.x > FLT_EPSILON || emission.y > FLT_EPSILON || emission.z > FLT_EPSILON) {
                            dst = (emission * 5.0f);
                        } else if (geom.kd.channels) {
                            int coordU = (int)(intersection.texcoord.x * geom.kd.width);
                            int coordV = (int)(intersection.texcoord.y * geom.kd.height);
                            int pixelID = coordV * geom.kd.width + coordU;
                            unsigned int colR = (unsigned int)geom.kd.image[pixelID * geom.kd.channels];
                            unsigned int colG = (unsigned int)geom.kd.image[pixelID * geom.kd.channels + 1];
                            unsigned int colB = (unsigned int)geom.kd.image[pixelID * geom.kd.channels + 2];
                            dst = glm::vec3(colR / 255.f, colG / 255.f, colB / 255.f);
                        }
                    } else if (material.emittance > 0.0f) {
                        dst = materialColor * material.emittance;
                    } else if (material.hasRefractive > 0.0f) {
                        dst = material.specular.color;
                    }
                } else {
                    dst = glm::vec3(0.0f);
                }
            }
        }
        for (int i = 0; i < num_paths; i++) shade_one(st, iter, i, st->isects[i], st->paths[i], st->depth);
    }
    if (stage_mask & 8) {
        auto end = std::stable_partition(st->paths.begin(), st->paths.begin() + num_paths,
                                         [](const PathSegment &p) { return p.remainingBounces > 0; });
        st->num_paths = (int)(end - st->paths.begin());
    }
    return st->num_paths;
}

void ref_pt_final_gather(void *h) {
    RefState *st = (RefState *)h;
    const float APPS_PI = 3.14159265358f;        // apps/src/pathtrace.cu:44
    for (int i = 0; i < st->pixelcount; i++) {
        if (st->opt.apps) st->image[st->paths[i].pixelIndex] += st->paths[i].color * APPS_PI;
        else st->image[st->paths[i].pixelIndex] += st->paths[i].color;
    }
}

// full iteration = pathtrace(pbo, frame, iter) without the PBO
int ref_pt_iterate(void *h, int iter) {
    RefState *st = (RefState *)h;
    ref_pt_generate(h, iter);
    while (true) {
        int n = ref_pt_bounce(h, iter, 15);
        if (n == 0) break;
    }
    ref_pt_final_gather(h);
    return (int)st->live_counts.size();
}

int ref_pt_live_counts(void *h, int *out, int cap) {
    RefState *st = (RefState *)h;
    int n = (int)st->live_counts.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = st->live_counts[i];
    return n;
}

void *ref_pt_paths(void *h) { return ((RefState *)h)->paths.data(); }
void *ref_pt_isects(void *h) { return ((RefState *)h)->isects.data(); }
float *ref_pt_image(void *h) { return (float *)((RefState *)h)->image.data(); }
int ref_pt_num_paths(void *h) { return ((RefState *)h)->num_paths; }
int ref_pt_pixelcount(void *h) { return ((RefState *)h)->pixelcount; }

// sendImageToPBO body (pathtrace.cu:74-88): rgba8 per pixel
void ref_pt_pbo(void *h, int iter, unsigned char *pbo) {
    RefState *st = (RefState *)h;
    for (int index = 0; index < st->pixelcount; index++) {
        glm::vec3 pix = st->image[index];
        glm::ivec3 color;
        color.x = glm::clamp((int)(pix.x / iter * 255.0), 0, 255);
        color.y = glm::clamp((int)(pix.y / iter * 255.0), 0, 255);
        color.z = glm::clamp((int)(pix.z / iter * 255.0), 0, 255);
        pbo[index * 4 + 3] = 0;
        pbo[index * 4 + 0] = color.x;
        pbo[index * 4 + 1] = color.y;
        pbo[index * 4 + 2] = color.z;
    }
}

int ref_sizeof_path() { return (int)sizeof(PathSegment); }
int ref_sizeof_isect() { return (int)sizeof(ShadeableIntersection); }

}  // extern "C"
