// pt_device.h -- device-side arithmetic of the MI355X path tracer (gfx950, wave64).
//
// Every function states which reference function it computes (file:line relative to the reference root).
// Result parity with the reference is a *bit-level* requirement here, not a tolerance: the shading RNG is seeded
// by a path's position in the material-sorted stream (src/pathtrace.cu:373), so one flipped hit/miss shifts
// every later stream index of that bounce.  Hence:
//   * this translation unit is compiled with -ffp-contract=off (no FMA contraction), IEEE division and sqrt
//     (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt), f32 denormals on;
//   * each expression is written in the order glm 0.9.6.3 evaluates it, e.g. mat4*vec4 is
//     (m0*x + m1*y) + (m2*z + m3*w)  (glm/detail/type_mat4x4.inl:617-628);
//   * sin/cos/pow are not the hardware approximations but the portable binary64 routines below, which agree with
//     glibc's correctly rounded results (tests/test_own_libm.py) and run the same on host and device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__
#define PT_HD __host__ __device__ __forceinline__      // also compiled for the host (BVH checks on the CPU, pt_bvh.h)

// ---- arithmetic level of this translation unit (ptx_options.arith, include/mi355x_pathtracer.h) ---------------------------------
// 0  EXACT: everything this header's first comment says; the product's default and the only level any parity claim is made for.
// 1  CONTRACTED: the same source compiled with -ffp-contract=fast (a*b+c becomes one fused multiply-add wherever the expression tree
//    allows, as nvcc's default --fmad=true does to the reference's kernels); division and square root stay IEEE (nvcc's defaults
//    --prec-div / --prec-sqrt too), sin / cos are an fp32 routine of about one ulp (as CUDA's sinf / cosf) instead of the correctly
//    rounded binary64 one.
// 2  FAST: level 1 + the hardware's approximate reciprocal / square root / reciprocal square root (v_rcp_f32, v_sqrt_f32, v_rsq_f32:
//    one ulp each, quotients a*rcp(b) about 2.5 ulp; -fno-hip-fp32-correctly-rounded-divide-sqrt) and its sine / cosine / exp2 / log2
//    instructions.
// Levels 1 and 2 are separate code objects (csrc/pt_arith.hip includes pt_engine.hip with PT_ARITH set and `ptd` renamed, so that no
// inline function of theirs can ever be the copy the exact translation unit links) and promise a STATISTICAL tolerance against
// level 0 (tests/test_gpu_parity.py::test_contracted_arithmetic_*), never bits.
#ifndef PT_ARITH
#define PT_ARITH 0
#endif

namespace ptd {

struct vec3 { float x, y, z; };

PT_HD vec3 V3(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD vec3 add(vec3 a, vec3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD vec3 sub(vec3 a, vec3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD vec3 mul(vec3 a, vec3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD vec3 scale(vec3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
PT_HD vec3 neg(vec3 a) { return V3(-a.x, -a.y, -a.z); }
// glm dot(vec3): tmp = x*y; tmp.x + tmp.y + tmp.z   (glm/detail/func_geometric.inl:65-72)
PT_HD float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// glm cross (func_geometric.inl:134-141)
PT_HD vec3 cross(vec3 x, vec3 y) { return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
// ---- correctly rounded square root and reciprocal without their range handling (device, round 4) ------------------------------
// `__builtin_sqrtf` and `1.0f / x` are IEEE-exact here (hipcc's correctly-rounded-divide-sqrt default), and on gfx950 the compiler expands
// each into a CORE -- v_sqrt_f32 and a one-ulp correction by two fused residuals; v_rcp_f32, one Newton step, three residual steps --
// wrapped in range handling: operands below 2^-96 are scaled up and the result down (v_cmp, v_mul, v_cndmask, ..., 16 instructions
// for a root where the core is 9), v_div_scale / v_div_fmas / v_div_fixup around a quotient (11 where the core is 6).  The path
// tracer's roots and reciprocals -- a normalize per transformed direction, a length per hit, 1/a per triangle: every seventh vector
// instruction of k_bounce -- see operands in the normal range practically always.  These two functions run THE SAME CORE, instruction
// for instruction as the compiler emits it (v_sqrt / v_rcp are deterministic, the fused steps are IEEE operations), behind one
// unsigned compare on the operand's bits, and hand everything else -- zero, subnormal, tiny, infinite, NaN, negative -- to the
// compiler's own expansion.  Same result for every operand by construction; `ptx_kat_fast_exact` compares them with the built-ins on
// the device over every exponent and on random operands (tests/test_gpu_parity.py), and every parity test runs on top of them.
#ifndef PT_FAST_EXACT
#define PT_FAST_EXACT 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && PT_ARITH >= 2
// FAST: the hardware's seeds as they are (one ulp; no range handling: v_rsq / v_rcp of 0 give inf, as 1 / sqrt(0) does)
__device__ __forceinline__ float pt_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float pt_rsqrt_glm(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float pt_rcp_pos(float a) { return __builtin_amdgcn_rcpf(a); }
#elif defined(__HIP_DEVICE_COMPILE__) && PT_FAST_EXACT
// 2^-96 <= x < +inf as ONE unsigned compare on the bits (negative, zero, subnormal, tiny, inf and NaN fail it)
__device__ __forceinline__ bool pt_in_core_range(float x) { return __float_as_uint(x) - 0x0f800000u < 0x70000000u; }
__device__ __forceinline__ float pt_sqrt_core(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);                                    // within one ulp
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float r1 = __builtin_fmaf(-sm, s, x);
    const float s1 = 0.0f >= r1 ? sm : s;
    const float r2 = __builtin_fmaf(-sp, s, x);
    return 0.0f < r2 ? sp : s1;
}
// 1 / t for 2^-48 <= t < 2^64 (a root of an operand that passed pt_in_core_range): the quotient's core with numerator 1
__device__ __forceinline__ float pt_rcp_core(float t) {
    float r = __builtin_amdgcn_rcpf(t);
    const float e = __builtin_fmaf(-t, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r;                                                                  // (= 1.0f * r)
    const float e2 = __builtin_fmaf(-t, q, 1.0f);
    q = __builtin_fmaf(e2, r, q);
    const float e3 = __builtin_fmaf(-t, q, 1.0f);
    return __builtin_fmaf(e3, r, q);
}
__device__ __forceinline__ float pt_sqrt(float x) { return __builtin_expect(pt_in_core_range(x), 1) ? pt_sqrt_core(x) : __builtin_sqrtf(x); }
__device__ __forceinline__ float pt_rsqrt_glm(float x) {                           // glm inversesqrt: 1 / sqrt(x), two roundings
    if (__builtin_expect(pt_in_core_range(x), 1)) return pt_rcp_core(pt_sqrt_core(x));
    return 1.0f / __builtin_sqrtf(x);
}
// 1 / a for FLT_EPSILON <= a (the caller's own test) and a < 2^60: in range for the core; anything larger takes the division
__device__ __forceinline__ float pt_rcp_pos(float a) { return __builtin_expect(a < 1.152921504606846976e18f, 1) ? pt_rcp_core(a) : 1.0f / a; }
#else
PT_HD float pt_sqrt(float x) { return __builtin_sqrtf(x); }
PT_HD float pt_rsqrt_glm(float x) { return 1.0f / __builtin_sqrtf(x); }
PT_HD float pt_rcp_pos(float a) { return 1.0f / a; }
#endif
// glm normalize = x * (1 / sqrt(dot(x,x)))  (func_geometric.inl:154-159, func_exponential.inl:62-68)
PT_HD vec3 normalize(vec3 a) { return scale(a, pt_rsqrt_glm(dot(a, a))); }
PT_HD float length(vec3 a) { return pt_sqrt(dot(a, a)); }
PT_HD float fmin_glm(float x, float y) { return x < y ? x : y; }   // glm min: x < y ? x : y
PT_HD float fmax_glm(float x, float y) { return x > y ? x : y; }   // glm max: x > y ? x : y

// multiplyMV (src/intersections.h:34-36): xyz of mat4*vec4, glm column-major m[c*4+r]
PT_HD vec3 multiplyMV(const float *__restrict__ m, vec3 v, float w) {
    vec3 r;
    r.x = (m[0] * v.x + m[4] * v.y) + (m[8] * v.z + m[12] * w);
    r.y = (m[1] * v.x + m[5] * v.y) + (m[9] * v.z + m[13] * w);
    r.z = (m[2] * v.x + m[6] * v.y) + (m[10] * v.z + m[14] * w);
    return r;
}

// ---- portable libm (same operation sequence as the CPU checker's copy; binary64, one rounding to binary32) ----
// fma(a, b, c) for a wave-uniform c held in a scalar register pair (the compiler's own choice, v_fmac_f64, needs c in vector registers)
PT_DEV double fma_sk(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
// sin and cos of a float argument, |x| <= 1e5: Cody-Waite reduction by pi/2, Taylor polynomials in Horner form with FUSED
// multiply-adds (25 binary64 instructions instead of 40; fused operations are IEEE operations, the checker's copy runs the same ones).
// On every binary32 with |x| <= 2 pi -- all the path tracer ever passes -- the results equal those of the separate-multiply-add form
// of rounds 1-3 (tools/sincos_fma_exhaustive.c: 2 173 837 242 arguments, 0 differ), so no fixture moved.
PT_DEV void sincos_own(float xf, float *s, float *c) {
    const double INVPIO2 = 0x1.45f306dc9c883p-1;
    const double PIO2_1 = 0x1.921fb54400000p+0;
    const double PIO2_1T = 0x1.0b4611a626331p-34;
    double x = (double)xf;
    if (!(x >= -1.0e5 && x <= 1.0e5)) { *s = __builtin_nanf(""); *c = __builtin_nanf(""); return; }
    double y = x * INVPIO2;
    int k = (int)(y + (y >= 0.0 ? 0.5 : -0.5));
    double kd = (double)k;
    double r = __builtin_fma(-kd, PIO2_1T, __builtin_fma(-kd, PIO2_1, x));
    double z = r * r;
    // (a Horner step as ONE v_fma_f64 whose addend is a scalar register pair -- written out, because the compiler prefers v_fmac_f64, whose
    // addend is its destination: it moves every coefficient into a vector pair first, two more vector instructions per step.  The
    // innermost step has two coefficients and a vector instruction reads one scalar pair: it stays a multiplication and an addition.)
    double ps = -0x1.ae7f3e733b81fp-41 + z * 0x1.952c77030ad4ap-49;
    ps = fma_sk(z, ps, 0x1.6124613a86d09p-33);
    ps = fma_sk(z, ps, -0x1.ae64567f544e4p-26);
    ps = fma_sk(z, ps, 0x1.71de3a556c734p-19);
    ps = fma_sk(z, ps, -0x1.a01a01a01a01ap-13);
    ps = fma_sk(z, ps, 0x1.1111111111111p-7);
    ps = fma_sk(z, ps, -0x1.5555555555555p-3);
    double sr = __builtin_fma(r, z * ps, r);
    double pc = -0x1.93974a8c07c9dp-37 + z * 0x1.ae7f3e733b81fp-45;
    pc = fma_sk(z, pc, 0x1.1eed8eff8d898p-29);
    pc = fma_sk(z, pc, -0x1.27e4fb7789f5cp-22);
    pc = fma_sk(z, pc, 0x1.a01a01a01a01ap-16);
    pc = fma_sk(z, pc, -0x1.6c16c16c16c17p-10);
    pc = fma_sk(z, pc, 0x1.5555555555555p-5);
    pc = __builtin_fma(z, pc, -0.5);
    double cr = __builtin_fma(z, pc, 1.0);
    // quadrant k & 3: (s, c) = (sr, cr), (cr, -sr), (-sr, -cr), (-cr, sr) -- odd k swaps, bit 1 of k negates the sine and bit 1 of
    // k + 1 the cosine (a negation is the sign bit: selects and two exclusive-ors instead of a four-way branch the wave would diverge on)
    const bool swap = (k & 1) != 0;
    const double s0 = swap ? cr : sr, c0 = swap ? sr : cr;
    const double sd = __hiloint2double(__double2hiint(s0) ^ (int)(((uint32_t)k & 2u) << 30), __double2loint(s0));
    const double cd = __hiloint2double(__double2hiint(c0) ^ (int)(((uint32_t)(k + 1) & 2u) << 30), __double2loint(c0));
    *s = (float)sd;
    *c = (float)cd;
}

#if PT_ARITH >= 1
// sin and cos in binary32 for |x| <= 1e5 (the path tracer passes |x| <= 2 pi): nearest multiple of pi/2 removed in two fused steps
// (Cody-Waite, pi/2 = hi + lo with hi's low bits zero), then the classic minimax polynomials on [-pi/4, pi/4] (Cephes sinf / cosf):
// about one ulp, the class of CUDA's sinf / cosf.  FAST takes the hardware's v_sin_f32 / v_cos_f32 (argument in revolutions).
PT_DEV void sincos_f32(float x, float *s, float *c) {
#if PT_ARITH >= 2
    const float rev = x * 0.15915494309189535f;
    *s = __builtin_amdgcn_sinf(rev);
    *c = __builtin_amdgcn_cosf(rev);
#else
    const float kf = __builtin_rintf(x * 0.6366197723675814f);
    const int k = (int)kf;
    float r = __builtin_fmaf(-kf, 1.5703125f, x);                     // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8
    r = __builtin_fmaf(-kf, 4.837512969970703125e-4f, r);
    r = __builtin_fmaf(-kf, 7.54978995489188e-8f, r);
    const float z = r * r;
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
    const float sr = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
    const float cr = __builtin_fmaf(z * z, pc, __builtin_fmaf(z, -0.5f, 1.0f));
    const bool swap = (k & 1) != 0;
    const float s0 = swap ? cr : sr, c0 = swap ? sr : cr;
    *s = __uint_as_float(__float_as_uint(s0) ^ (((uint32_t)k & 2u) << 30));
    *c = __uint_as_float(__float_as_uint(c0) ^ (((uint32_t)(k + 1) & 2u) << 30));
#endif
}
#endif
// what the samplers call: the correctly rounded binary64 routine (EXACT) or the binary32 one
PT_DEV void sincos_pt(float x, float *s, float *c) {
#if PT_ARITH >= 1
    sincos_f32(x, s, c);
#else
    sincos_own(x, s, c);
#endif
}

PT_DEV double pow5_own(double x) {
    double x2 = x * x;
    double x4 = x2 * x2;
    return x4 * x;
}

// powf(x, y) for x >= 0 via exp(y*log(x)) in binary64; powf(x, 0) = 1 for every x.
PT_DEV float powf_own(float xf, float yf) {
    if (yf == 0.0f) return 1.0f;
    if (xf != xf || yf != yf) return __builtin_nanf("");
    if (xf == 1.0f) return 1.0f;
    if (xf < 0.0f) return __builtin_nanf("");
    if (xf == 0.0f) return yf > 0.0f ? 0.0f : __builtin_inff();
    if (__builtin_isinf(xf)) return yf > 0.0f ? __builtin_inff() : 0.0f;
    if (__builtin_isinf(yf)) return ((xf > 1.0f) == (yf > 0.0f)) ? __builtin_inff() : 0.0f;
    double x = (double)xf;
    uint64_t u = (uint64_t)__double_as_longlong(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((long long)((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
    if (m > 0x1.6a09e667f3bcdp+0) { m = m * 0.5; e += 1; }
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
    const double LN2_HI = 0x1.62e42fee00000p-1, LN2_LO = 0x1.a39ef35793c76p-33;
    double ed = (double)e;
    double lg = (ed * LN2_HI + logm) + ed * LN2_LO;
    double a = (double)yf * lg;
    if (a > 89.0) return __builtin_inff();
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
    double two_k = __longlong_as_double((long long)((uint64_t)(k + 1023) << 52));
    return (float)(er * two_k);
}

// ---- hash + RNG -----------------------------------------------------------------------------------------------
// utilhash, src/intersections.h:12-20
PT_DEV uint32_t utilhash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

// thrust::minstd_rand (a = 48271, m = 2^31 - 1) + thrust::uniform_real_distribution<float>
struct Rng {
    uint32_t x;
    // makeSeededRandomEngine, src/pathtrace.cu:62-66
    PT_DEV void seed(int iter, int index, int depth) {
        uint32_t h = utilhash((1u << 31) | ((uint32_t)depth << 22) | (uint32_t)iter) ^ utilhash((uint32_t)index);
        x = h % 2147483647u;
        if (x == 0) x = 1;
    }
    PT_DEV uint32_t next() {
        uint64_t p = (uint64_t)x * 48271u;                       // < 2^47
        uint32_t r = (uint32_t)(p & 0x7fffffffu) + (uint32_t)(p >> 31);   // p mod (2^31-1) by folding
        r = (r & 0x7fffffffu) + (r >> 31);
        if (r >= 2147483647u) r -= 2147483647u;
        x = r;
        return r;
    }
    // float(x - min) / (1.0f + float(max - min)) * (b - a) + a, min = 1, max = m - 1: the divisor is 2^31
    PT_DEV float uniform(float a, float b) {
        float result = (float)(next() - 1u);
        result /= 2147483648.0f;
        return (result * (b - a)) + a;
    }
};

// ---- scene on the device ---------------------------------------------------------------------------------------
struct DTex { int32_t w, h, ch, pad; uint64_t off; };    // off = byte offset into the texture blob
struct DGeom {
    float xf[16], inv[16], invT[16];    // transform, inverseTransform, invTranspose (glm memory order)
    int32_t type, materialid, faceStart, faceCount;
    DTex tex[4];                        // kd, ks, ke, bump
};
struct DMaterial {                      // = struct Material, src/sceneStructs.h:71-81
    float color[3];
    float exponent;
    float speccolor[3];
    float hasReflective, hasRefractive, ior, emittance;
};
struct DCamera {                        // = struct Camera, src/sceneStructs.h:83-92
    int32_t resx, resy;
    float position[3], lookAt[3], view[3], up[3], right[3], fov[2], pixelLength[2];
};
struct alignas(16) BvhQuad { float x, y, z; int32_t w; };
struct alignas(16) BvhWide4 { int32_t a, b, c, d; };     // a quarter of a four-wide quantised node (16 words, bvhNearestWide)
#ifndef PT_MESH_CHUNK
#define PT_MESH_CHUNK 4
#endif
constexpr int MESH_CHUNK = PT_MESH_CHUNK;
#ifndef PT_DEFER_HITS
#define PT_DEFER_HITS 1       // chunked small meshes: accepted triangles' distances are evaluated after the chunk's tests (meshChunkTestLds)
#endif            // faces per lane when a small mesh's loop is spread over lanes
#ifndef PT_BVH_LEAF
#define PT_BVH_LEAF 4         // triangles per leaf at most (measured with the four-wide while-while walk, 20 448 triangles, k_mesh per iteration at 4K: 1: 0.62 ms, 2: 0.515, 4: 0.463, 6: 0.460, 8: 0.468)
#endif
constexpr int BVH_LEAF_MAX = PT_BVH_LEAF;
constexpr int BVH_MIN_FACES = 24;        // meshes smaller than this keep the plain loop
constexpr int BVH_TRI = 12;              // words per leaf triangle: v0, p1, p2 (the vertices as loaded), face index, two pads -- three 16-byte loads
constexpr int BVH_STACK = 32;            // entries per lane of k_mesh's traversal stack (trees deeper than 31 use the skip links)
struct DScene {
    const DGeom *__restrict__ geoms;
    const DMaterial *__restrict__ mats;
    const float *__restrict__ faces;    // 15 floats per face: 3 x (pos xyz, uv)
    const float *__restrict__ tri9;     // 9 floats per face: v0, e1 = v1 - v0, e2 = v2 - v0 (the subtractions the
                                        // triangle test starts with, done once at upload: same IEEE results)
    const uint8_t *__restrict__ texels;
    int32_t ngeoms, nmats;
    int32_t tri_lds;                    // != 0: the kernel has staged the scene tables at the start of its dynamic LDS; 2 (set as a
                                        // constant by the specialised kernel): the triangle tables too, whatever ntri is -- the
                                        // global-memory variants of the table reads are then not even compiled (triLds below)
                                        // (pt_lds): tri9 [ntri_lds*9], faces [ntri_lds*15], materials [nmats*11], gtab [ngeoms*40],
                                        // fnorm [ntri_lds*3], cnorm [ngeoms*18]
    int32_t ntri;
    int32_t ntri_lds;                   // triangles staged with them: ntri, or 0 when the mesh tables stay in global memory
    const float *__restrict__ gtab;     // 40 words per geom: inverseTransform rows 0-2 (12), transform rows 0-2 (12),
                                        // invTranspose rows 0-2 (12), type, materialid, faceStart, faceCount
    const float *__restrict__ aabb;     // 8 floats per geom: conservative world-space box as (centre xyz, pad, half extent xyz, pad), or NULL
    uint32_t cube_bits, sphere_bits, mesh_bits;   // bit i: geom i is a cube / sphere / mesh (unknown types are in none)
    const BvhQuad *__restrict__ bvh_nodes;          // threaded BVH of the larger meshes (pt_bvh.h), or NULL
    const float *__restrict__ bvh_tris;             // BVH_TRI words per leaf triangle
    const int32_t *__restrict__ bvh_root;           // per geom: root node, -1 = plain loop over its faces
    const int32_t *__restrict__ bvh_depth;          // per geom: depth of its tree (root = 0)
    int32_t bvh_stack;                              // entries per lane of the traversal stack the launch provides (k_mesh): trees of
                                                    // depth >= this use the skip links
    const BvhWide4 *__restrict__ bvh_wide;          // four-wide quantised nodes (4 x 16 B = one 64-byte line each) of the same trees, or NULL
    const int32_t *__restrict__ bvh_wroot;          // per geom: its wide root, -1 = none
    const int32_t *__restrict__ bvh_wneed;          // per geom: stack entries the wide walk of its tree can need
    const float *__restrict__ fnorm;    // 3 floats per face: its world-space geometric normal, computed at upload with the
                                        // arithmetic of meshIntersectionTest (src/intersections.h:237-243) -- used when the
                                        // geom has no bump map; cnorm: 18 floats per geom, the six face normals of a cube
                                        // (src/intersections.h:86), index (axis * 2 + (sign > 0)) * 3
    const float *__restrict__ cnorm;
    uint32_t bump_bits;                 // bit g: mesh g has a bump map (its normals are evaluated per hit, not tabulated)
    int32_t mesh_chunks;                // > 1: no mesh has a BVH and the longest has this many groups of MESH_CHUNK faces:
                                        // tileIntersect spreads every (ray, mesh) pair over that many lanes
    int32_t cull;                       // != 0: per-lane candidate lists from the world boxes (needs tri_lds, <= 32 geoms)
    const float *__restrict__ ldsblob;  // the tables below in ONE array, in the order and with the ntri_lds the launch stages them
                                        // (tri9, faces, materials, gtab, fnorm, cnorm): a workgroup copies it to LDS with all its loads in
                                        // flight at once -- one memory round trip, where table after table was six (it matters for the
                                        // short kernels of a small tile, which are chains of such round trips); NULL: table by table
};

// Words of dynamic LDS the staged scene tables take (tri9 + faces + fnorm = 27 per staged triangle, 11 per material, gtab 40 +
// cnorm 18 per geom), rounded to 16 bytes.  ONE definition for the kernels and for the host that sizes their launches: what
// follows the tables holds 64-bit LDS atomics, so the two must never disagree about where it starts.
__host__ __device__ constexpr int sceneTableWords(int ntri_lds, int nmats, int ngeoms) { return (ntri_lds * 27 + nmats * 11 + ngeoms * 58 + 3) & ~3; }
PT_DEV int sceneLdsWords(const DScene &sc) { return sceneTableWords(sc.ntri_lds, sc.nmats, sc.ngeoms); }
constexpr int GTAB_WORDS = 40;

// dynamic LDS of the kernels that use this header: [scene tables when sc.tri_lds][kernel-specific scratch]
extern __shared__ __attribute__((aligned(16))) int32_t pt_lds[];

// Copies the scene tables into LDS (call from `nthreads` threads of the workgroup with tid 0 .. nthreads - 1, then __syncthreads()).
__device__ __forceinline__ void stageSceneToLds(const DScene &sc, int tid, int nthreads) {
    float *l = reinterpret_cast<float *>(pt_lds);
    if (sc.ldsblob) {
        const int n = sc.ntri_lds * 27 + sc.nmats * 11 + sc.ngeoms * 58;
        for (int k = tid; k < n; k += 4 * nthreads) {           // four loads requested before the first is stored
            const int k1 = k + nthreads, k2 = k1 + nthreads, k3 = k2 + nthreads;
            const float a0 = sc.ldsblob[k], a1 = k1 < n ? sc.ldsblob[k1] : 0.f, a2 = k2 < n ? sc.ldsblob[k2] : 0.f, a3 = k3 < n ? sc.ldsblob[k3] : 0.f;
            l[k] = a0;
            if (k1 < n) l[k1] = a1;
            if (k2 < n) l[k2] = a2;
            if (k3 < n) l[k3] = a3;
        }
        return;
    }
    const int n9 = sc.ntri_lds * 9, n15 = sc.ntri_lds * 15, nm = sc.nmats * 11;
    for (int k = tid; k < n9; k += nthreads) l[k] = sc.tri9[k];
    for (int k = tid; k < n15; k += nthreads) l[n9 + k] = sc.faces[k];
    const float *m = reinterpret_cast<const float *>(sc.mats);
    for (int k = tid; k < nm; k += nthreads) l[n9 + n15 + k] = m[k];
    const int ng = sc.ngeoms * GTAB_WORDS;
    for (int k = tid; k < ng; k += nthreads) l[n9 + n15 + nm + k] = sc.gtab[k];
    const int nf = sc.ntri_lds * 3, nc = sc.ngeoms * 18;
    for (int k = tid; k < nf; k += nthreads) l[n9 + n15 + nm + ng + k] = sc.fnorm[k];
    for (int k = tid; k < nc; k += nthreads) l[n9 + n15 + nm + ng + nf + k] = sc.cnorm[k];
}
// are the triangle tables (tri9, faces, fnorm) staged in LDS?
PT_DEV bool triLds(const DScene &sc) { return sc.tri_lds == 2 || (sc.tri_lds && sc.ntri_lds); }
// precomputed normals: face `face` (global index) of the meshes / side `side` of cube `g`
PT_DEV vec3 faceNormalTab(const DScene &sc, int face) {
    if (triLds(sc)) {
        const float *l = reinterpret_cast<const float *>(pt_lds) + sc.ntri_lds * 24 + sc.nmats * 11 + sc.ngeoms * GTAB_WORDS + face * 3;
        return V3(l[0], l[1], l[2]);
    }
    const float *gp = sc.fnorm + (size_t)face * 3;
    return V3(gp[0], gp[1], gp[2]);
}
PT_DEV vec3 cubeNormalTab(const DScene &sc, int g, int side) {
    if (sc.tri_lds) {
        const float *l = reinterpret_cast<const float *>(pt_lds) + sc.ntri_lds * 27 + sc.nmats * 11 + sc.ngeoms * GTAB_WORDS + g * 18 + side * 3;
        return V3(l[0], l[1], l[2]);
    }
    const float *gp = sc.cnorm + (size_t)g * 18 + side * 3;
    return V3(gp[0], gp[1], gp[2]);
}

// type of geom g: from the staged per-geom table when there is one (an LDS read instead of a dependent global load in
// front of every diffuse scatter)
PT_DEV int geomType(const DScene &sc, int g) {
    if (sc.tri_lds) return __float_as_int(reinterpret_cast<const float *>(pt_lds)[sc.ntri_lds * 24 + sc.nmats * 11 + g * 40 + 36]);
    return sc.geoms[g].type;
}

// struct Material of material `id` (per-lane id: from LDS when staged, else global memory)
PT_DEV DMaterial getMaterial(const DScene &sc, int id) {
    DMaterial m;
    if (sc.tri_lds) {
        const float *l = reinterpret_cast<const float *>(pt_lds) + sc.ntri_lds * 24 + id * 11;
        m.color[0] = l[0]; m.color[1] = l[1]; m.color[2] = l[2]; m.exponent = l[3];
        m.speccolor[0] = l[4]; m.speccolor[1] = l[5]; m.speccolor[2] = l[6];
        m.hasReflective = l[7]; m.hasRefractive = l[8]; m.ior = l[9]; m.emittance = l[10];
    } else {
        m = sc.mats[id];
    }
    return m;
}
// word k of the 15-float record (3 x pos xyz, uv) of triangle `face` (global index)
PT_DEV float faceWord(const DScene &sc, int face, int k) {
    if (triLds(sc)) return reinterpret_cast<const float *>(pt_lds)[sc.ntri_lds * 9 + face * 15 + k];
    return sc.faces[(size_t)face * 15 + k];
}
PT_DEV vec3 faceVec(const DScene &sc, int face, int k) { return V3(faceWord(sc, face, k), faceWord(sc, face, k + 1), faceWord(sc, face, k + 2)); }

enum { G_SPHERE = 0, G_CUBE = 1, G_TRIANGLE = 2, G_OBJ = 3 };

struct Ray { vec3 o, d; };

// getPointOnRay, src/intersections.h:27-29
PT_DEV vec3 getPointOnRay(Ray r, float t) {
    vec3 nd = normalize(r.d);
    float tt = t - .0001f;
    return add(r.o, V3(tt * nd.x, tt * nd.y, tt * nd.z));
}

// What a geom test leaves behind for the normal, which only the nearest hit needs (computeIntersections keeps the
// normal of the winner only, src/pathtrace.cu:318-324): evaluating it after the loop is the same arithmetic, once.
struct Cand {
    int32_t axis; float sgn;     // cube: which component of the zero-initialised glm::vec3 n is +-1
    vec3 objP;                   // sphere: object-space hit point
    vec3 point;                  // cube / sphere: world-space hit point (only the per-function entry point reads it)
    bool outside;
    int32_t face;                // mesh: nearest face (index inside the geom)
    float u, v;                  // mesh: interpolated texcoord
};

// boxIntersectionTest without its normal, src/intersections.h:48-85,87.  Returns t (world distance) or -1.
PT_DEV float boxTestCore(const DGeom &box, Ray r, Cand &c) {
    Ray q;
    q.o = multiplyMV(box.inv, r.o, 1.0f);
    q.d = normalize(multiplyMV(box.inv, r.d, 0.0f));
    float tmin = -1e38f, tmax = 1e38f;
    int tmin_axis = -1, tmax_axis = -1;
    float tmin_s = 0.f, tmax_s = 0.f;
    const float qo[3] = {q.o.x, q.o.y, q.o.z};
    const float qd[3] = {q.d.x, q.d.y, q.d.z};
#pragma unroll
    for (int xyz = 0; xyz < 3; ++xyz) {
        float t1 = (-0.5f - qo[xyz]) / qd[xyz];
        float t2 = (+0.5f - qo[xyz]) / qd[xyz];
        float ta = fmin_glm(t1, t2);
        float tb = fmax_glm(t1, t2);
        float ns = t2 < t1 ? +1.f : -1.f;
        if (ta > 0 && ta > tmin) { tmin = ta; tmin_axis = xyz; tmin_s = ns; }
        if (tb < tmax) { tmax = tb; tmax_axis = xyz; tmax_s = ns; }
    }
    if (tmax >= tmin && tmax > 0) {
        c.outside = true;
        if (tmin <= 0) { tmin = tmax; tmin_axis = tmax_axis; tmin_s = tmax_s; c.outside = false; }
        c.axis = tmin_axis; c.sgn = tmin_s;
        c.point = multiplyMV(box.xf, getPointOnRay(q, tmin), 1.0f);
        return length(sub(r.o, c.point));
    }
    return -1.f;
}
// the normal of that hit, src/intersections.h:86
PT_DEV vec3 boxNormal(const DGeom &box, const Cand &c) {
    vec3 n = V3(c.axis == 0 ? c.sgn : 0.f, c.axis == 1 ? c.sgn : 0.f, c.axis == 2 ? c.sgn : 0.f);
    return normalize(multiplyMV(box.invT, n, 0.0f));
}

// sphereIntersectionTest without its normal, src/intersections.h:102-137,143
PT_DEV float sphereTestCore(const DGeom &sphere, Ray r, Cand &c) {
    const float radius = .5f;
    Ray rt;
    rt.o = multiplyMV(sphere.inv, r.o, 1.0f);
    rt.d = normalize(multiplyMV(sphere.inv, r.d, 0.0f));
    float vDotDirection = dot(rt.o, rt.d);
    float radicand = vDotDirection * vDotDirection - (dot(rt.o, rt.o) - radius * radius);   // powf(.5f,2) == .25f
    if (radicand < 0) return -1.f;
    float squareRoot = pt_sqrt(radicand);
    float firstTerm = -vDotDirection;
    float t1 = firstTerm + squareRoot;
    float t2 = firstTerm - squareRoot;
    float t;
    if (t1 < 0 && t2 < 0) {
        return -1.f;
    } else if (t1 > 0 && t2 > 0) {
        t = fmin_glm(t1, t2);
        c.outside = true;
    } else {
        t = fmax_glm(t1, t2);
        c.outside = false;
    }
    c.objP = getPointOnRay(rt, t);
    c.point = multiplyMV(sphere.xf, c.objP, 1.f);
    return length(sub(r.o, c.point));
}
// src/intersections.h:138-141
PT_DEV vec3 sphereNormal(const DGeom &sphere, const Cand &c) {
    vec3 normal = normalize(multiplyMV(sphere.invT, c.objP, 0.f));
    if (!c.outside) normal = neg(normal);
    return normal;
}

// Texel read as the reference indexes it (src/interactions.h:172-179): nearest, no wrap.  Indices outside the
// image (undefined behaviour in the reference) are clamped to the valid byte range.
PT_DEV uint32_t texel(const DScene &sc, const DTex &t, int pixelID, int c) {
    long long idx = (long long)pixelID * t.ch + c;
    long long n = (long long)t.w * t.h * t.ch;
    if (idx < 0) idx = 0;
    if (idx >= n) idx = n - 1;
    return (uint32_t)sc.texels[t.off + (uint64_t)idx];
}

PT_HD vec3 ld3(const float *__restrict__ p) { return V3(p[0], p[1], p[2]); }
PT_HD float __int_as_float_hd(int v) { float f; __builtin_memcpy(&f, &v, 4); return f; }
PT_HD int __float_as_int_hd(float f) { int v; __builtin_memcpy(&v, &f, 4); return v; }

// One triangle of glm::intersectRayTriangle (glm/gtx/intersect.inl:37-74, single sided: a < epsilon => miss) with
// e1 = v1 - v0 and e2 = v2 - v0 taken from the upload-time table.
PT_HD bool rayTriangle(vec3 orig, vec3 dir, vec3 v0, vec3 e1, vec3 e2, float &bx, float &by) {
    vec3 p = cross(dir, e2);
    float a = dot(e1, p);
    if (a < 1.1920928955078125e-07f) return false;     // FLT_EPSILON
    float f = pt_rcp_pos(a);                           // (a >= FLT_EPSILON here)
    vec3 s = sub(orig, v0);
    bx = f * dot(s, p);
    if (bx < 0.0f) return false;
    if (bx > 1.0f) return false;
    vec3 q = cross(s, e1);
    by = f * dot(dir, q);
    if (by < 0.0f) return false;
    if (by + bx > 1.0f) return false;
    float bz = f * dot(e2, q);
    return bz >= 0.0f;
}

// ---- BVH traversal (builder and layout: pt_bvh.h) ------------------------------------------------------------------

// The box test of both traversals.  It is not part of the reference's arithmetic -- it only decides which triangles are
// looked at -- so it is free to be any test that never misses a box the ray reaches WITHIN `slack`: planes through one FMA
// each, t = plane * (1/d) - o * (1/d), and every slab widened by slack * |1/d| (a distance `slack` along that axis).
// slack is per ray, bvhSlack() below; why it is what makes the tree equal the loop is derived in pt_bvh.h.
struct RaySlab { float ix, iy, iz, ox, oy, oz, sx, sy, sz; };
PT_HD RaySlab makeRaySlab(vec3 o, vec3 d, float slack) {
    const float tiny = 1e-20f;
    const float ddx = __builtin_fabsf(d.x) < tiny ? __builtin_copysignf(tiny, d.x) : d.x;
    const float ddy = __builtin_fabsf(d.y) < tiny ? __builtin_copysignf(tiny, d.y) : d.y;
    const float ddz = __builtin_fabsf(d.z) < tiny ? __builtin_copysignf(tiny, d.z) : d.z;
    RaySlab r;
    r.ix = 1.0f / ddx; r.iy = 1.0f / ddy; r.iz = 1.0f / ddz;
    r.ox = o.x * r.ix; r.oy = o.y * r.iy; r.oz = o.z * r.iz;
    r.sx = slack * __builtin_fabsf(r.ix); r.sy = slack * __builtin_fabsf(r.iy); r.sz = slack * __builtin_fabsf(r.iz);
    return r;
}
// false = skip the subtree: the widened box is missed, lies behind the origin, or starts beyond the best distance so far.
// (The best distance is the distance to a point INSIDE its triangle, hence inside every box around that triangle, so a box
// that starts beyond it cannot hold a nearer hit whatever the rounding of the triangle test; 1.0001 covers the rounding
// of the distance itself.)  Any NaN => visit.
PT_HD bool slabEntry(const BvhQuad &A, const BvhQuad &B, const RaySlab &r, float tmin, float &tn) {
    const float x0 = __builtin_fmaf(A.x, r.ix, -r.ox), x1 = __builtin_fmaf(B.x, r.ix, -r.ox);
    const float y0 = __builtin_fmaf(A.y, r.iy, -r.oy), y1 = __builtin_fmaf(B.y, r.iy, -r.oy);
    const float z0 = __builtin_fmaf(A.z, r.iz, -r.oz), z1 = __builtin_fmaf(B.z, r.iz, -r.oz);
    tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1) - r.sx, __builtin_fminf(y0, y1) - r.sy), __builtin_fminf(z0, z1) - r.sz);
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1) + r.sx, __builtin_fmaxf(y0, y1) + r.sy), __builtin_fmaxf(z0, z1) + r.sz);
    return !((tf < tn) || (tf < 0.0f) || (tn > tmin * 1.0001f));
}
// The per-ray slack: 2^11 u (|o - c| + 4 R), u = 2^-24, c and R the centre and half diagonal of the tree's root box
// (pt_bvh.h, "Why the tree equals the loop").
PT_HD float bvhSlack(const BvhQuad &A, const BvhQuad &B, vec3 o) {
    const vec3 c = V3(0.5f * (A.x + B.x), 0.5f * (A.y + B.y), 0.5f * (A.z + B.z));
    const vec3 h = V3(0.5f * (B.x - A.x), 0.5f * (B.y - A.y), 0.5f * (B.z - A.z));
    const float s = 1.220703125e-4f * (length(sub(o, c)) + 4.0f * length(h));       // 2^11 * 2^-24 = 2^-13
    return s == s ? s : 3.0e38f;                       // non-finite inputs: never cull
}

// Nearest face of the mesh whose tree starts at node `root`, object-space ray (o, d normalised as the reference
// does).  Returns the object-space distance (FLT_MAX: none); face = index inside the geom, b0/b1 = its barycentrics.
PT_HD float bvhNearest(const BvhQuad *__restrict__ nodes, const float *__restrict__ tris, int root, vec3 o, vec3 d,
                       int &face, float &b0o, float &b1o, int *visited = nullptr) {
    const RaySlab rs = makeRaySlab(o, d, bvhSlack(nodes[2 * root], nodes[2 * root + 1], o));
    float tmin = 3.402823466e+38f;
    face = -1; b0o = 0.f; b1o = 0.f;
    int n = root;
    while (n >= 0) {
        const BvhQuad A = nodes[2 * n], B = nodes[2 * n + 1];
        float tn_;
        const bool skip = !slabEntry(A, B, rs, tmin, tn_);
        if (visited) ++*visited;
        if (skip) { n = A.w; continue; }
        const int count = (int)((uint32_t)B.w >> 28), first = B.w & 0x0fffffff;
        if (count == 0) { n = n + 1; continue; }
        for (int k = 0; k < count; k++) {
            const float *T = tris + (size_t)(first + k) * BVH_TRI;
            // (e1 = p1 - v0, e2 = p2 - v0: the subtractions the upload-time table holds, done here -- same IEEE results, one load less)
            const vec3 v0 = V3(T[0], T[1], T[2]), p1 = V3(T[3], T[4], T[5]), p2 = V3(T[6], T[7], T[8]);
            const vec3 e1 = sub(p1, v0), e2 = sub(p2, v0);
            float b0, b1;
            if (rayTriangle(o, d, v0, e1, e2, b0, b1)) {
                const float w = 1 - b0 - b1;
                const vec3 p = add(add(scale(v0, w), scale(p1, b0)), scale(p2, b1));
                const float t = length(sub(o, p));
                int f;
                __builtin_memcpy(&f, &T[9], 4);
                if (t < tmin || (t == tmin && f < face)) { tmin = t; face = f; b0o = b0; b1o = b1; }
            }
        }
        n = A.w;
    }
    return tmin;
}

// The same search front to back: an explicit stack (entry k of this lane at stack[k * stride]) instead of the skip links,
// both children of a node tested together and the nearer one visited first, so that the best distance shrinks early and
// prunes most of the rest -- about half the box tests and a quarter of the dependent memory round trips of bvhNearest on
// the 20448-triangle mesh.  Visits another superset of the possible winners, applies the same per-triangle arithmetic
// and the same (distance, face index) minimum: same answer.  Needs depth <= cap - 1 (pt_bvh.h records the depth).
PT_HD float bvhNearestOrdered(const BvhQuad *__restrict__ nodes, const float *__restrict__ tris, int root, vec3 o, vec3 d,
                              int &face, float &b0o, float &b1o, int32_t *stack, int stride, int *visited = nullptr) {
    const RaySlab rs = makeRaySlab(o, d, bvhSlack(nodes[2 * root], nodes[2 * root + 1], o));
    float tmin = 3.402823466e+38f;
    face = -1; b0o = 0.f; b1o = 0.f;
    auto entry = [&](const BvhQuad &A, const BvhQuad &B, float &tn) { return slabEntry(A, B, rs, tmin, tn); };       // false = skip
    int sp = 0;
    int n = root;                 // node in hand (its box is tested when it is taken in hand), -1 = pop
    // The second quad of the node in hand -- leaf range or right child -- travels with it: it was fetched together with the box
    // when the node was tested, so a level costs ONE round trip to memory (the children's boxes), not two.
    BvhQuad Bn = nodes[2 * root + 1];
    {
        float tn;
        if (visited) ++*visited;
        if (!entry(nodes[2 * root], Bn, tn)) return tmin;
    }
    for (;;) {
        const BvhQuad B = Bn;
        const int count = (int)((uint32_t)B.w >> 28), first = B.w & 0x0fffffff;
        int next = -1;
        if (count) {
            for (int k = 0; k < count; k++) {
                const float *T = tris + (size_t)(first + k) * BVH_TRI;
                const vec3 v0 = V3(T[0], T[1], T[2]), p1 = V3(T[3], T[4], T[5]), p2 = V3(T[6], T[7], T[8]);
                const vec3 e1 = sub(p1, v0), e2 = sub(p2, v0);
                float b0, b1;
                if (rayTriangle(o, d, v0, e1, e2, b0, b1)) {
                    const float w = 1 - b0 - b1;
                    const vec3 p = add(add(scale(v0, w), scale(p1, b0)), scale(p2, b1));
                    const float t = length(sub(o, p));
                    int f;
                    __builtin_memcpy(&f, &T[9], 4);
                    if (t < tmin || (t == tmin && f < face)) { tmin = t; face = f; b0o = b0; b1o = b1; }
                }
            }
        } else {
            const int L = n + 1, R = B.w & 0x0fffffff;             // (the right child is named in the node: both children's
            const BvhQuad LA = nodes[2 * L], LB = nodes[2 * L + 1];   //  boxes are requested together, one round trip)
            const BvhQuad RA = nodes[2 * R], RB = nodes[2 * R + 1];
            float tl, tr;
            const bool hl = entry(LA, LB, tl), hr = entry(RA, RB, tr);
            if (visited) *visited += 2;
            if (hl && hr) {
                const bool left_first = tl <= tr;
                stack[sp * stride] = left_first ? R : L;
                sp++;
                next = left_first ? L : R;
                Bn = left_first ? LB : RB;
            } else if (hl) { next = L; Bn = LB; }
            else if (hr) { next = R; Bn = RB; }
        }
        // pop until a node whose box still matters (the best distance may have shrunk since it was pushed)
        while (next < 0) {
            if (sp == 0) return tmin;
            const int c = stack[--sp * stride];
            const BvhQuad CA = nodes[2 * c], CB = nodes[2 * c + 1];
            float tn;
            if (entry(CA, CB, tn)) { next = c; Bn = CB; }
        }
        n = next;
    }
}

// The same search over FOUR-WIDE, QUANTISED nodes (round 3).  Every inner node of the binary tree that is reached in an even number
// of steps from the root gets a wide node holding its (up to four) grandchildren -- or a child, where that child is a leaf.  A wide
// node is ONE 64-byte line: origin xyz, step xyz (floats), four references, and the entries' boxes as 8-bit grid coordinates
// (lo and hi per axis, four entries per 32-bit word): entry box = origin + step * q.  k_mesh walks one tree per lane, every lane
// somewhere else: its time is the number of cache lines its lanes pull through the CU's L1 (binary walk: two children = 64 B per
// level; four full-precision boxes per node brought only 11 %), so a node that costs one line per TWO levels is what pays.
// The grid boxes are rounded OUTWARDS, with the very fused multiply-add the device evaluates them with (pt_bvh.h checks every one),
// so each contains the binary node's stored (inflated) box: the skip rule -- slabEntry on a box that contains the subtree -- and with
// it the argument of pt_bvh.h hold unchanged; the per-triangle arithmetic and the (distance, face) minimum are the same, the answer
// is the same (checked against the loop and the two binary walks on every ray of tests/test_bvh.py).  Reference word: -1 = empty
// slot, bit 31 set = leaf (count << 24 | first triangle in the low 24 bits), else the entry's own wide node.  The entries of a node
// that are hit -- leaves and inner nodes alike -- are visited nearest first, the others pushed farthest first.  `nodes` / `root`:
// the binary tree, for the root box and the slack.
// The walk is written as a state (WideWalk) and two steps -- wideNodeStep: one four-wide node; wideLeafStep: the triangles of the leaf
// in hand, all of them or one -- so that the SAME code serves two schedules: bvhNearestWide below (one ray to its end: "while-while")
// and k_mesh's refilling waves (pt_engine.hip), where a lane whose walk has ended takes the next parked ray instead of idling.
constexpr int32_t WIDE_DONE = (int32_t)0x80000000;           // (a leaf reference with count 0: no leaf is encoded like this)
struct WideWalk {
    vec3 o, d;                                               // object-space ray (d normalised the way meshTestCore does)
    float ix, iy, iz, enx, eny, enz, efx, efy, efz;          // 1/d, and the slab origin terms with the slack inside (see wideNodeStep)
    float tmin; int32_t face; float b0, b1;                  // best so far: distance, face (index inside the geom), its barycentrics
    int32_t sp, n;                                           // stack entries in use; what is in hand: >= 0 a wide node, WIDE_DONE, else a leaf reference
};
// Sets the walk up at the root (A, B = the two quads of the binary tree's root node: its box, for the slack and the first test);
// n = WIDE_DONE when the ray misses the root box.
PT_HD void wideStart(WideWalk &w, const BvhQuad &A, const BvhQuad &B, int wroot, vec3 o, vec3 d, int *visited = nullptr) {
    const RaySlab rs = makeRaySlab(o, d, bvhSlack(A, B, o));
    w.o = o; w.d = d;
    w.tmin = 3.402823466e+38f; w.face = -1; w.b0 = 0.f; w.b1 = 0.f;
    w.ix = rs.ix; w.iy = rs.iy; w.iz = rs.iz;
    w.enx = -(rs.ox + rs.sx); w.eny = -(rs.oy + rs.sy); w.enz = -(rs.oz + rs.sz);
    w.efx = rs.sx - rs.ox; w.efy = rs.sy - rs.oy; w.efz = rs.sz - rs.oz;
    w.sp = 0;
    float tn;
    if (visited) ++*visited;
    w.n = slabEntry(A, B, rs, w.tmin, tn) ? wroot : WIDE_DONE;
}
// One wide node (w.n >= 0): four box tests, the entries that are hit sorted by entry distance, the nearest taken in hand, the others
// pushed farthest first -- leaves and inner entries alike.  Afterwards w.n is an inner node, a leaf reference or WIDE_DONE.
PT_HD void wideNodeStep(WideWalk &w, const BvhWide4 *__restrict__ wide, int32_t *stack, int stride, int *visited = nullptr, int *trace = nullptr) {
    const BvhWide4 *W = wide + (size_t)w.n * 4;
    const BvhWide4 Q0 = W[0], Q1 = W[1], Q2 = W[2], Q3 = W[3];     // one 64-byte line
    if (visited) *visited += 4;
    if (trace) trace[++trace[0]] = 0;
    const float ox = __int_as_float_hd(Q0.a), oy = __int_as_float_hd(Q0.b), oz = __int_as_float_hd(Q0.c);
    const float sx = __int_as_float_hd(Q0.d), sy = __int_as_float_hd(Q1.a), sz = __int_as_float_hd(Q1.b);
    const int r0 = Q1.c, r1 = Q1.d, r2 = Q2.a, r3 = Q2.b;
    const uint32_t lx = (uint32_t)Q2.c, ly = (uint32_t)Q2.d, lz = (uint32_t)Q3.a, hx = (uint32_t)Q3.b, hy = (uint32_t)Q3.c, hz = (uint32_t)Q3.d;
    const float none = __builtin_inff();
    float t0 = none, t1 = none, t2 = none, t3 = none;
    int c0 = -1, c1 = -1, c2 = -1, c3 = -1;
    // which of an entry's two planes per axis the ray meets first is the same for all four entries: the word that holds the
    // NEAR planes' grid coordinates is picked once per node (by the sign of 1/d, which is finite and not 0: makeRaySlab), and the slack
    // is inside the origin terms (en = -(o/d + slack/|d|), ef = -(o/d - slack/|d|)): a box is 12 conversions, 12 fused multiply-adds,
    // a max3, a min3 and three compares -- slabEntry's planes, interval and three reasons to skip, without its minima / maxima per axis.
    const bool negx = w.ix < 0.0f, negy = w.iy < 0.0f, negz = w.iz < 0.0f;
    const uint32_t nxw = negx ? hx : lx, fxw = negx ? lx : hx, nyw = negy ? hy : ly, fyw = negy ? ly : hy, nzw = negz ? hz : lz, fzw = negz ? lz : hz;
    const float tcut = w.tmin * 1.0001f;
#define PT_WIDE_ENTRY(K, REF, TK, CK)                                                                                          \
    if (REF != -1) {                                                                                                   \
        const float ax = __builtin_fmaf((float)((nxw >> (8 * K)) & 255u), sx, ox), bx = __builtin_fmaf((float)((fxw >> (8 * K)) & 255u), sx, ox); \
        const float ay = __builtin_fmaf((float)((nyw >> (8 * K)) & 255u), sy, oy), by = __builtin_fmaf((float)((fyw >> (8 * K)) & 255u), sy, oy); \
        const float az = __builtin_fmaf((float)((nzw >> (8 * K)) & 255u), sz, oz), bz = __builtin_fmaf((float)((fzw >> (8 * K)) & 255u), sz, oz); \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(ax, w.ix, w.enx), __builtin_fmaf(ay, w.iy, w.eny)), __builtin_fmaf(az, w.iz, w.enz)); \
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaf(bx, w.ix, w.efx), __builtin_fmaf(by, w.iy, w.efy)), __builtin_fmaf(bz, w.iz, w.efz)); \
        if (!((tf < tn) || (tf < 0.0f) || (tn > tcut))) { TK = tn; CK = REF; }                                         \
    }
    PT_WIDE_ENTRY(0, r0, t0, c0)
    PT_WIDE_ENTRY(1, r1, t1, c1)
    PT_WIDE_ENTRY(2, r2, t2, c2)
    PT_WIDE_ENTRY(3, r3, t3, c3)
#undef PT_WIDE_ENTRY
    // the entries that were hit, by entry distance: a five-exchange network on (distance, reference) pairs in registers (static
    // indices only); the slots that were not hit carry +inf and sink to the end.  (An entry whose distance is a NaN -- "visit" by
    // slabEntry's rule -- may stay behind an empty slot: every slot is looked at below, so it is visited all the same.)
#define PT_CX(ta, ca, tb, cb) do { const bool sw_ = tb < ta; const float tt_ = sw_ ? tb : ta; const int cc_ = sw_ ? cb : ca; \
                                   tb = sw_ ? ta : tb; cb = sw_ ? ca : cb; ta = tt_; ca = cc_; } while (0)
    PT_CX(t0, c0, t1, c1); PT_CX(t2, c2, t3, c3); PT_CX(t0, c0, t2, c2); PT_CX(t1, c1, t3, c3); PT_CX(t1, c1, t2, c2);
#undef PT_CX
    // nearest in hand, the others onto the stack farthest first
    int sp = w.sp;
    if (c3 != -1) { stack[sp * stride] = c3; sp++; }
    if (c2 != -1) { stack[sp * stride] = c2; sp++; }
    if (c1 != -1) { stack[sp * stride] = c1; sp++; }
    int n = c0;
    if (n == -1) n = sp ? stack[--sp * stride] : WIDE_DONE;
    w.sp = sp; w.n = n;
}
// The leaf in hand (w.n: bit 31, count << 24, first triangle): ALL its triangles (ONE = false), or the first of them, the rest staying
// in hand as a shorter leaf (ONE = true: k_mesh's refilling waves, where a lane with a one-triangle leaf should not wait for its
// neighbour's four).  The per-triangle arithmetic and the (distance, face index) minimum are meshTestCore's.
template <bool ONE>
PT_HD void wideLeafStep(WideWalk &w, const float *__restrict__ tris, int32_t *stack, int stride, int *visited = nullptr, int *trace = nullptr) {
    const int count = (int)(((uint32_t)w.n >> 24) & 0x7fu), first = w.n & 0x00ffffff;
    if (visited) *visited += (ONE ? 1 : count) << 16;       // (triangle tests, counted apart from the node visits in the low half)
    if (trace) trace[++trace[0]] = ONE ? 1 : count;
    const int todo = ONE ? 1 : count;
    for (int j = 0; j < todo; j++) {
        const float *T = tris + (size_t)(first + j) * BVH_TRI;
        // (e1 = p1 - v0, e2 = p2 - v0: the subtractions the upload-time table holds, done here -- same IEEE results, one load less)
        const vec3 v0 = V3(T[0], T[1], T[2]), p1 = V3(T[3], T[4], T[5]), p2 = V3(T[6], T[7], T[8]);
        const vec3 e1 = sub(p1, v0), e2 = sub(p2, v0);
        float b0, b1;
        if (rayTriangle(w.o, w.d, v0, e1, e2, b0, b1)) {
            const float wgt = 1 - b0 - b1;
            const vec3 p = add(add(scale(v0, wgt), scale(p1, b0)), scale(p2, b1));
            const float t = length(sub(w.o, p));
            int f;
            __builtin_memcpy(&f, &T[9], 4);
            if (t < w.tmin || (t == w.tmin && f < w.face)) { w.tmin = t; w.face = f; w.b0 = b0; w.b1 = b1; }
        }
    }
    if (ONE && count > 1) w.n = (int32_t)(0x80000000u | ((uint32_t)(count - 1) << 24) | (uint32_t)(first + 1));
    else w.n = w.sp ? stack[--w.sp * stride] : WIDE_DONE;
}

PT_HD float bvhNearestWide(const BvhQuad *__restrict__ nodes, const BvhWide4 *__restrict__ wide, const float *__restrict__ tris, int root,
                           int wroot, vec3 o, vec3 d, int &face, float &b0o, float &b1o, int32_t *stack, int stride, int *visited = nullptr,
                           int *trace = nullptr) {      // trace (host experiments only, tools/mesh_walk_sim.cpp): trace[0] = n, then n steps: 0 = a node, k > 0 = a leaf of k triangles
    // Two loops in turn ("while-while"): the lanes of a wave first walk INNER nodes together until every lane holds a leaf or is done;
    // then the lanes that hold a leaf test its triangles together and take their next entry from the stack.  The triangle code, an
    // order of magnitude longer than what most lanes of a wave need at any one node, is thereby issued once per leaf a lane visits, not
    // four times per node any lane of the wave visits.  A hit prunes through tmin: every box is tested against the best distance when
    // its node is visited.
    WideWalk w;
    wideStart(w, nodes[2 * root], nodes[2 * root + 1], wroot, o, d, visited);
    for (;;) {
        while (w.n >= 0) wideNodeStep(w, wide, stack, stride, visited, trace);
        if (w.n == WIDE_DONE) break;
        wideLeafStep<false>(w, tris, stack, stride, visited, trace);
    }
    face = w.face; b0o = w.b0; b1o = w.b1;
    return w.tmin;
}

// meshIntersectionTest up to the choice of the nearest face, src/intersections.h:207-233.  Returns the OBJECT-space
// distance, as the reference does.  (intersectionPoint, which the reference also fills, has no reader.)
// LDSF: the caller knows that the triangle tables are staged in LDS (sc.tri_lds && sc.ntri_lds): every table read is then a
// ds_read from a pointer the compiler can see is LDS, instead of a run-time choice per word
template <bool LDSF = false>
PT_DEV float meshTestCore(const DScene &sc, const DGeom &geom, Ray r, Cand &c, int bvhRoot = -1, int j0 = 0, int j1 = 0x7fffffff,
                          int32_t *stack = nullptr, int stride = 0, int wideRoot = -1) {
    Ray q;
    q.o = multiplyMV(geom.inv, r.o, 1.0f);
    q.d = normalize(multiplyMV(geom.inv, r.d, 0.0f));
    float tmin = 3.402823466e+38f;     // FLT_MAX
    int nearest = -1;
    if (bvhRoot >= 0) {
        // same per-triangle arithmetic, same winner (nearest distance, lowest face index): see pt_bvh.h
        float b0, b1;
        if (stack && wideRoot >= 0) tmin = bvhNearestWide(sc.bvh_nodes, sc.bvh_wide, sc.bvh_tris, bvhRoot, wideRoot, q.o, q.d, nearest, b0, b1, stack, stride);
        else if (stack) tmin = bvhNearestOrdered(sc.bvh_nodes, sc.bvh_tris, bvhRoot, q.o, q.d, nearest, b0, b1, stack, stride);
        else tmin = bvhNearest(sc.bvh_nodes, sc.bvh_tris, bvhRoot, q.o, q.d, nearest, b0, b1);
        if (nearest >= 0) {
            const int f = geom.faceStart + nearest;
            const float w = 1 - b0 - b1;
            c.u = (w * faceWord(sc, f, 3) + b0 * faceWord(sc, f, 8)) + b1 * faceWord(sc, f, 13);
            c.v = (w * faceWord(sc, f, 4) + b0 * faceWord(sc, f, 9)) + b1 * faceWord(sc, f, 14);
        }
        c.face = nearest;
        if (nearest == -1) return -1.f;
        return tmin;
    }
    if (j1 > geom.faceCount) j1 = geom.faceCount;
    for (int j = j0; j < j1; j++) {       // (the whole face list unless the caller spreads it over lanes)
        vec3 v0, e1, e2;
        if (LDSF || triLds(sc)) {      // broadcast ds_reads: every lane reads the same triangle
            const float *t9 = reinterpret_cast<const float *>(pt_lds) + (size_t)(geom.faceStart + j) * 9;
            v0 = V3(t9[0], t9[1], t9[2]); e1 = V3(t9[3], t9[4], t9[5]); e2 = V3(t9[6], t9[7], t9[8]);
        } else {
            const float *__restrict__ t9 = sc.tri9 + (size_t)(geom.faceStart + j) * 9;
            v0 = ld3(t9); e1 = ld3(t9 + 3); e2 = ld3(t9 + 6);
        }
        float b0, b1;
        if (rayTriangle(q.o, q.d, v0, e1, e2, b0, b1)) {
            const int f = geom.faceStart + j;
            // the face record (3 x pos xyz, uv) of the hit, through ONE pointer (LDS when the caller said so at compile time)
            const float *F = LDSF ? reinterpret_cast<const float *>(pt_lds) + sc.ntri_lds * 9 + f * 15
                                  : (triLds(sc) ? reinterpret_cast<const float *>(pt_lds) + sc.ntri_lds * 9 + f * 15 : sc.faces + (size_t)f * 15);
            vec3 p1 = V3(F[5], F[6], F[7]), p2 = V3(F[10], F[11], F[12]);
            float w = 1 - b0 - b1;
            vec3 p = add(add(scale(v0, w), scale(p1, b0)), scale(p2, b1));
            float t = length(sub(q.o, p));          // glm::distance(p, q.origin)
            if (t < tmin) {
                tmin = t;
                nearest = j;
                c.u = (w * F[3] + b0 * F[8]) + b1 * F[13];
                c.v = (w * F[4] + b0 * F[9]) + b1 * F[14];
            }
        }
    }
    c.face = nearest;
    if (nearest == -1) return -1.f;
    return tmin;
}
// geometric normal (+ optional tangent-space bump map) of the chosen face, src/intersections.h:237-279
// geoN = the unperturbed normal (the reference derives its unused `outside` flag from it, :243)
PT_DEV vec3 meshNormal(const DScene &sc, const DGeom &geom, const Cand &c, vec3 &geoN) {
    const int fi = geom.faceStart + c.face;
    float tri[15];
#pragma unroll
    for (int k = 0; k < 15; k++) tri[k] = faceWord(sc, fi, k);
    vec3 e1 = sub(ld3(tri + 5), ld3(tri));
    vec3 e2 = sub(ld3(tri + 10), ld3(tri));
    vec3 objN = normalize(cross(e1, e2));
    vec3 normal = normalize(multiplyMV(geom.invT, objN, 0.f));
    geoN = normal;
    const DTex &bump = geom.tex[3];
    if (geom.type == G_OBJ && bump.ch) {
        float dUV1x = tri[8] - tri[3], dUV1y = tri[9] - tri[4];
        float dUV2x = tri[13] - tri[3], dUV2y = tri[14] - tri[4];
        float f = 1.0f / (dUV1x * dUV2y - dUV2x * dUV1y);
        vec3 tangent, bitangent;
        tangent.x = f * (dUV2y * e1.x - dUV1y * e2.x);
        tangent.y = f * (dUV2y * e1.y - dUV1y * e2.y);
        tangent.z = f * (dUV2y * e1.z - dUV1y * e2.z);
        tangent = normalize(tangent);
        bitangent.x = f * (-dUV2x * e1.x + dUV1x * e2.x);
        bitangent.y = f * (-dUV2x * e1.y + dUV1x * e2.y);
        bitangent.z = f * (-dUV2x * e1.z + dUV1x * e2.z);
        bitangent = normalize(bitangent);
        vec3 T = normalize(multiplyMV(geom.xf, tangent, 0.f));
        vec3 B = normalize(multiplyMV(geom.xf, bitangent, 0.f));
        vec3 N = normal;
        int coordU = (int)(c.u * bump.w);
        int coordV = (int)(c.v * bump.h);
        int pixelID = coordV * bump.w + coordU;
        uint32_t colR = texel(sc, bump, pixelID, 0), colG = texel(sc, bump, pixelID, 1), colB = texel(sc, bump, pixelID, 2);
        vec3 tsn = normalize(V3(colR / 255.f, colG / 255.f, colB / 255.f));
        tsn = normalize(V3(tsn.x * 2.0f - 1.0f, tsn.y * 2.0f - 1.0f, tsn.z * 2.0f - 1.0f));
        vec3 w3 = V3(T.x * tsn.x + B.x * tsn.y + N.x * tsn.z,        // mat3(T,B,N) * v, glm type_mat3x3.inl:487
                     T.y * tsn.x + B.y * tsn.y + N.y * tsn.z,
                     T.z * tsn.x + B.z * tsn.y + N.z * tsn.z);
        normal = normalize(w3);
    }
    return normal;
}

struct Hit {
    float t;          // -1 = miss
    vec3 n;
    float u, v;
    int32_t geom, mat;
    int32_t ncode;    // cube hits: which of the cube's six tabulated normals n is (key aux & 7: axis + 1 | side << 2; 0: none recorded) -- what
                      // a stored path of a cubes-only material carries instead of the normal itself (cubeNormalByCode); 0 for every other hit
};

// triangleIntersectionLocalTest + objTriIntersectionTest, src/intersections.h:175-205, 284-315 -- DEAD CODE in the reference (the
// only call is commented out, src/pathtrace.cu:313) and on no path here either: restated as a known-answer function for SURVEY 8(a10)
// (ptx_kat_obj_tri_test; goldens produced by calling the reference's own functions, tests/golden/dead_tri_kat.npz).  Plane hit, then the
// three sub-triangle areas against the triangle's; nearest face by OBJECT-space distance, which is also what it returns.
PT_DEV float triangleLocalTest(vec3 ro, vec3 rd, vec3 v0, vec3 v1, vec3 v2, vec3 &intersectionPoint, vec3 &normal) {
    const vec3 planeNormal = normalize(cross(sub(v1, v0), sub(v2, v0)));
    const float t = dot(planeNormal, sub(v0, ro)) / dot(planeNormal, rd);
    if (t < 0) return -1.f;
    const vec3 p = add(ro, scale(rd, t));
    const float S = 0.5f * length(cross(sub(v0, v1), sub(v0, v2)));
    const float s1 = 0.5f * length(cross(sub(p, v1), sub(p, v2))) / S;
    const float s2 = 0.5f * length(cross(sub(p, v2), sub(p, v0))) / S;
    const float s3 = 0.5f * length(cross(sub(p, v0), sub(p, v1))) / S;
    const float sum = s1 + s2 + s3;
    if (s1 >= 0 && s1 <= 1 && s2 >= 0 && s2 <= 1 && s3 >= 0 && s3 <= 1 && __builtin_fabsf(sum - 1.0f) < 1.1920928955078125e-07f) {
        intersectionPoint = p;
        normal = planeNormal;
        return t;
    }
    return -1.f;
}
PT_DEV float objTriTest(const DScene &sc, const DGeom &geom, Ray r, vec3 &intersectionPoint, vec3 &normal, bool &outside) {
    float min_tri_t = 3.402823466e+38f;
    vec3 tmp_i = V3(0.f, 0.f, 0.f), tmp_n = V3(0.f, 0.f, 0.f), min_i = V3(0.f, 0.f, 0.f), min_n = V3(0.f, 0.f, 0.f);
    int nearest = -1;
    Ray q;
    q.o = multiplyMV(geom.inv, r.o, 1.0f);
    q.d = normalize(multiplyMV(geom.inv, r.d, 0.0f));
    for (int j = 0; j < geom.faceCount; j++) {
        const float *tri = sc.faces + (size_t)(geom.faceStart + j) * 15;
        const float tt = triangleLocalTest(q.o, q.d, V3(tri[0], tri[1], tri[2]), V3(tri[5], tri[6], tri[7]), V3(tri[10], tri[11], tri[12]), tmp_i, tmp_n);
        if (tt > 0 && tt < min_tri_t) { min_i = tmp_i; min_n = tmp_n; min_tri_t = tt; nearest = j; }
    }
    if (nearest == -1) return -1.f;
    intersectionPoint = multiplyMV(geom.xf, min_i, 1.f);
    normal = normalize(multiplyMV(geom.invT, min_n, 0.f));
    outside = dot(normal, r.d) < 0;
    return min_tri_t;
}

// Header of geom i (transform, inverseTransform, type, material, face range) through the SCALAR memory path: the
// index is wave-uniform and the table is immutable while a kernel runs, so reading it as constant address space
// lets the compiler use s_load and keep the matrices in SGPRs instead of one VGPR copy per lane.
PT_DEV void loadGeomHead(const DGeom *geoms, int i, DGeom &g) {
    typedef const __attribute__((address_space(4))) float cfloat;
    typedef const __attribute__((address_space(4))) int32_t cint;
    cfloat *pf = (cfloat *)(const float *)geoms[i].xf;
#pragma unroll
    for (int k = 0; k < 16; k++) g.xf[k] = pf[k];
#pragma unroll
    for (int k = 0; k < 16; k++) g.inv[k] = pf[16 + k];
    cint *pi = (cint *)(const int32_t *)&geoms[i].type;
    g.type = pi[0]; g.materialid = pi[1]; g.faceStart = pi[2]; g.faceCount = pi[3];
}

// body of computeIntersections, src/pathtrace.cu:270-343: nearest t > 0 over all geoms, lowest index wins ties.
// A miss leaves materialId = 0 (the reference's full-frame memset, :501), which is what the material sort sees.
PT_DEV void intersectScene(const DScene &sc, Ray ray, Hit &h) {
    float t_min = 3.402823466e+38f;
    h.t = -1.f; h.n = V3(0.f, 0.f, 0.f); h.u = 0.f; h.v = 0.f; h.geom = 0; h.mat = 0;
    int hit_geom_index = -1;
    Cand best;
    best.axis = -1; best.sgn = 0.f; best.objP = V3(0.f, 0.f, 0.f); best.outside = true; best.face = -1; best.u = best.v = 0.f;
    float tmp_u = 0.f, tmp_v = 0.f;          // the reference's tmp_uv survives from one mesh to the next geom
    for (int i = 0; i < sc.ngeoms; i++) {
        DGeom geom;
        loadGeomHead(sc.geoms, i, geom);
        float t = 0.f;
        Cand c;
        c.axis = -1; c.sgn = 0.f; c.objP = V3(0.f, 0.f, 0.f); c.outside = true; c.face = -1; c.u = tmp_u; c.v = tmp_v;
        bool tested = true;
        if (geom.type == G_CUBE) t = boxTestCore(geom, ray, c);
        else if (geom.type == G_SPHERE) t = sphereTestCore(geom, ray, c);
        else if (geom.type == G_OBJ) { t = meshTestCore(sc, geom, ray, c, sc.bvh_root ? sc.bvh_root[i] : -1); tmp_u = c.u; tmp_v = c.v; }
        else tested = false;    // TRIANGLE has no test routine in the reference: t keeps a value that never wins
        if (tested && t > 0.0f && t_min > t) {
            t_min = t;
            hit_geom_index = i;
            best = c;
        }
    }
    if (hit_geom_index != -1) {
        const DGeom &geom = sc.geoms[hit_geom_index];
        h.t = t_min;
        h.geom = hit_geom_index;
        h.mat = geom.materialid;
        h.u = best.u; h.v = best.v;
        if (geom.type == G_CUBE) h.n = boxNormal(geom, best);
        else if (geom.type == G_SPHERE) h.n = sphereNormal(geom, best);
        else { vec3 geoN; h.n = meshNormal(sc, geom, best, geoN); }
    }
}

// ---- per-lane candidate lists -------------------------------------------------------------------------------------
// computeIntersections asks every ray about every geom.  In a wave of incoherent rays each lane really needs two or
// three of them, but the wave pays for all.  Here every lane first collects the geoms whose conservative world-space
// box its ray can reach (a miss of that box implies a miss of the exact test: the box is inflated far beyond fp32
// error, see make_world_aabb in pt_engine.hip), then the wave loops "each lane takes ITS next cube or sphere" with the
// geom tables gathered per lane from LDS.  The exact tests and their arithmetic are unchanged, so is the result:
// nearest t > 0, lowest geom index on ties (the reference's strict `t_min > t` in index order).  The kernels go one
// step further (tileIntersect in pt_engine.hip): the (ray, geom) pairs of a whole 256-path tile are pooled in LDS and
// worked off by dense waves.

// xyz of mat4*vec4 for a matrix stored as 3 rows of 4 (same products and sums as multiplyMV)
PT_HD vec3 mulRows(const float *r, vec3 v, float w) {
    vec3 o;
    o.x = (r[0] * v.x + r[1] * v.y) + (r[2] * v.z + r[3] * w);
    o.y = (r[4] * v.x + r[5] * v.y) + (r[6] * v.z + r[7] * w);
    o.z = (r[8] * v.x + r[9] * v.y) + (r[10] * v.z + r[11] * w);
    return o;
}

// Candidate mask of one ray: bit i set <=> geom i's conservative world box is reached (uniform loop over geoms, two
// boxes per trip so their scalar loads overlap; 8 floats per box: min xyz, pad, max xyz, pad).  Which of the set bits
// are cubes, spheres or meshes is a per-scene constant (sc.cube_bits / sphere_bits / mesh_bits).
// SUBSET: only the geoms whose bit is set in `subset` are tested (a wave-uniform mask: the camera-ray bounce knows per tile which geoms
// its pixels can see at all, k_bounce's tile_geoms); the others' bits stay 0.
template <bool SUBSET = false>
PT_DEV uint32_t cullMask(const DScene &sc, Ray ray, uint32_t subset = 0xffffffffu) {
    typedef const __attribute__((address_space(4))) float cfloat;
    cfloat *ab = (cfloat *)sc.aabb;
    const float tiny = 1e-20f;
    const float ddx = __builtin_fabsf(ray.d.x) < tiny ? __builtin_copysignf(tiny, ray.d.x) : ray.d.x;
    const float ddy = __builtin_fabsf(ray.d.y) < tiny ? __builtin_copysignf(tiny, ray.d.y) : ray.d.y;
    const float ddz = __builtin_fabsf(ray.d.z) < tiny ? __builtin_copysignf(tiny, ray.d.z) : ray.d.z;
    const float ix = __builtin_amdgcn_rcpf(ddx), iy = __builtin_amdgcn_rcpf(ddy), iz = __builtin_amdgcn_rcpf(ddz);
    // This pre-test is not the reference's arithmetic (it only has to be conservative, and the boxes are inflated by 1e-3 +
    // 1e-4 |coordinate|, four orders of magnitude beyond its rounding).  The table holds a box as CENTRE and HALF EXTENT
    // (world_box_centre_half): with m = (c - o) / d the slab of an axis is [m - h / |d|, m + h / |d|] whatever the sign of d -- three
    // fused multiply-adds per axis (the centre and the half extent are scalar operands) and no per-axis minimum / maximum, which
    // issue at 0.64 of a multiply-add's rate on this chip: 13 instead of 16 vector instructions per geom for the stage that every
    // ray runs against every geom.
    const float ox = -(ray.o.x * ix), oy = -(ray.o.y * iy), oz = -(ray.o.z * iz);
    const float ax = __builtin_fabsf(ix), ay = __builtin_fabsf(iy), az = __builtin_fabsf(iz);
    uint32_t mask = 0;
    const int n = sc.ngeoms;
    auto slab = [&](const float *bx) {
        const float mx = __builtin_fmaf(bx[0], ix, ox), my = __builtin_fmaf(bx[1], iy, oy), mz = __builtin_fmaf(bx[2], iz, oz);
        const float x0 = __builtin_fmaf(bx[4], -ax, mx), x1 = __builtin_fmaf(bx[4], ax, mx);
        const float y0 = __builtin_fmaf(bx[5], -ay, my), y1 = __builtin_fmaf(bx[5], ay, my);
        const float z0 = __builtin_fmaf(bx[6], -az, mz), z1 = __builtin_fmaf(bx[6], az, mz);
        const float tn = __builtin_fmaxf(__builtin_fmaxf(x0, y0), z0);
        const float tf = __builtin_fminf(__builtin_fminf(x1, y1), z1);
        return !((tf < tn) || (tf < 0.0f));      // any NaN => not culled
    };
    if (SUBSET) {
        // (a scalar loop over the set bits, two boxes per trip like the plain loop; n <= 32 wherever the masks are in use)
        uint32_t todo = __builtin_amdgcn_readfirstlane(n >= 32 ? subset : subset & ((1u << n) - 1u));
        while (todo) {
            const int i = __builtin_ctz(todo);
            todo &= todo - 1;
            const int j = todo ? __builtin_ctz(todo) : i;
            todo &= todo - 1;                    // (0 & anything = 0: harmless when j == i)
            float bx[2][8];
#pragma unroll
            for (int k = 0; k < 8; k++) { bx[0][k] = ab[i * 8 + k]; bx[1][k] = ab[j * 8 + k]; }
            mask |= slab(bx[0]) ? (1u << i) : 0u;
            mask |= slab(bx[1]) ? (1u << j) : 0u;
        }
        return mask;
    }
    // From the last geom down, two boxes per trip, each verdict shifted in from the right (mask + mask + verdict: ONE add-with-carry
    // whose carry is the compare's result, instead of a move, a select and an or): geom 0's verdict arrives last and is bit 0.
    int i = n - 1;
    if (n & 1) {
        float bx[8];
#pragma unroll
        for (int k = 0; k < 8; k++) bx[k] = ab[i * 8 + k];
        mask = slab(bx) ? 1u : 0u;
        i--;
    }
    for (; i > 0; i -= 2) {
        float bx[2][8];
#pragma unroll
        for (int k = 0; k < 8; k++) { bx[0][k] = ab[i * 8 + k]; bx[1][k] = ab[(i - 1) * 8 + k]; }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned long long verdict = __builtin_amdgcn_ballot_w64(slab(bx[h]));
            // (written out: the compiler turns `mask + mask + verdict` back into a shift, a select and an or)
            asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(mask) : "s"(verdict) : "vcc");
        }
    }
    return mask;
}

// Result of one (ray, geom) test as a 64-bit key: fp32 bits of t (t > 0, so they order like t) << 32 | geom << 24 |
// 24 bits that let the winner rebuild its normal (cube: axis+1, sign, outside; mesh: face).  The minimum key over a
// ray's candidates is the reference's answer: nearest t, lowest geom index on ties.
constexpr unsigned long long KEY_NONE = ~0ull;
PT_DEV unsigned long long packKey(float t, int g, uint32_t aux) {
    return ((unsigned long long)(uint32_t)__float_as_int(t) << 32) | ((unsigned long long)(uint32_t)g << 24) | (aux & 0xffffffu);
}

// cube or sphere g (per-lane g: tables gathered from LDS) against one ray.  Accepts what the reference's
// `t > 0.0f && t_min > t` can accept (t_min starts at FLT_MAX).
PT_DEV unsigned long long primKey(const float *gtab, int g, Ray ray) {
    const float *G = gtab + g * GTAB_WORDS;
    float inv[12];
#pragma unroll
    for (int k = 0; k < 12; k++) inv[k] = G[k];
    const int type = __float_as_int(G[36]);
    Ray q;
    q.o = mulRows(inv, ray.o, 1.0f);
    q.d = normalize(mulRows(inv, ray.d, 0.0f));
    vec3 objP = V3(0.f, 0.f, 0.f);
    uint32_t aux = 0;
    bool hit = false;
    if (type == G_CUBE) {                      // boxIntersectionTest, src/intersections.h:54-84
        float tmin = -1e38f, tmax = 1e38f;
        int tmin_axis = -1, tmax_axis = -1;
        float tmin_s = 0.f, tmax_s = 0.f;
        const float qo[3] = {q.o.x, q.o.y, q.o.z};
        const float qd[3] = {q.d.x, q.d.y, q.d.z};
#pragma unroll
        for (int xyz = 0; xyz < 3; ++xyz) {
            float t1 = (-0.5f - qo[xyz]) / qd[xyz];
            float t2 = (+0.5f - qo[xyz]) / qd[xyz];
            float ta = fmin_glm(t1, t2);
            float tb = fmax_glm(t1, t2);
            float ns = t2 < t1 ? +1.f : -1.f;
            if (ta > 0 && ta > tmin) { tmin = ta; tmin_axis = xyz; tmin_s = ns; }
            if (tb < tmax) { tmax = tb; tmax_axis = xyz; tmax_s = ns; }
        }
        if (tmax >= tmin && tmax > 0) {
            bool outside = true;
            if (tmin <= 0) { tmin = tmax; tmin_axis = tmax_axis; tmin_s = tmax_s; outside = false; }
            aux = (uint32_t)(tmin_axis + 1) | (tmin_s > 0.f ? 4u : 0u) | (outside ? 8u : 0u);
            objP = getPointOnRay(q, tmin);
            hit = true;
        }
    } else {                                   // sphereIntersectionTest, src/intersections.h:104-135
        const float radius = .5f;
        float vDotDirection = dot(q.o, q.d);
        float radicand = vDotDirection * vDotDirection - (dot(q.o, q.o) - radius * radius);
        if (!(radicand < 0)) {
            float squareRoot = pt_sqrt(radicand);
            float firstTerm = -vDotDirection;
            float t1 = firstTerm + squareRoot;
            float t2 = firstTerm - squareRoot;
            if (!(t1 < 0 && t2 < 0)) {
                float t;
                bool outside;
                if (t1 > 0 && t2 > 0) { t = fmin_glm(t1, t2); outside = true; }
                else { t = fmax_glm(t1, t2); outside = false; }
                aux = outside ? 8u : 0u;
                objP = getPointOnRay(q, t);
                hit = true;
            }
        }
    }
    if (!hit) return KEY_NONE;
    float xf[12];
#pragma unroll
    for (int k = 0; k < 12; k++) xf[k] = G[12 + k];
    const vec3 point = mulRows(xf, objP, 1.0f);
    const float t = length(sub(ray.o, point));
    if (!(t > 0.0f && t < 3.402823466e+38f)) return KEY_NONE;
    return packKey(t, g, aux);
}

// mesh g against one ray (g may differ per lane: its header is gathered from LDS)
// One chunk of a small mesh (faces [j0, j0 + MESH_CHUNK) of the geom, tables in LDS) for tileIntersect.  Same tests, same
// comparisons in the same order as the loop of meshTestCore; what differs is WHEN the distance of an accepted triangle is
// evaluated.  In a wave whose lanes hold different rays some lane hits almost every triangle, so the loop pays for the hit
// path (two vertices, the barycentric point, a square root) at every triangle for all lanes.  Here the triangle tests run
// first and only remember hit bits and barycentrics; the hit path then runs once per hit a lane HAS -- in face order, with
// the loop's strict `t < tmin` -- i.e. about once per chunk instead of MESH_CHUNK times.
PT_DEV float meshChunkTestLds(const DScene &sc, const float *inv12, int faceStart, int faceCount, int j0, Ray r, int &nearest) {
    Ray q;
    q.o = mulRows(inv12, r.o, 1.0f);                 // = multiplyMV(geom.inverseTransform, ., .): same products, same sums
    q.d = normalize(mulRows(inv12, r.d, 0.0f));
    const float *L = reinterpret_cast<const float *>(pt_lds);
    float bx[MESH_CHUNK], by[MESH_CHUNK];
    uint32_t hits = 0;
#pragma unroll
    for (int k = 0; k < MESH_CHUNK; k++) {
        bx[k] = by[k] = 0.f;
        const int j = j0 + k;
        if (j < faceCount) {
            const float *t9 = L + (size_t)(faceStart + j) * 9;
            const vec3 v0 = V3(t9[0], t9[1], t9[2]), e1 = V3(t9[3], t9[4], t9[5]), e2 = V3(t9[6], t9[7], t9[8]);
            float b0, b1;
            if (rayTriangle(q.o, q.d, v0, e1, e2, b0, b1)) { hits |= 1u << k; bx[k] = b0; by[k] = b1; }
        }
    }
    float tmin = 3.402823466e+38f;
    nearest = -1;
    while (hits) {                                   // increasing face index, as the loop visits them
        const int k = __ffs((int)hits) - 1;
        hits &= hits - 1;
        float b0 = bx[0], b1 = by[0];
#pragma unroll
        for (int m = 1; m < MESH_CHUNK; m++) { b0 = k == m ? bx[m] : b0; b1 = k == m ? by[m] : b1; }
        const int f = faceStart + j0 + k;
        const float *t9 = L + (size_t)f * 9;
        const float *F = L + sc.ntri_lds * 9 + f * 15;
        const vec3 v0 = V3(t9[0], t9[1], t9[2]), p1 = V3(F[5], F[6], F[7]), p2 = V3(F[10], F[11], F[12]);
        const float w = 1 - b0 - b1;
        const vec3 p = add(add(scale(v0, w), scale(p1, b0)), scale(p2, b1));
        const float t = length(sub(q.o, p));          // glm::distance(p, q.origin)
        if (t < tmin) { tmin = t; nearest = j0 + k; }
    }
    return nearest == -1 ? -1.f : tmin;
}

template <bool LDSF = false>
PT_DEV unsigned long long meshKey(const DScene &sc, const float *gtab, int g, Ray ray, int chunk = -1, int32_t *stack = nullptr,
                                  int stride = 0) {
    const float *G = gtab + g * GTAB_WORDS;
    DGeom geom;
    // rows -> glm column-major for the shared mesh routine: only inv is read there
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) geom.inv[c * 4 + r] = G[r * 4 + c];
    geom.type = G_OBJ;
    geom.faceStart = __float_as_int(G[38]); geom.faceCount = __float_as_int(G[39]);
    Cand c;
    c.face = -1; c.u = 0.f; c.v = 0.f;
    // chunk >= 0: faces [chunk * MESH_CHUNK, +MESH_CHUNK) only; the minimum over a mesh's chunks of (t, geom, face) is
    // the loop's answer: nearest distance, lowest face index on ties
    float t;
    if (chunk >= 0) {
        if (chunk * MESH_CHUNK >= geom.faceCount) return KEY_NONE;
        if (LDSF && PT_DEFER_HITS) {
            float inv[12];
#pragma unroll
            for (int k = 0; k < 12; k++) inv[k] = G[k];
            int nearest;
            t = meshChunkTestLds(sc, inv, geom.faceStart, geom.faceCount, chunk * MESH_CHUNK, ray, nearest);
            c.face = nearest;
        } else
            t = meshTestCore<LDSF>(sc, geom, ray, c, -1, chunk * MESH_CHUNK, chunk * MESH_CHUNK + MESH_CHUNK);
    } else {
        // (a stack, when the caller has one and the tree fits it, buys the front-to-back search)
        // (four-wide nodes where the tree has them and the stack is long enough for their walk)
        const bool wideok = stack && sc.bvh_wroot && sc.bvh_wroot[g] >= 0 && sc.bvh_wneed[g] <= sc.bvh_stack;
        const bool ordered = wideok || (stack && sc.bvh_depth && sc.bvh_depth[g] < sc.bvh_stack);
        t = meshTestCore<LDSF>(sc, geom, ray, c, sc.bvh_root ? sc.bvh_root[g] : -1, 0, 0x7fffffff, ordered ? stack : nullptr, stride,
                               wideok ? sc.bvh_wroot[g] : -1);
    }
    if (!(t > 0.0f && t < 3.402823466e+38f)) return KEY_NONE;
    return packKey(t, g, (uint32_t)c.face);
}

// A cube hit's normal from its 3-bit code (the low bits of its key: axis + 1 | side << 2): the tabulated one, or -- no axis recorded,
// NaN inputs -- the reference's zero vector through the same arithmetic.  decodeKey's and, for a stored path that carries the code instead
// of the normal, the next bounce's (same table, same words).
PT_DEV vec3 cubeNormalByCode(const DScene &sc, const float *gtab, int g, int code) {
    const int axis = (code & 3) - 1;
    if (axis >= 0) return cubeNormalTab(sc, g, axis * 2 + ((code & 4) ? 1 : 0));
    const float *G = gtab + g * 40;
    float invT[12];
#pragma unroll
    for (int k = 0; k < 12; k++) invT[k] = G[24 + k];
    return normalize(mulRows(invT, V3(0.f, 0.f, 0.f), 0.0f));
}

// What the winning key stands for: t, geom, material, normal (and texcoords when the scene uses them).
// NO_MESH: the caller knows the winner is a cube or a sphere (pass 1 of the split bounce finishing a ray that reached no mesh box):
// the mesh branch -- barycentrics redone, bump map -- is not compiled into that call site.
template <bool NO_MESH = false>
PT_DEV void decodeKey(const DScene &sc, const float *gtab, unsigned long long key, Ray ray, bool need_uv, Hit &h) {
    h.t = -1.f; h.n = V3(0.f, 0.f, 0.f); h.u = 0.f; h.v = 0.f; h.geom = 0; h.mat = 0; h.ncode = 0;
    if (key == KEY_NONE) return;
    const int g = (int)((key >> 24) & 0xff);
    const uint32_t aux = (uint32_t)(key & 0xffffffu);
    const float *G = gtab + g * GTAB_WORDS;
    const int type = __float_as_int(G[36]);
    h.t = __int_as_float((int)(uint32_t)(key >> 32));
    h.geom = g;
    h.mat = __float_as_int(G[37]);
    if (!NO_MESH && type == G_OBJ) {
        Cand c;
        c.face = (int)aux; c.u = 0.f; c.v = 0.f;
        const DGeom &geom = sc.geoms[g];
        if (need_uv) {      // texcoords of the winning face: redo its barycentrics (same arithmetic, same bits)
            Ray q;
            q.o = multiplyMV(geom.inv, ray.o, 1.0f);
            q.d = normalize(multiplyMV(geom.inv, ray.d, 0.0f));
            const int f = geom.faceStart + c.face;
            vec3 v0 = faceVec(sc, f, 0);
            vec3 e1 = sub(faceVec(sc, f, 5), v0), e2 = sub(faceVec(sc, f, 10), v0);
            float b0 = 0.f, b1 = 0.f;
            rayTriangle(q.o, q.d, v0, e1, e2, b0, b1);
            float w = 1 - b0 - b1;
            c.u = (w * faceWord(sc, f, 3) + b0 * faceWord(sc, f, 8)) + b1 * faceWord(sc, f, 13);
            c.v = (w * faceWord(sc, f, 4) + b0 * faceWord(sc, f, 9)) + b1 * faceWord(sc, f, 14);
        }
        h.u = c.u; h.v = c.v;
        if ((sc.bump_bits >> g) & 1u) {
            vec3 geoN;
            h.n = meshNormal(sc, geom, c, geoN);
        } else {            // no bump map: the face's normal was computed at upload (same arithmetic, same bits)
            h.n = faceNormalTab(sc, __float_as_int(G[38]) + c.face);
        }
    } else {
        if (type == G_CUBE) {   // one of six normals per cube, computed at upload (normalize(invTranspose * +-e_axis))
            h.ncode = (int32_t)(aux & 7u);
            h.n = cubeNormalByCode(sc, gtab, g, h.ncode);
        } else {            // sphere: the object-space hit point is recomputed (same arithmetic as in primKey)
            float invT[12];
#pragma unroll
            for (int k = 0; k < 12; k++) invT[k] = G[24 + k];
            float inv[12];
#pragma unroll
            for (int k = 0; k < 12; k++) inv[k] = G[k];
            Ray q;
            q.o = mulRows(inv, ray.o, 1.0f);
            q.d = normalize(mulRows(inv, ray.d, 0.0f));
            const float radius = .5f;
            float vDotDirection = dot(q.o, q.d);
            float radicand = vDotDirection * vDotDirection - (dot(q.o, q.o) - radius * radius);
            float squareRoot = pt_sqrt(radicand);
            float firstTerm = -vDotDirection;
            float t1 = firstTerm + squareRoot;
            float t2 = firstTerm - squareRoot;
            const bool outside = (aux & 8u) != 0;
            float t = outside ? fmin_glm(t1, t2) : fmax_glm(t1, t2);
            vec3 objP = getPointOnRay(q, t);
            vec3 n = normalize(mulRows(invT, objP, 0.f));
            h.n = outside ? n : neg(n);
        }
    }
}

// Whole per-geom tests as the reference exposes them (used by the per-function parity entry point)
PT_DEV float boxIntersectionTest(const DGeom &g, Ray r, vec3 &point, vec3 &normal, bool &outside) {
    Cand c; c.axis = -1; c.sgn = 0.f; c.outside = true;
    float t = boxTestCore(g, r, c);
    if (t != -1.f) { outside = c.outside; point = c.point; normal = boxNormal(g, c); }
    return t;
}
PT_DEV float sphereIntersectionTest(const DGeom &g, Ray r, vec3 &point, vec3 &normal, bool &outside) {
    Cand c; c.objP = V3(0.f, 0.f, 0.f); c.outside = true;
    float t = sphereTestCore(g, r, c);
    if (t != -1.f) { outside = c.outside; point = c.point; normal = sphereNormal(g, c); }
    return t;
}
PT_DEV float meshIntersectionTest(const DScene &sc, const DGeom &g, Ray r, vec3 &point, vec3 &normal, float &u, float &v,
                                  bool &outside, int bvhRoot = -1) {
    Cand c; c.face = -1; c.u = u; c.v = v;
    float t = meshTestCore(sc, g, r, c, bvhRoot);
    u = c.u; v = c.v;
    if (t != -1.f) {
        Ray q;                                      // src/intersections.h:235,241 (the unused intersectionPoint)
        q.o = multiplyMV(g.inv, r.o, 1.0f);
        q.d = normalize(multiplyMV(g.inv, r.d, 0.0f));
        point = multiplyMV(g.xf, getPointOnRay(q, t), 1.f);
        vec3 geoN;
        normal = meshNormal(sc, g, c, geoN);
        outside = dot(geoN, r.d) < 0;
    }
    return t;
}

// ---- BSDF --------------------------------------------------------------------------------------------------------
#define PT_TWO_PI 6.2831853071795864769252867665590057683943f            // src/utilities.h:13
#define PT_SQRT_OF_ONE_THIRD 0.5773502691896257645091487805019574556476f // src/utilities.h:14

// calculateRandomDirectionInHemisphere, src/interactions.h:11-43
PT_DEV vec3 randomDirectionInHemisphere(vec3 normal, Rng &rng) {
    float up = pt_sqrt(rng.uniform(0.f, 1.f));
    float over = pt_sqrt(1 - up * up);
    float around = rng.uniform(0.f, 1.f) * PT_TWO_PI;
    vec3 notNormal;
    if (__builtin_fabsf(normal.x) < PT_SQRT_OF_ONE_THIRD) notNormal = V3(1, 0, 0);
    else if (__builtin_fabsf(normal.y) < PT_SQRT_OF_ONE_THIRD) notNormal = V3(0, 1, 0);
    else notNormal = V3(0, 0, 1);
    vec3 perp1 = normalize(cross(normal, notNormal));
    vec3 perp2 = normalize(cross(normal, perp1));
    float sn, cs;
    sincos_pt(around, &sn, &cs);
    vec3 a = scale(normal, up);
    vec3 b = scale(perp1, cs * over);
    vec3 c = scale(perp2, sn * over);
    return add(add(a, b), c);
}

// calculateJitteredDirectionHemisphere, src/interactions.h:46-85 -- DEAD CODE in the reference (JITTERED_SAMPLING 0; its call site does
// not compile) and on no path here: a known-answer function for SURVEY 8(a13) (ptx_kat_jittered_hemisphere; goldens produced by calling
// the reference's own function, tests/golden/jitter_kat.npz).  The sampler above on a stratified (iter % n, iter / n) cell.
PT_DEV vec3 jitteredDirectionInHemisphere(vec3 normal, Rng &rng, int iter, int max_iter) {
    const int sqrtVal = (int)(pt_sqrt((float)max_iter) + 0.5f);
    const float invSqrtVal = 1.f / (float)sqrtVal;
    const int x = iter % sqrtVal;
    const int y = (int)((float)iter / (float)sqrtVal);
    float x_point = ((float)x + rng.uniform(0.f, 1.f)) * invSqrtVal;
    x_point = fmin_glm(fmax_glm(x_point, 0.f), 1.f) ;
    float y_point = ((float)y + rng.uniform(0.f, 1.f)) * invSqrtVal;
    y_point = fmin_glm(fmax_glm(y_point, 0.f), 1.f);
    const float up = pt_sqrt(y_point);
    const float over = pt_sqrt(1.f - (up * up));
    const float around = x_point * PT_TWO_PI;
    vec3 notNormal;
    if (__builtin_fabsf(normal.x) < PT_SQRT_OF_ONE_THIRD) notNormal = V3(1, 0, 0);
    else if (__builtin_fabsf(normal.y) < PT_SQRT_OF_ONE_THIRD) notNormal = V3(0, 1, 0);
    else notNormal = V3(0, 0, 1);
    const vec3 perp1 = normalize(cross(normal, notNormal));
    const vec3 perp2 = normalize(cross(normal, perp1));
    float sn, cs;
    sincos_pt(around, &sn, &cs);
    return add(add(scale(normal, up), scale(perp1, cs * over)), scale(perp2, sn * over));
}

// glm reflect: I - N * dot(N, I) * 2
PT_DEV vec3 reflect(vec3 I, vec3 N) { return sub(I, scale(scale(N, dot(N, I)), 2.0f)); }
// glm refract (func_geometric.inl:190-197)
PT_DEV vec3 refract(vec3 I, vec3 N, float eta) {
    float dotValue = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - dotValue * dotValue);
    float f = eta * dotValue + pt_sqrt(k);
    vec3 r = sub(scale(I, eta), scale(N, f));
    return scale(r, (float)(k >= 0.0f));
}
// Schlick term with the reference's mixed precision (src/interactions.h:151-152, :190-191)
PT_DEV float schlick(float IoR1, float IoR2, float cosTheta) {
    float r0 = ((IoR1 - IoR2) / (IoR1 + IoR2)) * ((IoR1 - IoR2) / (IoR1 + IoR2));
    return (float)((double)r0 + (double)(1.0f - r0) * pow5_own(1.0 - (double)cosTheta));
}

struct PathState {
    vec3 o, d, color;
};

// scatterRay, src/interactions.h:111-256.  Returns true when the OBJ emissive-texel branch ended the path
// (the reference sets remainingBounces = 1 there and its caller decrements it to 0).
// TEX = false: the scene has no texture at all (every DTex::ch is 0), so the texel branches and the loads that decide them are
// left out.  The two diffuse cases -- an OBJ geom's and everything else's -- share one copy of the hemisphere sampler: a wave
// that holds both kinds of hit runs it once, and the kernel holds its binary64 sin/cos once.
template <bool TEX = true>
PT_DEV bool scatterRay(const DScene &sc, PathState &ps, vec3 intersect, const Hit &hit, const DMaterial &m, Rng &rng) {
    vec3 n = hit.n;
    if (m.hasReflective > 0) {
        vec3 reflectDir = reflect(ps.d, n);
        float spec = powf_own(fmax_glm(dot(neg(ps.d), reflectDir), 0.0f), m.exponent);
        ps.color = mul(ps.color, scale(V3(m.speccolor[0], m.speccolor[1], m.speccolor[2]), m.hasReflective * spec));
        ps.o = add(intersect, scale(n, 0.01f));
        ps.d = reflectDir;
    } else if (m.hasRefractive > 0) {
        float IoR1 = 1.0f;
        float IoR2 = m.ior;
        float cosTheta = dot(neg(ps.d), n);
        if (cosTheta < 0) {
            n = scale(n, -1.0f);
            IoR1 = IoR2;
            IoR2 = 1.0f;
            cosTheta = __builtin_fabsf(cosTheta);
        }
        float sinTheta = (float)__builtin_sqrt(1.0 - (double)(cosTheta * cosTheta));
        vec3 nd;
        if (IoR1 / IoR2 * sinTheta > 1.0f) {
            nd = reflect(ps.d, n);
        } else {
            float reflect_coeff = schlick(IoR1, IoR2, cosTheta);
            float random = rng.uniform(0.f, 1.f);
            if (random < reflect_coeff) nd = reflect(ps.d, n);
            else nd = refract(ps.d, n, IoR1 / IoR2);
        }
        ps.d = nd;
        ps.color = mul(ps.color, V3(m.speccolor[0], m.speccolor[1], m.speccolor[2]));
        ps.o = add(intersect, scale(nd, 0.01f));
    } else {
        vec3 diffuseColor = V3(m.color[0], m.color[1], m.color[2]);
        if (geomType(sc, hit.geom) == G_OBJ) {
            const DGeom &geom = sc.geoms[hit.geom];
            const DTex &kd = geom.tex[0], &ks = geom.tex[1], &ke = geom.tex[2];
            vec3 emission = V3(0.f, 0.f, 0.f);
            if (TEX && ke.ch) {
                int coordU = (int)(hit.u * ke.w);
                int coordV = (int)(hit.v * ke.h);
                int pixelID = coordV * ke.w + coordU;
                emission = V3(texel(sc, ke, pixelID, 0) / 255.f, texel(sc, ke, pixelID, 1) / 255.f, texel(sc, ke, pixelID, 2) / 255.f);
            }
            const float eps = 1.1920928955078125e-07f;
            if (TEX && (emission.x > eps || emission.y > eps || emission.z > eps)) {
                ps.color = mul(ps.color, scale(emission, 5.0f));
                return true;
            }
            float cosTheta = dot(neg(ps.d), n);
            float reflect_coeff = schlick(1.0f, m.ior, cosTheta);
            float random = rng.uniform(0.f, 1.f);
            if (random < reflect_coeff) {
                vec3 reflectDir = reflect(ps.d, n);
                float spec = 1.0f;      // glm::pow(x, 0.0f) == 1 for every x (src/interactions.h:203)
                vec3 specColor = V3(m.speccolor[0], m.speccolor[1], m.speccolor[2]);
                if (TEX && ks.ch) {
                    int coordU = (int)(hit.u * ks.w);
                    int coordV = (int)(hit.v * ks.h);
                    int pixelID = coordV * ks.w + coordU;
                    specColor = V3(texel(sc, ks, pixelID, 0) / 255.f, texel(sc, ks, pixelID, 1) / 255.f, texel(sc, ks, pixelID, 2) / 255.f);
                }
                specColor = scale(specColor, spec);
                ps.color = mul(ps.color, specColor);
                ps.o = add(intersect, scale(n, 0.01f));
                ps.d = reflectDir;
                return false;
            }
            if (TEX && kd.ch) {
                int coordU = (int)(hit.u * kd.w);
                int coordV = (int)(hit.v * kd.h);
                int pixelID = coordV * kd.w + coordU;
                diffuseColor = V3(texel(sc, kd, pixelID, 0) / 255.f, texel(sc, kd, pixelID, 1) / 255.f, texel(sc, kd, pixelID, 2) / 255.f);
            }
        }
        ps.color = mul(ps.color, diffuseColor);
        vec3 nd = randomDirectionInHemisphere(n, rng);
        ps.d = nd;
        ps.o = add(intersect, scale(nd, 0.01f));
    }
    return false;
}

// ConcentricSampleDisk, src/pathtrace.cu:183-197
PT_DEV void concentricSampleDisk(float px, float py, float &ox, float &oy) {
    float ux = 2.f * px - 1.f, uy = 2.f * py - 1.f;
    if (ux == 0 && uy == 0) { ox = 0; oy = 0; return; }
    float theta, r;
    if (__builtin_fabsf(ux) > __builtin_fabsf(uy)) {
        r = ux;
        theta = 0.785398f * (uy / ux);
    } else {
        r = uy;
        theta = 1.570796f - 0.785398f * (ux / uy);
    }
    float sn, cs;
    sincos_pt(theta, &sn, &cs);
    ox = r * cs;
    oy = r * sn;
}

// body of generateRayFromCamera, src/pathtrace.cu:208-254
PT_DEV void generateRay(const DCamera &cam, int iter, int traceDepth, bool aa, bool dof, int x, int y, PathState &ps) {
    int index = x + (y * cam.resx);
    vec3 origin = V3(cam.position[0], cam.position[1], cam.position[2]);
    float antia_x = (float)x;
    float antia_y = (float)y;
    if (aa) {
        Rng rngANTIA; rngANTIA.seed(iter, index, traceDepth);
        antia_x += rngANTIA.uniform(-0.5f, 0.5f);
        antia_y += rngANTIA.uniform(-0.5f, 0.5f);
    }
    vec3 right = V3(cam.right[0], cam.right[1], cam.right[2]);
    vec3 up = V3(cam.up[0], cam.up[1], cam.up[2]);
    vec3 view = V3(cam.view[0], cam.view[1], cam.view[2]);
    vec3 a = scale(scale(right, cam.pixelLength[0]), antia_x - (float)cam.resx * 0.5f);
    vec3 b = scale(scale(up, cam.pixelLength[1]), antia_y - (float)cam.resy * 0.5f);
    vec3 direction = normalize(sub(sub(view, a), b));
    if (dof) {
        const float lensRadius = 0.8f;
        const float focalDistance = 11.0f;
        Rng rng; rng.seed(iter, index, traceDepth);
        float s0 = rng.uniform(0.f, 1.f);
        float s1 = rng.uniform(0.f, 1.f);
        float lx, ly;
        concentricSampleDisk(s0, s1, lx, ly);
        lx = lensRadius * lx; ly = lensRadius * ly;
        float ft = __builtin_fabsf(focalDistance / direction.z);
        vec3 pFocus = add(origin, scale(direction, ft));
        origin = add(origin, V3(lx, ly, 0.f));
        direction = normalize(sub(pFocus, origin));
    }
    ps.o = origin;
    ps.d = direction;
    ps.color = V3(1.0f, 1.0f, 1.0f);
}

}  // namespace ptd
