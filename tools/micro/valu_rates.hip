// Microbenchmark (diagnostic, not part of the library): issue cost of the vector instructions k_bounce is made of, in cycles
// per wave-instruction per SIMD, at 1, 2 and 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define N_INNER 64          // instructions per loop trip = 8 * 8
template <int OP>
__global__ __launch_bounds__(256) void k(int trips, float *out, unsigned long long *cyc) {
    float a0 = threadIdx.x * 1e-3f + 1.f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    float b = 1.0001f, c = 1e-7f;
    double db = 1.0001, dc = 1e-7;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6}, pb = {b, b}, pc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < trips; i++) {
#define F8(ins) asm volatile(ins " %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); asm volatile(ins " %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); \
    asm volatile(ins " %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); asm volatile(ins " %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c)); \
    asm volatile(ins " %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c)); asm volatile(ins " %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c)); \
    asm volatile(ins " %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c)); asm volatile(ins " %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c));
#define G8(ins) asm volatile(ins " %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a1) : "v"(b)); \
    asm volatile(ins " %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a3) : "v"(b)); \
    asm volatile(ins " %0, %0, %1" : "+v"(a4) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a5) : "v"(b)); \
    asm volatile(ins " %0, %0, %1" : "+v"(a6) : "v"(b)); asm volatile(ins " %0, %0, %1" : "+v"(a7) : "v"(b));
#define U8(ins) asm volatile(ins " %0, %0" : "+v"(a0)); asm volatile(ins " %0, %0" : "+v"(a1)); asm volatile(ins " %0, %0" : "+v"(a2)); \
    asm volatile(ins " %0, %0" : "+v"(a3)); asm volatile(ins " %0, %0" : "+v"(a4)); asm volatile(ins " %0, %0" : "+v"(a5)); \
    asm volatile(ins " %0, %0" : "+v"(a6)); asm volatile(ins " %0, %0" : "+v"(a7));
#define D8(ins) asm volatile(ins " %0, %0, %1, %2" : "+v"(d0) : "v"(db), "v"(dc)); asm volatile(ins " %0, %0, %1, %2" : "+v"(d1) : "v"(db), "v"(dc)); \
    asm volatile(ins " %0, %0, %1, %2" : "+v"(d2) : "v"(db), "v"(dc)); asm volatile(ins " %0, %0, %1, %2" : "+v"(d3) : "v"(db), "v"(dc)); \
    asm volatile(ins " %0, %0, %1, %2" : "+v"(d4) : "v"(db), "v"(dc)); asm volatile(ins " %0, %0, %1, %2" : "+v"(d5) : "v"(db), "v"(dc)); \
    asm volatile(ins " %0, %0, %1, %2" : "+v"(d6) : "v"(db), "v"(dc)); asm volatile(ins " %0, %0, %1, %2" : "+v"(d7) : "v"(db), "v"(dc));
#define E8(ins) asm volatile(ins " %0, %0, %1" : "+v"(d0) : "v"(db)); asm volatile(ins " %0, %0, %1" : "+v"(d1) : "v"(db)); \
    asm volatile(ins " %0, %0, %1" : "+v"(d2) : "v"(db)); asm volatile(ins " %0, %0, %1" : "+v"(d3) : "v"(db)); \
    asm volatile(ins " %0, %0, %1" : "+v"(d4) : "v"(db)); asm volatile(ins " %0, %0, %1" : "+v"(d5) : "v"(db)); \
    asm volatile(ins " %0, %0, %1" : "+v"(d6) : "v"(db)); asm volatile(ins " %0, %0, %1" : "+v"(d7) : "v"(db));
#define P8(ins) asm volatile(ins " %0, %0, %1, %2" : "+v"(p0) : "v"(pb), "v"(pc)); asm volatile(ins " %0, %0, %1, %2" : "+v"(p1) : "v"(pb), "v"(pc)); \
    asm volatile(ins " %0, %0, %1, %2" : "+v"(p2) : "v"(pb), "v"(pc)); asm volatile(ins " %0, %0, %1, %2" : "+v"(p3) : "v"(pb), "v"(pc)); \
    asm volatile(ins " %0, %0, %1, %2" : "+v"(p4) : "v"(pb), "v"(pc)); asm volatile(ins " %0, %0, %1, %2" : "+v"(p5) : "v"(pb), "v"(pc)); \
    asm volatile(ins " %0, %0, %1, %2" : "+v"(p6) : "v"(pb), "v"(pc)); asm volatile(ins " %0, %0, %1, %2" : "+v"(p7) : "v"(pb), "v"(pc));
#define Q8(ins) asm volatile(ins " %0, %0, %1" : "+v"(p0) : "v"(pb)); asm volatile(ins " %0, %0, %1" : "+v"(p1) : "v"(pb)); \
    asm volatile(ins " %0, %0, %1" : "+v"(p2) : "v"(pb)); asm volatile(ins " %0, %0, %1" : "+v"(p3) : "v"(pb)); \
    asm volatile(ins " %0, %0, %1" : "+v"(p4) : "v"(pb)); asm volatile(ins " %0, %0, %1" : "+v"(p5) : "v"(pb)); \
    asm volatile(ins " %0, %0, %1" : "+v"(p6) : "v"(pb)); asm volatile(ins " %0, %0, %1" : "+v"(p7) : "v"(pb));
        if (OP == 0) { REP8(F8("v_fma_f32")) }
        if (OP == 1) { REP8(G8("v_mul_f32")) }
        if (OP == 2) { REP8(G8("v_add_f32")) }
        if (OP == 3) { REP8(U8("v_rcp_f32")) }
        if (OP == 4) { REP8(U8("v_sqrt_f32")) }
        if (OP == 5) { REP8(D8("v_fma_f64")) }
        if (OP == 6) { REP8(E8("v_mul_f64")) }
        if (OP == 7) { REP8(E8("v_add_f64")) }
        if (OP == 8) { REP8(P8("v_pk_fma_f32")) }
        if (OP == 9) { REP8(Q8("v_pk_mul_f32")) }
        if (OP == 10) { REP8(Q8("v_pk_add_f32")) }
        if (OP == 11) { REP8(F8("v_div_fixup_f32")) }
        if (OP == 12) { REP8(G8("v_mul_lo_u32")) }
        if (OP == 13) { REP8(G8("v_mul_hi_u32")) }
        if (OP == 14) { REP8(G8("v_max_f32")) }
        if (OP == 15) { REP8(G8("v_and_b32")) }
        if (OP == 16) { REP8(F8("v_fma_f32") G8("v_mul_f32")) }    // 128 instrs per trip: mixed
        if (OP == 17) { REP8(U8("v_mov_b32")) }
        if (OP == 18) { REP8(G8("v_min_f32")) }
        if (OP == 19) { REP8(F8("v_max3_f32")) }
        if (OP == 20) { REP8(F8("v_cndmask_b32 %0, %0, %1, vcc ; ")) }
        if (OP == 21) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a0), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a1), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a2), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a3), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a4), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a5), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a6), "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a7), "v"(b) : "vcc");) }
        if (OP == 22) { REP8(F8("v_div_fmas_f32")) }
        if (OP == 23) { REP8(G8("v_add_u32")) }
        if (OP == 24) { REP8(F8("v_mad_u32_u24")) }
        if (OP == 25) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d0) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d1) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d2) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d3) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d4) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d5) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d6) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d7) : "v"(b), "v"(c) : "vcc");) }
        if (OP == 26) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(d0) : "v"(db)); asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(d1) : "v"(db)); asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(d2) : "v"(db)); asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(d3) : "v"(db)); asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(d4) : "v"(db)); asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(d5) : "v"(db)); asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(d6) : "v"(db)); asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(d7) : "v"(db));) }
        if (OP == 27) { REP8(asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(a0) : "s20"); asm volatile("v_readlane_b32 s21, %0, 3" :: "v"(a1) : "s21"); asm volatile("v_readlane_b32 s22, %0, 3" :: "v"(a2) : "s22"); asm volatile("v_readlane_b32 s23, %0, 3" :: "v"(a3) : "s23"); asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(a4) : "s20"); asm volatile("v_readlane_b32 s21, %0, 3" :: "v"(a5) : "s21"); asm volatile("v_readlane_b32 s22, %0, 3" :: "v"(a6) : "s22"); asm volatile("v_readlane_b32 s23, %0, 3" :: "v"(a7) : "s23");) }
        if (OP == 28) { REP8(asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc"); asm volatile("s_add_u32 s21, s21, 1" ::: "s21", "scc"); asm volatile("s_add_u32 s22, s22, 1" ::: "s22", "scc"); asm volatile("s_add_u32 s23, s23, 1" ::: "s23", "scc"); asm volatile("s_and_b64 s[24:25], s[24:25], exec" ::: "s24", "s25", "scc"); asm volatile("s_or_b64 s[26:27], s[26:27], exec" ::: "s26", "s27", "scc"); asm volatile("s_mov_b32 s28, s20" ::: "s28"); asm volatile("s_nop 0");) }
        if (OP == 30) { REP8(F8("v_cndmask_b32_e64 %0, %0, %1, s[20:21] ; ")) }
        if (OP == 31) { REP8(F8("v_addc_co_u32_e64 %0, vcc, %0, %1, s[20:21] ; ")) }
        if (OP == 32) { REP8(F8("v_bfi_b32")) }
        if (OP == 33) { REP8(asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b), "v"(c) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b), "v"(c) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b), "v"(c) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b), "v"(c) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a4) : "v"(b), "v"(c) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a5) : "v"(b), "v"(c) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a6) : "v"(b), "v"(c) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a7) : "v"(b), "v"(c) : "vcc");) }
        if (OP == 34) { REP8(G8("v_xor_b32")) }
        if (OP == 35) { REP8(G8("v_lshlrev_b32")) }
        if (OP == 36) { REP8(F8("v_cmp_lt_f32_e64 s[22:23], %0, %1 ; ")) }
        if (OP == 37) { REP8(asm volatile("v_cmp_lt_f32_e64 s[22:23], %1, %2\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a0) : "v"(b), "v"(c) : "s22", "s23"); asm volatile("v_cmp_lt_f32_e64 s[24:25], %1, %2\n v_cndmask_b32_e64 %0, %0, %1, s[24:25]" : "+v"(a1) : "v"(b), "v"(c) : "s24", "s25"); asm volatile("v_cmp_lt_f32_e64 s[22:23], %1, %2\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a2) : "v"(b), "v"(c) : "s22", "s23"); asm volatile("v_cmp_lt_f32_e64 s[24:25], %1, %2\n v_cndmask_b32_e64 %0, %0, %1, s[24:25]" : "+v"(a3) : "v"(b), "v"(c) : "s24", "s25"); asm volatile("v_cmp_lt_f32_e64 s[22:23], %1, %2\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a4) : "v"(b), "v"(c) : "s22", "s23"); asm volatile("v_cmp_lt_f32_e64 s[24:25], %1, %2\n v_cndmask_b32_e64 %0, %0, %1, s[24:25]" : "+v"(a5) : "v"(b), "v"(c) : "s24", "s25"); asm volatile("v_cmp_lt_f32_e64 s[22:23], %1, %2\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a6) : "v"(b), "v"(c) : "s22", "s23"); asm volatile("v_cmp_lt_f32_e64 s[24:25], %1, %2\n v_cndmask_b32_e64 %0, %0, %1, s[24:25]" : "+v"(a7) : "v"(b), "v"(c) : "s24", "s25");) }
        if (OP == 38 || OP == 39) {      // k_bounce-like mix per 16 vector instructions: 8 plain (fma/mul/add f32, add_u32, and), 7 "second class" (max, cndmask, lshlrev, cmp, min3, mul_lo, f64 add), 1 rcp; OP 39 adds 8 scalar instructions
#define MIXV asm volatile("v_fma_f32 %0, %0, %8, %9\n v_max_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_add_f32 %4, %4, %9\n v_lshlrev_b32 %5, 1, %5\n v_add_u32 %6, %6, %7\n v_cmp_lt_f32_e64 s[22:23], %0, %8\n" \
                     "v_fma_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_and_b32 %3, %3, %5\n v_mul_lo_u32 %6, %6, %7\n v_mul_f32 %4, %4, %8\n v_add_f64 %10, %10, %11\n v_add_f32 %0, %0, %9\n v_rcp_f32 %7, %7" \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(d0), "v"(db) : "s22", "s23");
#define MIXS asm volatile("s_add_u32 s24, s24, 1\n s_and_b64 s[26:27], s[26:27], exec\n s_mov_b32 s28, s24\n s_lshl_b32 s29, s24, 2\n s_add_u32 s30, s30, s29\n s_cmp_lt_u32 s24, s30\n s_cselect_b32 s31, s24, s30\n s_or_b64 s[26:27], s[26:27], exec" ::: "s24", "s26", "s27", "s28", "s29", "s30", "s31", "scc");
            if (OP == 38) { MIXV MIXV MIXV MIXV }
            else { MIXV MIXS MIXV MIXS MIXV MIXS MIXV MIXS }
        }
        if (OP == 29) { REP8(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d0) : "v"(a0)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a1) : "v"(d1)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d2) : "v"(a2)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a3) : "v"(d3)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d4) : "v"(a4)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a5) : "v"(d5)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d6) : "v"(a6)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a7) : "v"(d7));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
}

template <int OP> double run(int wavesPerSimd, int trips, int per) {
    int ncu = 256;
    int blocks = ncu * wavesPerSimd;          // 256 threads = 4 waves = one per SIMD
    float *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&cyc, 8 * blocks * 4);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, trips, out, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, trips, out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, 8 * blocks * 4, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    s /= h.size();
    hipFree(out); hipFree(cyc);
    // s_memtime ticks at 100 MHz constant? (it is the shader clock on gfx9) -> report ticks per instruction per SIMD
    return s / ((double)trips * per) / wavesPerSimd;
}

int main() {
    const char *names[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_rcp_f32", "v_sqrt_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_pk_fma_f32", "v_pk_mul_f32",
                           "v_pk_add_f32", "v_div_fixup_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_max_f32", "v_and_b32", "fma+mul mix", "v_mov_b32", "v_min_f32", "v_max3_f32", "v_cndmask_b32", "v_cmp_lt_f32", "v_div_fmas_f32", "v_add_u32", "v_mad_u32_u24", "v_mad_u64_u32",
                           "v_lshl_add_u64", "v_readlane_b32", "SALU mix (8)", "cvt f64<->f32", "v_cndmask_e64 sgpr", "v_addc_co_u32 sgpr", "v_bfi_b32", "cmp+cndmask vcc (2)", "v_xor_b32", "v_lshlrev_b32", "v_cmp_e64 -> sgpr", "cmp+cndmask sgpr (2)", "mix: 64 vector", "mix: 64 vector + 32 scalar"};
    printf("%-18s %8s %8s %8s   (s_memtime ticks per wave-instruction per SIMD)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
#define ROW(OP, per) printf("%-18s %8.2f %8.2f %8.2f\n", names[OP], run<OP>(1, 2000, per), run<OP>(2, 2000, per), run<OP>(4, 2000, per));
    ROW(0, 64) ROW(1, 64) ROW(2, 64) ROW(3, 64) ROW(4, 64) ROW(5, 64) ROW(6, 64) ROW(7, 64) ROW(8, 64) ROW(9, 64) ROW(10, 64) ROW(11, 64) ROW(12, 64) ROW(13, 64)
    ROW(14, 64) ROW(15, 64) ROW(16, 128) ROW(17, 64) ROW(18, 64) ROW(19, 64) ROW(20, 64) ROW(21, 64) ROW(22, 64) ROW(23, 64) ROW(24, 64) ROW(25, 64) ROW(26, 64) ROW(27, 64) ROW(28, 64) ROW(29, 64) ROW(30, 64) ROW(31, 64) ROW(32, 64) ROW(33, 128) ROW(34, 64) ROW(35, 64) ROW(36, 64) ROW(37, 128) ROW(38, 64) ROW(39, 64)
    printf("%-18s %8.2f %8.2f   (6 and 8 waves per SIMD, ticks per VECTOR instruction)\n", "mix: 64 vector", run<38>(6, 2000, 64), run<38>(8, 2000, 64));
    printf("%-18s %8.2f %8.2f\n", "mix + scalar", run<39>(6, 2000, 64), run<39>(8, 2000, 64));
    printf("%-18s %8.2f %8.2f\n", "v_fma_f32", run<0>(6, 2000, 64), run<0>(8, 2000, 64));
    printf("%-18s %8.2f %8.2f\n", "v_max_f32", run<14>(6, 2000, 64), run<14>(8, 2000, 64));
    // the same by the wall clock (events around one launch): wave-instructions per SIMD and cycle at 2.38 GHz, whatever the waves' own clocks say
    for (int w : {1, 2, 4, 8}) {
        int blocks = 256 * w; float *out; unsigned long long *cyc;
        hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&cyc, 8 * blocks * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms[3];
        hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, 20000, out, cyc); hipDeviceSynchronize();
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, 100000, out, cyc); hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms[0], e0, e1);
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k<14>, dim3(blocks), dim3(256), 0, 0, 100000, out, cyc); hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms[1], e0, e1);
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k<38>, dim3(blocks), dim3(256), 0, 0, 100000, out, cyc); hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms[2], e0, e1);
        const double instr_per_simd = (double)w * 100000.0 * 64.0;
        printf("wall clock, %d waves per SIMD: v_fma_f32 %.3f ms = %.2f cycles per instruction per SIMD; v_max_f32 %.3f ms = %.2f; mix %.3f ms = %.2f\n", w,
               ms[0], ms[0] * 1e-3 * 2.38e9 / instr_per_simd, ms[1], ms[1] * 1e-3 * 2.38e9 / instr_per_simd, ms[2], ms[2] * 1e-3 * 2.38e9 / instr_per_simd);
        hipFree(out); hipFree(cyc);
    }
    {   // tick calibration: wall time of a long v_fma run against its ticks
        int blocks = 256; float *out; unsigned long long *cyc;
        hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&cyc, 8 * blocks * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, 20000, out, cyc); hipDeviceSynchronize();
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, 200000, out, cyc); hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4); hipMemcpy(h.data(), cyc, 8 * blocks * 4, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v; s /= h.size();
        printf("calibration: %.0f s_memtime ticks in %.3f ms of kernel time = %.1f MHz tick rate; v_fma_f32 one wave per SIMD = %.2f ns each\n", s, ms, s / ms / 1e3, ms * 1e6 / (200000.0 * 64));
    }
    return 0;
}
