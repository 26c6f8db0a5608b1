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
                           "v_pk_add_f32", "v_div_fixup_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_max_f32", "v_and_b32", "fma+mul mix", "v_mov_b32"};
    printf("%-18s %8s %8s %8s   (s_memtime ticks per wave-instruction per SIMD)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
#define ROW(OP, per) printf("%-18s %8.2f %8.2f %8.2f\n", names[OP], run<OP>(1, 2000, per), run<OP>(2, 2000, per), run<OP>(4, 2000, per));
    ROW(0, 64) ROW(1, 64) ROW(2, 64) ROW(3, 64) ROW(4, 64) ROW(5, 64) ROW(6, 64) ROW(7, 64) ROW(8, 64) ROW(9, 64) ROW(10, 64) ROW(11, 64) ROW(12, 64) ROW(13, 64)
    ROW(14, 64) ROW(15, 64) ROW(16, 128) ROW(17, 64)
    return 0;
}
