/* sincos_host.c -- the product's sincos_fast (mygpuraytracer_amd/csrc/pt_sincos_fast.h) compiled for the host as a small shared
 * library: the GPU test compares the device's results with it value by value (tests/test_gpu_parity.py). */
#include "../mygpuraytracer_amd/csrc/pt_sincos_fast.h"

void host_sincos_fast(int n, const float *x, float *s, float *c, int *ok) {
    for (int i = 0; i < n; i++) {
        float sn = 0.f, cs = 0.f;
        ok[i] = sincos_fast(x[i], &sn, &cs, 0);
        s[i] = sn; c[i] = cs;
    }
}
