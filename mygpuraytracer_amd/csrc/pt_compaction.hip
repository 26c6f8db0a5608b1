// pt_compaction.hip -- scan / stream compaction on int arrays behind include/mi355x_stream_compaction.h.
//
// Replaces the reference's stream_compaction/{cpu,naive,efficient,thrust,common}.cu.  The reference's GPU scans
// issue one launch per tree level (2*log2(n)+1 launches for the Blelloch version, efficient.cu:46-61); here a
// scan is three launches regardless of n, written for 64-wide wavefronts:
//   k_block_scan : each 256-thread workgroup scans 2048 elements (8 per lane as two 16-byte loads, wave scan by
//                  __shfl_up, 4 wave totals through LDS) and emits its total;
//   k_sums_scan  : one workgroup scans the block totals (running carry, any count);
//   k_add_offsets: adds each block's offset.
// Compaction fuses kernMapToBoolean (common.cu:25-34) into the first pass and ends with kernScatter (:40-49).
// HBM traffic: scan reads n and writes n ints twice (8+8 B/elem); compaction adds one read and <= one write.
#include <hip/hip_runtime.h>
#include <chrono>
#include <string>
#include <string.h>

#include "../../include/mi355x_pathtracer.h"
#include "../../include/mi355x_stream_compaction.h"

extern "C" void ptx_internal_set_error(const char *msg);

namespace {

constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 8;
constexpr int SC_BLOCK = SC_THREADS * SC_ITEMS;     // 2048 elements per workgroup

thread_local float g_gpu_ms = 0.f, g_cpu_ms = 0.f;

int sc_fail(const std::string &m) { ptx_internal_set_error(m.c_str()); return PTX_ERR_HIP; }
#define SC_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return sc_fail(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}

// exclusive scan inside each 2048-element block; MAP: scan (x != 0) instead of x
template <bool MAP>
__global__ __launch_bounds__(SC_THREADS) void k_block_scan(int n, const int *__restrict__ in, int *__restrict__ out,
                                                            int *__restrict__ block_sums) {
    __shared__ int wave_tot[SC_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * SC_BLOCK + (long long)tid * SC_ITEMS;
    int v[SC_ITEMS];
    if (base + SC_ITEMS <= n && ((((uintptr_t)(in + base)) & 15) == 0)) {
        const int4 a = *reinterpret_cast<const int4 *>(in + base);
        const int4 b = *reinterpret_cast<const int4 *>(in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < SC_ITEMS; k++) v[k] = (base + k < n) ? in[base + k] : 0;
    }
    if (MAP) {
#pragma unroll
        for (int k = 0; k < SC_ITEMS; k++) v[k] = v[k] != 0 ? 1 : 0;
    }
    int sum = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) sum += v[k];
    const int incl = wave_inclusive_scan(sum, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; w++) wave_off += wave_tot[w];
    int run = wave_off + incl - sum;
    int o[SC_ITEMS];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) { o[k] = run; run += v[k]; }
    if (base + SC_ITEMS <= n && ((((uintptr_t)(out + base)) & 15) == 0)) {
        *reinterpret_cast<int4 *>(out + base) = make_int4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<int4 *>(out + base + 4) = make_int4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int k = 0; k < SC_ITEMS; k++) if (base + k < n) out[base + k] = o[k];
    }
    if (tid == SC_THREADS - 1) block_sums[blockIdx.x] = run;
}

// exclusive scan of the block totals, in place, by one workgroup with a running carry
__global__ __launch_bounds__(1024) void k_sums_scan(int nblocks, int *__restrict__ sums) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + tid;
        const int x = i < nblocks ? sums[i] : 0;
        const int incl = wave_inclusive_scan(x, lane);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int off = carry_s;
        for (int w = 0; w < wave; w++) off += wave_tot[w];
        if (i < nblocks) sums[i] = off + incl - x;
        __syncthreads();
        if (tid == 1023) carry_s = off + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SC_THREADS) void k_add_offsets(int n, int *__restrict__ out, const int *__restrict__ sums) {
    const int off = sums[blockIdx.x];
    const long long base = (long long)blockIdx.x * SC_BLOCK + (long long)threadIdx.x * SC_ITEMS;
    if (off == 0) return;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) if (base + k < n) out[base + k] += off;
}

// kernScatter (common.cu:40-49) with the boolean recomputed from the data; also publishes the count
__global__ __launch_bounds__(SC_THREADS) void k_scatter(int n, int *__restrict__ out, const int *__restrict__ in,
                                                         const int *__restrict__ indices, int *__restrict__ count) {
    const int i = blockIdx.x * SC_THREADS + threadIdx.x;
    if (i < n) {
        const int x = in[i];
        if (x != 0) out[indices[i]] = x;
        if (i == n - 1) *count = indices[i] + (x != 0 ? 1 : 0);
    }
}

int scan_device(int n, int *d_out, const int *d_in, int *d_sums, hipStream_t st, bool map) {
    const int nblocks = (n + SC_BLOCK - 1) / SC_BLOCK;
    if (map) hipLaunchKernelGGL(k_block_scan<true>, dim3(nblocks), dim3(SC_THREADS), 0, st, n, d_in, d_out, d_sums);
    else hipLaunchKernelGGL(k_block_scan<false>, dim3(nblocks), dim3(SC_THREADS), 0, st, n, d_in, d_out, d_sums);
    if (nblocks > 1) {
        hipLaunchKernelGGL(k_sums_scan, dim3(1), dim3(1024), 0, st, nblocks, d_sums);
        hipLaunchKernelGGL(k_add_offsets, dim3(nblocks), dim3(SC_THREADS), 0, st, n, d_out, d_sums);
    }
    SC_CHECK(hipGetLastError());
    return PTX_OK;
}

// host-pointer scan shared by the three reference entry points (naive.cu:32, efficient.cu:35, thrust.cu:20)
int host_scan(int n, int *odata, const int *idata) {
    if (n <= 0) return PTX_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        ptx_internal_set_error("no HIP device available; the GPU scan has no CPU path (use sc_cpu_scan)");
        return PTX_ERR_NODEVICE;
    }
    int *d_in = nullptr, *d_out = nullptr, *d_ws = nullptr;
    hipEvent_t e0, e1;
    SC_CHECK(hipMalloc(&d_in, sizeof(int) * (size_t)n));
    SC_CHECK(hipMalloc(&d_out, sizeof(int) * (size_t)n));
    SC_CHECK(hipMalloc(&d_ws, sc_scan_workspace_bytes(n)));
    SC_CHECK(hipMemcpy(d_in, idata, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    SC_CHECK(hipEventCreate(&e0)); SC_CHECK(hipEventCreate(&e1));
    SC_CHECK(hipEventRecord(e0, 0));
    int rc = scan_device(n, d_out, d_in, d_ws, 0, false);
    SC_CHECK(hipEventRecord(e1, 0));
    SC_CHECK(hipEventSynchronize(e1));
    SC_CHECK(hipEventElapsedTime(&g_gpu_ms, e0, e1));
    if (rc == PTX_OK) SC_CHECK(hipMemcpy(odata, d_out, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(d_in); hipFree(d_out); hipFree(d_ws);
    return rc;
}

struct CpuTimer {
    std::chrono::high_resolution_clock::time_point t0 = std::chrono::high_resolution_clock::now();
    ~CpuTimer() { g_cpu_ms = std::chrono::duration<float, std::milli>(std::chrono::high_resolution_clock::now() - t0).count(); }
};

}  // namespace

extern "C" {

// ---- StreamCompaction::CPU (host code by definition: these are the reference's CPU entry points) ---------------
// cpu.cu:20-32
void sc_cpu_scan(int n, int *odata, const int *idata) {
    CpuTimer tm;
    if (n <= 0) return;
    odata[0] = idata[0];
    for (int i = 1; i < n; i++) odata[i] = odata[i - 1] + idata[i];
    for (int i = 0; i < n; i++) odata[i] -= idata[i];
}

// cpu.cu:39-51
int sc_cpu_compact_without_scan(int n, int *odata, const int *idata) {
    CpuTimer tm;
    int num = 0;
    for (int i = 0; i < n; i++) if (idata[i] != 0) odata[num++] = idata[i];
    return num;
}

// cpu.cu:58-95: map, scan, scatter
int sc_cpu_compact_with_scan(int n, int *odata, const int *idata) {
    if (n <= 0) return 0;
    int *flags = new int[n];
    int *pos = new int[n];
    int num = 0;
    {
        CpuTimer tm;
        for (int i = 0; i < n; i++) flags[i] = idata[i] == 0 ? 0 : 1;
        pos[0] = flags[0];
        for (int i = 1; i < n; i++) pos[i] = pos[i - 1] + flags[i];
        for (int i = 0; i < n; i++) pos[i] -= flags[i];
        for (int i = 0; i < n; i++) if (flags[i] == 1) { odata[pos[i]] = idata[i]; num++; }
    }
    delete[] flags;
    delete[] pos;
    return num;
}

int sc_naive_scan(int n, int *odata, const int *idata) { return host_scan(n, odata, idata); }
int sc_efficient_scan(int n, int *odata, const int *idata) { return host_scan(n, odata, idata); }
int sc_thrust_scan(int n, int *odata, const int *idata) { return host_scan(n, odata, idata); }

// efficient.cu:79-136
int sc_efficient_compact(int n, int *odata, const int *idata) {
    if (n <= 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        ptx_internal_set_error("no HIP device available; the GPU compaction has no CPU path (use sc_cpu_compact_*)");
        return -1;
    }
    int *d_in = nullptr, *d_out = nullptr, *d_ws = nullptr, *d_count = nullptr;
    hipEvent_t e0, e1;
#define SCC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { sc_fail(std::string(#expr) + ": " + hipGetErrorString(e_)); return -1; } } while (0)
    SCC(hipMalloc(&d_in, sizeof(int) * (size_t)n));
    SCC(hipMalloc(&d_out, sizeof(int) * (size_t)n));
    SCC(hipMalloc(&d_count, sizeof(int)));
    SCC(hipMalloc(&d_ws, sc_scan_workspace_bytes(n)));
    SCC(hipMemcpy(d_in, idata, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    SCC(hipEventCreate(&e0)); SCC(hipEventCreate(&e1));
    SCC(hipEventRecord(e0, 0));
    int rc = sc_compact_device(n, d_out, d_in, d_count, d_ws, nullptr);
    SCC(hipEventRecord(e1, 0));
    SCC(hipEventSynchronize(e1));
    SCC(hipEventElapsedTime(&g_gpu_ms, e0, e1));
    int count = -1;
    if (rc == PTX_OK) {
        SCC(hipMemcpy(&count, d_count, sizeof(int), hipMemcpyDeviceToHost));
        if (count > 0) SCC(hipMemcpy(odata, d_out, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost));
    }
#undef SCC
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(d_in); hipFree(d_out); hipFree(d_ws); hipFree(d_count);
    return count;
}

unsigned long long sc_scan_workspace_bytes(int n) {
    // block totals + (for compaction) the index array
    const unsigned long long nblocks = n > 0 ? ((unsigned long long)n + SC_BLOCK - 1) / SC_BLOCK : 1;
    const unsigned long long head = (nblocks + 16 + 3) & ~3ull;          // keeps the index array 16-byte aligned
    return sizeof(int) * (head + (unsigned long long)(n > 0 ? n : 0));
}

int sc_scan_device(int n, int *d_odata, const int *d_idata, void *d_workspace, void *stream) {
    if (n <= 0) return PTX_OK;
    if (!d_odata || !d_idata || !d_workspace) { ptx_internal_set_error("null device pointer"); return PTX_ERR_INVALID; }
    return scan_device(n, d_odata, d_idata, (int *)d_workspace, (hipStream_t)stream, false);
}

int sc_compact_device(int n, int *d_odata, const int *d_idata, int *d_count, void *d_workspace, void *stream) {
    if (!d_count) { ptx_internal_set_error("null device pointer"); return PTX_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    if (n <= 0) { SC_CHECK(hipMemsetAsync(d_count, 0, sizeof(int), st)); return PTX_OK; }
    if (!d_odata || !d_idata || !d_workspace) { ptx_internal_set_error("null device pointer"); return PTX_ERR_INVALID; }
    const int nblocks = (n + SC_BLOCK - 1) / SC_BLOCK;
    int *d_sums = (int *)d_workspace;
    int *d_indices = d_sums + ((nblocks + 16 + 3) & ~3);
    int rc = scan_device(n, d_indices, d_idata, d_sums, st, true);
    if (rc != PTX_OK) return rc;
    hipLaunchKernelGGL(k_scatter, dim3((n + SC_THREADS - 1) / SC_THREADS), dim3(SC_THREADS), 0, st, n, d_odata, d_idata, d_indices, d_count);
    SC_CHECK(hipGetLastError());
    return PTX_OK;
}

float sc_last_gpu_ms(void) { return g_gpu_ms; }
float sc_last_cpu_ms(void) { return g_cpu_ms; }

// common.h:21-31
int sc_ilog2(int x) { int lg = 0; while (x >>= 1) ++lg; return lg; }
int sc_ilog2ceil(int x) { return x == 1 ? 0 : sc_ilog2(x - 1) + 1; }

}  // extern "C"
