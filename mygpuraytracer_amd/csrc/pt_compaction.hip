// pt_compaction.hip -- scan / stream compaction on int arrays behind include/mi355x_stream_compaction.h.
//
// Replaces the reference's stream_compaction/{cpu,naive,efficient,thrust,common}.cu.  The reference's GPU scans
// issue one launch per tree level (2*log2(n)+1 launches for the Blelloch version, efficient.cu:46-61) and its
// compaction is map + scan + scatter over three int arrays (efficient.cu:79-136).  Here either is ONE pass over the
// data -- a chained scan with decoupled look-back, written for 64-wide wavefronts:
//   * a workgroup (512 threads for the scan, 1024 for compaction) takes the next 16384-element tile (ticket from an
//     atomic counter, so a tile's predecessors are always resident or done), loaded as coalesced non-temporal
//     16-byte loads, 8 or 4 per lane;
//   * scan inside the tile: 4 per lane per load in registers, wave scan in DPP (round 5; __shfl_up before), the 64 (load, wave) totals
//     scanned by wave 0 through LDS;
//   * the tile publishes {flag, total} as one 8-byte word (relaxed agent-scope store: one granule, no fence needed,
//     visible across the XCDs' L2s) and its wave 0 looks back over its predecessors' words, 64 per step, adding
//     tile totals until it meets one that already knows its inclusive prefix, then publishes its own;
//   * scan: the prefixes are stored; compaction (kernMapToBoolean fused, common.cu:25-34): the survivors of the
//     tile are packed in LDS and stored as one contiguous run (kernScatter, common.cu:40-49, without the index array).
// HBM traffic is the algorithmic minimum: scan 4 B read + 4 B written per element, compaction 4 B read + 4 B per
// survivor, plus 8 B per tile of status words (zeroed by a memset node in front of the kernel).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <string>
#include <string.h>

#include "../../include/mi355x_pathtracer.h"
#include "../../include/mi355x_stream_compaction.h"

extern "C" void ptx_internal_set_error(const char *msg);

namespace {

constexpr int SC_TILE = 16384;                                 // elements (64 KB) per workgroup: one ticket each, and one
                                                               // address takes only ~80 M atomics/s across the XCDs
// 64 (round, wave) totals per tile either way (one lane each in the second-level scan); measured on 2^28 ints:
// scan 1024 x 4: 4.39 TB/s, 512 x 8: 4.57 TB/s; compaction 1024 x 4: 3.75 TB/s, 512 x 8: 3.60 TB/s (copy: 4.9-5.0);
// at least 4 waves per SIMD (<= 128 registers): asking for 8 spills and is 5 % slower
constexpr int SC_SCAN_THREADS = 512, SC_COMPACT_THREADS = 1024;
constexpr int SC_HEAD = 64;                                    // workspace: [ticket, padding to 64 B][status word per tile]

typedef unsigned long long u64;
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr u64 ST_AGGREGATE = 1ull << 32;                       // low word = this tile's total
constexpr u64 ST_INCLUSIVE = 2ull << 32;                       // low word = total of this tile and everything before it

thread_local float g_gpu_ms = 0.f, g_cpu_ms = 0.f;

int sc_fail(const std::string &m) { ptx_internal_set_error(m.c_str()); return PTX_ERR_HIP; }
#define SC_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return sc_fail(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// inclusive prefix sum over the lanes of a wave in the vector ALU's own lane network (DPP), as pt_engine.hip's waveInclusiveScan: four
// shifted adds inside the rows of 16 lanes, lane 15 of a row broadcast into the next row (rows 1 and 3), lane 31 into the upper half.
// Six dependent vector instructions where __shfl_up made six ds_bpermute round trips through the LDS crossbar (rounds 1-4) -- at frame
// sizes (2 M elements) the kernel is a chain of such latencies, not a stream.
__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1 (lanes without a source add 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return v;
}

// sum over the wave, in every lane: the scan's last lane through the scalar path (v_readlane), no LDS round trip
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_inclusive_scan(v, 0), 63); }

// kernMapToBoolean (common.cu:25-34): bools[i] = idata[i] != 0.  The first n4 quads go as 16-byte accesses (both arrays 16-byte
// aligned), the rest one by one; HBM-bound, 8 B per element.
__global__ __launch_bounds__(256) void k_map_to_boolean(int n, int n4, int *__restrict__ bools, const int *__restrict__ idata) {
    const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (int q = t0; q < n4; q += stride) {
        const v4i a = reinterpret_cast<const v4i *>(idata)[q];
        v4i b;
        b.x = a.x != 0; b.y = a.y != 0; b.z = a.z != 0; b.w = a.w != 0;
        reinterpret_cast<v4i *>(bools)[q] = b;
    }
    for (int i = 4 * n4 + t0; i < n; i += stride) bools[i] = idata[i] != 0 ? 1 : 0;
}
// kernScatter (common.cu:40-49): bools[i] == 1 => odata[indices[i]] = idata[i]
__global__ __launch_bounds__(256) void k_scatter(int n, int *__restrict__ odata, const int *__restrict__ idata, const int *__restrict__ bools,
                                                  const int *__restrict__ indices) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (bools[i] == 1) odata[indices[i]] = idata[i];
}

// COMPACT = false: out[i] = in[0] + .. + in[i-1].  COMPACT = true: out = the non-zero elements of in, in order;
// *count = how many.  A tile waits only for tiles with lower tickets, which are resident or done and post their totals
// before they wait for anything themselves: every wait ends.
template <bool COMPACT, int SC_THREADS>
__global__ __launch_bounds__(SC_THREADS, 4) void k_onepass(int n, const int *__restrict__ in, int *__restrict__ out,
                                                            unsigned *__restrict__ ticket, u64 *__restrict__ status,
                                                            int *__restrict__ count) {
    constexpr int SC_WAVES = SC_THREADS / 64, SC_ROUNDS = SC_TILE / (SC_THREADS * 4);       // 16-byte loads per lane
    __shared__ int s_tile, s_excl, s_total;
    __shared__ int s_wtot[SC_ROUNDS * SC_WAVES];              // (round, wave) totals, then their exclusive prefixes
    __shared__ int s_stage[COMPACT ? SC_TILE : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = s_tile;
    const long long tbase = (long long)tile * SC_TILE;
    const bool whole = tbase + SC_TILE <= n;

    int x[SC_ROUNDS][4];
    if (whole && (((uintptr_t)in) & 15) == 0) {
#pragma unroll
        for (int r = 0; r < SC_ROUNDS; r++) {
            // streamed once: non-temporal loads and stores (+5 % on 1 GiB)
            const v4i a = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(in + tbase + (r * SC_THREADS + tid) * 4));
            x[r][0] = a.x; x[r][1] = a.y; x[r][2] = a.z; x[r][3] = a.w;
        }
    } else {
#pragma unroll
        for (int r = 0; r < SC_ROUNDS; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const long long i = tbase + (r * SC_THREADS + tid) * 4 + k;
                x[r][k] = i < n ? in[i] : 0;
            }
    }
    // what is summed: the value, or 1 per survivor
    int sum[SC_ROUNDS], incl[SC_ROUNDS];
#pragma unroll
    for (int r = 0; r < SC_ROUNDS; r++) {
        sum[r] = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sum[r] += COMPACT ? (x[r][k] != 0 ? 1 : 0) : x[r][k];
        incl[r] = wave_inclusive_scan(sum[r], lane);
        if (lane == 63) s_wtot[r * SC_WAVES + wave] = incl[r];
    }
    __syncthreads();

    if (wave == 0) {
        // the 64 (round, wave) totals are one wave's worth: scan them, then look back for what precedes the tile
        static_assert(SC_ROUNDS * SC_WAVES == 64, "one lane per (round, wave) total");
        const int t = s_wtot[lane];
        const int ti = wave_inclusive_scan(t, lane);
        s_wtot[lane] = ti - t;
        const int total = __builtin_amdgcn_readlane(ti, 63);
        int excl = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&status[0], ST_INCLUSIVE | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&status[tile], ST_AGGREGATE | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // lane l looks at tile j - l; every predecessor holds a ticket, so its word is only a matter of time
            for (int j = tile - 1;; j -= 64) {
                const int idx = j - lane;
                int first, part;
                for (;;) {
                    const u64 st = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ST_INCLUSIVE;
                    const u64 inclusive = __ballot((st >> 32) == 2);
                    const u64 pending = __ballot((st >> 32) == 0);
                    first = inclusive ? __builtin_ctzll(inclusive) : 64;
                    const u64 needed = first >= 63 ? ~0ull : ((2ull << first) - 1);       // lanes 0 .. first
                    if ((pending & needed) == 0) { part = lane <= first ? (int)(unsigned)st : 0; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                excl += wave_sum(part);
                if (first < 64) break;
            }
            if (lane == 0) __hip_atomic_store(&status[tile], ST_INCLUSIVE | (unsigned)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) { s_excl = excl; s_total = total; }
    }
    __syncthreads();
    const int excl = s_excl;
    int off[SC_ROUNDS];
#pragma unroll
    for (int r = 0; r < SC_ROUNDS; r++) off[r] = s_wtot[r * SC_WAVES + wave] + incl[r] - sum[r];

    if (!COMPACT) {
        if (whole && (((uintptr_t)out) & 15) == 0) {
#pragma unroll
            for (int r = 0; r < SC_ROUNDS; r++) {
                int4 o;
                o.x = excl + off[r]; o.y = o.x + x[r][0]; o.z = o.y + x[r][1]; o.w = o.z + x[r][2];
                __builtin_nontemporal_store(v4i{o.x, o.y, o.z, o.w}, reinterpret_cast<v4i *>(out + tbase + (r * SC_THREADS + tid) * 4));
            }
        } else {
#pragma unroll
            for (int r = 0; r < SC_ROUNDS; r++) {
                int run = excl + off[r];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const long long i = tbase + (r * SC_THREADS + tid) * 4 + k;
                    if (i < n) out[i] = run;
                    run += x[r][k];
                }
            }
        }
    } else {
        const int total = s_total;
#pragma unroll
        for (int r = 0; r < SC_ROUNDS; r++) {
            int rank = off[r];
#pragma unroll
            for (int k = 0; k < 4; k++) if (x[r][k] != 0) s_stage[rank++] = x[r][k];
        }
        __syncthreads();
        for (int i = tid; i < total; i += SC_THREADS) __builtin_nontemporal_store(s_stage[i], out + (long long)excl + i);
        if (tid == 0 && tbase + SC_TILE >= n) *count = excl + total;
    }
}

inline int sc_tiles(int n) { return (int)(((long long)n + SC_TILE - 1) / SC_TILE); }

int onepass_device(int n, int *d_out, const int *d_in, int *d_count, void *d_ws, hipStream_t st, bool compact) {
    const int ntiles = sc_tiles(n);
    unsigned *ticket = (unsigned *)d_ws;
    u64 *status = (u64 *)((char *)d_ws + SC_HEAD);
    SC_CHECK(hipMemsetAsync(d_ws, 0, SC_HEAD + sizeof(u64) * (size_t)ntiles, st));
    if (compact) hipLaunchKernelGGL((k_onepass<true, SC_COMPACT_THREADS>), dim3(ntiles), dim3(SC_COMPACT_THREADS), 0, st, n, d_in, d_out, ticket, status, d_count);
    else hipLaunchKernelGGL((k_onepass<false, SC_SCAN_THREADS>), dim3(ntiles), dim3(SC_SCAN_THREADS), 0, st, n, d_in, d_out, ticket, status, d_count);
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
    int rc = onepass_device(n, d_out, d_in, nullptr, d_ws, 0, false);
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
    // the ticket counter and one status word per tile; the workspace must be 8-byte aligned (any hipMalloc is)
    return SC_HEAD + sizeof(u64) * (unsigned long long)(n > 0 ? sc_tiles(n) : 1);
}

int sc_scan_device(int n, int *d_odata, const int *d_idata, void *d_workspace, void *stream) {
    if (n <= 0) return PTX_OK;
    if (!d_odata || !d_idata || !d_workspace) { ptx_internal_set_error("null device pointer"); return PTX_ERR_INVALID; }
    if (((uintptr_t)d_workspace) & 7) { ptx_internal_set_error("workspace must be 8-byte aligned"); return PTX_ERR_INVALID; }
    return onepass_device(n, d_odata, d_idata, nullptr, d_workspace, (hipStream_t)stream, false);
}

// d_odata must not alias d_idata: a tile's survivors land where an earlier position's tile may still be reading
int sc_compact_device(int n, int *d_odata, const int *d_idata, int *d_count, void *d_workspace, void *stream) {
    if (!d_count) { ptx_internal_set_error("null device pointer"); return PTX_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    if (n <= 0) { SC_CHECK(hipMemsetAsync(d_count, 0, sizeof(int), st)); return PTX_OK; }
    if (!d_odata || !d_idata || !d_workspace) { ptx_internal_set_error("null device pointer"); return PTX_ERR_INVALID; }
    if (((uintptr_t)d_workspace) & 7) { ptx_internal_set_error("workspace must be 8-byte aligned"); return PTX_ERR_INVALID; }
    return onepass_device(n, d_odata, d_idata, d_count, d_workspace, st, true);
}

// StreamCompaction::Common::kernMapToBoolean / kernScatter (stream_compaction/common.cu:25-49) on device arrays.  The library's own
// compaction fuses both into k_onepass and never materialises bools[] or indices[]; these exist for callers that built their own
// pipeline on the reference's two kernels (map -> their scan -> scatter).
int sc_map_to_boolean_device(int n, int *d_bools, const int *d_idata, void *stream) {
    if (n <= 0) return PTX_OK;
    if (!d_bools || !d_idata) { ptx_internal_set_error("null device pointer"); return PTX_ERR_INVALID; }
    const int n4 = (((uintptr_t)d_bools | (uintptr_t)d_idata) & 15) == 0 ? n / 4 : 0;
    const int work = n4 + (n - 4 * n4);
    hipLaunchKernelGGL(k_map_to_boolean, dim3((unsigned)std::min(8192, (work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, n4, d_bools, d_idata);
    SC_CHECK(hipGetLastError());
    return PTX_OK;
}

int sc_scatter_device(int n, int *d_odata, const int *d_idata, const int *d_bools, const int *d_indices, void *stream) {
    if (n <= 0) return PTX_OK;
    if (!d_odata || !d_idata || !d_bools || !d_indices) { ptx_internal_set_error("null device pointer"); return PTX_ERR_INVALID; }
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)std::min(8192, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, d_odata, d_idata, d_bools, d_indices);
    SC_CHECK(hipGetLastError());
    return PTX_OK;
}

float sc_last_gpu_ms(void) { return g_gpu_ms; }
float sc_last_cpu_ms(void) { return g_cpu_ms; }

// common.h:21-31
int sc_ilog2(int x) { int lg = 0; while (x >>= 1) ++lg; return lg; }
int sc_ilog2ceil(int x) { return x == 1 ? 0 : sc_ilog2(x - 1) + 1; }

}  // extern "C"
