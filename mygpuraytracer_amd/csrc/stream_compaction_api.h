// stream_compaction_api.h -- C++ veneer with the reference's own names over the C ABI of the scan / compaction library
// (include/mi355x_stream_compaction.h).  Header-only.
//
// A user of the reference's stream_compaction/ library replaces
//     #include <stream_compaction/cpu.h> <stream_compaction/naive.h> <stream_compaction/efficient.h> <stream_compaction/thrust.h>
// by this header and links libmi355x_pathtracer.so; calls such as
//     StreamCompaction::Efficient::scan(n, odata, idata);                                  // efficient.h:9
//     int kept = StreamCompaction::Efficient::compact(n, odata, idata);                    // efficient.h:11
//     float ms = StreamCompaction::Efficient::timer().getGpuElapsedTimeForPreviousOperation();
//     StreamCompaction::CPU::compactWithoutScan(n, odata, idata);                          // cpu.h:11
// compile and mean what they meant: host pointers in and out, exclusive prefix sums, non-zero elements kept in order, one
// timer per namespace holding the time of that namespace's previous operation (common.h:48-132).  A failing GPU call
// prints and exits like checkCUDAError (common.h:17-20, common.cu:3-17); use the C ABI for return codes.
// Common::kernMapToBoolean / kernScatter (common.h:38-41) are __global__ kernels in the reference, launched by its own
// compaction with <<<grid, block>>>; here they are ordinary functions on device pointers that enqueue the whole array on a
// stream (default: the null stream), because this header must also compile in translation units without HIP:
//     StreamCompaction::Common::kernMapToBoolean(n, dev_bools, dev_idata);              // was <<<blocks, 128>>>(n, ...)
// Not carried over: startGpuTimer/endGpuTimer and their CPU twins (the library times its operations itself).
#pragma once
#include <cstdio>
#include <cstdlib>

#include "../../include/mi355x_pathtracer.h"
#include "../../include/mi355x_stream_compaction.h"

inline int ilog2(int x) { return sc_ilog2(x); }              // common.h:21-27
inline int ilog2ceil(int x) { return sc_ilog2ceil(x); }      // common.h:29-31

namespace StreamCompaction {
namespace Common {

class PerformanceTimer {                                     // common.h:48-132, the two getters
public:
    float getCpuElapsedTimeForPreviousOperation() { return cpu_ms_; }
    float getGpuElapsedTimeForPreviousOperation() { return gpu_ms_; }
    PerformanceTimer() = default;
    PerformanceTimer(const PerformanceTimer &) = delete;
    PerformanceTimer &operator=(const PerformanceTimer &) = delete;
    void mi355x_set(float cpu_ms, float gpu_ms) { if (cpu_ms >= 0.f) cpu_ms_ = cpu_ms; if (gpu_ms >= 0.f) gpu_ms_ = gpu_ms; }
private:
    float cpu_ms_ = 0.f, gpu_ms_ = 0.f;
};

inline void mi355x_check(int rc, const char *what) {
    if (rc == 0) return;
    fprintf(stderr, "mi355x stream compaction error (%s): %s\n", what, ptx_last_error());
    exit(EXIT_FAILURE);
}

// common.h:38-41, common.cu:25-49 -- device pointers; the launch configuration is the library's business
inline void kernMapToBoolean(int n, int *bools, const int *idata, void *stream = nullptr) {
    mi355x_check(sc_map_to_boolean_device(n, bools, idata, stream), "Common::kernMapToBoolean");
}
inline void kernScatter(int n, int *odata, const int *idata, const int *bools, const int *indices, void *stream = nullptr) {
    mi355x_check(sc_scatter_device(n, odata, idata, bools, indices, stream), "Common::kernScatter");
}

}  // namespace Common

namespace CPU {                                              // cpu.h:5-15
inline Common::PerformanceTimer &timer() { static Common::PerformanceTimer t; return t; }
inline void scan(int n, int *odata, const int *idata) { sc_cpu_scan(n, odata, idata); timer().mi355x_set(sc_last_cpu_ms(), -1.f); }
inline int compactWithoutScan(int n, int *odata, const int *idata) {
    const int k = sc_cpu_compact_without_scan(n, odata, idata);
    timer().mi355x_set(sc_last_cpu_ms(), -1.f);
    return k;
}
inline int compactWithScan(int n, int *odata, const int *idata) {
    const int k = sc_cpu_compact_with_scan(n, odata, idata);
    timer().mi355x_set(sc_last_cpu_ms(), -1.f);
    return k;
}
}  // namespace CPU

namespace Naive {                                            // naive.h:5-9
inline Common::PerformanceTimer &timer() { static Common::PerformanceTimer t; return t; }
inline void scan(int n, int *odata, const int *idata) {
    Common::mi355x_check(sc_naive_scan(n, odata, idata), "Naive::scan");
    timer().mi355x_set(-1.f, sc_last_gpu_ms());
}
}  // namespace Naive

namespace Efficient {                                        // efficient.h:5-13
inline Common::PerformanceTimer &timer() { static Common::PerformanceTimer t; return t; }
inline void scan(int n, int *odata, const int *idata) {
    Common::mi355x_check(sc_efficient_scan(n, odata, idata), "Efficient::scan");
    timer().mi355x_set(-1.f, sc_last_gpu_ms());
}
inline int compact(int n, int *odata, const int *idata) {
    const int k = sc_efficient_compact(n, odata, idata);
    Common::mi355x_check(k < 0 ? 1 : 0, "Efficient::compact");
    timer().mi355x_set(-1.f, sc_last_gpu_ms());
    return k;
}
}  // namespace Efficient

namespace Thrust {                                           // thrust.h:5-9
inline Common::PerformanceTimer &timer() { static Common::PerformanceTimer t; return t; }
inline void scan(int n, int *odata, const int *idata) {
    Common::mi355x_check(sc_thrust_scan(n, odata, idata), "Thrust::scan");
    timer().mi355x_set(-1.f, sc_last_gpu_ms());
}
}  // namespace Thrust
}  // namespace StreamCompaction
