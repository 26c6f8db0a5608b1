// sc_veneer_check.cpp -- the reference's stream_compaction calls, spelled as its users spell them, through
// mygpuraytracer_amd/csrc/stream_compaction_api.h.  `cpu` runs only the CPU namespace (no GPU needed); without it every
// namespace runs and must agree with the CPU one.  Built and run by tests/test_abi.py (cpu) and the GPU tier.
#include <cstdio>
#include <cstring>
#include <vector>

#include "../mygpuraytracer_amd/csrc/stream_compaction_api.h"

int main(int argc, char **argv) {
    const bool cpu_only = argc > 1 && !strcmp(argv[1], "cpu");
    int bad = 0;
    for (int n : {1, 7, 256, 1000, 16384, 16385, 1 << 20, (1 << 21) + 3}) {
        std::vector<int> a(n), want(n), keep(n), got(n);
        unsigned s = 12345u + (unsigned)n;
        for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; a[i] = (s >> 28) < 7 ? 0 : (int)((s >> 20) & 63) - 20; }
        long long run = 0;
        int kept = 0;
        for (int i = 0; i < n; i++) { want[i] = (int)run; run += a[i]; if (a[i]) keep[kept++] = a[i]; }
        StreamCompaction::CPU::scan(n, got.data(), a.data());
        bad += memcmp(got.data(), want.data(), sizeof(int) * n) != 0;
        bad += StreamCompaction::CPU::timer().getCpuElapsedTimeForPreviousOperation() < 0.f;
        bad += StreamCompaction::CPU::compactWithoutScan(n, got.data(), a.data()) != kept || memcmp(got.data(), keep.data(), sizeof(int) * kept) != 0;
        bad += StreamCompaction::CPU::compactWithScan(n, got.data(), a.data()) != kept || memcmp(got.data(), keep.data(), sizeof(int) * kept) != 0;
        if (cpu_only) continue;
        StreamCompaction::Naive::scan(n, got.data(), a.data());
        bad += memcmp(got.data(), want.data(), sizeof(int) * n) != 0;
        StreamCompaction::Efficient::scan(n, got.data(), a.data());
        bad += memcmp(got.data(), want.data(), sizeof(int) * n) != 0;
        bad += !(StreamCompaction::Efficient::timer().getGpuElapsedTimeForPreviousOperation() > 0.f);
        StreamCompaction::Thrust::scan(n, got.data(), a.data());
        bad += memcmp(got.data(), want.data(), sizeof(int) * n) != 0;
        bad += StreamCompaction::Efficient::compact(n, got.data(), a.data()) != kept || memcmp(got.data(), keep.data(), sizeof(int) * kept) != 0;
    }
    bad += ilog2(1024) != 10 || ilog2ceil(1025) != 11;
    printf("%s: %d mismatches\n", cpu_only ? "cpu" : "all", bad);
    return bad ? 1 : 0;
}
