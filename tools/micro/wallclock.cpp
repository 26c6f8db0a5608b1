// prints the rates of the two device timers (kHz): wall_clock64() and s_memtime / clock64()
#include <hip/hip_runtime.h>
#include <cstdio>
int main() {
    int wc = 0, ci = 0;
    hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
    hipDeviceGetAttribute(&ci, hipDeviceAttributeClockInstructionRate, 0);
    printf("wall_clock64 kHz %d, clock64 kHz %d\n", wc, ci);
    return 0;
}
