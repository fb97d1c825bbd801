// Shader-clock probe: one wave spins for `ticks` ticks of the constant 100 MHz counter (s_memrealtime) and reports how many
// shader-clock cycles (s_memtime) passed meanwhile -> the frequency the CU runs at where the probe sits in a stream of work.
// out[3 * slot + 0] = shader cycles, + 1 = 100 MHz ticks, + 2 = realtime stamp at the start.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void clock_probe_kernel(unsigned long long* out, int slot, int ticks) {
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r1 = r0;
    while ((long long)(r1 - r0) < ticks) r1 = wall_clock64();
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) { out[3 * slot] = c1 - c0; out[3 * slot + 1] = r1 - r0; out[3 * slot + 2] = r0; }
}
extern "C" int clock_probe(void* out, int slot, int ticks, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, slot, ticks);
    return (int)hipGetLastError();
}
