// How many cycles does one dropout hash cost a wave?  fmix32 (two v_mul_lo_u32) against candidates made of full-rate instructions only.
// hipcc --offload-arch=gfx950 -O3 tools/probes/hash_rate_probe.hip -o tools/probes/hash_rate_probe && ./tools/probes/hash_rate_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
// two rounds of xorshift + 24-bit multiply (the multiply sees the low 24 bits; the shift before it folds the high bits in)
__device__ __forceinline__ uint32_t mix24(uint32_t h) { h ^= h >> 15; h = mul24(h, 0x6B43A9u) ^ (h >> 9); h ^= h >> 13; h = mul24(h, 0xB2AE35u) ^ (h >> 11); h ^= h >> 16; return h; }
template <int WHICH>
__global__ void k(uint32_t* out, int iters, uint32_t key, uint32_t thr) {
    uint32_t lo = threadIdx.x * 977u + blockIdx.x * 131071u, cnt = 0;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t x = (lo + (uint32_t)(i * 16 + j)) ^ key;
            const uint32_t h = WHICH == 0 ? fmix32(x) : mix24(x);
            cnt += ((h >> 8) >= thr) ? 1u : 0u;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = cnt;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint32_t)(t1 - t0);
}
int main() {
    uint32_t* d; hipMalloc(&d, ((1 << 20) + 4) * 4);
    const int iters = 4096;
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, d, iters, 0x1234567u, 1677721u);
            else hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, d, iters, 0x1234567u, 1677721u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            uint32_t clk, c0; hipMemcpy(&clk, d + (1 << 20), 4, hipMemcpyDeviceToHost); hipMemcpy(&c0, d, 4, hipMemcpyDeviceToHost);
            // 1024 workgroups x 4 waves over 256 CUs x 4 SIMDs = 4 waves per SIMD; a wave's hashes: iters * 16
            const double hashes_per_simd = 4.0 * iters * 16;
            printf("%s: %.3f ms; %.1f ns per wave-hash per SIMD (= %.1f cycles at 2.4 GHz); kept %u of %d\n", which == 0 ? "fmix32 (2 x v_mul_lo_u32)" : "mix24 (2 x v_mul_u32_u24)",
                   ms, ms * 1e6 / hashes_per_simd, ms * 1e6 / hashes_per_simd * 2.4, c0, iters * 16);
        }
    }
    return 0;
}
