// EXPERIMENT (not part of libmade_hip.so): where do f32 atomic adds of a weight-gradient flush run, and how fast?
//   1. XCC_ID (s_getreg hwreg 20) of every workgroup of a 256-workgroup launch against blockIdx % 8 (the round-robin placement the kernels assume);
//   2. the flush pattern of gemm_tn_256_grouped_kernel (256 workgroups x 512 threads, each adding a 256 x 256 f32 tile, eight workgroups per tile):
//      agent-scope atomics, eight workgroups of a tile spread over the eight XCDs (what the kernel does today), against
//      workgroup-scope atomics (performed in the issuing XCD's L2) with the eight workgroups of a tile on ONE XCD; sums checked on the host.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/xcc_atomics_probe tools/probes/xcc_atomics_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }

__global__ void where_kernel(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

template <int SCOPE>       // 0: agent scope, tile = blockIdx / 8 (its eight workgroups on eight XCDs); 1: workgroup scope, tile owned by the real XCD
__global__ __launch_bounds__(512) void flush_kernel(float* C, int* claim) {
    __shared__ int s_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int wn = wave >> 2, wk = wave & 3;
    int tile;
    if (SCOPE == 0) tile = blockIdx.x >> 3;
    else {
        if (threadIdx.x == 0) { const int x = xcc_id(); const int u = atomicAdd(claim + x, 1); s_tile = x * 4 + ((u >> 3) & 3); }
        __syncthreads();
        tile = s_tile;
    }
    float* T = C + (size_t)tile * 65536;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = wn * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh, k = wk * 64 + j * 32 + r;
                if (SCOPE == 0) __hip_atomic_fetch_add(T + n * 256 + k, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_fetch_add(T + n * 256 + k, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
}

int main() {
    int* d_where; CK(hipMalloc(&d_where, 256 * 4));
    std::vector<int> where(256);
    int mism = 0, cnt[8] = {0};
    for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL(where_kernel, dim3(256), dim3(512), 0, 0, d_where);
        CK(hipMemcpy(where.data(), d_where, 256 * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < 256; ++b) { mism += where[b] != (b & 7); if (rep == 0) cnt[where[b] & 7]++; }
    }
    printf("XCC_ID != blockIdx %% 8 in %d of %d workgroups (20 launches of 256 x 512 threads); first launch per XCD:", mism, 20 * 256);
    for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
    printf("\n  first 16 workgroups:");
    for (int b = 0; b < 16; ++b) printf(" %d", where[b]);
    printf("\n");

    float* C; int* claim;
    CK(hipMalloc(&C, 32 * 65536 * 4)); CK(hipMalloc(&claim, 8 * 4));
    std::vector<float> h(32 * 65536);
    for (int scope = 0; scope < 2; ++scope) {
        CK(hipMemset(C, 0, 32 * 65536 * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 20;
        float ms = 0.f, tot = 0.f;
        for (int i = 0; i < reps + 2; ++i) {
            CK(hipMemsetAsync(claim, 0, 32, 0));
            CK(hipEventRecord(e0, 0));
            if (scope == 0) hipLaunchKernelGGL(flush_kernel<0>, dim3(256), dim3(512), 0, 0, C, claim);
            else hipLaunchKernelGGL(flush_kernel<1>, dim3(256), dim3(512), 0, 0, C, claim);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (i >= 2) tot += ms;
        }
        CK(hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; double sum = 0;
        for (float v : h) { sum += v; bad += v != 8.0f * (reps + 2); }
        printf("%s-scope atomics: %.1f us per launch (67 MB of f32 adds = %.2f TB/s), %zu of %zu sums wrong (total %.0f, expected %.0f)\n",
               scope == 0 ? "agent" : "workgroup", tot / reps * 1e3, 67.1e6 / (tot / reps * 1e-3) / 1e12, bad, h.size(), sum, 8.0 * (reps + 2) * h.size());
    }
    return 0;
}
