// Probe: cost of an atomic-counter grid barrier across all CUs of an MI355X (one workgroup per CU), with a bounded spin.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/grid_barrier_probe.hip -o tools/probes/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int SLEEP>
__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned target, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(cnt, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(SLEEP);
            if (++spins > (1u << 22)) { atomicExch(err, 1u); ok = false; break; }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

// barrier among the workgroups that share blockIdx % 8 (one XCD under round-robin dispatch): own counter, own cache line
template <int SLEEP>
__global__ __launch_bounds__(1024) void probe_xcd(unsigned* cnt, unsigned* err, float* data, int nbar, int fence) {
    const int xcd = blockIdx.x & 7, nloc = gridDim.x >> 3, j = blockIdx.x >> 3;
    unsigned* c = cnt + xcd * 64;
    for (int i = 0; i < nbar; ++i) {
        if (threadIdx.x == 0) __hip_atomic_store(&data[((i & 1) * 8 + xcd) * 64 + j], (float)(i + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (fence) __threadfence();
            atomicAdd(c, 1u);
            unsigned spins = 0;
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(i + 1) * nloc) {
                __builtin_amdgcn_s_sleep(SLEEP);
                if (++spins > (1u << 22)) { atomicExch(err, 1u); return; }
            }
            if (fence) __threadfence();
            const float v = __hip_atomic_load(&data[((i & 1) * 8 + xcd) * 64 + (j + 1) % nloc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (float)(i + (j + 1) % nloc)) atomicExch(err, 2u);
        }
        __syncthreads();
    }
}

template <int SLEEP>
__global__ __launch_bounds__(1024) void probe(unsigned* cnt, unsigned* err, float* data, int nbar, int work) {
    float acc = 0.f;
    for (int i = 0; i < nbar; ++i) {
        // a little cross-block traffic: every block writes one value, reads its neighbour's after the barrier
        if (threadIdx.x == 0) data[(i & 1) * gridDim.x + blockIdx.x] = (float)(i + blockIdx.x);
        for (int w = 0; w < work; ++w) acc += __sinf(acc + w);
        if (!grid_barrier<SLEEP>(cnt, (unsigned)(i + 1) * gridDim.x, err)) return;
        if (threadIdx.x == 0) {
            const float v = __hip_atomic_load(&data[(i & 1) * gridDim.x + (blockIdx.x + 1) % gridDim.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (float)(i + (blockIdx.x + 1) % gridDim.x)) atomicExch(err, 2u);
        }
    }
    if (acc == 12345.f) data[0] = acc;
}

int main(int argc, char** argv) {
    int nblk = argc > 1 ? atoi(argv[1]) : 256, nbar = argc > 2 ? atoi(argv[2]) : 100, slp = argc > 3 ? atoi(argv[3]) : 1, nthr = argc > 4 ? atoi(argv[4]) : 256;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    unsigned *cnt, *err; float* data;
    hipMalloc(&cnt, 4096); hipMalloc(&err, 4); hipMalloc(&data, 2 * 8 * 64 * 4 + 2 * nblk * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cnt, 0, 4096); hipMemset(err, 0, 4);
        hipEventRecord(s);
        if (slp >= 100) hipLaunchKernelGGL(probe_xcd<1>, dim3(nblk), dim3(nthr), 0, 0, cnt, err, data, nbar, slp - 100);
        else if (slp == 1) hipLaunchKernelGGL(probe<1>, dim3(nblk), dim3(nthr), 0, 0, cnt, err, data, nbar, 0);
        else if (slp == 8) hipLaunchKernelGGL(probe<8>, dim3(nblk), dim3(nthr), 0, 0, cnt, err, data, nbar, 0);
        else hipLaunchKernelGGL(probe<32>, dim3(nblk), dim3(nthr), 0, 0, cnt, err, data, nbar, 0);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        unsigned h; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
        printf("sleep %d threads %d blocks %d barriers %d: %.1f us total, %.2f us per barrier, err %u\n", slp, nthr, nblk, nbar, ms * 1e3, ms * 1e3 / nbar, h);
    }
    return 0;
}
