// Does a kernel see everything the previous kernel of ITS stream stored, when a second stream keeps the chip busy?
// K1: 128 workgroups, the first 32 lanes of each store 16 bytes (value = iteration) at the very end; K2: 128 workgroups read the whole
// array at their very start and count entries that are not `iteration`.  Stream B runs a big streaming kernel the whole time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;
__global__ void k1(int* buf, int iter, int spin) {
    __shared__ float s[1024];
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    s[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x >= 32) return;
    i32x4 v; v[0] = v[1] = v[2] = v[3] = iter + (s[5] < -1.f ? 1 : 0);
    // row = lane / 2 of a 16-row tile, 32 bytes per row piece, rows 1 KB apart (as the 16 x 16-tile kernels store)
    const int row = (blockIdx.x >> 5) * 16 + (threadIdx.x >> 1), col = (blockIdx.x & 31) * 16 + (threadIdx.x & 1) * 8;
    *(i32x4*)((short*)buf + row * 512 + col) = v;
}
__global__ void k2(const int* buf, int iter, int* err) {
    // wave w of workgroup (x, y) reads rows 16 y + 4 w .. + 3 (1 KB = 256 ints each, 4 ints per lane) at once
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, y = blockIdx.x >> 5;
    int bad = 0;
    i32x4 v[4];
    for (int i = 0; i < 4; ++i) v[i] = *(const i32x4*)(buf + (16 * y + 4 * wave + i) * 256 + lane * 4);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) bad += v[i][j] != iter;
    if (bad) atomicAdd(err, bad);
}
__global__ void heavy(const float4* a, float4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 t = a[i]; t.x += 1.f; b[i] = t; }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000, load = argc > 2 ? atoi(argv[2]) : 1, spin = argc > 3 ? atoi(argv[3]) : 200;
    int *buf, *err; float4 *ha, *hb; size_t n = (size_t)64 << 20;   // 1 GiB each
    hipMalloc(&buf, 64 * 512 * 2); hipMalloc(&err, 4); hipMemset(err, 0, 4); hipMemset(buf, 0xff, 64 * 512 * 2);
    hipMalloc(&ha, n * 16); hipMalloc(&hb, n * 16); hipMemset(ha, 0, n * 16);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    for (int it = 0; it < iters; ++it) {
        if (load && (it % 8) == 0) hipLaunchKernelGGL(heavy, dim3(4096), dim3(256), 0, sb, ha, hb, n / 16);
        hipLaunchKernelGGL(k1, dim3(128), dim3(256), 0, sa, buf, it, spin);
        hipLaunchKernelGGL(k2, dim3(128), dim3(256), 0, sa, buf, it, err);
    }
    hipDeviceSynchronize();
    int e = 0; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
    printf("iterations %d, second stream %s, spin %d: stale ints seen by the next kernel: %d\n", iters, load ? "busy" : "idle", spin, e);
    return 0;
}
