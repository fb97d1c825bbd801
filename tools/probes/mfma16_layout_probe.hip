// Checks the operand / result maps xpool_sims64_kernel's pass 1 assumes for v_mfma_f32_16x16x32_bf16 with exact integer data:
// A = K rows out of an LDS image swizzled like issue_k's (row r, 16-byte chunk c at slot c ^ (r & 15)), B = Q rows from global memory,
// D[seg][video] expected at lane (g4 = seg / 4 % 4 ... ) -- prints every mismatch.  hipcc -O3 --offload-arch=gfx950 -o p mfma16_layout_probe.hip && ./p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const __bf16* K, const __bf16* Q, float* out) {   // K [16][256], Q [16][256] -> out[seg][video]
    __shared__ __attribute__((aligned(16))) unsigned char lds[16 * 512];
    const int lane = threadIdx.x, v16 = lane & 15, g4 = lane >> 4;
    for (int i = lane; i < 16 * 32; i += 64) {                       // (row, chunk) -> slot chunk ^ (row & 15)
        const int row = i / 32, c = i % 32;
        *(bf16x8*)(lds + row * 512 + ((c ^ (row & 15)) << 4)) = *(const bf16x8*)(K + row * 256 + c * 8);
    }
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    const unsigned kx0 = v16 * 512 + ((((g4 ^ (v16 & 3)) | (v16 & 12))) << 4);
    for (int ks = 0; ks < 8; ++ks) {
        const unsigned addr = (kx0 ^ ((ks & 3) << 6)) + (ks >> 2) * 256;
        const bf16x8 a = *(const bf16x8*)(lds + addr);
        const bf16x8 b = *(const bf16x8*)(Q + v16 * 256 + ks * 32 + g4 * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
    for (int e = 0; e < 4; ++e) out[(4 * g4 + e) * 16 + v16] = acc[e];
}
int main() {
    std::vector<__bf16> K(16 * 256), Q(16 * 256);
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 256; ++c) { K[r * 256 + c] = (__bf16)(float)((r * 7 + c * 3) % 5 - 2); Q[r * 256 + c] = (__bf16)(float)((r * 5 + c) % 7 - 3); }
    __bf16 *dK, *dQ; float* dO;
    hipMalloc(&dK, K.size() * 2); hipMalloc(&dQ, Q.size() * 2); hipMalloc(&dO, 256 * 4);
    hipMemcpy(dK, K.data(), K.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dQ, Q.data(), Q.size() * 2, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dK, dQ, dO);
    std::vector<float> o(256); hipMemcpy(o.data(), dO, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int s = 0; s < 16; ++s) for (int v = 0; v < 16; ++v) {
        float ref = 0; for (int c = 0; c < 256; ++c) ref += (float)K[s * 256 + c] * (float)Q[v * 256 + c];
        if (ref != o[s * 16 + v]) { if (bad < 8) printf("seg %d video %d: got %g want %g\n", s, v, o[s * 16 + v], ref); ++bad; }
    }
    printf("%d of 256 mismatches\n", bad);
    return 0;
}
