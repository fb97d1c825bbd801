// Does a packed-FP32 instruction count as a wait state between a VALU write and a DPP read of the same VGPR?  (gfx950 / ROCm 7.2: the
// compiler's hazard recogniser needs 2 wait states there and, in made_dec_stage_bwd's SLP build, fills them with `v_pk_mul_f32` + one SALU
// instruction instead of `s_nop 1` -- the first of the four row sums of csrc/decoder.hip's LayerNorm backward.)  Each variant writes a fresh
// value into a register, executes the fillers, reads the register through DPP quad_perm:[1,0,3,2] and checks that it sees the NEIGHBOUR'S
// fresh value; run alone and beside a co-runner that keeps the other SIMD slots busy.
//   hipcc -O3 --offload-arch=gfx950 -o p dpp_pk_hazard_probe.hip && ./p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int V>
__global__ void probe(unsigned* bad, int iters) {
    const unsigned lane = threadIdx.x & 63;
    unsigned nbad = 0;
    f32x2 p = {1.5f, 2.5f}, q = {0.5f, 0.25f}, r2 = {3.f, 4.f};
    for (int k = 0; k < iters; ++k) {
        unsigned x = (lane * 2654435761u) ^ (unsigned)(k * 40503 + blockIdx.x), stale = 0xDEAD0000u | lane;
        unsigned t = stale, d;
        if (V == 0)        // the reference: two explicit wait states
            asm volatile("v_mov_b32 %0, %2\n\ts_nop 1\n\tv_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(t), "=v"(d) : "v"(x));
        else if (V == 1)   // what the compiler emitted: a packed multiply and a scalar move as the two wait states
            asm volatile("v_mov_b32 %0, %3\n\tv_pk_mul_f32 %2, %4, %5\n\ts_mov_b32 s20, 0x85ebca6b\n\tv_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(t), "=&v"(d), "=&v"(p) : "v"(x), "v"(q), "v"(r2) : "s20");
        else if (V == 2)   // a packed multiply alone (one wait state short if it counts as one, two short if it counts as none)
            asm volatile("v_mov_b32 %0, %3\n\tv_pk_mul_f32 %2, %4, %5\n\tv_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(t), "=&v"(d), "=&v"(p) : "v"(x), "v"(q), "v"(r2));
        else               // no wait state at all (what a violated hazard looks like, if it shows at all)
            asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(t), "=v"(d) : "v"(x));
        const unsigned want = ((lane ^ 1u) * 2654435761u) ^ (unsigned)(k * 40503 + blockIdx.x);
        nbad += d != want;
    }
    if (nbad) atomicAdd(bad + V, nbad);
    if (p[0] == 123.f) bad[7] = 1;
}
__global__ void corun(const f32x4* src, float* sink, int n, int loops) {        // copy loop + MFMAs (the register-staged Linear's mix)
    __shared__ f32x4 tile[256];
    float acc = 0; f32x16 c = {0};
    for (int l = 0; l < loops; ++l) {
        f32x4 v = src[(blockIdx.x * 256 + threadIdx.x + l * 7919) % n];
        tile[threadIdx.x] = v; __syncthreads(); f32x4 w = tile[(threadIdx.x * 17) & 255]; acc += w[0] * v[1] + w[2];
        bf16x8 fa = *(const bf16x8*)&tile[(threadIdx.x * 3) & 255], fb = *(const bf16x8*)&tile[(threadIdx.x * 7 + 1) & 255];
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c, 0, 0, 0);
        __syncthreads();
    }
    if (acc + c[0] == 123.456f) sink[0] = acc;
}
int main() {
    unsigned* bad; float* sink; f32x4* big;
    hipMalloc(&bad, 64); hipMalloc(&sink, 4); hipMalloc(&big, 64 << 20); hipMemset(big, 0, 64 << 20);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int co = 0; co < 2; ++co) {
        hipMemset(bad, 0, 64);
        for (int rep = 0; rep < 200; ++rep) {
            if (co && rep % 10 == 0) corun<<<256, 256, 0, s2>>>(big, sink, (64 << 20) / 16, 3000);
            probe<0><<<1024, 256, 0, s1>>>(bad, 2000); probe<1><<<1024, 256, 0, s1>>>(bad, 2000);
            probe<2><<<1024, 256, 0, s1>>>(bad, 2000); probe<3><<<1024, 256, 0, s1>>>(bad, 2000);
        }
        hipDeviceSynchronize();
        unsigned h[8]; hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
        printf("%s: stale DPP reads of 200 x 1024 x 256 x 2000 -- s_nop 1: %u | v_pk_mul_f32 + s_mov: %u | v_pk_mul_f32 only: %u | nothing between: %u\n",
               co ? "beside the co-runner" : "alone", h[0], h[1], h[2], h[3]);
    }
    return 0;
}
