// Stand-alone check of DESIGN.md's packed-FP32 claim (round 4, 3d-1): do v_pk_{mul,add,fma}_f32 results depend on what shares the CU?
//   hipcc -O3 --offload-arch=gfx950 -o pk_f32_corun_probe pk_f32_corun_probe.hip && ./pk_f32_corun_probe [launches=10000]
// Kernel under test, two modes: (1) a dependent chain of hand-placed v_pk_fma_f32 / v_pk_mul_f32 on fixed inputs; (2) packed f32
// arithmetic (explicit 2-vectors, compiler-scheduled, its hazard recogniser in play) on the result of a bf16 MFMA through an LDS round trip
// and a DPP row sum -- the instruction mix of made_dec_stage_bwd's tail.  Each mode runs `launches` times on stream 1 while ONE co-runner
// loops on stream 2 -- a register-staged copy loop (global -> VGPR -> LDS -> VALU, one 256-thread block per CU), then the same loop with bf16
// MFMAs on the staged tile (the instruction mix of the register-staged Linear that exposed the deviation); every launch's output is compared
// bit for bit with the solo run by a compare kernel.  Prints the mismatch counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void pk_chain(const float* in, float* out, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {in[2 * i], in[2 * i + 1]}, b = {in[2 * i + 1] * 0.5f + 0.25f, in[2 * i] * 0.75f}, c = {1.f, -1.f}, d = {0.999f, 0.998f};
    for (int k = 0; k < iters; ++k) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(c) : "v"(d));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    }
    out[2 * i] = c.x + a.x; out[2 * i + 1] = c.y + a.y;
}
__global__ void pk_after_mfma(const float* in, float* out, int iters) {
    __shared__ f32x4 tile[256];
    int t = threadIdx.x, i = blockIdx.x * blockDim.x + t;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[(i * 8 + j) & 0xFFFF]; b[j] = (__bf16)in[(i * 8 + j + 4099) & 0xFFFF]; }
    f32x4 acc = {0, 0, 0, 0}; f32x2 s = {in[i & 0xFFFF], 0.5f}, r = {0.f, 0.f};
    for (int k = 0; k < iters; ++k) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        tile[t] = acc; __syncthreads(); f32x4 o = tile[(t * 5 + k) & 255]; __syncthreads();
        f32x2 p = {acc[0], acc[1]}, q = {o[2], o[3]};
        r = p * s + r; r = r * q; r = r + f32x2{0.125f, -0.25f};                      // v_pk_fma / v_pk_mul / v_pk_add
        float rs = r.x + __shfl_xor(r.x, 1) + __shfl_xor(r.y, 2);                      // DPP / swizzle row sum
        acc = acc * 0.5f + f32x4{rs, r.y, r.x, rs} * 1e-3f; r = r * 0.25f;
    }
    out[2 * i] = r.x + acc[0]; out[2 * i + 1] = r.y + acc[3];
}
template <bool MFMA>
__global__ void corun(const f32x4* src, float* sink, int n, int loops) {
    __shared__ f32x4 tile[256];
    float acc = 0;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 c = {0};
    for (int l = 0; l < loops; ++l) {
        f32x4 v = src[(blockIdx.x * 256 + threadIdx.x + l * 7919) % n];
        tile[threadIdx.x] = v; __syncthreads(); f32x4 w = tile[(threadIdx.x * 17) & 255]; acc += w[0] * v[1] + w[2];
        if (MFMA) {
            bf16x8 fa = *(const bf16x8*)&tile[(threadIdx.x * 3) & 255], fb = *(const bf16x8*)&tile[(threadIdx.x * 7 + 1) & 255];
            for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c, 0, 0, 0);
        }
        __syncthreads();
    }
    if (acc + c[0] + c[7] == 123.456f) sink[0] = acc;
}
__global__ void compare(const unsigned* got, const unsigned* ref, int n, unsigned* bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && got[i] != ref[i]) atomicAdd(bad, 1u);
}
int main(int argc, char** argv) {
    int launches = argc > 1 ? atoi(argv[1]) : 10000, N = 256 * 256 * 4, iters = 64;
    std::vector<float> h(1 << 20); srand(7); for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *in, *ref, *out, *sink; f32x4* big; unsigned* bad;
    CK(hipMalloc(&in, h.size() * 4)); CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ref, N * 8)); CK(hipMalloc(&out, N * 8)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&big, 64 << 20)); CK(hipMemset(big, 0, 64 << 20)); CK(hipMalloc(&bad, 8));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    for (int mode = 1; mode <= 2; ++mode) {
        auto run = [&](float* dst) { if (mode == 1) pk_chain<<<N / 256, 256, 0, s1>>>(in, dst, iters); else pk_after_mfma<<<N / 256, 256, 0, s1>>>(in, dst, iters); };
        run(ref); CK(hipStreamSynchronize(s1));
        for (int co = 0; co <= 2; ++co) {                                                  // without, then beside each co-runner
            CK(hipMemset(bad, 0, 8));
            for (int l = 0; l < launches; ++l) {
                if (co == 1 && l % 50 == 0) corun<false><<<256, 256, 0, s2>>>(big, sink, (64 << 20) / 16, 4000);   // keeps stream 2 busy throughout
                if (co == 2 && l % 50 == 0) corun<true><<<256, 256, 0, s2>>>(big, sink, (64 << 20) / 16, 2000);
                run(out); compare<<<(2 * N + 255) / 256, 256, 0, s1>>>((unsigned*)out, (unsigned*)ref, 2 * N, bad);
            }
            CK(hipDeviceSynchronize());
            unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            printf("mode %d (%s) %s: %u differing words over %d launches x %d words\n", mode, mode == 1 ? "v_pk chain, inline asm" : "packed f32 after MFMA + LDS + DPP",
                   co == 0 ? "alone" : (co == 1 ? "beside the copy-loop co-runner" : "beside the copy + MFMA co-runner"), hb, launches, 2 * N);
        }
    }
    return 0;
}
