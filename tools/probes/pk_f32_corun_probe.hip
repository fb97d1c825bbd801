// Stand-alone check of DESIGN.md's packed-FP32 claim (round 4, 3d-1): do v_pk_{mul,add,fma}_f32 results depend on what shares the CU?
//   hipcc -O3 --offload-arch=gfx950 -o pk_f32_corun_probe pk_f32_corun_probe.hip && ./pk_f32_corun_probe [launches=10000]
// Kernel under test, three modes (mode 3 = the LayerNorm-backward row of made_dec_stage_bwd as it is written in csrc/decoder.hip -- four row sums
// through DPP, 1 / sqrt, the bf16 conversions -- left to the SLP vectoriser, which packs it): (1) a dependent chain of hand-placed v_pk_fma_f32 / v_pk_mul_f32 on fixed inputs; (2) packed f32
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
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_f32(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1, 0xF>(v, v); v += dpp_f32<0x4E, 0xF>(v, v); v += dpp_f32<0x124, 0xF>(v, v); v += dpp_f32<0x128, 0xF>(v, v);
    v += dpp_f32<0x142, 0xA>(0.f, v); v += dpp_f32<0x143, 0xC>(0.f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__global__ void ln_bwd_rows(const __bf16* xa, const __bf16* dy, const float* gamma, __bf16* out, int rows) {   // one wave per row of 512
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float xv[8], gy[8], gm[8], o[8];
    const bf16x8 a = *(const bf16x8*)(xa + (size_t)row * 512 + lane * 8), b = *(const bf16x8*)(dy + (size_t)row * 512 + lane * 8);
    for (int j = 0; j < 8; ++j) { xv[j] = (float)a[j]; gy[j] = (float)b[j]; gm[j] = gamma[lane * 8 + j]; }
    float sx = 0.f, sxx = 0.f, sg = 0.f, sgx = 0.f;
    for (int j = 0; j < 8; ++j) { const float g = gy[j] * gm[j]; sx += xv[j]; sxx += xv[j] * xv[j]; sg += g; sgx += g * xv[j]; }
    sx = wave_sum(sx); sxx = wave_sum(sxx); sg = wave_sum(sg); sgx = wave_sum(sgx);
    const float mean = sx * (1.f / 512), var = fmaxf(sxx * (1.f / 512) - mean * mean, 0.f), rstd = 1.0f / sqrtf(var + 1e-5f);
    const float s1 = sg * (1.f / 512), s2 = rstd * (sgx - mean * sg) * (1.f / 512);
    for (int j = 0; j < 8; ++j) { const float xh = (xv[j] - mean) * rstd; o[j] = rstd * (gy[j] * gm[j] - s1 - xh * s2); }
    bf16x8 t; for (int j = 0; j < 8; ++j) t[j] = (__bf16)o[j];
    *(bf16x8*)(out + (size_t)row * 512 + lane * 8) = t;
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
__global__ void corun_mfma_dense(float* sink, int loops) {       // nothing but back-to-back bf16 MFMAs (what a GEMM's slab phase looks like to a co-resident wave)
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    bf16x8 a, b; for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * (threadIdx.x ^ j)); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int l = 0; l < loops; ++l) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 123.456f) sink[0] = c0[0];
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
    __bf16 *bxa, *bdy; CK(hipMalloc(&bxa, 4096 * 512 * 2)); CK(hipMalloc(&bdy, 4096 * 512 * 2));
    { std::vector<__bf16> t(4096 * 512); for (auto& x : t) x = (__bf16)((float)rand() / RAND_MAX * 2.f - 1.f); CK(hipMemcpy(bxa, t.data(), t.size() * 2, hipMemcpyHostToDevice));
      for (auto& x : t) x = (__bf16)(((float)rand() / RAND_MAX * 2.f - 1.f) * 0.3f); CK(hipMemcpy(bdy, t.data(), t.size() * 2, hipMemcpyHostToDevice)); }
    for (int mode = 1; mode <= 3; ++mode) {
        auto run = [&](float* dst) { if (mode == 1) pk_chain<<<N / 256, 256, 0, s1>>>(in, dst, iters); else if (mode == 2) pk_after_mfma<<<N / 256, 256, 0, s1>>>(in, dst, iters);
                                     else ln_bwd_rows<<<512, 256, 0, s1>>>(bxa, bdy, in, (__bf16*)dst, 2048); };
        run(ref); CK(hipStreamSynchronize(s1));
        for (int co = 0; co <= 3; ++co) {                                                  // without, then beside each co-runner
            CK(hipMemset(bad, 0, 8));
            for (int l = 0; l < launches; ++l) {
                if (co == 1 && l % 50 == 0) corun<false><<<256, 256, 0, s2>>>(big, sink, (64 << 20) / 16, 4000);   // keeps stream 2 busy throughout
                if (co == 2 && l % 50 == 0) corun<true><<<256, 256, 0, s2>>>(big, sink, (64 << 20) / 16, 2000);
                if (co == 3 && l % 50 == 0) corun_mfma_dense<<<512, 256, 0, s2>>>(sink, 6000);
                run(out); compare<<<(2 * N + 255) / 256, 256, 0, s1>>>((unsigned*)out, (unsigned*)ref, 2 * N, bad);
            }
            CK(hipDeviceSynchronize());
            unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            printf("mode %d (%s) %s: %u differing words over %d launches x %d words\n", mode, mode == 1 ? "v_pk chain, inline asm" : mode == 2 ? "packed f32 after MFMA + LDS + DPP" : "LayerNorm-backward rows of decoder.hip (SLP-packed, DPP sums, 1/sqrt)",
                   co == 0 ? "alone" : (co == 1 ? "beside the copy-loop co-runner" : co == 2 ? "beside the copy + MFMA co-runner" : "beside the dense-MFMA co-runner"), hb, launches, 2 * N);
        }
    }
    return 0;
}
