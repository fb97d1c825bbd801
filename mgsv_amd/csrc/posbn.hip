// BatchNorm1d over the token axis + ReLU, forward and backward: the two normalisations inside the EmbeddingNet aggregator
// (agg_module = "mlp", reference model/model_Base.py:216-249).  The reference hands nn.BatchNorm1d a [B, T, F] tensor, so the
// "channel" is the token position t and the statistics of position t run over the B * F values found there.  HBM-bound: one
// workgroup per position walks its B rows of F values with 16-byte accesses (the rows of one position are T * ld apart, each row
// contiguous); a position's B * F values (256 KB at B = 128, F = 1024, bf16) stay in L2 between the passes.
#include "common.h"

namespace {

constexpr int PT = 1024;                   // 16 waves per position

__device__ __forceinline__ f32x4 ld4(const void* p, int dtype, int64_t idx) {
    f32x4 v;
    if (dtype == MADE_F32) {
        v = *(const f32x4*)((const float*)p + idx);
    } else {
        bf16x4 t = *(const bf16x4*)((const bf16_t*)p + idx);
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    return v;
}
__device__ __forceinline__ void st4(void* p, int dtype, int64_t idx, f32x4 v) {
    if (dtype == MADE_F32) {
        *(f32x4*)((float*)p + idx) = v;
    } else {
        bf16x4 t;
        t[0] = (bf16_t)v[0]; t[1] = (bf16_t)v[1]; t[2] = (bf16_t)v[2]; t[3] = (bf16_t)v[3];
        *(bf16x4*)((bf16_t*)p + idx) = t;
    }
}

// sum of `v` over the workgroup, returned to every thread (sm: PT / 64 floats; two barriers)
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                        // the previous reduction's readers are done with sm
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < PT / 64; ++w) s += sm[w];
    return s;
}

struct PosBnFwdArgs {
    const void* x; int xdt; int64_t ldx;
    const float* weight; const float* bias; float* running_mean; float* running_var; float momentum, eps;
    float* save_mean; float* save_rstd;
    void* y; int ydt; int64_t ldy;
    int B, T, F;
};

template <bool BATCH_STATS>
__global__ __launch_bounds__(PT) void posbn_relu_fwd_kernel(const PosBnFwdArgs a) {
    __shared__ float sm[PT / 64];
    const int t = blockIdx.x;
    const int f4 = a.F >> 2, n4 = a.B * f4;
    const float n = (float)a.B * (float)a.F;
    float mean, rstd;
    if (BATCH_STATS) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n4; i += PT) {
            const int b = i / f4, c = (i - b * f4) << 2;
            const f32x4 v = ld4(a.x, a.xdt, ((int64_t)b * a.T + t) * a.ldx + c);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        mean = block_sum(s, sm) / n;
        float q = 0.f;                      // second pass for the variance: E[(x - mean)^2], no cancellation
        for (int i = threadIdx.x; i < n4; i += PT) {
            const int b = i / f4, c = (i - b * f4) << 2;
            const f32x4 v = ld4(a.x, a.xdt, ((int64_t)b * a.T + t) * a.ldx + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[j] - mean; q += d * d; }
        }
        const float var = block_sum(q, sm) / n;
        rstd = 1.0f / sqrtf(var + a.eps);
        if (threadIdx.x == 0 && a.running_mean) {          // torch.nn.BatchNorm1d: the running variance takes the UNBIASED estimate
            a.running_mean[t] = (1.f - a.momentum) * a.running_mean[t] + a.momentum * mean;
            a.running_var[t] = (1.f - a.momentum) * a.running_var[t] + a.momentum * var * (n / fmaxf(n - 1.f, 1.f));
        }
    } else {
        mean = a.running_mean[t];
        rstd = 1.0f / sqrtf(a.running_var[t] + a.eps);
    }
    if (threadIdx.x == 0) { a.save_mean[t] = mean; a.save_rstd[t] = rstd; }
    const float sc = rstd * a.weight[t], sh = a.bias[t] - mean * sc;
    for (int i = threadIdx.x; i < n4; i += PT) {
        const int b = i / f4, c = (i - b * f4) << 2;
        const int64_t row = (int64_t)b * a.T + t;
        f32x4 v = ld4(a.x, a.xdt, row * a.ldx + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j] * sc + sh, 0.f);
        st4(a.y, a.ydt, row * a.ldy + c, v);
    }
}

struct PosBnBwdArgs {
    const void* x; int xdt; int64_t ldx;
    const void* y; int ydt; int64_t ldy;
    const void* dy; int dydt; int64_t lddy;
    const float* weight; const float* save_mean; const float* save_rstd;
    void* dx; int dxdt; int64_t lddx;
    float* dweight; float* dbias;
    int B, T, F;
};

template <bool BATCH_STATS>
__global__ __launch_bounds__(PT) void posbn_relu_bwd_kernel(const PosBnBwdArgs a) {
    __shared__ float sm[PT / 64];
    const int t = blockIdx.x;
    const int f4 = a.F >> 2, n4 = a.B * f4;
    const float n = (float)a.B * (float)a.F;
    const float mean = a.save_mean[t], rstd = a.save_rstd[t];
    float s1 = 0.f, s2 = 0.f;               // sum g, sum g * xhat  (g = dy where the ReLU let the value through)
    for (int i = threadIdx.x; i < n4; i += PT) {
        const int b = i / f4, c = (i - b * f4) << 2;
        const int64_t row = (int64_t)b * a.T + t;
        const f32x4 xv = ld4(a.x, a.xdt, row * a.ldx + c), yv = ld4(a.y, a.ydt, row * a.ldy + c), gv = ld4(a.dy, a.dydt, row * a.lddy + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = yv[j] > 0.f ? gv[j] : 0.f;
            s1 += g;
            s2 += g * ((xv[j] - mean) * rstd);
        }
    }
    s1 = block_sum(s1, sm);
    s2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
        if (a.dweight) unsafeAtomicAdd(a.dweight + t, s2);
        if (a.dbias) unsafeAtomicAdd(a.dbias + t, s1);
    }
    const float k = a.weight[t] * rstd, m1 = s1 / n, m2 = s2 / n;
    for (int i = threadIdx.x; i < n4; i += PT) {
        const int b = i / f4, c = (i - b * f4) << 2;
        const int64_t row = (int64_t)b * a.T + t;
        const f32x4 xv = ld4(a.x, a.xdt, row * a.ldx + c), yv = ld4(a.y, a.ydt, row * a.ldy + c), gv = ld4(a.dy, a.dydt, row * a.lddy + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = yv[j] > 0.f ? gv[j] : 0.f;
            o[j] = BATCH_STATS ? k * (g - m1 - (xv[j] - mean) * rstd * m2) : k * g;
        }
        st4(a.dx, a.dxdt, row * a.lddx + c, o);
    }
}

bool dtype_ok(int32_t d) { return d == MADE_F32 || d == MADE_BF16; }

}  // namespace

extern "C" int made_posbn_relu_fwd(const void* x, int32_t x_dtype, int64_t ldx, const float* weight, const float* bias,
                                   float* running_mean, float* running_var, float momentum, float eps, int32_t batch_stats,
                                   float* save_mean, float* save_rstd, void* y, int32_t y_dtype, int64_t ldy,
                                   int64_t B, int64_t T, int64_t F, void* stream) {
    MADE_REQUIRE(x && weight && bias && save_mean && save_rstd && y, "made_posbn_relu_fwd: null pointer");
    MADE_REQUIRE(batch_stats || (running_mean && running_var), "made_posbn_relu_fwd: running statistics needed when batch_stats == 0");
    MADE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "made_posbn_relu_fwd: running_mean / running_var come together");
    MADE_REQUIRE(dtype_ok(x_dtype) && dtype_ok(y_dtype), "made_posbn_relu_fwd: dtype must be f32 or bf16");
    MADE_UNSUPPORTED(F > 0 && F % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= F && ldy >= F && B * (F / 4) < (1ll << 31) && T < (1ll << 31),
                     "made_posbn_relu_fwd: F=%lld ldx=%lld ldy=%lld unsupported", (long long)F, (long long)ldx, (long long)ldy);
    if (B <= 0 || T <= 0) return MADE_OK;
    PosBnFwdArgs a{x, x_dtype, ldx, weight, bias, running_mean, running_var, momentum, eps, save_mean, save_rstd, y, y_dtype, ldy, (int)B, (int)T, (int)F};
    if (batch_stats) hipLaunchKernelGGL(posbn_relu_fwd_kernel<true>, dim3((unsigned)T), dim3(PT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(posbn_relu_fwd_kernel<false>, dim3((unsigned)T), dim3(PT), 0, (hipStream_t)stream, a);
    return made_check_launch("made_posbn_relu_fwd");
}

extern "C" int made_posbn_relu_bwd(const void* x, int32_t x_dtype, int64_t ldx, const void* y, int32_t y_dtype, int64_t ldy,
                                   const void* dy, int32_t dy_dtype, int64_t lddy, const float* weight, const float* save_mean,
                                   const float* save_rstd, int32_t batch_stats, void* dx, int32_t dx_dtype, int64_t lddx,
                                   float* dweight, float* dbias, int64_t B, int64_t T, int64_t F, void* stream) {
    MADE_REQUIRE(x && y && dy && weight && save_mean && save_rstd && dx, "made_posbn_relu_bwd: null pointer");
    MADE_REQUIRE(dtype_ok(x_dtype) && dtype_ok(y_dtype) && dtype_ok(dy_dtype) && dtype_ok(dx_dtype), "made_posbn_relu_bwd: dtype must be f32 or bf16");
    MADE_UNSUPPORTED(F > 0 && F % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ldx >= F && ldy >= F && lddy >= F &&
                     lddx >= F && B * (F / 4) < (1ll << 31) && T < (1ll << 31), "made_posbn_relu_bwd: F=%lld / strides unsupported", (long long)F);
    if (B <= 0 || T <= 0) return MADE_OK;
    PosBnBwdArgs a{x, x_dtype, ldx, y, y_dtype, ldy, dy, dy_dtype, lddy, weight, save_mean, save_rstd, dx, dx_dtype, lddx, dweight, dbias, (int)B, (int)T, (int)F};
    if (batch_stats) hipLaunchKernelGGL(posbn_relu_bwd_kernel<true>, dim3((unsigned)T), dim3(PT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(posbn_relu_bwd_kernel<false>, dim3((unsigned)T), dim3(PT), 0, (hipStream_t)stream, a);
    return made_check_launch("made_posbn_relu_bwd");
}
