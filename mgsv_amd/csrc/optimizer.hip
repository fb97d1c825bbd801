// Train-step tail of the MaDe path on gfx950: per-group gradient-norm clipping + Adam on the flat f32 master buffer, and the
// re-derivation of the kernel-facing parameter copies (compute-dtype W and W^T) in one launch each.
// Replaces, per step, the reference's three nn.utils.clip_grad_norm_ calls, optim.Adam.step() over ~200 tensors and the
// implicit per-tensor work of autograd (reference train-MaDe.py:262-266,375-381): three launches instead of several hundred.
#include "common.h"

namespace {

constexpr int OT = 256;

struct AdamArgs {
    float* param; const float* grad; float* m; float* v; int64_t n;
    MadeAdamGroup g[MADE_ADAM_MAX_GROUPS]; int n_groups;
    float beta1, beta2, eps, bc1, bc2_sqrt, grad_scale;
    float* norm_sq;
    const MadeAdamDeviceState* st;          // non-NULL: step count / learning rates / bias corrections live in device memory
};

__device__ __forceinline__ int group_of(const AdamArgs& a, int64_t i) {
    int gi = -1;
#pragma unroll
    for (int k = 0; k < MADE_ADAM_MAX_GROUPS; ++k)
        if (k < a.n_groups && i >= a.g[k].begin && i < a.g[k].end) gi = k;
    return gi;
}

// squared L2 norm of every group's (scaled) gradient: per-workgroup partial sums
__global__ __launch_bounds__(OT) void grad_norm_kernel(const AdamArgs a) {
    __shared__ float red[MADE_ADAM_MAX_GROUPS][OT / 64];
    float acc[MADE_ADAM_MAX_GROUPS];
#pragma unroll
    for (int k = 0; k < MADE_ADAM_MAX_GROUPS; ++k) acc[k] = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * OT + threadIdx.x) * 4; i < a.n; i += (int64_t)gridDim.x * OT * 4) {
        if (group_of(a, i) < 0 && group_of(a, i + 3) < 0) continue;   // (a call may cover some of the groups only: nothing is read outside them)
        const f32x4 g4 = *(const f32x4*)(a.grad + i);           // n is a multiple of 4 (parameters start on 64-element boundaries)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gi = group_of(a, i + j);
            const float g = g4[j] * a.grad_scale;
#pragma unroll
            for (int k = 0; k < MADE_ADAM_MAX_GROUPS; ++k) acc[k] += (gi == k) ? g * g : 0.f;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < MADE_ADAM_MAX_GROUPS; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[k][wave] = s;
    }
    __syncthreads();
    // per-workgroup partials, combined in a fixed order by norm_finish_kernel: every rank of a data-parallel job computes
    // bit-identical norms from the all-reduced gradients (atomics would let the ranks drift apart)
    if (threadIdx.x < MADE_ADAM_MAX_GROUPS) {
        float s = 0.f;
        for (int w = 0; w < OT / 64; ++w) s += red[threadIdx.x][w];
        a.norm_sq[MADE_ADAM_MAX_GROUPS + (int64_t)blockIdx.x * MADE_ADAM_MAX_GROUPS + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(OT) void norm_finish_kernel(float* norm_ws, int nblocks, MadeAdamDeviceState* st, float beta1, float beta2, int advance) {
    __shared__ float red[OT];
    if (st && advance && threadIdx.x == 0) {   // this single-workgroup launch also advances the device-side step count
        const int64_t step = ++st->step;
        st->bc1 = (float)(1.0 - pow((double)beta1, (double)step));
        st->bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    }
    for (int k = 0; k < MADE_ADAM_MAX_GROUPS; ++k) {
        float s = 0.f;
        for (int b = threadIdx.x; b < nblocks; b += OT) s += norm_ws[MADE_ADAM_MAX_GROUPS + (int64_t)b * MADE_ADAM_MAX_GROUPS + k];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = OT / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) norm_ws[k] = red[0];
        __syncthreads();
    }
}

// g' = clip_coef(group) * grad_scale * g;  m, v, p updated as torch.optim.Adam does (amsgrad off, weight_decay 0)
__global__ __launch_bounds__(OT) void adam_update_kernel(const AdamArgs a) {
    float coef[MADE_ADAM_MAX_GROUPS];
#pragma unroll
    for (int k = 0; k < MADE_ADAM_MAX_GROUPS; ++k) {
        coef[k] = 1.f;
        if (k < a.n_groups && a.g[k].max_norm > 0.f) {
            const float c = a.g[k].max_norm / (sqrtf(a.norm_sq[k]) + 1e-6f);       // nn.utils.clip_grad_norm_
            coef[k] = c < 1.f ? c : 1.f;
        }
    }
    float lrs[MADE_ADAM_MAX_GROUPS];
#pragma unroll
    for (int k = 0; k < MADE_ADAM_MAX_GROUPS; ++k) lrs[k] = a.st ? a.st->lr[k] : a.g[k].lr;
    const float bc1 = a.st ? a.st->bc1 : a.bc1, bc2_sqrt = a.st ? a.st->bc2_sqrt : a.bc2_sqrt;
    for (int64_t i = ((int64_t)blockIdx.x * OT + threadIdx.x) * 4; i < a.n; i += (int64_t)gridDim.x * OT * 4) {
        const int g0 = group_of(a, i), g3 = group_of(a, i + 3);
        if (g0 < 0 && g3 < 0) continue;
        f32x4 g4 = *(const f32x4*)(a.grad + i), m4 = *(const f32x4*)(a.m + i), v4 = *(const f32x4*)(a.v + i), p4 = *(const f32x4*)(a.param + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gi = group_of(a, i + j);
            if (gi < 0) continue;
            float cf = 1.f, lr = 0.f;
#pragma unroll
            for (int k = 0; k < MADE_ADAM_MAX_GROUPS; ++k)
                if (gi == k) { cf = coef[k]; lr = lrs[k]; }
            const float g = g4[j] * a.grad_scale * cf;
            m4[j] = a.beta1 * m4[j] + (1.f - a.beta1) * g;
            v4[j] = a.beta2 * v4[j] + (1.f - a.beta2) * g * g;
            const float denom = sqrtf(v4[j]) / bc2_sqrt + a.eps;
            p4[j] -= (lr / bc1) * (m4[j] / denom);
        }
        *(f32x4*)(a.m + i) = m4; *(f32x4*)(a.v + i) = v4; *(f32x4*)(a.param + i) = p4;
    }
}

// one 64 x 64 tile of one matrix per workgroup: W (cast) and W^T (cast, through LDS so both sides stay coalesced).
// (first version: 32 x 32 tiles, scalar accesses and a binary search over the descriptor table in global memory -- seven DEPENDENT
//  loads in front of every workgroup's first useful one: 101 us for 21 M parameters, a third of the HBM rate.)  Here the descriptor
//  is found by one parallel pass (thread i tests descriptor i), rows move as 16-byte loads / 8-byte bf16 stores.
constexpr int RPT = 64;
__global__ __launch_bounds__(OT) void repack_kernel(const MadeRepackDesc* descs, int n_desc) {
    __shared__ float tile[RPT][RPT + 1];
    __shared__ int s_desc;
    const int64_t t = blockIdx.x;
    for (int i = threadIdx.x; i < n_desc; i += OT) {
        const int64_t b = descs[i].tile_begin;
        const int64_t e = i + 1 < n_desc ? descs[i + 1].tile_begin : INT64_MAX;
        if (b <= t && t < e) s_desc = i;
    }
    __syncthreads();
    const MadeRepackDesc d = descs[s_desc];
    const int64_t local = t - d.tile_begin;
    const int64_t tc_n = (d.cols + RPT - 1) / RPT;
    const int64_t r0 = (local / tc_n) * RPT, c0 = (local % tc_n) * RPT;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 column groups of 4 x 16 rows
    const bool bf = d.dtype == MADE_BF16;
#pragma unroll
    for (int k = 0; k < RPT / 16; ++k) {
        const int rr = ty + 16 * k;
        const int64_t r = r0 + rr, c = c0 + 4 * tx;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < d.rows && c < d.cols) {
            const float* sp = d.src + r * d.cols + c;
            if (c + 4 <= d.cols && (((uintptr_t)sp & 15) == 0)) {
                const f32x4 q = *(const f32x4*)sp;
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c + j < d.cols) v[j] = sp[j];
            }
            if (d.w) {
                if (bf && c + 4 <= d.cols && (((r * d.cols + c) & 3) == 0)) {
                    bf16x4 o; o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
                    *(bf16x4*)((bf16_t*)d.w + r * d.cols + c) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (c + j < d.cols) store_from_f32(d.w, d.dtype, r * d.cols + c + j, v[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[rr][4 * tx + j] = v[j];
    }
    __syncthreads();
    if (d.wt) {
#pragma unroll
        for (int k = 0; k < RPT / 16; ++k) {
            const int cc = ty + 16 * k;
            const int64_t c = c0 + cc, r = r0 + 4 * tx;
            if (c < d.cols && r < d.rows) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = tile[4 * tx + j][cc];
                if (bf && r + 4 <= d.rows && (((c * d.wt_ld + r) & 3) == 0)) {
                    bf16x4 o; o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
                    *(bf16x4*)((bf16_t*)d.wt + c * d.wt_ld + r) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (r + j < d.rows) store_from_f32(d.wt, d.dtype, c * d.wt_ld + r + j, v[j]);
                }
            }
        }
    }
}

}  // namespace

static int adam_launch(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const MadeAdamGroup* groups,
                       int32_t n_groups, float beta1, float beta2, float eps, int64_t step, MadeAdamDeviceState* state_device, int advance,
                       float grad_scale, float* norm_ws, void* stream, const char* what) {
    AdamArgs a;
    a.param = param; a.grad = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n; a.n_groups = n_groups;
    for (int k = 0; k < MADE_ADAM_MAX_GROUPS; ++k) {
        if (k < n_groups) a.g[k] = groups[k];
        else { a.g[k].begin = a.g[k].end = 0; a.g[k].lr = 0.f; a.g[k].max_norm = 0.f; }
    }
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale;
    a.bc1 = state_device ? 1.f : (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = state_device ? 1.f : (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.norm_sq = norm_ws;
    a.st = state_device;
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = (n / 4 + OT - 1) / OT;
    if (nb > MADE_ADAM_NORM_BLOCKS) nb = MADE_ADAM_NORM_BLOCKS;
    hipLaunchKernelGGL(grad_norm_kernel, dim3((unsigned)nb), dim3(OT), 0, st, a);
    hipLaunchKernelGGL(norm_finish_kernel, dim3(1), dim3(OT), 0, st, norm_ws, (int)nb, state_device, beta1, beta2, advance);
    hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)nb), dim3(OT), 0, st, a);
    return made_check_launch(what);
}

extern "C" int made_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                              const MadeAdamGroup* groups, int32_t n_groups, float beta1, float beta2, float eps, int64_t step,
                              float grad_scale, float* norm_ws, void* stream) {
    MADE_REQUIRE(param && grad && exp_avg && exp_avg_sq && groups && norm_ws, "made_adam_step: null pointer");
    MADE_REQUIRE(n_groups >= 1 && n_groups <= MADE_ADAM_MAX_GROUPS, "made_adam_step: n_groups=%d out of range", n_groups);
    MADE_REQUIRE(n > 0 && n % 4 == 0 && step >= 1, "made_adam_step: n must be a positive multiple of 4 and step >= 1");
    return adam_launch(param, grad, exp_avg, exp_avg_sq, n, groups, n_groups, beta1, beta2, eps, step, nullptr, 0, grad_scale, norm_ws, stream,
                       "made_adam_step");
}

extern "C" int made_adam_step_device(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                     const MadeAdamGroup* groups, int32_t n_groups, float beta1, float beta2, float eps,
                                     MadeAdamDeviceState* state_device, int32_t advance_state, float grad_scale, float* norm_ws, void* stream) {
    MADE_REQUIRE(param && grad && exp_avg && exp_avg_sq && groups && norm_ws && state_device, "made_adam_step_device: null pointer");
    MADE_REQUIRE(n_groups >= 1 && n_groups <= MADE_ADAM_MAX_GROUPS, "made_adam_step_device: n_groups=%d out of range", n_groups);
    MADE_REQUIRE(n > 0 && n % 4 == 0, "made_adam_step_device: n must be a positive multiple of 4");
    return adam_launch(param, grad, exp_avg, exp_avg_sq, n, groups, n_groups, beta1, beta2, eps, 0, state_device, advance_state != 0, grad_scale, norm_ws, stream,
                       "made_adam_step_device");
}

extern "C" int made_repack(const MadeRepackDesc* descs_device, int32_t n_desc, int64_t total_tiles, void* stream) {
    MADE_REQUIRE(descs_device != nullptr && n_desc >= 1 && total_tiles >= 1, "made_repack: bad arguments");
    MADE_UNSUPPORTED(total_tiles < (1LL << 31), "made_repack: too many tiles");
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)total_tiles), dim3(OT), 0, (hipStream_t)stream, descs_device, (int)n_desc);
    return made_check_launch("made_repack");
}
