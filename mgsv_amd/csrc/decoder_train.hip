// The moment-DETR decoder of the TRAINING step with one moment query (Q = 1), all layers in one launch per direction:
// reference music_detr/transformer.py:119-145 (stack) and :273-307 (forward_post layer) under model.train().
//
// With Q = 1 every sample's chain through the decoder is independent of the other samples: a layer is eight mat-vecs against the
// layer's weights, three LayerNorms and one cross-attention of 8 head queries over the sample's own memory rows.  Launched as
// 64-row GEMMs that chain is ~250 dependent launches per step (forward + backward) of ~9 us each, 16 workgroups wide: the
// training step's largest single cost on the device AND on the host.  Here one workgroup owns one sample and walks the whole
// stack; the only traffic is the weights (shared by all workgroups: L2 hits after the first reader) and the sample's memory
// rows.  Every intermediate the backward needs is written to the same [layer, sample, ...] stacks the unfused path fills, so the
// layer-batched weight-gradient products (mgsv_amd/trainer.py) are unchanged.
//
// Work distribution inside the workgroup (8 waves):
//   mat-vec   y[n] = sum_k W[n, k] x[k]: a row of W is read by 16 lanes (16 bytes per lane and load, the 16 lanes cover 256
//             contiguous bytes), v_dot2_f32_bf16 against the lane's slice of x, four DPP steps finish the row; 32 rows of W are in
//             flight per step of the workgroup, the next batch's loads are issued before the current batch is multiplied.
//   weighted row sum  out[h][:] = sum_r w[h][r] M[r][:] (P.V of the cross-attention, the per-head folds of W_k / W_v): a wave
//             reads a whole row per load, every lane keeps its columns' sums for all heads, waves meet through LDS float atomics.
// The memory-space formulation of the cross-attention is the unfused path's: q'_h = W_k,h^T qc_h, scores against (memory + pos),
// pooled_h = P_h memory, v_h = W_v,h pooled_h + s_h b_v,h  (s_h = sum of the dropped weights).
#include "common.h"
#include <type_traits>

namespace {

constexpr int NT = 512, NW = NT / 64;

template <typename TC> struct TT;
template <> struct TT<bf16_t> { static constexpr int PER = 8; typedef bf16x8 frag; };
template <> struct TT<float>  { static constexpr int PER = 4; typedef f32x4 frag; };
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ float fdot(bf16x8 w, bf16x8 x, float acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16x2 a = {w[2 * i], w[2 * i + 1]}, b = {x[2 * i], x[2 * i + 1]};
        acc = __builtin_amdgcn_fdot2_f32_bf16(a, b, acc, false);
    }
    return acc;
}
__device__ __forceinline__ float fdot(f32x4 w, f32x4 x, float acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_fmaf(w[i], x[i], acc);
    return acc;
}
__device__ __forceinline__ void to_frag(const float* p, bf16x8& f) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    f[0] = (bf16_t)a[0]; f[1] = (bf16_t)a[1]; f[2] = (bf16_t)a[2]; f[3] = (bf16_t)a[3];
    f[4] = (bf16_t)b[0]; f[5] = (bf16_t)b[1]; f[6] = (bf16_t)b[2]; f[7] = (bf16_t)b[3];
}
__device__ __forceinline__ void to_frag(const float* p, f32x4& f) { f = *(const f32x4*)p; }
__device__ __forceinline__ float frag_get(const bf16x8& f, int i) { return (float)f[i]; }
__device__ __forceinline__ float frag_get(const f32x4& f, int i) { return f[i]; }

// Pointers that come out of a descriptor in memory (MadeDecTrainLayer) are generic to the compiler: it would emit FLAT loads, which
// count against vmcnt AND lgkmcnt -- every LDS wait would then drain the weight loads in flight.  All of them are global.
template <typename F, typename T> __device__ __forceinline__ F gload(const T* p) {
    return *(const __attribute__((address_space(1))) F*)p;
}

// The operand streams (weight rows, memory rows) are read with BUFFER loads: a scalar resource + one 32-bit byte offset per lane
// (+ an immediate).  Global loads through pointers made the compiler keep a 64-bit address per lane and per load in flight, hoist
// them out of the layer loop and spill them -- and a reload from scratch in front of every load serialises the whole stream.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
template <typename F> __device__ __forceinline__ F bload(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(F, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
// a private copy of the thread index per call: what a primitive derives from it (lane offsets, row numbers) cannot be merged with
// another call's or hoisted out of the layer loop, so nothing of one phase stays live in another
__device__ __forceinline__ int opaque_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// sum over the 16 lanes of a DPP row; every lane of the row gets it
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1, 0xF>(v, v);
    v += dpp_f32<0x4E, 0xF>(v, v);
    v += dpp_f32<0x124, 0xF>(v, v);
    v += dpp_f32<0x128, 0xF>(v, v);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {      // red: NW floats; all threads call
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];
    return s;
}

// y[n] = sum_k W[n * ldw + k] x[k], n in [0, N), K = KC * nch; x: f32 in LDS.  N % (32 * UNR) == 0.  Raw sums land in y (LDS).
// The caller synchronises before x is valid and after y is written.
// hd_rows > 0: row n multiplies the vector xl + (n / hd_rows) * KC instead (one x per head; nch == 1).
template <typename TC, int KC, int UNR>
__device__ __forceinline__ void matvec(const TC* __restrict__ W, int64_t ldw, int N, int nch, const float* xl, float* yl, int hd_rows = 0) {
    typedef typename TT<TC>::frag frag;
    constexpr int PER = TT<TC>::PER, NLD = KC / (16 * PER);
    const int tx = opaque_tid();
    const int lane = tx & 63, wave = tx >> 6, g = lane >> 4, j = lane & 15;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(W);
    const int nb = N / (32 * UNR);
    float acc[UNR];
    frag wf[2][UNR][NLD];
    auto load = [&](int buf, int it, int c) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            // 32-bit element offsets from the (uniform) base: the loads take the scalar-base form instead of a 64-bit address per lane
            const uint32_t off = ((uint32_t)((it * UNR + u) * 32 + wave * 4 + g) * (uint32_t)ldw + (uint32_t)(c * KC + PER * j)) * (uint32_t)sizeof(TC);
#pragma unroll
            for (int i = 0; i < NLD; ++i) wf[buf][u][i] = bload<frag>(rs, off + 256 * i);
        }
    };
    const int total = nb * nch;                      // (row batch, K chunk) pairs, chunk fastest
    frag xf[NLD];
    auto step = [&](auto BUF, int t) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value;
        const int it = t / nch, c = t - it * nch;
        if ((nch > 1 || t == 0) && hd_rows == 0) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) to_frag(xl + c * KC + PER * (j + 16 * i), xf[i]);
        }
        if (c == 0) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) acc[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (hd_rows > 0) {
                const int head = ((it * UNR + u) * 32 + wave * 4 + g) / hd_rows;
#pragma unroll
                for (int i = 0; i < NLD; ++i) to_frag(xl + head * KC + PER * (j + 16 * i), xf[i]);
            }
#pragma unroll
            for (int i = 0; i < NLD; ++i) acc[u] = fdot(wf[buf][u][i], xf[i], acc[u]);
        }
        if (c == nch - 1) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const float s = row16_sum(acc[u]);
                if (j == 0) yl[(it * UNR + u) * 32 + wave * 4 + g] = s;
            }
        }
    };
    // (the prefetches are unconditional -- past the end they re-read the last batch: a load under a branch makes the waitcnt pass
    // assume it may not have been issued, and every wait for the current batch then drains the prefetch as well)
    // sched_barrier: hipcc otherwise sinks the next batch's loads below the current batch's arithmetic (one buffer, load-all /
    // wait-all / multiply: every batch pays a full memory round trip)
    load(0, 0, 0);
    for (int t = 0; t < total; t += 2) {
        { const int t1 = t + 1 < total ? t + 1 : total - 1; load(1, t1 / nch, t1 % nch); }
        __builtin_amdgcn_sched_barrier(0);
        step(std::integral_constant<int, 0>{}, t);
        __builtin_amdgcn_sched_barrier(0);
        { const int t2 = t + 2 < total ? t + 2 : total - 1; load(0, t2 / nch, t2 % nch); }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < total) step(std::integral_constant<int, 1>{}, t + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// S[h][r] = sum_k M[r * ld + k] x[h][k] for NH vectors x (f32 in LDS, [NH][KC]); rows r in [0, R) (any R).  sl: LDS, row pitch ldsl.
// UNR rows per 16-lane group and batch (UNR * NLD loads in flight per lane, twice).
template <typename TC, int KC, int NH, int UNR>
__device__ __forceinline__ void multidot(const TC* __restrict__ M, int64_t ld, int R, const float* xl, float* sl, int ldsl) {
    typedef typename TT<TC>::frag frag;
    constexpr int PER = TT<TC>::PER, NLD = KC / (16 * PER);
    const int tx = opaque_tid();
    const int lane = tx & 63, wave = tx >> 6, g = lane >> 4, j = lane & 15;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(M);
    frag xf[NH][NLD];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < NLD; ++i) to_frag(xl + h * KC + PER * (j + 16 * i), xf[h][i]);
    const int nit = (R + 32 * UNR - 1) / (32 * UNR);
    frag mf[2][UNR][NLD];
    auto load = [&](int buf, int it) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            int r = (it * UNR + u) * 32 + wave * 4 + g; r = r < R ? r : R - 1;
            const uint32_t off = ((uint32_t)r * (uint32_t)ld + (uint32_t)(PER * j)) * (uint32_t)sizeof(TC);
#pragma unroll
            for (int i = 0; i < NLD; ++i) mf[buf][u][i] = bload<frag>(rs, off + 256 * i);
        }
    };
    auto step = [&](auto BUF, int it) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int r = (it * UNR + u) * 32 + wave * 4 + g;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < NLD; ++i) a = fdot(mf[buf][u][i], xf[h][i], a);
                a = row16_sum(a);
                if (j == 0 && r < R) sl[h * ldsl + r] = a;
            }
        }
    };
    load(0, 0);
    for (int it = 0; it < nit; it += 2) {
        load(1, it + 1 < nit ? it + 1 : nit - 1);
        __builtin_amdgcn_sched_barrier(0);
        step(std::integral_constant<int, 0>{}, it);
        __builtin_amdgcn_sched_barrier(0);
        load(0, it + 2 < nit ? it + 2 : nit - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < nit) step(std::integral_constant<int, 1>{}, it + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Column layout of the weighted row sums: a row of DC elements is covered by LPR = DC / PER lanes with one 16-byte fragment each;
// wider rows give a lane NF fragments, narrower ones put RPW rows side by side in one wave-wide load.
template <typename TC, int DC> struct RowLayout {
    static constexpr int PER = TT<TC>::PER;
    static constexpr int LPR = DC / PER;
    static constexpr int NF = LPR > 64 ? LPR / 64 : 1;
    static constexpr int RPW = LPR < 64 ? 64 / LPR : 1;
};

// out[h][c] += sum_r wt[r * ldw + h] * M[r * ld + c], c in [0, DC), h in [0, NH); out: f32 in LDS (zeroed by the caller, who also
// synchronises around the call); wt: f32 in LDS (row-major [R][ldw]).  Waves stride the rows (RB rows per wave and batch in flight,
// twice), lanes own columns, the waves meet through LDS float atomics.  A row whose weights are all zero (a masked key) may hold
// anything, NaN included: it is not multiplied.
template <typename TC, int DC, int NH, int RB>
__device__ __forceinline__ void wsum(const TC* __restrict__ M, int64_t ld, int R, const float* wt, int ldw, float* outl, int ldo) {
    typedef typename TT<TC>::frag frag;
    typedef RowLayout<TC, DC> RL;
    constexpr int PER = RL::PER, LPR = RL::LPR, NF = RL::NF, RPW = RL::RPW;
    const int tx = opaque_tid();
    const int lane = tx & 63, wave = tx >> 6;
    const int sub = RPW > 1 ? lane / LPR : 0, lc = RPW > 1 ? lane % LPR : lane;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(M);
    float acc[NH][NF * PER];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int e = 0; e < NF * PER; ++e) acc[h][e] = 0.f;
    const int step_r = NW * RPW;                                     // rows the workgroup covers per row slot
    const int nit = (R + step_r * RB - 1) / (step_r * RB);
    frag mf[2][RB][NF];
    auto load = [&](int buf, int it) __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            int r = (it * RB + rb) * step_r + wave * RPW + sub; r = r < R ? r : R - 1;
#pragma unroll
            for (int f = 0; f < NF; ++f) mf[buf][rb][f] = bload<frag>(rs, ((uint32_t)r * (uint32_t)ld + (uint32_t)(PER * lc)) * (uint32_t)sizeof(TC) + 1024 * f);
        }
    };
    auto step = [&](auto BUF, int it) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = (it * RB + rb) * step_r + wave * RPW + sub;
            const int rc = r < R ? r : R - 1;                        // (unconditional loads: a conditional one costs a branch each)
            float w[NH];
            bool live = false;
            if constexpr (NH % 4 == 0) {
#pragma unroll
                for (int h = 0; h < NH; h += 4) {
                    const f32x4 t = *(const f32x4*)(wt + rc * ldw + h);    // ldw % 4 == 0 on this path
                    w[h] = t[0]; w[h + 1] = t[1]; w[h + 2] = t[2]; w[h + 3] = t[3];
                }
            } else {
#pragma unroll
                for (int h = 0; h < NH; ++h) w[h] = wt[rc * ldw + h];
            }
#pragma unroll
            for (int h = 0; h < NH; ++h) { w[h] = r < R ? w[h] : 0.f; live = live || w[h] != 0.f; }
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    float m = frag_get(mf[buf][rb][f], e);
                    m = live ? m : 0.f;
#pragma unroll
                    for (int h = 0; h < NH; ++h) acc[h][f * PER + e] = __builtin_fmaf(w[h], m, acc[h][f * PER + e]);
                }
        }
    };
    load(0, 0);                                                      // (R >= 1; rows past the end are clamped inside load)
    for (int it = 0; it < nit; it += 2) {
        load(1, it + 1 < nit ? it + 1 : nit - 1);
        __builtin_amdgcn_sched_barrier(0);
        step(std::integral_constant<int, 0>{}, it);
        __builtin_amdgcn_sched_barrier(0);
        load(0, it + 2 < nit ? it + 2 : nit - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < nit) step(std::integral_constant<int, 1>{}, it + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < PER; ++e) atomicAdd(outl + h * ldo + PER * (lc + 64 * f) + e, acc[h][f * PER + e]);
}

// Per-head fold  out[h][c] = sum_j w[h * hd + j] * M[(h * hd + j) * ld + c]  (c in [0, DC), 8 heads): wave h owns head h -- its hd
// rows are contiguous -- so no sums cross waves.  w: f32 in LDS; out: f32 in LDS, row pitch ldo (stored, not accumulated).
template <typename TC, int DC, int RB>
__device__ __forceinline__ void headfold(const TC* __restrict__ M, int64_t ld, int hd, const float* w, float* outl, int ldo) {
    typedef typename TT<TC>::frag frag;
    typedef RowLayout<TC, DC> RL;
    constexpr int PER = RL::PER, LPR = RL::LPR, NF = RL::NF, RPW = RL::RPW;
    const int tx = opaque_tid();
    const int lane = tx & 63, h = tx >> 6;
    const int sub = RPW > 1 ? lane / LPR : 0, lc = RPW > 1 ? lane % LPR : lane;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(M);
    float acc[NF * PER];
#pragma unroll
    for (int e = 0; e < NF * PER; ++e) acc[e] = 0.f;
    const int nit = hd / (RB * RPW);                                  // hd is a multiple of RB * RPW (checked by the host)
    const uint32_t hbase = (uint32_t)(h * hd) * (uint32_t)ld;
    frag mf[2][RB][NF];
    auto load = [&](int buf, int it) __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = (it * RB + rb) * RPW + sub;
#pragma unroll
            for (int f = 0; f < NF; ++f) mf[buf][rb][f] = bload<frag>(rs, (hbase + (uint32_t)r * (uint32_t)ld + (uint32_t)(PER * lc)) * (uint32_t)sizeof(TC) + 1024 * f);
        }
    };
    auto step = [&](auto BUF, int it) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const float wj = w[h * hd + (it * RB + rb) * RPW + sub];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int e = 0; e < PER; ++e) acc[f * PER + e] = __builtin_fmaf(wj, frag_get(mf[buf][rb][f], e), acc[f * PER + e]);
        }
    };
    load(0, 0);
    for (int it = 0; it < nit; it += 2) {
        load(1, it + 1 < nit ? it + 1 : nit - 1);
        __builtin_amdgcn_sched_barrier(0);
        step(std::integral_constant<int, 0>{}, it);
        __builtin_amdgcn_sched_barrier(0);
        load(0, it + 2 < nit ? it + 2 : nit - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < nit) step(std::integral_constant<int, 1>{}, it + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (RPW > 1) {                                                    // RPW == 2: the two half-waves hold the even / odd rows' sums
#pragma unroll
        for (int e = 0; e < NF * PER; ++e) acc[e] += __shfl_xor(acc[e], 32);
    }
    if (sub == 0) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < PER; ++e) outl[h * ldo + PER * (lc + 64 * f) + e] = acc[f * PER + e];
    }
}

__device__ __forceinline__ float drop_apply(float v, uint64_t seed, uint32_t site, uint32_t thr, float sc, uint64_t idx) {
    return (made_rng_mix(seed, site, idx) >> 8) >= thr ? v * sc : 0.f;
}

template <typename TC> __device__ __forceinline__ float rnd(float v) { return to_f32(from_f32<TC>(v)); }

// LayerNorm of the D values x (LDS) -> y[n] returned for n = tid (tid < D), statistics over the workgroup
template <int D>
__device__ __forceinline__ float layernorm_row(const float* xl, const float* __restrict__ g, const float* __restrict__ b, float eps, float* red) {
    const int n = threadIdx.x;
    const float v = n < D ? xl[n] : 0.f;
    const float mean = block_sum(v, red) * (1.f / D);
    const float d = n < D ? v - mean : 0.f;
    const float var = block_sum(d * d, red) * (1.f / D);
    const float rstd = 1.0f / sqrtf(var + eps);
    return n < D ? d * rstd * gload<float>(g + n) + gload<float>(b + n) : 0.f;
}

template <typename TC, int D>
__global__ __launch_bounds__(NT) void dec_train_fwd_kernel(const MadeDecTrainArgs a) {
    constexpr int MAXF = 2048;                                       // widest FFN hidden layer kept in LDS
    constexpr int NLD = D / (16 * TT<TC>::PER);                      // 16-byte loads per lane and row
    constexpr int MVU = 16 / NLD;                                    // rows per lane in flight: 16 loads, twice (double buffer)
    constexpr int NHS = sizeof(TC) == 2 ? 4 : 2;                     // heads per pass over the keys
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xt = lds;                    // [D]      the layer's input / running content query
    float* va = xt + D;                 // [MAXF]   mat-vec input
    float* vb = va + MAXF;              // [MAXF]   mat-vec raw output
    float* vc = vb + MAXF;              // [D]      residual carrier
    float* qp_l = vc + D;               // [H][D]   q' of the cross-attention
    float* pool_l = qp_l + a.H * D;     // [H][D]
    float* red = pool_l + a.H * D;      // [64]
    float* sc_l = red + 64;             // [H][Lp]  scores, then P^T as [L][H]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int H = a.H, hd = D / H, L = a.L, Fd = a.Fd, B = a.B;
    const int Lp = (L + 3) & ~3;
    const uint64_t seed = a.drop.p > 0.f ? made_drop_seed(a.drop) : 0;
    const uint32_t thr = made_drop_threshold(a.drop.p);
    const float dsc = a.drop.p > 0.f ? 1.f / (1.f - a.drop.p) : 1.f;
    const bool dropping = a.drop.p > 0.f;
    TC* const tgt_s = (TC*)a.tgt; TC* const qkv_s = (TC*)a.qkv; TC* const att_s = (TC*)a.att; TC* const ta_s = (TC*)a.t_a;
    TC* const t1_s = (TC*)a.t1; TC* const t1q_s = (TC*)a.t1q; TC* const qc_s = (TC*)a.qc; TC* const pooled_s = (TC*)a.pooled;
    TC* const attc_s = (TC*)a.attc; TC* const tb_s = (TC*)a.t_b; TC* const t2_s = (TC*)a.t2; TC* const h_s = (TC*)a.h;
    TC* const tc_s = (TC*)a.t_c; TC* const hs_s = (TC*)a.hs; TC* const gq_s = (TC*)a.GQ;
    const TC* const qpos = (const TC*)a.query_pos;
    const TC* const mem = (const TC*)a.mem + (int64_t)b * L * D;
    const TC* const mempos = (const TC*)a.mempos + (int64_t)b * L * D;
    const float* const kmask = a.key_mask ? a.key_mask + (int64_t)b * L : nullptr;
    const int64_t BD = (int64_t)B * D;

    int n_stamp = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (a.stamps && b == 0 && tid == 0) a.stamps[n_stamp++] = __builtin_readcyclecounter();
    };
    if (tid < D) xt[tid] = to_f32(tgt_s[(int64_t)b * D + tid]);
    __syncthreads();
    stamp();

    for (int l = 0; l < a.n_layers; ++l) {
        const MadeDecTrainLayer& ly = a.layers[l];
        // ---- self-attention with one query = its value path: v = W_v tgt + b_v, one attention-weight draw per head
        matvec<TC, D, MVU>((const TC*)ly.sa_v_w, D, D, 1, xt, vb);
        __syncthreads();
        if (tid < D) {
            const int n = tid, h = n / hd;
            const float v = rnd<TC>(vb[n] + gload<float>(ly.sa_v_b + n));
            qkv_s[((int64_t)l * B + b) * 3 * D + 2 * D + n] = from_f32<TC>(v);
            float g = v;
            if (dropping) g = drop_apply(v, seed, ly.site_sa_attn, thr, dsc, (uint64_t)b * H + h);
            g = rnd<TC>(g);
            att_s[(l * BD) + (int64_t)b * D + n] = from_f32<TC>(g);
            va[n] = g;
        }
        __syncthreads();
        stamp();   // 1: sa_v done
        matvec<TC, D, MVU>((const TC*)ly.sa_out_w, D, D, 1, va, vb);
        __syncthreads();
        if (tid < D) {
            const int n = tid;
            float v = vb[n] + gload<float>(ly.sa_out_b + n);
            if (dropping) v = drop_apply(v, seed, ly.site_drop1, thr, dsc, (uint64_t)b * D + n);
            v = rnd<TC>(v + xt[n]);
            ta_s[l * BD + (int64_t)b * D + n] = from_f32<TC>(v);
            vc[n] = v;
        }
        __syncthreads();
        stamp();   // 2: sa_out done
        {
            const float y = layernorm_row<D>(vc, ly.ln1_g, ly.ln1_b, a.eps, red);
            if (tid < D) {
                const float t1 = rnd<TC>(y);
                const float t1q = rnd<TC>(y + to_f32(qpos[tid]));
                t1_s[l * BD + (int64_t)b * D + tid] = from_f32<TC>(t1);
                t1q_s[l * BD + (int64_t)b * D + tid] = from_f32<TC>(t1q);
                va[tid] = t1q;
                vc[tid] = t1;                                        // residual of the cross-attention block
            }
        }
        __syncthreads();
        stamp();   // 3: ln1 done
        // ---- cross-attention in memory space
        matvec<TC, D, MVU>((const TC*)ly.ca_q_w, D, D, 1, va, vb);
        for (int i = tid; i < H * D; i += NT) pool_l[i] = 0.f;
        __syncthreads();
        if (tid < D) {
            const float v = rnd<TC>(vb[tid] + gload<float>(ly.ca_q_b + tid));
            qc_s[l * BD + (int64_t)b * D + tid] = from_f32<TC>(v);
            va[tid] = v;
        }
        __syncthreads();
        stamp();   // 4: ca_q done
        // q'_h = W_k,h^T qc_h: rows (h, j) of W_k weighted by qc[(h, j)]
        headfold<TC, D, 8>((const TC*)ly.ca_k_w, D, hd, va, qp_l, D);
        __syncthreads();
        for (int i = tid; i < H * D; i += NT) {
            const float v = rnd<TC>(qp_l[i]);
            qp_l[i] = v;
            gq_s[(((int64_t)b * 2 + 1) * a.n_layers + l) * H * D + i] = from_f32<TC>(v);
        }
        __syncthreads();
        stamp();   // 5: q' fold done
        // scores; the softmax of head h is wave h's
        for (int h0 = 0; h0 < H; h0 += NHS) multidot<TC, D, NHS, 2>(mempos, D, L, qp_l + h0 * D, sc_l + h0 * Lp, Lp);
        __syncthreads();
        stamp();   // 6: scores done
        for (int h = tid >> 6; h < H; h += NW) {
            const int lane = tid & 63;
            float* srow = sc_l + h * Lp;
            float mx = -INFINITY;
            for (int k = lane; k < L; k += 64) {
                float s = srow[k] * a.scale;
                if (kmask && kmask[k] == 0.f) s = -INFINITY;
                srow[k] = s;
                mx = fmaxf(mx, s);
            }
            mx = wave_max(mx);
            const float muse = mx == -INFINITY ? 0.f : mx;
            float sum = 0.f, dsum = 0.f;
            for (int k = lane; k < L; k += 64) {
                float p = expf(srow[k] - muse);
                sum += p;
                if (dropping) p = drop_apply(p, seed, ly.site_ca_attn, thr, dsc, ((uint64_t)b * H + h) * (uint64_t)L + k);
                dsum += p;
                srow[k] = p;
            }
            sum = wave_sum(sum); dsum = wave_sum(dsum);
            const float inv = sum > 0.f ? 1.f / sum : 0.f;
            for (int k = lane; k < L; k += 64) srow[k] *= inv;
            if (lane == 0) {
                const float sh = dropping ? dsum * inv : 1.f;
                red[32 + h] = sh;
                a.s_sum[((int64_t)l * B + b) * H + h] = sh;
            }
        }
        __syncthreads();
        stamp();   // 7: softmax done
        // pooled_h = P_h memory (masked keys carry weight 0 but may hold anything: their rows are skipped by a zero weight only if
        // finite -- the encoder leaves padded rows finite, see made_layernorm_add)
        {
            // transpose P to [L][H] in place is not possible: use the tail of the score buffer
            float* pt = sc_l + H * Lp;
            for (int i = tid; i < L * H; i += NT) { const int k = i / H, h = i - k * H; pt[i] = sc_l[h * Lp + k]; }
            __syncthreads();
            wsum<TC, D, 8, 8>(mem, D, L, pt, H, pool_l, D);
        }
        __syncthreads();
        for (int i = tid; i < H * D; i += NT) {
            const float v = rnd<TC>(pool_l[i]);
            pool_l[i] = v;
            pooled_s[((int64_t)l * B + b) * H * D + i] = from_f32<TC>(v);
        }
        __syncthreads();
        stamp();   // 8: pooled done
        // v_h = W_v,h pooled_h + s_h b_v,h
        matvec<TC, D, MVU>((const TC*)ly.ca_v_w, D, D, 1, pool_l, vb, hd);
        __syncthreads();
        if (tid < D) {
            const int n = tid, h = n / hd;
            const float v = rnd<TC>(vb[n] + red[32 + h] * gload<float>(ly.ca_v_b + n));
            attc_s[l * BD + (int64_t)b * D + n] = from_f32<TC>(v);
            va[n] = v;
        }
        __syncthreads();
        stamp();   // 9: v-proj done
        matvec<TC, D, MVU>((const TC*)ly.ca_out_w, D, D, 1, va, vb);
        __syncthreads();
        if (tid < D) {
            const int n = tid;
            float v = vb[n] + gload<float>(ly.ca_out_b + n);
            if (dropping) v = drop_apply(v, seed, ly.site_drop2, thr, dsc, (uint64_t)b * D + n);
            v = rnd<TC>(v + vc[n]);
            tb_s[l * BD + (int64_t)b * D + n] = from_f32<TC>(v);
            xt[n] = v;
        }
        __syncthreads();
        stamp();   // 10: ca_out done
        {
            const float y = layernorm_row<D>(xt, ly.ln2_g, ly.ln2_b, a.eps, red);
            if (tid < D) {
                const float t2 = rnd<TC>(y);
                t2_s[l * BD + (int64_t)b * D + tid] = from_f32<TC>(t2);
                va[tid] = t2;
                vc[tid] = t2;
            }
        }
        __syncthreads();
        stamp();   // 11: ln2 done
        // ---- feed-forward
        matvec<TC, D, MVU>((const TC*)ly.ff1_w, D, Fd, 1, va, vb);
        __syncthreads();
        for (int n = tid; n < Fd; n += NT) {
            float v = fmaxf(vb[n] + gload<float>(ly.ff1_b + n), 0.f);
            if (dropping) v = drop_apply(v, seed, ly.site_ffn_act, thr, dsc, (uint64_t)b * Fd + n);
            v = rnd<TC>(v);
            h_s[((int64_t)l * B + b) * Fd + n] = from_f32<TC>(v);
            va[n] = v;
        }
        __syncthreads();
        stamp();   // 12: ff1 done
        matvec<TC, D, MVU>((const TC*)ly.ff2_w, Fd, D, Fd / D, va, vb);
        __syncthreads();
        if (tid < D) {
            const int n = tid;
            float v = vb[n] + gload<float>(ly.ff2_b + n);
            if (dropping) v = drop_apply(v, seed, ly.site_drop3, thr, dsc, (uint64_t)b * D + n);
            v = rnd<TC>(v + vc[n]);
            tc_s[l * BD + (int64_t)b * D + n] = from_f32<TC>(v);
            vc[n] = v;
        }
        __syncthreads();
        stamp();   // 13: ff2 done
        {
            const float y = layernorm_row<D>(vc, ly.ln3_g, ly.ln3_b, a.eps, red);
            if (tid < D) {
                const float t3 = rnd<TC>(y);
                tgt_s[(l + 1) * BD + (int64_t)b * D + tid] = from_f32<TC>(t3);
                xt[tid] = t3;
            }
        }
        __syncthreads();
        {
            const float y = layernorm_row<D>(xt, a.norm_g, a.norm_b, a.eps, red);
            if (tid < D) hs_s[l * BD + (int64_t)b * D + tid] = from_f32<TC>(y);
        }
        __syncthreads();
        stamp();   // 14: ln3 + output norm done
    }
}

size_t fwd_lds_bytes(int D, int H, int L) {
    const int Lp = (L + 3) & ~3;
    return sizeof(float) * ((size_t)D + 2048 + 2048 + D + 2 * (size_t)H * D + 64 + (size_t)H * Lp + (size_t)L * H);
}

}  // namespace

extern "C" int made_dec_train_fwd(const MadeDecTrainArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_dec_train_fwd: null args");
    const MadeDecTrainArgs& a = *args;
    MADE_REQUIRE(a.layers && a.tgt && a.qkv && a.att && a.t_a && a.t1 && a.t1q && a.qc && a.pooled && a.attc && a.t_b && a.t2 && a.h && a.t_c &&
                 a.hs && a.GQ && a.s_sum && a.mem && a.mempos && a.query_pos && a.norm_g && a.norm_b, "made_dec_train_fwd: null pointer");
    MADE_REQUIRE(a.dtype == MADE_F32 || a.dtype == MADE_BF16, "made_dec_train_fwd: dtype must be f32 or bf16");
    MADE_REQUIRE(a.drop.p >= 0.f && a.drop.p < 1.f, "made_dec_train_fwd: dropout p out of [0,1)");
    MADE_UNSUPPORTED((a.D == 256 || a.D == 512) && a.H == 8 && a.Fd % a.D == 0 && a.Fd <= 2048 && a.L >= 1 && a.n_layers >= 1,
                     "made_dec_train_fwd: D=%d H=%d Fd=%d unsupported (D in {256, 512}, 8 heads, Fd a multiple of D up to 2048)", a.D, a.H, a.Fd);
    if (a.B <= 0) return MADE_OK;
    const size_t lds = fwd_lds_bytes(a.D, a.H, a.L);
    MADE_UNSUPPORTED(lds <= 160 * 1024 - 512, "made_dec_train_fwd: L=%d needs %zu bytes of LDS", a.L, lds);
    hipStream_t st = (hipStream_t)stream;
#define MADE_DT_LAUNCH(TC_, D_)                                                                                         \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            if (hipFuncSetAttribute((const void*)dec_train_fwd_kernel<TC_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) { \
                made_set_error("made_dec_train_fwd: cannot reserve LDS"); return MADE_ERR_HIP; }                         \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((dec_train_fwd_kernel<TC_, D_>), dim3((unsigned)a.B), dim3(NT), lds, st, a);                 \
    } while (0)
    if (a.dtype == MADE_BF16) { if (a.D == 512) MADE_DT_LAUNCH(bf16_t, 512); else MADE_DT_LAUNCH(bf16_t, 256); }
    else { if (a.D == 512) MADE_DT_LAUNCH(float, 512); else MADE_DT_LAUNCH(float, 256); }
#undef MADE_DT_LAUNCH
    return made_check_launch("made_dec_train_fwd");
}
