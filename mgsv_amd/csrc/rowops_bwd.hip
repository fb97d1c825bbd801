// Row kernels of the training path (HBM-bound): backward of LayerNorm, of the masked mean + L2 normalisation that
// produces the clip-level vectors, of the row L2 normalisations, of the symmetric cross entropy, of the X-Pool tail
// (LayerNorm3 + cosine) and of the softmax inside the wide-head attention.  One wave per row, 16-byte accesses, wave
// shuffles for the row reductions; parameter gradients are accumulated in registers over the rows a wave owns and added
// to the f32 gradient buffers with one atomic per column per workgroup.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int RT = 256;                    // 4 waves per workgroup
constexpr int MAXV = 8;

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
__device__ __forceinline__ bool keep_at(uint64_t seed, uint32_t site, uint32_t thr, uint64_t idx) {
    return (made_rng_mix(seed, site, idx) >> 8) >= thr;
}

// combine per-wave column partials through LDS and add them to a global f32 vector.  Every workgroup ends with one atomic per
// column on the SAME D addresses, and same-address atomics serialise in L2: the kernels that flush are launched with few, large
// (16-wave) workgroups when there are many rows
template <int NV, int NW = 4>
__device__ __forceinline__ void flush_cols(float* __restrict__ dst, const f32x4* acc, int D, int lane, int wave, float* sm /* [NW][NV*256] */) {
    if (dst == nullptr) return;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[wave * (NV * 256) + (i * WAVE + lane) * 4 + j] = acc[i][j];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += NW * 64) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 4)
            s += (sm[w * (NV * 256) + c] + sm[(w + 1) * (NV * 256) + c]) + (sm[(w + 2) * (NV * 256) + c] + sm[(w + 3) * (NV * 256) + c]);
        unsafeAtomicAdd(dst + c, s);
    }
    __syncthreads();
}

// ---- element-wise gate: out = dropout(x * act'(G) * scale) ------------------------------------------
// (the same options a made_linear epilogue carries, for the places of a backward chain where there is no GEMM to carry them)
__device__ __forceinline__ float gate_grad(float g, int gate) {
    switch (gate) {
        case MADE_GATE_RELU_OUT: return g != 0.f ? 1.f : 0.f;
        case MADE_GATE_GELU_Z: {
            const float cdf = 0.5f * (1.f + erff(g * 0.70710678118654752440f));
            return cdf + g * 0.39894228040143267794f * expf(-0.5f * g * g);
        }
        case MADE_GATE_QUICKGELU_Z: {
            const float sg = 1.f / (1.f + expf(-1.702f * g));
            return sg * (1.f + 1.702f * g * (1.f - sg));
        }
        case MADE_GATE_SIGMOID_OUT: return g * (1.f - g);
        default: return 1.f;
    }
}

__global__ __launch_bounds__(RT) void gate_rows_kernel(const void* x, int xdt, int64_t ldx, const void* G, int gdt, int64_t ldg, int gate,
                                                       float scale, MadeDropout drop, int64_t drop_ld, int drop_col_div, void* out, int odt,
                                                       int64_t ldo, const float* row_skip, int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (RT / 64) + (threadIdx.x >> 6);
    if (row >= rows || (row_skip && row_skip[row] == 0.f)) return;
    const uint32_t thr = made_drop_threshold(drop.p);
    const uint64_t drop_seed = drop.p > 0.f ? made_drop_seed(drop) : 0;
    const float dsc = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    for (int c = lane * 4; c < cols; c += 256) {
        f32x4 v = ld4(x, xdt, row * ldx + c);
        if (G) {
            const f32x4 g = ld4(G, gdt, row * ldg + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= gate_grad(g[j], gate);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= scale;
        if (drop.p > 0.f) {
            const uint64_t base = (uint64_t)row * (uint64_t)drop_ld;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = keep_at(drop_seed, drop.site, thr, base + (uint64_t)((c + j) / drop_col_div)) ? v[j] * dsc : 0.f;
        }
        st4(out, odt, row * ldo + c, v);
    }
}

// ---- LayerNorm backward ---------------------------------------------------------------------------
//   xhat = (x - mean) * rstd,  g = dy * gamma
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ add];   dgamma += sum_rows dy * xhat;   dbeta += sum_rows dy
struct LnBwdArgs {
    const void* x; int xdt; int64_t ldx, rpb, xbs;
    const float* gamma;
    const void* dy; int dydt; int64_t lddy;
    const void* add; int adt; int64_t ldadd;
    void* dx; int dxdt; int64_t lddx;
    void* dxd; int64_t lddxd; MadeDropout drop; int64_t drop_ld;
    float* dgamma; float* dbeta;
    int64_t rows; int D; float eps; const float* row_skip;
};

template <int NV, int NW>
__global__ __launch_bounds__(NW * 64) void layernorm_bwd_kernel(const LnBwdArgs a) {
    __shared__ float sm[NW * NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; }
    const uint32_t thr = made_drop_threshold(a.drop.p);
    const uint64_t drop_seed = a.drop.p > 0.f ? made_drop_seed(a.drop) : 0;
    const float dsc = a.drop.p > 0.f ? 1.f / (1.f - a.drop.p) : 1.f;
    const int D = a.D;
    // A wave's rows are base + k * stride.  It looks 64 of them ahead (one row_skip load per lane, one ballot), then walks the valid
    // ones two at a time with the loads of both in flight: padded tokens cost nothing, and a row's latency is hidden by its pair.
    const int64_t stride = (int64_t)gridDim.x * NW;
    auto load_row = [&](int64_t row, f32x4 (&xv)[NV], f32x4 (&gy)[NV]) __attribute__((always_inline)) {
        const int64_t xoff = a.rpb > 0 ? (row / a.rpb) * a.xbs + (row % a.rpb) * a.ldx : row * a.ldx;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            const int cc = c < D ? c : 0;
            xv[i] = ld4(a.x, a.xdt, xoff + cc);
            gy[i] = ld4(a.dy, a.dydt, row * a.lddy + cc);
        }
    };
    auto process = [&](int64_t row, f32x4 (&xv)[NV], f32x4 (&gy)[NV]) __attribute__((always_inline)) {
        const bool skip = false;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c >= D) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { xv[i][j] = 0.f; gy[i][j] = 0.f; }
            }
            sum += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
        }
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = xv[i][j] - mean; sq += d * d; }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + a.eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < D) {
                const f32x4 gm = *(const f32x4*)(a.gamma + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xv[i][j] - mean) * rstd;
                    const float g = gy[i][j] * gm[j];
                    dg[i][j] += gy[i][j] * xh;
                    db[i][j] += gy[i][j];
                    xv[i][j] = xh;                 // keep xhat
                    gy[i][j] = g;                  // keep g
                    s1 += g; s2 += g * xh;
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < D) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = skip ? 0.f : rstd * (gy[i][j] - s1 - xv[i][j] * s2);
                if (a.add && !skip) {
                    const f32x4 ad = ld4(a.add, a.adt, row * a.ldadd + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += ad[j];
                }
                st4(a.dx, a.dxdt, row * a.lddx + c, o);
                if (a.dxd) {
                    if (a.drop.p > 0.f) {
                        const uint64_t base = (uint64_t)row * (uint64_t)a.drop_ld + (uint64_t)c;
                        const uint32_t kb = made_keep_bits<4>(drop_seed, a.drop.site, thr, base);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = ((kb >> j) & 1u) ? o[j] * dsc : 0.f;
                    }
                    st4(a.dxd, a.dxdt, row * a.lddxd + c, o);
                }
            }
        }
    };
    for (int64_t base = (int64_t)blockIdx.x * NW + wave; base < a.rows; base += 64 * stride) {
        const int64_t cand = base + (int64_t)lane * stride;
        const bool ok = cand < a.rows && (!a.row_skip || a.row_skip[cand] != 0.f);
        uint64_t todo = __ballot(ok);
        while (todo) {
            const int j0 = __builtin_ctzll(todo);
            todo &= todo - 1;
            const bool two = todo != 0;
            const int j1 = two ? __builtin_ctzll(todo) : j0;
            if (two) todo &= todo - 1;
            const int64_t r0 = base + (int64_t)j0 * stride, r1 = base + (int64_t)j1 * stride;
            f32x4 xa[NV], ga[NV], xb[NV], gb[NV];
            load_row(r0, xa, ga);
            load_row(r1, xb, gb);
            process(r0, xa, ga);
            if (two) process(r1, xb, gb);
        }
    }
    flush_cols<NV, NW>(a.dgamma, dg, D, lane, wave, sm);
    flush_cols<NV, NW>(a.dbeta, db, D, lane, wave, sm);
}

// ---- LayerNorm backward, the wide-access form for the step's large launches: bf16 rows of 512 columns -------------------------
// A lane owns 8 CONSECUTIVE columns, so every tensor of a row moves as one 16-byte access per lane (1 KB per wave instruction;
// the general kernel above moves 8 bytes per lane and needs two per tensor), and a wave keeps FOUR rows in flight: the loads of
// all four are requested before the first reduction, and the four rows' dependent reduction chains interleave.
// (Round 6 dealt the valid rows out evenly instead -- a workgroup compacts its ~74 valid rows of 136 into an LDS list, every wave gets the same number
// to within one and keeps two or three rows in flight: the body drops from 19 - 26 to 16 - 21 us, but then all 256 workgroups reach the
// parameter-gradient flush together and its 256 same-address atomics per cache line, which the ragged finish of this kernel hides under its
// stragglers, cost 5.7 us instead of 1.3: 21.4 / 21.9 / 26.1 us against 20.3 / 20.6 / 27.4 for the three forms the step launches.  Not kept:
// profiles/r06_lnbwd_even_rows.txt, tools/lnbwd_variants.py.)
template <int NW, int RF>
__global__ __launch_bounds__(NW * 64) void layernorm_bwd_v8_kernel(const LnBwdArgs a) {
    constexpr int D = 512;
    __shared__ float sm[NW * D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = lane * 8;
    float dg[8], db[8], gm[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[j] = 0.f; db[j] = 0.f; }
    {
        const f32x4 g0 = *(const f32x4*)(a.gamma + c0), g1 = *(const f32x4*)(a.gamma + c0 + 4);
        gm[0] = g0[0]; gm[1] = g0[1]; gm[2] = g0[2]; gm[3] = g0[3]; gm[4] = g1[0]; gm[5] = g1[1]; gm[6] = g1[2]; gm[7] = g1[3];
    }
    const uint32_t thr = made_drop_threshold(a.drop.p);
    const uint64_t drop_seed = a.drop.p > 0.f ? made_drop_seed(a.drop) : 0;
    const float dsc = a.drop.p > 0.f ? 1.f / (1.f - a.drop.p) : 1.f;
    const bf16_t* X = (const bf16_t*)a.x; const bf16_t* DY = (const bf16_t*)a.dy; const bf16_t* AD = (const bf16_t*)a.add;
    bf16_t* DX = (bf16_t*)a.dx; bf16_t* DXD = (bf16_t*)a.dxd;
    const int64_t stride = (int64_t)gridDim.x * NW;
    for (int64_t base = (int64_t)blockIdx.x * NW + wave; base < a.rows; base += 64 * stride) {
        const int64_t cand = base + (int64_t)lane * stride;
        const bool ok = cand < a.rows && (!a.row_skip || a.row_skip[cand] != 0.f);
        uint64_t todo = __ballot(ok);
        while (todo) {
            int64_t rw[RF]; bool on[RF];
#pragma unroll
            for (int k = 0; k < RF; ++k) {
                on[k] = todo != 0;
                const int j = on[k] ? __builtin_ctzll(todo) : 0;
                if (on[k]) todo &= todo - 1;
                rw[k] = on[k] ? base + (int64_t)j * stride : base;          // (an idle slot re-reads a valid row and stores nothing)
            }
            bf16x8 xr[RF], gr[RF], ar[RF];
#pragma unroll
            for (int k = 0; k < RF; ++k) {
                xr[k] = *(const bf16x8*)(X + rw[k] * a.ldx + c0);
                gr[k] = *(const bf16x8*)(DY + rw[k] * a.lddy + c0);
                if (AD) ar[k] = *(const bf16x8*)(AD + rw[k] * a.ldadd + c0);
            }
            // the dropout draws need nothing of the rows: made while the loads are in flight
            uint32_t kbits[RF];
#pragma unroll
            for (int k = 0; k < RF; ++k) {
                kbits[k] = 0xFFu;
                if (a.drop.p > 0.f) {
                    const uint64_t dbase = (uint64_t)rw[k] * (uint64_t)a.drop_ld + (uint64_t)c0;
                    kbits[k] = made_keep_bits<8>(drop_seed, a.drop.site, thr, dbase);
                }
            }
#pragma unroll
            for (int k = 0; k < RF; ++k) asm volatile("" : "+v"(kbits[k]));      // (keep the draws in front of the first use of the loads)
            // all four row sums (x, x^2, g, g x; g = dy * gamma) in ONE round of independent wave reductions per row
            float xv[RF][8], gy[RF][8], mean[RF], rstd[RF], s1[RF], s2[RF], t0[RF], t1[RF], t2[RF], t3[RF];
#pragma unroll
            for (int k = 0; k < RF; ++k) {
                float sx = 0.f, sxx = 0.f, sg = 0.f, sgx = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xv[k][j] = (float)xr[k][j]; gy[k][j] = (float)gr[k][j];
                    const float g = gy[k][j] * gm[j];
                    sx += xv[k][j]; sxx += xv[k][j] * xv[k][j]; sg += g; sgx += g * xv[k][j];
                }
                t0[k] = sx; t1[k] = sxx; t2[k] = sg; t3[k] = sgx;
            }
#pragma unroll
            for (int k = 0; k < RF; ++k) { t0[k] = wave_sum(t0[k]); t1[k] = wave_sum(t1[k]); t2[k] = wave_sum(t2[k]); t3[k] = wave_sum(t3[k]); }
#pragma unroll
            for (int k = 0; k < RF; ++k) {
                mean[k] = t0[k] * (1.f / D);
                const float var = fmaxf(t1[k] * (1.f / D) - mean[k] * mean[k], 0.f);
                rstd[k] = 1.0f / sqrtf(var + a.eps);
                s1[k] = t2[k] * (1.f / D);
                s2[k] = rstd[k] * (t3[k] - mean[k] * t2[k]) * (1.f / D);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (xv[k][j] - mean[k]) * rstd[k];
                    if (on[k]) { dg[j] += gy[k][j] * xh; db[j] += gy[k][j]; }
                    xv[k][j] = xh; gy[k][j] *= gm[j];
                }
            }
#pragma unroll
            for (int k = 0; k < RF; ++k) {
                if (!on[k]) continue;
                bf16x8 o, od;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = rstd[k] * (gy[k][j] - s1[k] - xv[k][j] * s2[k]);
                    if (AD) v += (float)ar[k][j];
                    o[j] = (bf16_t)v;
                    v = ((kbits[k] >> j) & 1u) ? v * dsc : 0.f;
                    od[j] = (bf16_t)v;
                }
                *(bf16x8*)(DX + rw[k] * a.lddx + c0) = o;
                if (DXD) *(bf16x8*)(DXD + rw[k] * a.lddxd + c0) = od;
            }
        }
    }
    // parameter gradients: the waves' column partials meet in LDS, one atomic per column and workgroup
    for (int pass = 0; pass < 2; ++pass) {
        float* dst = pass == 0 ? a.dgamma : a.dbeta;
        if (dst == nullptr) continue;
        const float* src = pass == 0 ? dg : db;
        f32x4 p0, p1;
        p0[0] = src[0]; p0[1] = src[1]; p0[2] = src[2]; p0[3] = src[3]; p1[0] = src[4]; p1[1] = src[5]; p1[2] = src[6]; p1[3] = src[7];
        *(f32x4*)(sm + wave * D + c0) = p0; *(f32x4*)(sm + wave * D + c0 + 4) = p1;
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += NW * 64) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += sm[w * D + c];
            unsafeAtomicAdd(dst + c, t);
        }
        __syncthreads();
    }
}

// ---- two chained LayerNorms backward in one pass (the decoder's per-layer norm 3 followed by the shared output norm) -----------
//   forward: t3 = LN_a(x_a) (+ nothing),  hs = LN_b(t3)      x_b = the saved t3
//   g  = LN_b'(dy; x_b, gamma_b) + add        (add: the gradient arriving at t3 from the next layer)
//   dx = LN_a'(g;  x_a, gamma_a);  dx_drop = dropout(dx);  parameter gradients of both norms accumulated
// One wave per row, every row of the launch independent: a 64-row link of the decoder's backward chain instead of two.
struct LnBwd2Args {
    const void* xa; const float* gamma_a; const void* xb; const float* gamma_b;
    const void* dy; const void* add; void* dx; void* dxd;
    int64_t ldxa, ldxb, lddy, ldadd, lddx, lddxd;
    int dt; MadeDropout drop; int64_t drop_ld;
    float *dgamma_a, *dbeta_a, *dgamma_b, *dbeta_b;
    int64_t rows; int D; float eps;
};

template <int NV>
__device__ __forceinline__ void ln_bwd_row(f32x4 (&xv)[NV], f32x4 (&gy)[NV], const float* gamma, int D, int lane, float eps,
                                           f32x4 (&dg)[NV], f32x4 (&db)[NV], f32x4 (&o)[NV]) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if ((i * WAVE + lane) * 4 >= D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { xv[i][j] = 0.f; gy[i][j] = 0.f; }
        }
        sum += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if ((i * WAVE + lane) * 4 < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = xv[i][j] - mean; sq += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * WAVE + lane) * 4;
        if (c < D) {
            const f32x4 gm = *(const f32x4*)(gamma + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xv[i][j] - mean) * rstd;
                const float g = gy[i][j] * gm[j];
                dg[i][j] += gy[i][j] * xh;
                db[i][j] += gy[i][j];
                xv[i][j] = xh;
                gy[i][j] = g;
                s1 += g; s2 += g * xh;
            }
        }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[i][j] = rstd * (gy[i][j] - s1 - xv[i][j] * s2);
}

template <int NV>
__global__ __launch_bounds__(RT) void layernorm_bwd2_kernel(const LnBwd2Args a) {
    __shared__ float sm[4 * NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = a.D;
    f32x4 dga[NV], dba[NV], dgb[NV], dbb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { dga[i][j] = 0.f; dba[i][j] = 0.f; dgb[i][j] = 0.f; dbb[i][j] = 0.f; }
    const uint32_t thr = made_drop_threshold(a.drop.p);
    const uint64_t drop_seed = a.drop.p > 0.f ? made_drop_seed(a.drop) : 0;
    const float dsc = a.drop.p > 0.f ? 1.f / (1.f - a.drop.p) : 1.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < a.rows; row += (int64_t)gridDim.x * 4) {
        f32x4 xa[NV], xb[NV], gy[NV], ad[NV], g[NV], o[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {                          // every load of the row before the first reduction
            const int c = (i * WAVE + lane) * 4;
            const int cc = c < D ? c : 0;
            xb[i] = ld4(a.xb, a.dt, row * a.ldxb + cc);
            gy[i] = ld4(a.dy, a.dt, row * a.lddy + cc);
            xa[i] = ld4(a.xa, a.dt, row * a.ldxa + cc);
            if (a.add) ad[i] = ld4(a.add, a.dt, row * a.ldadd + cc);
            else { ad[i][0] = ad[i][1] = ad[i][2] = ad[i][3] = 0.f; }
        }
        ln_bwd_row<NV>(xb, gy, a.gamma_b, D, lane, a.eps, dgb, dbb, g);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) g[i][j] += ad[i][j];
        ln_bwd_row<NV>(xa, g, a.gamma_a, D, lane, a.eps, dga, dba, o);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < D) {
                st4(a.dx, a.dt, row * a.lddx + c, o[i]);
                if (a.dxd) {
                    f32x4 od = o[i];
                    if (a.drop.p > 0.f) {
                        const uint64_t base = (uint64_t)row * (uint64_t)a.drop_ld + (uint64_t)c;
#pragma unroll
                        for (int j = 0; j < 4; ++j) od[j] = keep_at(drop_seed, a.drop.site, thr, base + j) ? od[j] * dsc : 0.f;
                    }
                    st4(a.dxd, a.dt, row * a.lddxd + c, od);
                }
            }
        }
    }
    flush_cols<NV, 4>(a.dgamma_a, dga, D, lane, wave, sm);
    flush_cols<NV, 4>(a.dbeta_a, dba, D, lane, wave, sm);
    flush_cols<NV, 4>(a.dgamma_b, dgb, D, lane, wave, sm);
    flush_cols<NV, 4>(a.dbeta_b, dbb, D, lane, wave, sm);
}

// ---- clip-level vector backward: vec = normalize(masked mean(local)) --------------------------------
//   dmean = (dvec - vhat (vhat . dvec)) / max(|mean|, eps);  out[b,t,:] = mask[b,t] * (in1 + in2 + dmean / count_b)
struct PoolBwdArgs {
    const float* mean; const float* dvec; const float* mask;
    const void* in1; int in1dt; int64_t in1_bs, in1_ld;
    const void* in2; int in2dt; int64_t in2_bs, in2_ld;
    void* out; int odt; int64_t out_bs, out_ld;
    int64_t B, T; int D; float eps;
};

// One wave takes PB_RPW consecutive tokens of ONE sample: the sample's constants (token count, |mean|, vhat . dvec: a strided pass over
// the mask, two row loads and three dependent wave reductions) are made once per wave, not once per token as in the first version
// (one token per wave: 152 us for 32768 x 512 in the step, a tenth of the HBM rate), and padded tokens cost a store of zeros, no loads.
constexpr int PB_RPW = 8;
template <int NV>
__global__ __launch_bounds__(RT) void pool_bwd_kernel(const PoolBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tiles = (a.T + 4 * PB_RPW - 1) / (4 * PB_RPW);
    const int64_t b = blockIdx.x / tiles;
    const int64_t t0 = (blockIdx.x % tiles) * (4 * PB_RPW) + wave * PB_RPW;
    if (b >= a.B || t0 >= a.T) return;
    const int D = a.D;
    float mk[PB_RPW];
#pragma unroll
    for (int r = 0; r < PB_RPW; ++r) { const int64_t t = t0 + r < a.T ? t0 + r : a.T - 1; mk[r] = a.mask[b * a.T + t]; }
    float cnt = 0.f;
    for (int64_t j = lane; j < a.T; j += 64) cnt += a.mask[b * a.T + j];
    cnt = wave_sum(cnt);
    f32x4 mv[NV], gv[NV];
    float nn = 0.f, dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * WAVE + lane) * 4;
        const int cc = c < D ? c : 0;
        mv[i] = *(const f32x4*)(a.mean + b * D + cc);
        gv[i] = *(const f32x4*)(a.dvec + b * D + cc);
        if (c >= D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { mv[i][j] = 0.f; gv[i][j] = 0.f; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { nn += mv[i][j] * mv[i][j]; dot += mv[i][j] * gv[i][j]; }
    }
    const float nrm = fmaxf(sqrtf(wave_sum(nn)), a.eps);
    dot = wave_sum(dot) / (nrm * nrm);                       // (vhat . dvec) / nrm
    const float inv = 1.f / (nrm * cnt);
    f32x4 dm[NV];                                            // dmean / count of this sample, this lane's columns
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) dm[i][j] = (gv[i][j] - mv[i][j] * dot) * inv;
#pragma unroll
    for (int r = 0; r < PB_RPW; ++r) {
        const int64_t t = t0 + r;
        if (t >= a.T) break;
        const bool valid = mk[r] != 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < D) {
                f32x4 o = dm[i];
                if (valid) {
                    if (a.in1) {
                        const f32x4 x = ld4(a.in1, a.in1dt, b * a.in1_bs + t * a.in1_ld + c);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] += x[j];
                    }
                    if (a.in2) {
                        const f32x4 x = ld4(a.in2, a.in2dt, b * a.in2_bs + t * a.in2_ld + c);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] += x[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = 0.f;
                }
                st4(a.out, a.odt, b * a.out_bs + t * a.out_ld + c, o);
            }
        }
    }
}

// ---- y = x / max(|x|, eps)  ->  dx = (dy - yhat (yhat . dy)) / max(|x|, eps) -------------------------------
struct L2BwdArgs {
    const void* x; int xdt; int64_t ldx;
    const float* dy; int64_t lddy; int64_t dy_rows_per;
    float* dx; int64_t lddx; int accumulate;
    void* dxa; int adt; int64_t lddxa;
    int64_t rows; int D; float eps;
};

template <int NV>
__global__ __launch_bounds__(RT) void l2norm_bwd_kernel(const L2BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int D = a.D;
    f32x4 xv[NV], gv[NV];
    float nn = 0.f, dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * WAVE + lane) * 4;
        const int cc = c < D ? c : 0;
        xv[i] = ld4(a.x, a.xdt, row * a.ldx + cc);
        gv[i] = *(const f32x4*)(a.dy + (row / a.dy_rows_per) * a.lddy + cc);
        if (c >= D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { xv[i][j] = 0.f; gv[i][j] = 0.f; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { nn += xv[i][j] * xv[i][j]; dot += xv[i][j] * gv[i][j]; }
    }
    const float nrm = fmaxf(sqrtf(wave_sum(nn)), a.eps);
    dot = wave_sum(dot) / (nrm * nrm);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * WAVE + lane) * 4;
        if (c < D) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (gv[i][j] - xv[i][j] * dot) / nrm;
            if (a.dxa) st4(a.dxa, a.adt, row * a.lddxa + c, o);
            if (a.dx) {
                float* p = a.dx + row * a.lddx + c;
                if (a.accumulate) {
                    const f32x4 old = *(const f32x4*)p;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += old[j];
                }
                *(f32x4*)p = o;
            }
        }
    }
}

// ---- symmetric cross entropy backward ---------------------------------------------------------------------
// z = sims * exp(logit_scale); L = w/2 (mean_i (lse_row_i - z_ii) + mean_j (lse_col_j - z_jj))
//   dz_ij = w/(2n) (exp(z_ij - lse_row_i) + exp(z_ij - lse_col_j) - 2 [i == j]);  dsims = dz * e^ls;  dls = sum dz z
__global__ __launch_bounds__(RT) void clip_lse_kernel(const float* sims, int64_t ld, int n, const float* logit_scale, float* lse,
                                                      const float* row_exclude) {
    const int lane = threadIdx.x & 63;
    const int line = blockIdx.x * 4 + (threadIdx.x >> 6);         // 0..n-1 rows, n..2n-1 columns
    if (line >= 2 * n) return;
    const float gsc = expf(logit_scale[0]);
    const bool col = line >= n;
    const int i = col ? line - n : line;
    const float* ex = (!col && row_exclude) ? row_exclude + (int64_t)i * n : nullptr;     // same-track negatives leave the row softmax
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) {
        const float z = (col ? sims[(int64_t)j * ld + i] : sims[(int64_t)i * ld + j]) * gsc;
        mx = fmaxf(mx, (ex && ex[j] != 0.f) ? -INFINITY : z);
    }
    mx = wave_max(mx);
    float se = 0.f;
    for (int j = lane; j < n; j += 64) {
        const float z = (col ? sims[(int64_t)j * ld + i] : sims[(int64_t)i * ld + j]) * gsc;
        se += (ex && ex[j] != 0.f) ? 0.f : expf(z - mx);
    }
    se = wave_sum(se);
    if (lane == 0) lse[line] = mx + logf(se);
}

__global__ __launch_bounds__(RT) void clip_bwd_kernel(const float* sims, int64_t ld, int n, const float* logit_scale, float weight,
                                                      const float* upstream, const float* lse, float* dsims, float* dsims_t,
                                                      int accumulate, float* dls, const float* row_exclude) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float gsc = expf(logit_scale[0]);
    const float w = weight * (upstream ? upstream[0] : 1.f) / (2.f * (float)n);
    float acc = 0.f;
    for (int j = lane; j < n; j += 64) {
        const float z = sims[(int64_t)i * ld + j] * gsc;
        const float prow = (row_exclude && row_exclude[(int64_t)i * n + j] != 0.f) ? 0.f : expf(z - lse[i]);
        const float dz = w * (prow + expf(z - lse[n + j]) - (i == j ? 2.f : 0.f));
        acc += dz * z;
        const float g = dz * gsc;
        if (accumulate) {
            dsims[(int64_t)i * n + j] += g;
            if (dsims_t) dsims_t[(int64_t)j * n + i] += g;
        } else {
            dsims[(int64_t)i * n + j] = g;
            if (dsims_t) dsims_t[(int64_t)j * n + i] = g;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0 && dls) unsafeAtomicAdd(dls, acc);
}

// ---- X-Pool tail backward: sims[n,m] = cos(video_n, LayerNorm3(y[m,n])) -------------------------------------------
struct XtailBwdArgs {
    const void* y; int ydt; int64_t ldy;
    const float* gamma; const float* beta; const float* video; int64_t ldv;
    const float* dsims; int64_t ldds;
    void* dy; int dydt; int64_t lddy;
    void* dyd; MadeDropout drop;
    float* dgamma; float* dbeta; float* dvideo; int64_t lddv;
    const float* dpool; int64_t lddp; float dpool_scale;      // extra gradient of the pooled rows: dpool[m, :] * scale for every n
    int64_t rows, Nv; int D; float eps;
};

template <int NV, int NW>
__global__ __launch_bounds__(NW * 64) void xpool_tail_bwd_kernel(const XtailBwdArgs a) {
    __shared__ float sm[NW * NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = a.D;
    f32x4 dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; }
    const uint32_t thr = made_drop_threshold(a.drop.p);
    const uint64_t drop_seed = a.drop.p > 0.f ? made_drop_seed(a.drop) : 0;
    const float dsc = a.drop.p > 0.f ? 1.f / (1.f - a.drop.p) : 1.f;
    for (int64_t row = (int64_t)blockIdx.x * NW + wave; row < a.rows; row += (int64_t)gridDim.x * NW) {
        const int64_t m = row / a.Nv, n = row % a.Nv;
        f32x4 xh[NV], pv[NV], vd[NV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            const int cc = c < D ? c : 0;
            xh[i] = ld4(a.y, a.ydt, row * a.ldy + cc);
            vd[i] = *(const f32x4*)(a.video + n * a.ldv + cc);
            if (c >= D) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { xh[i][j] = 0.f; vd[i][j] = 0.f; }
            }
            sum += (xh[i][0] + xh[i][1]) + (xh[i][2] + xh[i][3]);
        }
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = xh[i][j] - mean; sq += d * d; }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + a.eps);
        float pp = 0.f, vv = 0.f, pdv = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < D) {
                const f32x4 gm = *(const f32x4*)(a.gamma + c), bt = *(const f32x4*)(a.beta + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[i][j] = (xh[i][j] - mean) * rstd;
                    pv[i][j] = xh[i][j] * gm[j] + bt[j];
                    pp += pv[i][j] * pv[i][j]; vv += vd[i][j] * vd[i][j]; pdv += pv[i][j] * vd[i][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) pv[i][j] = 0.f;
            }
        }
        const float np = sqrtf(wave_sum(pp)), nvn = sqrtf(wave_sum(vv));
        const float sim = wave_sum(pdv) / (np * nvn);
        const float ds = a.dsims[n * a.ldds + m];
        // dp = ds * (vhat - sim * phat) / |p| ;  dvideo_n += ds * (phat - sim * vhat) / |v|
        float s1 = 0.f, s2 = 0.f;
        f32x4 g[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < D) {
                const f32x4 gm = *(const f32x4*)(a.gamma + c);
                f32x4 dvd, dpx = {0.f, 0.f, 0.f, 0.f};
                if (a.dpool) dpx = *(const f32x4*)(a.dpool + m * a.lddp + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ph = pv[i][j] / np, vh = vd[i][j] / nvn;
                    const float dp = ds * (vh - sim * ph) / np + dpx[j] * a.dpool_scale;
                    dvd[j] = ds * (ph - sim * vh) / nvn;
                    dg[i][j] += dp * xh[i][j];
                    db[i][j] += dp;
                    g[i][j] = dp * gm[j];
                    s1 += g[i][j]; s2 += g[i][j] * xh[i][j];
                }
                if (a.dvideo) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) unsafeAtomicAdd(a.dvideo + n * a.lddv + c + j, dvd[j]);
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < D) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rstd * (g[i][j] - s1 - xh[i][j] * s2);
                st4(a.dy, a.dydt, row * a.lddy + c, o);
                if (a.dyd) {
                    if (a.drop.p > 0.f) {
                        const uint64_t base = (uint64_t)row * (uint64_t)D + (uint64_t)c;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = keep_at(drop_seed, a.drop.site, thr, base + j) ? o[j] * dsc : 0.f;
                    }
                    st4(a.dyd, a.dydt, row * a.lddy + c, o);
                }
            }
        }
    }
    flush_cols<NV, NW>(a.dgamma, dg, D, lane, wave, sm);
    flush_cols<NV, NW>(a.dbeta, db, D, lane, wave, sm);
}

// ---- softmax backward of the wide-head attention (scores materialised: [rows, L] with few rows per batch) ---------
//   P = softmax(scale * S + mask);  Pd = dropout(P);  dP = dropout'(dPd + extra);  dS = scale * P * (dP - sum_k P_k dP_k)
// Outputs in the compute dtype: Pd [rows, ldo], dS [rows, ldo] and dS^T [batch, L, ldt] (rows of one batch along ldt).
// Columns L .. ldo-1 of Pd / dS are written as zeros (the TN products read whole 16-byte chunks).
struct SmBwdArgs {
    const float* S; int64_t lds_; const float* dP; int64_t lddp;
    const float* mask; int64_t rows_per_mask;            // mask row = row / rows_per_mask
    const float* extra; float scale; MadeDropout drop;
    void* Pd; void* dS; void* dSt; int odt; int64_t ldo, ldt, obs, tbs;
    int64_t rows, rpb, L;
};

__global__ __launch_bounds__(RT) void softmax_bwd_kernel(const SmBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const float* s = a.S + row * a.lds_;
    const float* g = a.dP + row * a.lddp;
    const float* mk = a.mask ? a.mask + (row / a.rows_per_mask) * a.L : nullptr;
    const float ex = a.extra ? a.extra[row] : 0.f;
    const uint32_t thr = made_drop_threshold(a.drop.p);
    const uint64_t drop_seed = a.drop.p > 0.f ? made_drop_seed(a.drop) : 0;
    const float dsc = a.drop.p > 0.f ? 1.f / (1.f - a.drop.p) : 1.f;
    const uint64_t base = (uint64_t)row * (uint64_t)a.L;
    const int64_t z = row / a.rpb, i = row % a.rpb;
    constexpr int SM_MAXC = 17;                              // rows of up to 1088 keys live in registers: ONE pass over S / dP / mask
    if (a.L <= 64 * SM_MAXC) {
        // (the three-pass form below reads S, dP and the mask with conditional 4-byte loads three times over: ~20 us for a 542-key row
        // of the decoder's memory-space attention, all of it latency)
        const int L = (int)a.L, nch = (L + 63) >> 6;
        float sv[SM_MAXC], gv[SM_MAXC];
        uint32_t okbits = 0;
#pragma unroll
        for (int c = 0; c < SM_MAXC; ++c) {
            if (c < nch) {
                const int k = c * 64 + lane, kc = k < L ? k : L - 1;          // unconditional loads from a clamped index
                sv[c] = s[kc]; gv[c] = g[kc];
                const float m = mk ? mk[kc] : 1.f;
                okbits |= (k < L && m != 0.f) ? (1u << c) : 0u;
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < SM_MAXC; ++c)
            if (c < nch) mx = fmaxf(mx, (okbits >> c) & 1u ? sv[c] * a.scale : -INFINITY);
        mx = wave_max(mx);
        float se = 0.f, dot = 0.f;
#pragma unroll
        for (int c = 0; c < SM_MAXC; ++c) {
            if (c < nch) {
                const bool ok = (okbits >> c) & 1u;
                const float p = ok ? expf(sv[c] * a.scale - mx) : 0.f;
                float dp = ok ? gv[c] + ex : 0.f;              // masked keys may hold stale (non-finite) products: never touch them
                bool kp = true;
                if (a.drop.p > 0.f) { kp = keep_at(drop_seed, a.drop.site, thr, base + (uint64_t)(c * 64 + lane)); dp = kp ? dp * dsc : 0.f; }
                se += p; dot += p * dp;
                sv[c] = p; gv[c] = dp;
                okbits = kp ? okbits : (okbits & ~(1u << c));  // (bit c now: the key is valid AND kept)
            }
        }
        se = wave_sum(se); dot = wave_sum(dot);
        const float inv = 1.f / se;
        dot *= inv;
#pragma unroll
        for (int c = 0; c < SM_MAXC; ++c) {
            const int k = c * 64 + lane;
            if (k < a.ldo) {
                float pd = 0.f, ds = 0.f;
                if (c < nch && k < L) {
                    const float p = sv[c] * inv;
                    pd = (okbits >> c) & 1u ? p * dsc : 0.f;  // dropped (or masked: p == 0) keys carry no weight
                    if (!(a.drop.p > 0.f)) pd = p;
                    ds = a.scale * p * (gv[c] - dot);
                }
                store_from_f32(a.Pd, a.odt, z * a.obs + i * a.ldo + k, pd);
                store_from_f32(a.dS, a.odt, z * a.obs + i * a.ldo + k, ds);
                if (a.dSt && k < L) store_from_f32(a.dSt, a.odt, z * a.tbs + (int64_t)k * a.ldt + i, ds);
            }
        }
        return;
    }
    float mx = -INFINITY;
    for (int64_t k = lane; k < a.L; k += 64) {
        const bool ok = mk == nullptr || mk[k] != 0.f;
        mx = fmaxf(mx, ok ? s[k] * a.scale : -INFINITY);
    }
    mx = wave_max(mx);
    float se = 0.f, dot = 0.f;
    for (int64_t k = lane; k < a.L; k += 64) {
        const bool ok = mk == nullptr || mk[k] != 0.f;
        const float p = ok ? expf(s[k] * a.scale - mx) : 0.f;
        float dp = ok ? g[k] + ex : 0.f;              // masked keys may hold stale (non-finite) products: never touch them
        if (a.drop.p > 0.f) dp = keep_at(drop_seed, a.drop.site, thr, base + (uint64_t)k) ? dp * dsc : 0.f;
        se += p; dot += p * dp;
    }
    se = wave_sum(se); dot = wave_sum(dot);
    const float inv = 1.f / se;
    dot *= inv;
    for (int64_t k = lane; k < a.ldo; k += 64) {
        float pd = 0.f, ds = 0.f;
        if (k < a.L) {
            const bool ok = mk == nullptr || mk[k] != 0.f;
            const float p = ok ? expf(s[k] * a.scale - mx) * inv : 0.f;
            float dp = ok ? g[k] + ex : 0.f;
            pd = p;
            if (a.drop.p > 0.f) {
                const bool kp = keep_at(drop_seed, a.drop.site, thr, base + (uint64_t)k);
                dp = kp ? dp * dsc : 0.f;
                pd = kp ? p * dsc : 0.f;
            }
            ds = a.scale * p * (dp - dot);
        }
        store_from_f32(a.Pd, a.odt, z * a.obs + i * a.ldo + k, pd);
        store_from_f32(a.dS, a.odt, z * a.obs + i * a.ldo + k, ds);
        if (a.dSt && k < a.L) store_from_f32(a.dSt, a.odt, z * a.tbs + k * a.ldt + i, ds);
    }
}

// ---- out[row, h*hd + j] (+)= s[row, h] * bias[h*hd + j]  (the value bias of the memory-space cross-attention under dropout)
__global__ void head_bias_kernel(void* x, int xdt, int64_t ldx, const float* s, const float* bias, int64_t rows, int H, int hd) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * H * hd) return;
    const int64_t row = idx / (H * hd);
    const int c = (int)(idx % (H * hd));
    const float v = load_as_f32(x, xdt, row * ldx + c) + s[row * H + c / hd] * bias[c];
    store_from_f32(x, xdt, row * ldx + c, v);
}
// backward: dbias[c] += sum_rows s[row,h] * dy[row,c];  ds[row,h] = sum_j dy[row, h*hd+j] * bias[h*hd+j]
__global__ __launch_bounds__(RT) void head_bias_bwd_kernel(const void* dy, int dt, int64_t ld, const float* s, const float* bias,
                                                           float* dbias, float* ds, int64_t rows, int H, int hd) {
    const int lane = threadIdx.x & 63;
    const int64_t rh = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);         // (row, h)
    if (rh >= rows * H) return;
    const int64_t row = rh / H;
    const int h = (int)(rh % H);
    float acc = 0.f;
    for (int j = lane; j < hd; j += 64) {
        const float g = load_as_f32(dy, dt, row * ld + h * hd + j);
        acc += g * bias[h * hd + j];
        unsafeAtomicAdd(dbias + h * hd + j, g * s[rh]);
    }
    acc = wave_sum(acc);
    if (lane == 0) ds[rh] = acc;
}

// ---- out = a + b (+ c), any mix of f32 / bf16, contiguous [n] ------------------------------------------------------
__global__ void add3_kernel(void* out, int odt, const void* a, int adt, const void* b, int bdt, const void* c, int cdt, int64_t n,
                            int64_t b_mod) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = load_as_f32(a, adt, i);
    if (b) v += load_as_f32(b, bdt, b_mod > 0 ? i % b_mod : i);
    if (c) v += load_as_f32(c, cdt, i);
    store_from_f32(out, odt, i, v);
}

// ---- out[c] += sum_rows x[row, c]  (f32 accumulate; gradient of a vector broadcast over rows) -------------------------
__global__ __launch_bounds__(RT) void colsum_kernel(const void* x, int dt, int64_t ld, int64_t rows, int64_t cols, float* out) {
    const int64_t c = (int64_t)blockIdx.x * RT + threadIdx.x;
    if (c >= cols) return;
    const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) acc += load_as_f32(x, dt, r * ld + c);
    unsafeAtomicAdd(out + c, acc);
}

inline unsigned blocks4(int64_t rows) { return (unsigned)((rows + 3) / 4); }
inline int nv_of(int64_t D) { return D <= 512 ? 2 : (D <= 1024 ? 4 : 8); }

#define DISPATCH_NVB(D, CALL)                                       \
    switch (nv_of(D)) {                                             \
        case 2: { constexpr int NV = 2; CALL; } break;              \
        case 4: { constexpr int NV = 4; CALL; } break;              \
        default: { constexpr int NV = 8; CALL; } break;             \
    }

}  // namespace

extern "C" int made_layernorm_bwd(const void* x, int32_t x_dtype, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride,
                                  const float* gamma, const void* dy, int32_t dy_dtype, int64_t lddy,
                                  const void* add, int32_t add_dtype, int64_t ld_add,
                                  void* dx, int32_t dx_dtype, int64_t lddx,
                                  void* dx_drop, int64_t lddxd, const MadeDropout* drop, int64_t drop_ld,
                                  float* dgamma, float* dbeta, int64_t rows, int64_t D, float eps, const float* row_skip,
                                  void* stream) {
    MADE_REQUIRE(x && gamma && dy && dx, "made_layernorm_bwd: null pointer");
    MADE_UNSUPPORTED(D > 0 && D % 4 == 0 && D <= 64 * 4 * MAXV, "made_layernorm_bwd: D=%lld must be a multiple of 4 and <= %d", (long long)D, 64 * 4 * MAXV);
    MADE_UNSUPPORTED(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ld_add % 4 == 0 && lddxd % 4 == 0 && x_batch_stride % 4 == 0,
                     "made_layernorm_bwd: row strides must be multiples of 4");
    if (rows <= 0) return MADE_OK;
    LnBwdArgs a;
    a.x = x; a.xdt = x_dtype; a.ldx = ldx; a.rpb = x_rows_per_batch; a.xbs = x_batch_stride;
    a.gamma = gamma; a.dy = dy; a.dydt = dy_dtype; a.lddy = lddy;
    a.add = add; a.adt = add_dtype; a.ldadd = ld_add;
    a.dx = dx; a.dxdt = dx_dtype; a.lddx = lddx;
    a.dxd = dx_drop; a.lddxd = lddxd;
    a.drop.seed = 0; a.drop.site = 0; a.drop.p = 0.f; a.drop.seed_device = nullptr;
    if (drop) a.drop = *drop;
    a.drop_ld = drop_ld > 0 ? drop_ld : D;
    a.dgamma = dgamma; a.dbeta = dbeta; a.rows = rows; a.D = (int)D; a.eps = eps; a.row_skip = row_skip;
    const bool v8 = rows > 256 && D == 512 && x_rows_per_batch == 0 && x_dtype == MADE_BF16 && dy_dtype == MADE_BF16 && dx_dtype == MADE_BF16 &&
                    (add == nullptr || add_dtype == MADE_BF16) && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && ld_add % 8 == 0 && lddxd % 8 == 0 &&
                    ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)add % 16) == 0 &&
                    ((uintptr_t)dx_drop % 16) == 0 && made_variant_env("MADE_LNBWD_V8_OFF") == nullptr;
    if (v8) {                                                   // bf16 rows of 512: 16-byte accesses, four rows in flight per wave
        // measured (tools/r03_micro.py, 18 979 valid rows, dropout + flush): 16-wave workgroups with two rows in flight per wave, one per CU:
        // 29.4 us; 8-wave workgroups with four rows in flight: 31.5 (256) / 34.0 us (512 workgroups); the 8-byte kernel: 36.9 us
        static const int rf = [] { const char* e = made_variant_env("MADE_LNBWD_RF"); return e ? atoi(e) : 2; }();          // knobs for measurements
        static const int nb_cap8 = [] { const char* e = made_variant_env("MADE_LNBWD_NB"); return e ? atoi(e) : 256; }();
        if (rf == 2) {
            int64_t nb = (rows + 15) / 16;
            if (nb > nb_cap8) nb = nb_cap8;
            hipLaunchKernelGGL((layernorm_bwd_v8_kernel<16, 2>), dim3((unsigned)nb), dim3(1024), 0, (hipStream_t)stream, a);
        } else {
            int64_t nb = (rows + 7) / 8;
            if (nb > nb_cap8) nb = nb_cap8;
            hipLaunchKernelGGL((layernorm_bwd_v8_kernel<8, 4>), dim3((unsigned)nb), dim3(512), 0, (hipStream_t)stream, a);
        }
    } else if (rows > 256 && D <= 1024) {                       // each workgroup flushes 2*D same-address atomics: few, large workgroups
        int64_t nb = (rows + 15) / 16;
        static const int nb_cap = [] { const char* e = made_variant_env("MADE_LNBWD_NB"); return e ? atoi(e) : 256; }();   // one 16-wave workgroup per CU (7.18 vs 7.27 ms per training step against 512); knob for measurements
        if (nb > nb_cap) nb = nb_cap;
        // (D > 512: eight waves per workgroup -- with sixteen the 128-register budget of four waves per SIMD spilled 132 bytes)
        if (nv_of(D) <= 2) hipLaunchKernelGGL((layernorm_bwd_kernel<2, 16>), dim3((unsigned)nb), dim3(1024), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((layernorm_bwd_kernel<4, 8>), dim3((unsigned)nb), dim3(512), 0, (hipStream_t)stream, a);
    } else {
        int64_t nb = (rows + 3) / 4;
        if (nb > 1024) nb = 1024;
        DISPATCH_NVB(D, hipLaunchKernelGGL((layernorm_bwd_kernel<NV, 4>), dim3((unsigned)nb), dim3(RT), 0, (hipStream_t)stream, a));
    }
    return made_check_launch("made_layernorm_bwd");
}

extern "C" int made_layernorm_bwd2(const void* xa, const float* gamma_a, int64_t ldxa, const void* xb, const float* gamma_b, int64_t ldxb,
                                   const void* dy, int64_t lddy, const void* add, int64_t ld_add, void* dx, int64_t lddx,
                                   void* dx_drop, int64_t lddxd, const MadeDropout* drop, int64_t drop_ld, int32_t dtype,
                                   float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b, int64_t rows, int64_t D, float eps,
                                   void* stream) {
    MADE_REQUIRE(xa && gamma_a && xb && gamma_b && dy && dx, "made_layernorm_bwd2: null pointer");
    MADE_REQUIRE(dtype == MADE_F32 || dtype == MADE_BF16, "made_layernorm_bwd2: bad dtype %d", dtype);
    MADE_UNSUPPORTED(D > 0 && D % 4 == 0 && D <= 64 * 4 * 4, "made_layernorm_bwd2: D=%lld must be a multiple of 4 and <= 1024", (long long)D);
    MADE_UNSUPPORTED(ldxa % 4 == 0 && ldxb % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ld_add % 4 == 0 && lddxd % 4 == 0,
                     "made_layernorm_bwd2: row strides must be multiples of 4");
    if (rows <= 0) return MADE_OK;
    LnBwd2Args a;
    a.xa = xa; a.gamma_a = gamma_a; a.xb = xb; a.gamma_b = gamma_b; a.dy = dy; a.add = add; a.dx = dx; a.dxd = dx_drop;
    a.ldxa = ldxa; a.ldxb = ldxb; a.lddy = lddy; a.ldadd = ld_add; a.lddx = lddx; a.lddxd = lddxd;
    a.dt = dtype;
    a.drop.seed = 0; a.drop.site = 0; a.drop.p = 0.f; a.drop.seed_device = nullptr;
    if (drop) a.drop = *drop;
    a.drop_ld = drop_ld > 0 ? drop_ld : D;
    a.dgamma_a = dgamma_a; a.dbeta_a = dbeta_a; a.dgamma_b = dgamma_b; a.dbeta_b = dbeta_b;
    a.rows = rows; a.D = (int)D; a.eps = eps;
    int64_t nb = (rows + 3) / 4;
    if (nb > 1024) nb = 1024;
    switch ((D + 255) / 256) {
        case 1: hipLaunchKernelGGL((layernorm_bwd2_kernel<1>), dim3((unsigned)nb), dim3(RT), 0, (hipStream_t)stream, a); break;
        case 2: hipLaunchKernelGGL((layernorm_bwd2_kernel<2>), dim3((unsigned)nb), dim3(RT), 0, (hipStream_t)stream, a); break;
        default: hipLaunchKernelGGL((layernorm_bwd2_kernel<4>), dim3((unsigned)nb), dim3(RT), 0, (hipStream_t)stream, a); break;
    }
    return made_check_launch("made_layernorm_bwd2");
}

extern "C" int made_gate_rows(const void* x, int32_t x_dtype, int64_t ldx, const void* G, int32_t g_dtype, int64_t ldg, int32_t gate,
                              float scale, const MadeDropout* drop, int64_t drop_ld, int64_t drop_col_div, void* out, int32_t out_dtype,
                              int64_t ldo, const float* row_skip, int64_t rows, int64_t cols, void* stream) {
    MADE_REQUIRE(x && out, "made_gate_rows: null pointer");
    MADE_REQUIRE(gate == MADE_GATE_NONE || G != nullptr, "made_gate_rows: gate without G");
    MADE_UNSUPPORTED(cols > 0 && cols % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (G == nullptr || ldg % 4 == 0), "made_gate_rows: bad cols/strides");
    if (rows <= 0) return MADE_OK;
    MadeDropout d; d.seed = 0; d.site = 0; d.p = 0.f; d.seed_device = nullptr;
    if (drop) d = *drop;
    MADE_REQUIRE(d.p >= 0.f && d.p < 1.f, "made_gate_rows: dropout p out of [0,1)");
    hipLaunchKernelGGL(gate_rows_kernel, dim3(blocks4(rows)), dim3(RT), 0, (hipStream_t)stream, x, x_dtype, ldx, G, g_dtype, ldg, gate, scale,
                       d, drop_ld > 0 ? drop_ld : cols, (int)(drop_col_div > 0 ? drop_col_div : 1), out, out_dtype, ldo, row_skip, rows, (int)cols);
    return made_check_launch("made_gate_rows");
}

extern "C" int made_pool_bwd(const float* mean, const float* dvec, const float* mask,
                             const void* in1, int32_t in1_dtype, int64_t in1_bs, int64_t in1_ld,
                             const void* in2, int32_t in2_dtype, int64_t in2_bs, int64_t in2_ld,
                             void* out, int32_t out_dtype, int64_t out_bs, int64_t out_ld,
                             int64_t B, int64_t T, int64_t D, float eps, void* stream) {
    MADE_REQUIRE(mean && dvec && mask && out, "made_pool_bwd: null pointer");
    MADE_UNSUPPORTED(D > 0 && D % 4 == 0 && D <= 64 * 4 * MAXV && in1_ld % 4 == 0 && in2_ld % 4 == 0 && out_ld % 4 == 0 &&
                     in1_bs % 4 == 0 && in2_bs % 4 == 0 && out_bs % 4 == 0, "made_pool_bwd: bad D/strides");
    if (B * T <= 0) return MADE_OK;
    PoolBwdArgs a{mean, dvec, mask, in1, in1_dtype, in1_bs, in1_ld, in2, in2_dtype, in2_bs, in2_ld, out, out_dtype, out_bs, out_ld, B, T, (int)D, eps};
    const int64_t pb_tiles = (T + 4 * PB_RPW - 1) / (4 * PB_RPW);
    DISPATCH_NVB(D, hipLaunchKernelGGL((pool_bwd_kernel<NV>), dim3((unsigned)(B * pb_tiles)), dim3(RT), 0, (hipStream_t)stream, a));
    return made_check_launch("made_pool_bwd");
}

extern "C" int made_l2norm_bwd(const void* x, int32_t x_dtype, int64_t ldx, const float* dy, int64_t lddy, int64_t dy_rows_per,
                               float* dx, int64_t lddx, int32_t accumulate, void* dx_alt, int32_t alt_dtype, int64_t lddxa,
                               int64_t rows, int64_t D, float eps, void* stream) {
    MADE_REQUIRE(x && dy && (dx || dx_alt), "made_l2norm_bwd: null pointer");
    MADE_UNSUPPORTED(D > 0 && D % 4 == 0 && D <= 64 * 4 * MAXV && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && lddxa % 4 == 0,
                     "made_l2norm_bwd: bad D/strides");
    if (rows <= 0) return MADE_OK;
    L2BwdArgs a{x, x_dtype, ldx, dy, lddy, dy_rows_per > 0 ? dy_rows_per : 1, dx, lddx, accumulate, dx_alt, alt_dtype, lddxa, rows, (int)D, eps};
    DISPATCH_NVB(D, hipLaunchKernelGGL((l2norm_bwd_kernel<NV>), dim3(blocks4(rows)), dim3(RT), 0, (hipStream_t)stream, a));
    return made_check_launch("made_l2norm_bwd");
}

extern "C" int made_clip_loss_bwd(const float* sims, int64_t ld, int64_t n, const float* logit_scale, float weight,
                                  const float* upstream, float* lse_ws, float* dsims, float* dsims_t, int32_t accumulate,
                                  float* d_logit_scale, const float* row_exclude, void* stream) {
    MADE_REQUIRE(sims && logit_scale && lse_ws && dsims, "made_clip_loss_bwd: null pointer");
    MADE_REQUIRE(n > 0 && n <= (1 << 20) && ld >= n, "made_clip_loss_bwd: n=%lld out of range", (long long)n);
    hipLaunchKernelGGL(clip_lse_kernel, dim3(blocks4(2 * n)), dim3(RT), 0, (hipStream_t)stream, sims, ld, (int)n, logit_scale, lse_ws, row_exclude);
    hipLaunchKernelGGL(clip_bwd_kernel, dim3(blocks4(n)), dim3(RT), 0, (hipStream_t)stream, sims, ld, (int)n, logit_scale, weight,
                       upstream, lse_ws, dsims, dsims_t, accumulate, d_logit_scale, row_exclude);
    return made_check_launch("made_clip_loss_bwd");
}

extern "C" int made_xpool_tail_bwd(const void* y, int32_t y_dtype, int64_t ldy, const float* gamma, const float* beta,
                                   const float* video, int64_t ld_video, const float* dsims, int64_t ld_dsims,
                                   void* dy, int32_t dy_dtype, int64_t lddy, void* dy_drop, const MadeDropout* drop,
                                   float* dgamma, float* dbeta, float* dvideo, int64_t ld_dvideo,
                                   const float* dpool, int64_t ld_dpool, float dpool_scale,
                                   int64_t Nm, int64_t Nv, int64_t D, float eps, void* stream) {
    MADE_REQUIRE(y && gamma && beta && video && dsims && dy, "made_xpool_tail_bwd: null pointer");
    MADE_UNSUPPORTED(dpool == nullptr || (ld_dpool % 4 == 0 && ((uintptr_t)dpool % 16) == 0), "made_xpool_tail_bwd: dpool must keep 16-byte alignment");
    MADE_UNSUPPORTED(D > 0 && D % 4 == 0 && D <= 64 * 4 * MAXV && ldy % 4 == 0 && ld_video % 4 == 0 && lddy % 4 == 0,
                     "made_xpool_tail_bwd: bad D/strides");
    const int64_t rows = Nm * Nv;
    if (rows <= 0) return MADE_OK;
    XtailBwdArgs a;
    a.y = y; a.ydt = y_dtype; a.ldy = ldy; a.gamma = gamma; a.beta = beta; a.video = video; a.ldv = ld_video;
    a.dsims = dsims; a.ldds = ld_dsims; a.dy = dy; a.dydt = dy_dtype; a.lddy = lddy; a.dyd = dy_drop;
    a.drop.seed = 0; a.drop.site = 0; a.drop.p = 0.f; a.drop.seed_device = nullptr;
    if (drop) a.drop = *drop;
    a.dgamma = dgamma; a.dbeta = dbeta; a.dvideo = dvideo; a.lddv = ld_dvideo;
    a.dpool = dpool; a.lddp = ld_dpool; a.dpool_scale = dpool_scale;
    a.rows = rows; a.Nv = Nv; a.D = (int)D; a.eps = eps;
    if (rows > 256 && D <= 1024) {
        int64_t nb = (rows + 15) / 16;
        static const int nb_cap = [] { const char* e = made_variant_env("MADE_LNBWD_NB"); return e ? atoi(e) : 256; }();   // one 16-wave workgroup per CU (7.18 vs 7.27 ms per training step against 512); knob for measurements
        if (nb > nb_cap) nb = nb_cap;
        if (nv_of(D) <= 2) hipLaunchKernelGGL((xpool_tail_bwd_kernel<2, 16>), dim3((unsigned)nb), dim3(1024), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((xpool_tail_bwd_kernel<4, 8>), dim3((unsigned)nb), dim3(512), 0, (hipStream_t)stream, a);   // (as made_layernorm_bwd)
    } else {
        int64_t nb = (rows + 3) / 4;
        if (nb > 1024) nb = 1024;
        DISPATCH_NVB(D, hipLaunchKernelGGL((xpool_tail_bwd_kernel<NV, 4>), dim3((unsigned)nb), dim3(RT), 0, (hipStream_t)stream, a));
    }
    return made_check_launch("made_xpool_tail_bwd");
}

extern "C" int made_softmax_bwd(const float* S, int64_t ld_s, const float* dP, int64_t ld_dp, const float* mask,
                                int64_t rows_per_mask, const float* extra, float scale, const MadeDropout* drop,
                                void* Pd, void* dS, void* dSt, int32_t out_dtype, int64_t ldo, int64_t ldt,
                                int64_t out_batch_stride, int64_t t_batch_stride,
                                int64_t rows, int64_t rows_per_batch, int64_t L, void* stream) {
    MADE_REQUIRE(S && dP && Pd && dS, "made_softmax_bwd: null pointer");
    MADE_REQUIRE(rows >= 0 && L > 0 && ldo >= L && rows_per_batch > 0 && rows_per_mask > 0, "made_softmax_bwd: bad dims");
    MADE_REQUIRE(dSt == nullptr || ldt >= rows_per_batch, "made_softmax_bwd: ldt too small");
    if (rows == 0) return MADE_OK;
    SmBwdArgs a;
    a.S = S; a.lds_ = ld_s; a.dP = dP; a.lddp = ld_dp; a.mask = mask; a.rows_per_mask = rows_per_mask; a.extra = extra;
    a.scale = scale; a.drop.seed = 0; a.drop.site = 0; a.drop.p = 0.f; a.drop.seed_device = nullptr;
    if (drop) a.drop = *drop;
    a.Pd = Pd; a.dS = dS; a.dSt = dSt; a.odt = out_dtype; a.ldo = ldo; a.ldt = ldt; a.rows = rows; a.rpb = rows_per_batch; a.L = L;
    a.obs = out_batch_stride > 0 ? out_batch_stride : rows_per_batch * ldo;
    a.tbs = t_batch_stride > 0 ? t_batch_stride : L * ldt;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(blocks4(rows)), dim3(RT), 0, (hipStream_t)stream, a);
    return made_check_launch("made_softmax_bwd");
}

extern "C" int made_head_bias(void* x, int32_t x_dtype, int64_t ldx, const float* s, const float* bias,
                              int64_t rows, int64_t H, int64_t hd, void* stream) {
    MADE_REQUIRE(x && s && bias && H > 0 && hd > 0, "made_head_bias: bad arguments");
    const int64_t n = rows * H * hd;
    if (n <= 0) return MADE_OK;
    hipLaunchKernelGGL(head_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, ldx, s, bias, rows, (int)H, (int)hd);
    return made_check_launch("made_head_bias");
}

extern "C" int made_head_bias_bwd(const void* dy, int32_t dtype, int64_t ld, const float* s, const float* bias,
                                  float* dbias, float* ds, int64_t rows, int64_t H, int64_t hd, void* stream) {
    MADE_REQUIRE(dy && s && bias && dbias && ds && H > 0 && hd > 0, "made_head_bias_bwd: bad arguments");
    if (rows <= 0) return MADE_OK;
    hipLaunchKernelGGL(head_bias_bwd_kernel, dim3(blocks4(rows * H)), dim3(RT), 0, (hipStream_t)stream, dy, dtype, ld, s, bias, dbias, ds, rows, (int)H, (int)hd);
    return made_check_launch("made_head_bias_bwd");
}

extern "C" int made_add3(void* out, int32_t out_dtype, const void* a, int32_t a_dtype, const void* b, int32_t b_dtype,
                         const void* c, int32_t c_dtype, int64_t n, int64_t b_mod, void* stream) {
    MADE_REQUIRE(out && a, "made_add3: null pointer");
    if (n <= 0) return MADE_OK;
    hipLaunchKernelGGL(add3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, out_dtype, a, a_dtype, b, b_dtype, c, c_dtype, n, b_mod);
    return made_check_launch("made_add3");
}

extern "C" int made_colsum(const void* x, int32_t dtype, int64_t ld, int64_t rows, int64_t cols, float* out, void* stream) {
    MADE_REQUIRE(x && out && cols > 0, "made_colsum: bad arguments");
    if (rows <= 0) return MADE_OK;
    int64_t ny = (rows + 7) / 8;                           // few rows per thread: the loads of a column run in parallel
    if (ny > 512) ny = 512;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((cols + RT - 1) / RT), (unsigned)ny), dim3(RT), 0, (hipStream_t)stream, x, dtype, ld, rows, cols, out);
    return made_check_launch("made_colsum");
}
