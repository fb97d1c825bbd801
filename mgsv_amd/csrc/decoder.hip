// made_dec_stage: one stage of the moment-DETR decoder's chain of 64-row Linears (reference music_detr/transformer.py:273-307 with
// Q = 1: B*Q rows, every Linear depends on the one before it).
//
// Round 1 ran each such Linear as a split-K GEMM + a finish launch (bias / residual / LayerNorm), 12 launches per layer.  A
// LayerNorm needs whole rows, which a column-split GEMM does not have -- but the NEXT Linear's workgroups each read whole
// rows of their input anyway (K = D).  So the LayerNorm moves into the prologue of its consumer: a stage is
//     x  = LayerNorm(Zin)            raw f32 rows of the previous stage (or x = Zin when there is no norm)
//     x2 = LayerNorm2(x)             optional (the decoder's shared output norm -> hs[l - 1])
//     A  = x (+ add)                 bf16, e.g. + query_pos
//     out = act(A W^T + bias) (+ R | + x)   f32 raw rows for the next norm, or bf16
// computed by (N / 16) x (M / 16) workgroups of 16 rows x 16 columns; every workgroup normalises its 16 rows itself (under the
// flight of its own weight fragments), the rows r with r % gridDim.x == blockIdx.x are also written out as x / x2 (the residual of a
// later stage, the decoder output).  No split-K, no finish launch, no partial sums in HBM.  The four waves split K four ways (as
// linear_t16_kernel does, v_mfma_f32_16x16x32_bf16): weight fragments come straight from global memory (16 contiguous bytes of one
// row per lane), A fragments from the normalised LDS tile, partial tiles meet in LDS.
// Tile size (round 3): such a stage costs t = 3.4 us + (bytes ONE workgroup pulls in) / ~33 GB/s -- the arithmetic is nothing -- so
// the 64 x 32 tiles of round 2 (64 KB of rows + 32 KB of weights per workgroup at K = 512) became 16 x 16 (16 + 16 KB) with eight
// times as many workgroups; the column tiles that share a weight slice are blockIdx.x apart = on one XCD under the observed placement.
#include "common.h"
#include <type_traits>

namespace {

constexpr int DS_BM = 16, DS_BN = 16, DS_THREADS = 256, DS_RPW = 4, DS_CT_LD = DS_BN + 4;   // 4 waves, 4 rows each in the prologue

__device__ __forceinline__ float ds_act(float x, int act) {   // (ReLU is all the decoder uses; erf / exp code would double the kernel)
    return act == MADE_ACT_RELU ? fmaxf(x, 0.f) : x;
}

// ZB16: Zin holds bf16 rows (the training chain keeps its pre-norm rows in the compute dtype: they are also what the LayerNorm
// backward reads); TRAIN: the stateless dropout of include/made_hip.h after the activation, and the GEMM input A = x + add is
// also written out (a_out: the backward's weight-gradient operand).
template <int NV, bool ZB16, bool TRAIN>                        // K = 64 * NV (NV = 4: D = 256, NV = 8: D = 512)
__global__ __launch_bounds__(DS_THREADS) void dec_stage_kernel(const MadeDecStageArgs a) {
    constexpr int K = 64 * NV;
    constexpr int LDA = K * 2 + 16;                             // bytes per row of the LDS A tile (padded: conflict-free 16-byte reads)
    constexpr int STEPS = NV / 2;                               // 32-deep MFMA k-steps per wave (K / 4 / 32)
    __shared__ __attribute__((aligned(16))) unsigned char dlds[DS_BM * LDA];
    __shared__ __attribute__((aligned(16))) float Ct[4 * DS_BM * DS_CT_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int M = (int)a.M, N = (int)a.N;
    const int n0 = blockIdx.x * DS_BN, m0 = blockIdx.y * DS_BM;

    // What such a one-shot kernel pays for is dependent memory round trips (about 1 us each, and vmcnt counts STORES too: a load
    // issued after a store waits for the store to retire) and spills.  Hence: every global load is requested before the first
    // wait, every global store of the prologue comes after its last load.
    // ---- 1. this wave's weight fragments (a quarter of K), in flight before anything else
    const int kw = wave * (K / 4) + kq * 8;
    int gn = n0 + r16; gn = gn < N ? gn : N - 1;
    const bf16_t* pw = (const bf16_t*)a.W + (int64_t)gn * a.ldw + kw;
    bf16x8 fw[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) fw[s] = *(const bf16x8*)(pw + s * 32);

    // ---- 2. LayerNorm prologue: wave w normalises rows 4w .. 4w + 3, all 64 lanes on one row at a time (lane l: columns
    // NV*l .. NV*l + NV - 1): the norm parameters of a lane's columns are loaded once and shared by its rows, the rows' reductions
    // are independent chains the scheduler interleaves.
    constexpr int RPW = DS_RPW;
    const int c0 = NV * lane;
    const bool has_ln = a.ln_g != nullptr, has_ln2 = a.ln2_g != nullptr && a.x2_out != nullptr;
    float v[RPW][NV];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        int gm = m0 + wave * RPW + i; gm = gm < M ? gm : M - 1;
        if constexpr (ZB16) {
            const bf16_t* zp = (const bf16_t*)a.Zin + (int64_t)gm * a.ldz + c0;
            if constexpr (NV == 8) {
                const bf16x8 t = *(const bf16x8*)zp;
#pragma unroll
                for (int u = 0; u < 8; ++u) v[i][u] = (float)t[u];
            } else {
                const bf16x4 t = *(const bf16x4*)zp;
#pragma unroll
                for (int u = 0; u < 4; ++u) v[i][u] = (float)t[u];
            }
        } else {
            const float* zp = (const float*)a.Zin + (int64_t)gm * a.ldz + c0;
#pragma unroll
            for (int j = 0; j < NV; j += 4) {
                const f32x4 t = *(const f32x4*)(zp + j);
                v[i][j] = t[0]; v[i][j + 1] = t[1]; v[i][j + 2] = t[2]; v[i][j + 3] = t[3];
            }
        }
    }
    float g1[NV], b1[NV], g2[NV], b2[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { g1[j] = 1.f; b1[j] = 0.f; g2[j] = 1.f; b2[j] = 0.f; }
    if (has_ln) {
#pragma unroll
        for (int j = 0; j < NV; j += 4) {
            const f32x4 g = *(const f32x4*)(a.ln_g + c0 + j), b = *(const f32x4*)(a.ln_b + c0 + j);
#pragma unroll
            for (int u = 0; u < 4; ++u) { g1[j + u] = g[u]; b1[j + u] = b[u]; }
        }
    }
    if (has_ln2) {
#pragma unroll
        for (int j = 0; j < NV; j += 4) {
            const f32x4 g = *(const f32x4*)(a.ln2_g + c0 + j), b = *(const f32x4*)(a.ln2_b + c0 + j);
#pragma unroll
            for (int u = 0; u < 4; ++u) { g2[j + u] = g[u]; b2[j + u] = b[u]; }
        }
    }
    const int add_mod = (int)a.add_row_mod;
    const bool add_same = a.add != nullptr && add_mod == 1;     // one vector for every row (Q = 1)
    float ad[RPW][NV];
    if (a.add) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            if (add_same && i > 0) {
#pragma unroll
                for (int u = 0; u < NV; ++u) ad[i][u] = ad[0][u];
                continue;
            }
            int gm = m0 + wave * RPW + i; gm = gm < M ? gm : M - 1;
            const bf16_t* ap = (const bf16_t*)a.add + (int64_t)(add_same ? 0 : gm % add_mod) * K + c0;
            if constexpr (NV == 8) {
                const bf16x8 t = *(const bf16x8*)ap;
#pragma unroll
                for (int u = 0; u < 8; ++u) ad[i][u] = (float)t[u];
            } else {
                const bf16x4 t = *(const bf16x4*)ap;
#pragma unroll
                for (int u = 0; u < 4; ++u) ad[i][u] = (float)t[u];
            }
        }
    }
    float y2[RPW][NV];                                           // the second norm's output, stored after the last load
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int rl = wave * RPW + i;                          // row inside the tile (wave-uniform)
        if (has_ln) {
            if constexpr (ZB16) {
                // training chain: sum x and sum x^2 in ONE round of two independent wave reductions (bf16 rows: the difference to the
                // centred form is far below their rounding)
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j) { s += v[i][j]; q += v[i][j] * v[i][j]; }
                s = wave_sum(s); q = wave_sum(q);
                const float mean = s * (1.f / K);
                const float rstd = rsqrtf(fmaxf(q * (1.f / K) - mean * mean, 0.f) + a.eps);
#pragma unroll
                for (int j = 0; j < NV; ++j) v[i][j] = (v[i][j] - mean) * rstd * g1[j] + b1[j];
            } else {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j) s += v[i][j];
                const float mean = wave_sum(s) * (1.f / K);
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j) { v[i][j] -= mean; q += v[i][j] * v[i][j]; }
                const float rstd = rsqrtf(wave_sum(q) * (1.f / K) + a.eps);
#pragma unroll
                for (int j = 0; j < NV; ++j) v[i][j] = v[i][j] * rstd * g1[j] + b1[j];
            }
        }
        if (has_ln2) {                                          // the decoder's output norm on top of this layer's norm
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) s += v[i][j];
            const float mean = wave_sum(s) * (1.f / K);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) { const float d = v[i][j] - mean; q += d * d; }
            const float rstd = rsqrtf(wave_sum(q) * (1.f / K) + a.eps);
#pragma unroll
            for (int j = 0; j < NV; ++j) y2[i][j] = (v[i][j] - mean) * rstd * g2[j] + b2[j];
        }
        // the GEMM input: x as stored (bf16) + add, rounded once
        if constexpr (NV == 8) {
            bf16x8 t;
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = a.add ? (bf16_t)((float)(bf16_t)v[i][u] + ad[i][u]) : (bf16_t)v[i][u];
            *(bf16x8*)(dlds + rl * LDA + c0 * 2) = t;
        } else {
            bf16x4 t;
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = a.add ? (bf16_t)((float)(bf16_t)v[i][u] + ad[i][u]) : (bf16_t)v[i][u];
            *(bf16x4*)(dlds + rl * LDA + c0 * 2) = t;
        }
    }
    // the epilogue's inputs (thread t < 32 finishes row t / 2, 8 columns)
    const int cc = tid & 1, row = (tid >> 1) & 15;
    const int n = n0 + cc * 8;
    int nvalid = N - n; nvalid = nvalid > 8 ? 8 : nvalid;
    const int ml = m0 + row;
    const int mlc = ml < M ? ml : M - 1;
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = 0.f;
    if (a.bias) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int nj = n + j < N ? n + j : N - 1; bv[j] = a.bias[nj]; }
    }
    const bool r_vec = a.R && nvalid == 8 && (a.ldr % 8 == 0) && (((uintptr_t)a.R & 15) == 0);
    bf16x8 rpre;
    if (r_vec) rpre = *(const bf16x8*)((const bf16_t*)a.R + (int64_t)mlc * a.ldr + n);

    // the prologue's global stores (x, x2), after the last load of this wave
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int gm = m0 + wave * RPW + i;
        const bool writer = (gm < M) && (gm % (int)gridDim.x == (int)blockIdx.x);
        if (!writer) continue;
        if (a.x_out) {
            bf16_t* xp = (bf16_t*)a.x_out + (int64_t)gm * a.ldx + c0;
            if constexpr (NV == 8) { bf16x8 t; for (int u = 0; u < 8; ++u) t[u] = (bf16_t)v[i][u]; *(bf16x8*)xp = t; }
            else { bf16x4 t; for (int u = 0; u < 4; ++u) t[u] = (bf16_t)v[i][u]; *(bf16x4*)xp = t; }
        }
        if (TRAIN && a.a_out) {                                   // A = bf16(x) + add, exactly what went into the LDS tile
            bf16_t* ap = (bf16_t*)a.a_out + (int64_t)gm * a.lda_out + c0;
            if constexpr (NV == 8) { bf16x8 t; for (int u = 0; u < 8; ++u) t[u] = a.add ? (bf16_t)((float)(bf16_t)v[i][u] + ad[i][u]) : (bf16_t)v[i][u]; *(bf16x8*)ap = t; }
            else { bf16x4 t; for (int u = 0; u < 4; ++u) t[u] = a.add ? (bf16_t)((float)(bf16_t)v[i][u] + ad[i][u]) : (bf16_t)v[i][u]; *(bf16x4*)ap = t; }
        }
        if (has_ln2) {
            bf16_t* yp = (bf16_t*)a.x2_out + (int64_t)gm * a.ldx2 + c0;
            if constexpr (NV == 8) { bf16x8 t; for (int u = 0; u < 8; ++u) t[u] = (bf16_t)y2[i][u]; *(bf16x8*)yp = t; }
            else { bf16x4 t; for (int u = 0; u < 4; ++u) t[u] = (bf16_t)y2[i][u]; *(bf16x4*)yp = t; }
        }
    }
    __syncthreads();

    // ---- 3. 16 x 16 partial tile of this wave's K quarter (C layout: lane (r16, kq) holds rows 4 kq .. 4 kq + 3 of column r16)
    f32x4 acc;
    acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const bf16x8 fa = *(const bf16x8*)(dlds + r16 * LDA + (kw + s * 32) * 2);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fw[s], acc, 0, 0, 0);
    }

    // ---- 4. the four partial tiles meet in LDS; bias, activation, residual, store
    float xres[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xres[j] = 0.f;
    if (a.res_from_x && n + 8 <= K) {                           // residual = the normalised input itself (N == K): still in the A tile
        const bf16x8 t = *(const bf16x8*)(dlds + row * LDA + n * 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) xres[j] = (float)t[j];
    }
    {
        float* mine = Ct + wave * (DS_BM * DS_CT_LD);
#pragma unroll
        for (int e = 0; e < 4; ++e) mine[(kq * 4 + e) * DS_CT_LD + r16] = acc[e];
    }
    __syncthreads();
    if (tid >= 32 || nvalid <= 0 || ml >= M) return;
    float v8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v8[j] = 0.f;
    float rv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rv[j] = xres[j];
    const bool vec = nvalid == 8;
    if (a.R) {
        if (r_vec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rv[j] = (float)rpre[j];
        } else {
            for (int j = 0; j < nvalid; ++j) rv[j] = (float)((const bf16_t*)a.R)[(int64_t)ml * a.ldr + n + j];
        }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float* cp = Ct + w * (DS_BM * DS_CT_LD) + row * DS_CT_LD + cc * 8;
        const f32x4 q0 = *(const f32x4*)cp, q1 = *(const f32x4*)(cp + 4);
        v8[0] += q0[0]; v8[1] += q0[1]; v8[2] += q0[2]; v8[3] += q0[3];
        v8[4] += q1[0]; v8[5] += q1[1]; v8[6] += q1[2]; v8[7] += q1[3];
    }
    const int act = a.act;
#pragma unroll
    for (int j = 0; j < 8; ++j) v8[j] = ds_act(v8[j] + bv[j], act);
    if constexpr (TRAIN) {
        if (a.drop.p > 0.f) {                                   // element index row * drop_ld + col (/ drop_col_div: one draw per head)
            const uint32_t thr = made_drop_threshold(a.drop.p);
            const float sc = 1.f / (1.f - a.drop.p);
            const uint64_t seed = made_drop_seed(a.drop);
            const uint64_t rb = (uint64_t)ml * (uint64_t)a.drop_ld;
            const int div = a.drop_col_div > 1 ? a.drop_col_div : 1;
            if (div == 1) {
                const uint32_t kb = made_keep_bits<8>(seed, a.drop.site, thr, rb + (uint64_t)n);
#pragma unroll
                for (int j = 0; j < 8; ++j) v8[j] = ((kb >> j) & 1u) ? v8[j] * sc : 0.f;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v8[j] = (made_rng_mix(seed, a.drop.site, rb + (uint64_t)((n + j) / div)) >> 8) >= thr ? v8[j] * sc : 0.f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v8[j] += rv[j];
    if (a.out_dtype == MADE_F32) {
        float* op = (float*)a.out + (int64_t)ml * a.ldo + n;
        if (vec && (a.ldo % 4 == 0) && (((uintptr_t)a.out & 15) == 0)) {
            f32x4 o0, o1;
            o0[0] = v8[0]; o0[1] = v8[1]; o0[2] = v8[2]; o0[3] = v8[3]; o1[0] = v8[4]; o1[1] = v8[5]; o1[2] = v8[6]; o1[3] = v8[7];
            *(f32x4*)op = o0; *(f32x4*)(op + 4) = o1;
        } else {
            for (int j = 0; j < nvalid; ++j) op[j] = v8[j];
        }
    } else {
        bf16_t* op = (bf16_t*)a.out + (int64_t)ml * a.ldo + n;
        if (vec && (a.ldo % 8 == 0) && (((uintptr_t)a.out & 15) == 0)) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v8[j];
            *(bf16x8*)op = o;
        } else {
            for (int j = 0; j < nvalid; ++j) op[j] = (bf16_t)v8[j];
        }
    }
}

// =================================================================================================
// made_dec_stage_bwd: the mirror image for the backward chain -- a LayerNorm BACKWARD in the prologue of the Linear (dX product)
// that consumes its result (reference music_detr/transformer.py:273-307 read backwards: x = LN(t), t = res + dropout(branch)):
//     g   = dy (+ add)                                   or, with a second norm stacked on the first (norm 3 + the shared output norm):
//     g   = LN_b'(dy; xb, gamma_b) + add
//     dx  = LN_a'(g; xa, gamma_a)                        -> dx_out (the residual path's gradient), parameter gradients accumulated
//     A   = dropout_a(dx)                                -> a_out (the branch's output gradient: a weight-gradient operand) and LDS
//     out = dropout_o((A W^T) * gate'(G) * gate_scale) + R
// Every workgroup (64 rows x 32 output columns) redoes the row work for its 64 rows -- 2 x 64 KB from L2 under the flight of its own
// weight fragments, as made_dec_stage does for the forward norm -- so a 64-row link of the chain is ONE launch instead of two.
// One row of a LayerNorm backward with ALL FOUR row sums taken in one round (sum x, sum x^2, sum g, sum g x; g = dy * gamma): the four
// wave reductions are independent and pipeline, where mean -> variance -> (mean g, mean g xhat) is three dependent rounds of ~150
// cycles each, eight rows per wave (this prologue is what a 64-row stage of the backward chain spends its time in).
//   mean = Sx / K,  var = Sxx / K - mean^2,  s1 = Sg / K,  s2 = rstd (Sgx - mean Sg) / K,  dx = rstd (g - s1 - xhat s2)
template <int NV>
__device__ __forceinline__ void dsb_ln_bwd_row(float (&xv)[NV], float (&gy)[NV], const float (&gm)[NV], float eps, float (&dg)[NV], float (&db)[NV], float (&o)[NV]) {
    constexpr int K = 64 * NV;
    float sx = 0.f, sxx = 0.f, sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float g = gy[j] * gm[j];
        sx += xv[j]; sxx += xv[j] * xv[j]; sg += g; sgx += g * xv[j];
    }
    sx = wave_sum(sx); sxx = wave_sum(sxx); sg = wave_sum(sg); sgx = wave_sum(sgx);
    const float mean = sx * (1.f / K);
    const float var = fmaxf(sxx * (1.f / K) - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float s1 = sg * (1.f / K);
    const float s2 = rstd * (sgx - mean * sg) * (1.f / K);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float xh = (xv[j] - mean) * rstd;
        dg[j] += gy[j] * xh; db[j] += gy[j];
        o[j] = rstd * (gy[j] * gm[j] - s1 - xh * s2);
    }
}

template <int NV>
__device__ __forceinline__ void dsb_load_row(const void* p, int64_t off, float (&v)[NV]) {
    if constexpr (NV == 8) {
        const bf16x8 t = *(const bf16x8*)((const bf16_t*)p + off);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (float)t[u];
    } else {
        const bf16x4 t = *(const bf16x4*)((const bf16_t*)p + off);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (float)t[u];
    }
}
template <int NV>
__device__ __forceinline__ void dsb_store_row(void* p, int64_t off, const float (&v)[NV]) {
    if constexpr (NV == 8) { bf16x8 t; for (int u = 0; u < 8; ++u) t[u] = (bf16_t)v[u]; *(bf16x8*)((bf16_t*)p + off) = t; }
    else { bf16x4 t; for (int u = 0; u < 4; ++u) t[u] = (bf16_t)v[u]; *(bf16x4*)((bf16_t*)p + off) = t; }
}

template <int NV, int MODE>                                   // MODE 0: one norm; 1: one norm, g = dy + add; 2: two stacked norms (+ add)
__global__ __launch_bounds__(DS_THREADS) void dec_stage_bwd_kernel(const MadeDecStageBwdArgs a) {
    constexpr bool TWO = MODE == 2, ADD1 = MODE == 1;
    constexpr int K = 64 * NV;
    constexpr int LDA = K * 2 + 16;
    constexpr int STEPS = NV / 2;                               // 32-deep MFMA k-steps per wave (K / 4 / 32)
    __shared__ __attribute__((aligned(16))) unsigned char dlds[DS_BM * LDA];
    __shared__ __attribute__((aligned(16))) float Ct[4 * DS_BM * DS_CT_LD];
    __shared__ float pgrad[4 * (TWO ? 4 : 2) * K];              // [4 waves][2 or 4 vectors][K]: parameter-gradient partials (workgroups x = 0 only)
    constexpr int PV = TWO ? 4 : 2;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int M = (int)a.M, N = (int)a.N;
    const int n0 = blockIdx.x * DS_BN, m0 = blockIdx.y * DS_BM;
    constexpr bool two = TWO;

    // ---- 1. this wave's weight fragments (a quarter of K), in flight before anything else
    const int kw = wave * (K / 4) + kq * 8;
    int gn = n0 + r16; gn = gn < N ? gn : N - 1;
    const bf16_t* pw = (const bf16_t*)a.W + (int64_t)gn * a.ldw + kw;
    bf16x8 fw[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) fw[s] = *(const bf16x8*)(pw + s * 32);
    // the epilogue's inputs (thread t < 32 finishes row t / 2, 8 columns)
    const int cc = tid & 1, row = (tid >> 1) & 15;
    const int n = n0 + cc * 8;
    int nvalid = N - n; nvalid = nvalid > 8 ? 8 : nvalid;
    const int ml = m0 + row;
    const int mlc = ml < M ? ml : M - 1;
    const bool e_vec = nvalid == 8;
    bf16x8 rpre, gpre;
    const bool r_vec = a.R && e_vec && (a.ldr % 8 == 0) && (((uintptr_t)a.R & 15) == 0);
    const bool g_vec = a.G && e_vec && (a.ldg % 8 == 0) && (((uintptr_t)a.G & 15) == 0);
    if (r_vec) rpre = *(const bf16x8*)((const bf16_t*)a.R + (int64_t)mlc * a.ldr + n);
    if (g_vec) gpre = *(const bf16x8*)((const bf16_t*)a.G + (int64_t)mlc * a.ldg + n);

    // ---- 2. LayerNorm backward of this wave's 4 rows (lane l: columns NV*l .. NV*l + NV - 1)
    const int c0 = NV * lane;
    float gma[NV], gmb[NV];
#pragma unroll
    for (int j = 0; j < NV; j += 4) {
        const f32x4 g = *(const f32x4*)(a.gamma_a + c0 + j);
        gma[j] = g[0]; gma[j + 1] = g[1]; gma[j + 2] = g[2]; gma[j + 3] = g[3];
        if (two) { const f32x4 h = *(const f32x4*)(a.gamma_b + c0 + j); gmb[j] = h[0]; gmb[j + 1] = h[1]; gmb[j + 2] = h[2]; gmb[j + 3] = h[3]; }
        else { gmb[j] = gmb[j + 1] = gmb[j + 2] = gmb[j + 3] = 1.f; }
    }
    float dga[NV], dba[NV], dgb[NV], dbb[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { dga[j] = 0.f; dba[j] = 0.f; dgb[j] = 0.f; dbb[j] = 0.f; }
    const uint32_t thr = made_drop_threshold(a.drop_a.p);
    const uint64_t seed_a = a.drop_a.p > 0.f ? made_drop_seed(a.drop_a) : 0;
    const float sc_a = a.drop_a.p > 0.f ? 1.f / (1.f - a.drop_a.p) : 1.f;
    constexpr int RPW = DS_RPW;
    // every row of this wave is requested before the first reduction (raw bf16 vectors: 4 registers per row and tensor)
    typedef typename std::conditional<NV == 8, bf16x8, bf16x4>::type raw_t;
    raw_t rxa[RPW], rdy[RPW], rxb[TWO ? RPW : 1], rad[(TWO || ADD1) ? RPW : 1];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        int gm = m0 + wave * RPW + i; gm = gm < M ? gm : M - 1;
        rxa[i] = *(const raw_t*)((const bf16_t*)a.xa + (int64_t)gm * a.ldxa + c0);
        rdy[i] = *(const raw_t*)((const bf16_t*)a.dy + (int64_t)gm * a.lddy + c0);
    }
    if constexpr (TWO) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            int gm = m0 + wave * RPW + i; gm = gm < M ? gm : M - 1;
            rxb[i] = *(const raw_t*)((const bf16_t*)a.xb + (int64_t)gm * a.ldxb + c0);
            rad[i] = *(const raw_t*)((const bf16_t*)(a.add ? a.add : a.xb) + (int64_t)gm * (a.add ? a.ldadd : a.ldxb) + c0);
        }
    }
    if constexpr (ADD1) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            int gm = m0 + wave * RPW + i; gm = gm < M ? gm : M - 1;
            rad[i] = *(const raw_t*)((const bf16_t*)a.add + (int64_t)gm * a.ldadd + c0);
        }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int rl = wave * RPW + i;
        int gm = m0 + rl; gm = gm < M ? gm : M - 1;
        const bool real = m0 + rl < M;
        float xa[NV], gy[NV], o[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) { xa[j] = (float)rxa[i][j]; gy[j] = (float)rdy[i][j]; }
        if constexpr (ADD1) {
#pragma unroll
            for (int j = 0; j < NV; ++j) gy[j] += (float)rad[i][j];
        }
        if constexpr (TWO) {
            float xb[NV], ad[NV], g[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) { xb[j] = (float)rxb[i][j]; ad[j] = a.add ? (float)rad[i][j] : 0.f; }
            float tg[NV], tb[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) { tg[j] = 0.f; tb[j] = 0.f; }
            dsb_ln_bwd_row<NV>(xb, gy, gmb, a.eps, tg, tb, g);
#pragma unroll
            for (int j = 0; j < NV; ++j) { if (real) { dgb[j] += tg[j]; dbb[j] += tb[j]; } gy[j] = g[j] + ad[j]; }
        }
        float tg[NV], tb[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) { tg[j] = 0.f; tb[j] = 0.f; }
        dsb_ln_bwd_row<NV>(xa, gy, gma, a.eps, tg, tb, o);
#pragma unroll
        for (int j = 0; j < NV; ++j) { if (real) { dga[j] += tg[j]; dba[j] += tb[j]; } }
        // the branch's gradient: dropout of this site's forward mask (element index row * drop_a_ld + col)
        float od[NV];
        const uint64_t dbase = (uint64_t)gm * (uint64_t)a.drop_a_ld + (uint64_t)c0;
        const uint32_t kb = a.drop_a.p > 0.f ? made_keep_bits<NV>(seed_a, a.drop_a.site, thr, dbase) : 0xFFu;
#pragma unroll
        for (int j = 0; j < NV; ++j) od[j] = ((kb >> j) & 1u) ? o[j] * sc_a : 0.f;
        dsb_store_row<NV>(dlds, (int64_t)rl * (LDA / 2) + c0, od);               // (LDA bytes = LDA / 2 bf16 elements per row)
        const bool writer = real && ((m0 + rl) % (int)gridDim.x == (int)blockIdx.x);
        if (writer) {
            if (a.dx_out) dsb_store_row<NV>(a.dx_out, (int64_t)gm * a.lddx + c0, o);
            if (a.a_out) dsb_store_row<NV>(a.a_out, (int64_t)gm * a.lda_out + c0, od);
        }
    }
    // parameter gradients: the four waves' column partials of the workgroups x = 0 (one per 16 rows) meet in LDS, one atomic per column
    if (blockIdx.x == 0) {
        float* mine = pgrad + wave * PV * K;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            mine[c0 + j] = dga[j]; mine[K + c0 + j] = dba[j];
            if constexpr (TWO) { mine[2 * K + c0 + j] = dgb[j]; mine[3 * K + c0 + j] = dbb[j]; }
        }
    }
    __syncthreads();

    // ---- 3. 16 x 16 partial tile of this wave's K quarter (C layout: lane (r16, kq) holds rows 4 kq .. 4 kq + 3 of column r16)
    f32x4 acc;
    acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const bf16x8 fa = *(const bf16x8*)(dlds + r16 * LDA + (kw + s * 32) * 2);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fw[s], acc, 0, 0, 0);
    }
    {
        float* mine = Ct + wave * (DS_BM * DS_CT_LD);
#pragma unroll
        for (int e = 0; e < 4; ++e) mine[(kq * 4 + e) * DS_CT_LD + r16] = acc[e];
    }
    if (blockIdx.x == 0) {                                      // (after the MFMAs are issued: the atomics' round trip is nobody's business)
        for (int c = tid; c < PV * K; c += DS_THREADS) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += pgrad[w * PV * K + c];
            float* dst = c < K ? a.dgamma_a : (c < 2 * K ? a.dbeta_a : (c < 3 * K ? a.dgamma_b : a.dbeta_b));
            if (dst) unsafeAtomicAdd(dst + (c % K), t);
        }
    }
    __syncthreads();
    if (tid >= 32 || nvalid <= 0 || ml >= M) return;

    // ---- 4. the four partial tiles meet; gate, dropout, residual, store
    float v8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v8[j] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float* cp = Ct + w * (DS_BM * DS_CT_LD) + row * DS_CT_LD + cc * 8;
        const f32x4 q0 = *(const f32x4*)cp, q1 = *(const f32x4*)(cp + 4);
        v8[0] += q0[0]; v8[1] += q0[1]; v8[2] += q0[2]; v8[3] += q0[3];
        v8[4] += q1[0]; v8[5] += q1[1]; v8[6] += q1[2]; v8[7] += q1[3];
    }
    if (a.G) {                                                  // ReLU gate from the saved (dropped) activation: act'(G) * gate_scale
        float g8[8];
        if (g_vec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) g8[j] = (float)gpre[j];
        } else {
            for (int j = 0; j < 8; ++j) g8[j] = j < nvalid ? (float)((const bf16_t*)a.G)[(int64_t)ml * a.ldg + n + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = g8[j] != 0.f ? v8[j] * a.gate_scale : 0.f;
    }
    if (a.drop_o.p > 0.f) {
        const uint32_t thr_o = made_drop_threshold(a.drop_o.p);
        const float sc = 1.f / (1.f - a.drop_o.p);
        const uint64_t seed_o = made_drop_seed(a.drop_o);
        const uint64_t rb = (uint64_t)ml * (uint64_t)a.drop_o_ld;
        const int div = a.drop_o_col_div > 1 ? a.drop_o_col_div : 1;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v8[j] = (made_rng_mix(seed_o, a.drop_o.site, rb + (uint64_t)((n + j) / div)) >> 8) >= thr_o ? v8[j] * sc : 0.f;
    }
    if (a.R) {
        if (r_vec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] += (float)rpre[j];
        } else {
            for (int j = 0; j < nvalid; ++j) v8[j] += (float)((const bf16_t*)a.R)[(int64_t)ml * a.ldr + n + j];
        }
    }
    bf16_t* op = (bf16_t*)a.out + (int64_t)ml * a.ldo + n;
    if (e_vec && (a.ldo % 8 == 0) && (((uintptr_t)a.out & 15) == 0)) {
        bf16x8 o8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = (bf16_t)v8[j];
        *(bf16x8*)op = o8;
    } else {
        for (int j = 0; j < nvalid; ++j) op[j] = (bf16_t)v8[j];
    }
}

}  // namespace

extern "C" int made_dec_stage_bwd(const MadeDecStageBwdArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_dec_stage_bwd: null args");
    const MadeDecStageBwdArgs& a = *args;
    MADE_REQUIRE(a.xa && a.gamma_a && a.dy && a.W && a.out, "made_dec_stage_bwd: null xa, gamma_a, dy, W or out");
    MADE_REQUIRE((a.xb == nullptr) == (a.gamma_b == nullptr), "made_dec_stage_bwd: xb and gamma_b come together");
    MADE_REQUIRE(a.M > 0 && a.N > 0, "made_dec_stage_bwd: bad dims M=%lld N=%lld", (long long)a.M, (long long)a.N);
    MADE_UNSUPPORTED(a.K == 256 || a.K == 512, "made_dec_stage_bwd: K=%lld (the row width of the LayerNorm) must be 256 or 512", (long long)a.K);
    MADE_UNSUPPORTED(a.ldxa % 8 == 0 && a.lddy % 8 == 0 && a.ldxb % 8 == 0 && a.ldadd % 8 == 0 && a.lddx % 8 == 0 && a.lda_out % 8 == 0 && a.ldw % 8 == 0 &&
                     ((uintptr_t)a.xa % 16) == 0 && ((uintptr_t)a.dy % 16) == 0 && ((uintptr_t)a.xb % 16) == 0 && ((uintptr_t)a.add % 16) == 0 &&
                     ((uintptr_t)a.dx_out % 16) == 0 && ((uintptr_t)a.a_out % 16) == 0 && ((uintptr_t)a.W % 16) == 0,
                     "made_dec_stage_bwd: rows must be 16-byte aligned");
    MADE_REQUIRE(a.drop_a.p >= 0.f && a.drop_a.p < 1.f && a.drop_o.p >= 0.f && a.drop_o.p < 1.f, "made_dec_stage_bwd: dropout p out of [0,1)");
    const dim3 grid((unsigned)((a.N + DS_BN - 1) / DS_BN), (unsigned)((a.M + DS_BM - 1) / DS_BM)), block(DS_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (a.K == 512) {
        if (a.xb) hipLaunchKernelGGL((dec_stage_bwd_kernel<8, 2>), grid, block, 0, st, a);
        else if (a.add) hipLaunchKernelGGL((dec_stage_bwd_kernel<8, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((dec_stage_bwd_kernel<8, 0>), grid, block, 0, st, a);
    } else {
        if (a.xb) hipLaunchKernelGGL((dec_stage_bwd_kernel<4, 2>), grid, block, 0, st, a);
        else if (a.add) hipLaunchKernelGGL((dec_stage_bwd_kernel<4, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((dec_stage_bwd_kernel<4, 0>), grid, block, 0, st, a);
    }
    return made_check_launch("made_dec_stage_bwd");
}

extern "C" int made_dec_stage(const MadeDecStageArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_dec_stage: null args");
    const MadeDecStageArgs& a = *args;
    MADE_REQUIRE(a.Zin && a.W && a.out, "made_dec_stage: null Zin, W or out");
    MADE_REQUIRE(a.M > 0 && a.N > 0, "made_dec_stage: bad dims M=%lld N=%lld", (long long)a.M, (long long)a.N);
    MADE_UNSUPPORTED(a.K == 256 || a.K == 512, "made_dec_stage: K=%lld (the row width of the LayerNorm) must be 256 or 512", (long long)a.K);
    MADE_UNSUPPORTED(a.ldz % 4 == 0 && ((uintptr_t)a.Zin % 16) == 0, "made_dec_stage: Zin rows must be 16-byte aligned");
    MADE_UNSUPPORTED(a.ldw % 8 == 0 && ((uintptr_t)a.W % 16) == 0, "made_dec_stage: W rows must be 16-byte aligned");
    MADE_REQUIRE((a.ln_g == nullptr) == (a.ln_b == nullptr) && (a.ln2_g == nullptr) == (a.ln2_b == nullptr), "made_dec_stage: gamma and beta come together");
    MADE_REQUIRE(a.out_dtype == MADE_F32 || a.out_dtype == MADE_BF16, "made_dec_stage: bad out_dtype %d", a.out_dtype);
    MADE_UNSUPPORTED(a.act == MADE_ACT_NONE || a.act == MADE_ACT_RELU, "made_dec_stage: act=%d (ReLU or none)", a.act);
    if (a.x_out) MADE_UNSUPPORTED(a.ldx % 8 == 0 && ((uintptr_t)a.x_out % 16) == 0, "made_dec_stage: x_out rows must be 16-byte aligned");
    if (a.x2_out) MADE_UNSUPPORTED(a.ldx2 % 8 == 0 && ((uintptr_t)a.x2_out % 16) == 0, "made_dec_stage: x2_out rows must be 16-byte aligned");
    if (a.add) MADE_REQUIRE(a.add_row_mod >= 1 && ((uintptr_t)a.add % 16) == 0, "made_dec_stage: add needs add_row_mod >= 1 and 16-byte alignment");
    if (a.res_from_x) MADE_REQUIRE(a.N == a.K && a.add == nullptr && a.R == nullptr, "made_dec_stage: res_from_x needs N == K, no add, no R");
    const bool train = a.drop.p > 0.f || a.a_out != nullptr;
    MADE_REQUIRE(a.zin_dtype == MADE_F32 || a.zin_dtype == MADE_BF16, "made_dec_stage: bad zin_dtype %d", a.zin_dtype);
    MADE_REQUIRE(a.drop.p >= 0.f && a.drop.p < 1.f, "made_dec_stage: dropout p out of [0,1)");
    if (a.a_out) MADE_UNSUPPORTED(a.lda_out % 8 == 0 && ((uintptr_t)a.a_out % 16) == 0, "made_dec_stage: a_out rows must be 16-byte aligned");
    if (a.zin_dtype == MADE_BF16) MADE_UNSUPPORTED(a.ldz % 8 == 0, "made_dec_stage: bf16 Zin rows must be 16-byte aligned");
    // two instances per width: the eval chain (f32 raw rows, no dropout) and the training chain (bf16 rows, dropout, a_out)
    MADE_UNSUPPORTED((a.zin_dtype == MADE_BF16) == train || !train, "made_dec_stage: dropout / a_out need bf16 Zin rows");
    const bool tr = a.zin_dtype == MADE_BF16;
    const dim3 grid((unsigned)((a.N + DS_BN - 1) / DS_BN), (unsigned)((a.M + DS_BM - 1) / DS_BM)), block(DS_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (a.K == 512) {
        if (tr) hipLaunchKernelGGL((dec_stage_kernel<8, true, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((dec_stage_kernel<8, false, false>), grid, block, 0, st, a);
    } else {
        if (tr) hipLaunchKernelGGL((dec_stage_kernel<4, true, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((dec_stage_kernel<4, false, false>), grid, block, 0, st, a);
    }
    return made_check_launch("made_dec_stage");
}
