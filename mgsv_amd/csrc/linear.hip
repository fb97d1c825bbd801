// made_linear: fused Linear (GEMM + prologue/epilogue) on MFMA, gfx950.
//
// Tile: 128 x 128 outputs per 256-thread workgroup (4 waves, 2 x 2, each 64 x 64 = 2 x 2 MFMA
// 32x32 tiles), K consumed in 128-byte slabs (64 bf16 / 32 f32 per row).  Both operands are
// K-contiguous ([M,K] activations, [N,K] nn.Linear weights), staged global -> registers -> LDS
// (the activation prologue: row mask, +A2, f32->bf16 conversion happens in registers) into a
// two-stage LDS ring: the next slab's global loads are in flight during the current slab's MFMAs
// and there is one barrier per slab.  LDS rows are padded to 144 B so the 16-byte fragment reads
// of a wave are bank-conflict free.  The epilogue parks the accumulators in LDS as an f32 tile and
// streams it out row-contiguously with 16-byte loads/stores (bias, activation, residual, row mask,
// per-batch addressing), so the output side is coalesced whatever the MFMA register layout.
//   bf16 : v_mfma_f32_32x32x16_bf16, one per 16-byte fragment pair
//   f32  : v_mfma_f32_32x32x2_f32, four per fragment pair (exact f32 FMA chain)
// Transposed output segments swap the MFMA operands so that lanes run along the row (time) axis
// and the per-batch transposed store stays coalesced.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int KB = 128;               // bytes of K per row per stage, in the compute type
constexpr int LDS_ROW = KB + 16;      // padded row stride in bytes
constexpr int NTHREADS = 256;

template <typename TC> struct Frag;
template <> struct Frag<float>  { typedef f32x4  type; };
template <> struct Frag<bf16_t> { typedef bf16x8 type; };

template <typename TC>
__device__ __forceinline__ typename Frag<TC>::type zero_frag() {
    typename Frag<TC>::type z;
#pragma unroll
    for (int i = 0; i < elem_traits<TC>::per16; ++i) z[i] = (TC)0.f;
    return z;
}

// Load one 16-byte compute-type fragment of A' = (A [+ A2]); p / p2 point at its first element.
template <typename TA, typename TC>
__device__ __forceinline__ typename Frag<TC>::type load_a_frag(const TA* __restrict__ p, const TA* __restrict__ p2) {
    constexpr int n = elem_traits<TC>::per16;
    float v[n];
    if constexpr (sizeof(TA) == 4) {
#pragma unroll
        for (int i = 0; i < n; i += 4) {
            f32x4 t = *(const f32x4*)(p + i);
            v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
        }
    } else {
        static_assert(n == 8, "bf16 activations need bf16 compute");
        bf16x8 t = *(const bf16x8*)p;
        if (!p2) return t;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
    }
    if (p2) {
        if constexpr (sizeof(TA) == 4) {
#pragma unroll
            for (int i = 0; i < n; i += 4) {
                f32x4 t = *(const f32x4*)(p2 + i);
                v[i] += t[0]; v[i + 1] += t[1]; v[i + 2] += t[2]; v[i + 3] += t[3];
            }
        } else {
            bf16x8 t = *(const bf16x8*)p2;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += (float)t[i];
        }
    }
    typename Frag<TC>::type f;
#pragma unroll
    for (int i = 0; i < n; ++i) f[i] = from_f32<TC>(v[i]);
    return f;
}

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case MADE_ACT_RELU: return fmaxf(x, 0.f);
        case MADE_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
        case MADE_ACT_QUICKGELU: return x / (1.f + expf(-1.702f * x));
        case MADE_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        default: return x;
    }
}

// erf for the bf16 fast paths (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7: far inside a bf16 output's half-ulp of 2e-3 and the
// f32 output tolerance; ~14 instructions against erff's ~50).  The f32 parity kernels keep erff.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = fmaf(-p * t, e, 1.f);
    return copysignf(r, x);
}
__device__ __forceinline__ float apply_act_fast(float x, int act) {
    switch (act) {
        case MADE_ACT_RELU: return fmaxf(x, 0.f);
        case MADE_ACT_GELU: return 0.5f * x * (1.f + fast_erf(x * 0.70710678118654752440f));
        case MADE_ACT_QUICKGELU: return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
        case MADE_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        default: return x;
    }
}
__device__ __forceinline__ float act_grad(float g, int gate);
__device__ __forceinline__ float act_grad_fast(float g, int gate) {
    switch (gate) {
        case MADE_GATE_GELU_Z: {
            const float cdf = 0.5f * (1.f + fast_erf(g * 0.70710678118654752440f));
            return fmaf(g * 0.39894228040143267794f, __builtin_amdgcn_exp2f(-0.5f * 1.4426950408889634f * g * g), cdf);
        }
        default: return act_grad(g, gate);
    }
}

// bias of 8 consecutive columns.  Loads under a per-element condition are serialised by hipcc (one round trip each: 8 of them cost
// the 64-row kernels 3-4 us of a 10 us launch); ONE uniform branch, clamped addresses, columns past N are never stored.
__device__ __forceinline__ void load_bias8(const float* __restrict__ bias, int n, int N, float* bv) {
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = 0.f;
    if (bias) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int nj = n + j < N ? n + j : N - 1; bv[j] = bias[nj]; }
    }
}

// 8 consecutive f32 -> output dtype, vector store when `vec_ok`
__device__ __forceinline__ void store8(void* out, int odt, int64_t off, const float* v, int nvalid, bool vec_ok) {
    if (vec_ok && nvalid == 8) {
        if (odt == MADE_F32) {
            f32x4 a, b;
            a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
            b[0] = v[4]; b[1] = v[5]; b[2] = v[6]; b[3] = v[7];
            *(f32x4*)((float*)out + off) = a;
            *(f32x4*)((float*)out + off + 4) = b;
        } else {
            bf16x8 t;
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = (bf16_t)v[j];
            *(bf16x8*)((bf16_t*)out + off) = t;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)                          // (a run-time trip count would index v[] dynamically: scratch memory)
            if (j < nvalid) store_from_f32(out, odt, off + j, v[j]);
    }
}

// 8 consecutive f32 from `p` in a runtime dtype (vector load when `vec`)
__device__ __forceinline__ void load8(const void* p, int dt, int64_t off, float* v, int nvalid, bool vec) {
    if (vec && nvalid == 8) {
        if (dt == MADE_F32) {
            f32x4 r0 = *(const f32x4*)((const float*)p + off), r1 = *(const f32x4*)((const float*)p + off + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = r0[j]; v[4 + j] = r1[j]; }
        } else {
            bf16x8 rb = *(const bf16x8*)((const bf16_t*)p + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)rb[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = j < nvalid ? load_as_f32(p, dt, off + j) : 0.f;
    }
}

__device__ __forceinline__ float act_grad(float g, int gate) {
    switch (gate) {
        case MADE_GATE_RELU_OUT: return g != 0.f ? 1.f : 0.f;                    // g = saved output (after ReLU and dropout)
        case MADE_GATE_GELU_Z: {                                                // g = saved pre-activation
            const float cdf = 0.5f * (1.f + erff(g * 0.70710678118654752440f));
            return cdf + g * 0.39894228040143267794f * expf(-0.5f * g * g);
        }
        case MADE_GATE_QUICKGELU_Z: {
            const float sg = 1.f / (1.f + expf(-1.702f * g));
            return sg * (1.f + 1.702f * g * (1.f - sg));
        }
        case MADE_GATE_SIGMOID_OUT: return g * (1.f - g);                        // g = saved sigmoid output
        default: return 1.f;
    }
}

// The element-wise tail of made_linear on 8 consecutive outputs of row m starting at column n:
//   z = acc + bias  [-> Zout]   v = act(z)   v *= act'(G) * gate_scale   v = dropout(v)   v += R   row mask
// rpre / gpre / om: residual, gate tensor and output row mask of these 8 outputs when the caller loaded them ahead of its K loop
// (the 64-row kernels are one dependent memory round trip after another; whatever the epilogue needs is requested up front)
// (by value: a pointer that is selected at run time -- `have ? &x : nullptr` -- pins x in scratch memory)
struct EpiPre {
    bool has_r = false, has_g = false, has_om = false, has_seed = false;
    bf16x8 r, g;
    float om = 1.f;
    uint64_t seed = 0;
};

template <bool TRAIN>
__device__ __forceinline__ void epilogue8(const MadeLinearArgs& a, int m, int n, int nvalid, float* v, const float* bv,
                                          int rmod, bool r_vec, const EpiPre pre = EpiPre()) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += bv[j];
    if (TRAIN && a.Zout) {
        const bool zv = (a.ldz % 8 == 0) && (((uintptr_t)a.Zout & 15) == 0);
        store8(a.Zout, a.z_dtype, (int64_t)m * a.ldz + n, v, nvalid, zv);
    }
    // one (wave-uniform) switch per 8 outputs, not one per output: the epilogue of a one-wave-per-SIMD kernel has nothing to hide
    // a chain of taken branches behind
    switch (a.act) {
        case MADE_ACT_NONE: break;
        case MADE_ACT_RELU:
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            break;
        default:
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j], a.act);
            break;
    }
    if (TRAIN && a.gate != MADE_GATE_NONE) {
        float g[8];
        if (pre.has_g) {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = (float)pre.g[j];
        } else {
            load8(a.G, a.g_dtype, (int64_t)m * a.ldg + n, g, nvalid, (a.ldg % 8 == 0) && (((uintptr_t)a.G & 15) == 0));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= act_grad(g[j], a.gate) * a.gate_scale;
    }
    if (TRAIN && a.drop.p > 0.f) {
        const uint32_t thr = made_drop_threshold(a.drop.p);
        const float sc = 1.f / (1.f - a.drop.p);
        const uint64_t seed = pre.has_seed ? pre.seed : made_drop_seed(a.drop);   // (a device-side seed is a memory round trip: the 16-row kernel reads it up front)
        if (a.drop_col_div > 1) {                            // one draw per group of columns (per attention head): index row * drop_ld + col / div
            const uint64_t rb = (uint64_t)m * (uint64_t)a.drop_ld;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = (made_rng_mix(seed, a.drop.site, rb + (uint64_t)((n + j) / a.drop_col_div)) >> 8) >= thr ? v[j] * sc : 0.f;
        } else {
            const uint64_t base = (uint64_t)m * (uint64_t)a.drop_ld + (uint64_t)n;
            const uint32_t kb = made_keep_bits<8>(seed, a.drop.site, thr, base);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ((kb >> j) & 1u) ? v[j] * sc : 0.f;
        }
    }
    if (a.R) {
        float rv[8];
        if (pre.has_r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rv[j] = (float)pre.r[j];
        } else {
            const int rr = rmod > 0 ? m % rmod : m;
            load8(a.R, a.r_dtype, (int64_t)rr * a.ldr + n, rv, nvalid, r_vec);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rv[j];
    }
    if (pre.has_om ? (pre.om == 0.f) : (a.out_row_mask && a.out_row_mask[m] == 0.f)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
}

template <typename TA, typename TC, bool X3 = false>      // X3 (f32 compute only): split-bf16 products (common.h, made_set_f32_products)
__global__ __launch_bounds__(NTHREADS, 2) void linear_kernel(const MadeLinearArgs a) {
    typedef typename Frag<TC>::type frag_t;
    constexpr int PER16 = elem_traits<TC>::per16;
    constexpr int KE = KB / (int)sizeof(TC);          // elements of K per stage
    constexpr int STAGE = 2 * BM * LDS_ROW;           // bytes of one (A, W) stage
    constexpr int CT_LD = BN + 4;                     // f32 row stride of the epilogue tile
    static_assert(BM * CT_LD * 4 <= 2 * STAGE, "epilogue tile must fit in the staging LDS");

    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
    const int n_tiles = (N + BN - 1) / BN;
    const int tile_m = blockIdx.x / n_tiles, tile_n = blockIdx.x % n_tiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int splitk = a.split_k > 1 ? a.split_k : 1;
    const int64_t z = splitk > 1 ? 0 : blockIdx.z;         // grid.z = K split index when split_k > 1, else problem index
    // row gather: logical row r (r < *n_rows) lives at physical row row_index[r]; tiles past the valid rows have nothing to do
    int Mv = M;
    if (a.n_rows) { const int nv = *a.n_rows; Mv = nv < M ? nv : M; }
    if (m0 >= Mv) return;

    // segment of this column tile
    int si = 0;
#pragma unroll
    for (int s = 1; s < 4; ++s)
        if (s < a.nseg && n0 >= a.seg[s].col_begin) si = s;
    const MadeLinearSeg seg = a.seg[si];
    const bool transposed = seg.transposed != 0;

    // ---- padded tiles: if every row of this tile is padding, there is nothing to compute.  Every wave looks at all 128
    // rows itself (2 per lane), so the answer is wave-uniform and identical in the 4 waves: no barrier on this path.
    if (a.tile_skip_mask) {
        const int g0 = m0 + lane, g1 = m0 + 64 + lane;
        const float v0 = g0 < M ? a.tile_skip_mask[g0] : 0.f, v1 = g1 < M ? a.tile_skip_mask[g1] : 0.f;
        if (!__any(v0 != 0.f || v1 != 0.f)) {
            if (a.out_row_mask && !seg.transposed && (a.split_k <= 1)) {       // consumers expect zeros in masked rows
                const int rpb0 = (int)seg.rows_per_batch;
                for (int idx = tid; idx < BM * (BN / 8); idx += NTHREADS) {
                    const int row = idx / (BN / 8), c8 = idx % (BN / 8);
                    const int m = m0 + row, n = n0 + c8 * 8;
                    if (m >= M || n >= N) continue;
                    int64_t orow;
                    if (rpb0 > 0) { const int b = m / rpb0, t = m - b * rpb0; orow = (int64_t)b * seg.out_batch_stride + (int64_t)t * seg.ldo; }
                    else orow = (int64_t)m * seg.ldo;
                    const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    int nv = N - n; nv = nv > 8 ? 8 : nv;
                    store8(seg.out, seg.out_dtype, blockIdx.z * seg.out_z_stride + orow + (n - (int)seg.col_begin), zero8, nv, false);
                }
            }
            return;
        }
    }

    // ---- staging assignment: thread owns 16-byte chunk kc of rows srow0 + 32*i (i = 0..3) of A and of W.
    // All row-dependent address math is done once here, not per K slab.
    const int kc = tid & 7, srow0 = tid >> 3;
    const TA* pa[4];
    const TA* pa2[4];
    const TC* pw[4];
    bool ka[4], kw[4];                                 // keep flags: rows past the edge / masked rows are zeroed AFTER the load
    const bool has_a2 = seg.use_a2 && a.A2 && !a.a2_replace;
    {
        const bool repl = seg.use_a2 && a.A2 && a.a2_replace;
        const TA* A = (repl ? (const TA*)a.A2 : (const TA*)a.A) + z * a.a_z_stride;
        const int64_t lda = repl ? a.lda2 : a.lda;
        const TA* A2 = has_a2 ? (const TA*)a.A2 : nullptr;
        const TC* W = (const TC*)a.W + z * a.w_z_stride;
        const int a2mod = (int)a.a2_row_mod;
        const int kce = kc * PER16 < K ? kc * PER16 : 0;      // rows shorter than one slab (K < 64 / 32): stay inside the row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = m0 + srow0 + 32 * i, gn = n0 + srow0 + 32 * i;
            const int gml = gm < Mv ? gm : Mv - 1, gnc = gn < N ? gn : N - 1;     // clamped: every lane always loads
            const int gmc = a.row_index ? a.row_index[gml] : gml;                 // physical row
            ka[i] = gm < Mv && (a.a_row_mask == nullptr || a.a_row_mask[gmc] != 0.f);
            kw[i] = gn < N;
            pa[i] = A + (int64_t)gmc * lda + kce;
            pa2[i] = A2 ? A2 + (int64_t)(a2mod > 0 ? gmc % a2mod : gmc) * a.lda2 + kce : nullptr;
            pw[i] = W + (int64_t)gnc * a.ldw + kce;
        }
    }
    frag_t ra[4], rw[4];
    // branch-free staging loads (conditional loads make hipcc wait for each load in turn): the K offset is clamped into
    // the row, out-of-range pieces are zeroed on the registers
    auto load_stage = [&](int k0) {
        const bool kin = k0 + kc * PER16 < K;
        const int kofs = kin ? k0 : 0;
        if (has_a2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = load_a_frag<TA, TC>(pa[i] + kofs, pa2[i] + kofs);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = load_a_frag<TA, TC>(pa[i] + kofs, nullptr);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) rw[i] = *(const frag_t*)(pw[i] + kofs);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = keep_or_zero(ra[i], ka[i] && kin);
            rw[i] = keep_or_zero(rw[i], kw[i] && kin);
        }
    };
    auto store_stage = [&](unsigned char* st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(frag_t*)(st + (srow0 + 32 * i) * LDS_ROW + kc * 16) = ra[i];
            *(frag_t*)(st + BM * LDS_ROW + (srow0 + 32 * i) * LDS_ROW + kc * 16) = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- main loop: two LDS stages, next slab's global loads in flight during the MFMAs, one barrier per slab
    const int nk_all = (K + KE - 1) / KE;
    const int per_split = (nk_all + splitk - 1) / splitk;
    const int kt0 = splitk > 1 ? (int)blockIdx.z * per_split : 0;
    const int nk = splitk > 1 ? (kt0 + per_split < nk_all ? per_split : (nk_all > kt0 ? nk_all - kt0 : 0)) : nk_all;
    load_stage(kt0 * KE);
    store_stage(lds);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned char* cur = lds + (kt & 1) * STAGE;
        if (kt + 1 < nk) load_stage((kt0 + kt + 1) * KE);
        const unsigned char* la = cur + (wm * 64 + r) * LDS_ROW + hh * 16;
        const unsigned char* lw = cur + BM * LDS_ROW + (wn * 64 + r) * LDS_ROW + hh * 16;
        if constexpr (sizeof(TC) == 4 && X3) {
            // split-bf16 products (common.h): two 8-deep steps per product, every fragment split once
#pragma unroll
            for (int ks = 0; ks < 4; ks += 2) {
                SplitF32x4 sa[2][2], sw2[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        sa[u][t] = made_split4(*(const f32x4*)(la + t * 32 * LDS_ROW + (ks + u) * 32));
                        sw2[u][t] = made_split4(*(const f32x4*)(lw + t * 32 * LDS_ROW + (ks + u) * 32));
                    }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = transposed ? made_mfma_x3_16(sw2[0][nt], sw2[1][nt], sa[0][mt], sa[1][mt], acc[mt][nt])
                                                 : made_mfma_x3_16(sa[0][mt], sa[1][mt], sw2[0][nt], sw2[1][nt], acc[mt][nt]);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_t fa[2], fw[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *(const frag_t*)(la + t * 32 * LDS_ROW + ks * 32);
                fw[t] = *(const frag_t*)(lw + t * 32 * LDS_ROW + ks * 32);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if constexpr (sizeof(TC) == 2) {
                        if (transposed)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nt], fa[mt], acc[mt][nt], 0, 0, 0);
                        else
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt], fw[nt], acc[mt][nt], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (transposed)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[nt][e], fa[mt][e], acc[mt][nt], 0, 0, 0);
                            else
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][e], fw[nt][e], acc[mt][nt], 0, 0, 0);
                        }
                    }
                }
        }
        }
        if (kt + 1 < nk) store_stage(lds + ((kt + 1) & 1) * STAGE);   // other stage: nobody reads it now
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS tile (f32) -> row-contiguous, vectorised global I/O ----------
    // normal:     Ct[row = m_local][col = n_local]
    // transposed: Ct[row = n_local][col = m_local]   (MFMA operands were swapped above)
    float* Ct = (float*)lds;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int row, col;
                if (transposed) { row = wn * 64 + nt * 32 + acc_row(e, hh); col = wm * 64 + mt * 32 + r; }
                else            { row = wm * 64 + mt * 32 + acc_row(e, hh); col = wn * 64 + nt * 32 + r; }
                Ct[row * CT_LD + col] = acc[mt][nt][e];
            }
    __syncthreads();

    if (splitk > 1) {
        // raw f32 partial sums -> split_ws[split][M][N]; made_splitk_finish adds them up and runs the epilogue
        float* wsp = a.split_ws + (int64_t)blockIdx.z * M * N;
        const int cc = tid & 15;
        const int n = n0 + cc * 8;
        const bool vec = (N % 4 == 0);
        for (int i = 0; i < 8; ++i) {
            const int row = (tid >> 4) + 16 * i;
            const int m = m0 + row;
            if (m >= M) break;
            const float* cp = Ct + row * CT_LD + cc * 8;
            if (vec && n + 8 <= N) {
                *(f32x4*)(wsp + (int64_t)m * N + n) = *(const f32x4*)cp;
                *(f32x4*)(wsp + (int64_t)m * N + n + 4) = *(const f32x4*)(cp + 4);
            } else {
                for (int j = 0; j < 8 && n + j < N; ++j) wsp[(int64_t)m * N + n + j] = cp[j];
            }
        }
        return;
    }
    unsigned char* outp = (unsigned char*)seg.out;
    const int64_t out_z = z * seg.out_z_stride;
    const int rpb = (int)seg.rows_per_batch, rmod = (int)a.r_row_mod;
    const int colb = (int)seg.col_begin;
    const int odt = seg.out_dtype;

    if (!transposed) {
        const int cc = tid & 15;                       // 8-column chunk inside the tile row
        const int n = n0 + cc * 8;
        int nvalid = N - n; nvalid = nvalid > 8 ? 8 : nvalid;
        if (nvalid > 0) {
            float bv[8];
            load_bias8(a.bias, n, N, bv);
            const bool out_vec = (seg.ldo % 8 == 0) && (seg.out_batch_stride % 8 == 0) && (seg.out_z_stride % 8 == 0) &&
                                 (((uintptr_t)outp & 15) == 0) && (colb % 8 == 0);
            const bool r_vec = a.R && (a.ldr % 8 == 0) && (((uintptr_t)a.R & 15) == 0);
#pragma unroll 2
            for (int i = 0; i < 8; ++i) {
                const int row = (tid >> 4) + 16 * i;
                const int ml = m0 + row;
                if (ml >= Mv) break;
                const int m = a.row_index ? a.row_index[ml] : ml;                   // physical row from here on
                const float* cp = Ct + row * CT_LD + cc * 8;
                f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
                float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                epilogue8<true>(a, m, n, nvalid, v, bv, rmod, r_vec);
                int64_t orow;
                if (rpb > 0) {
                    const int b = m / rpb, t = m - b * rpb;
                    orow = (int64_t)b * seg.out_batch_stride + (int64_t)t * seg.ldo;
                } else {
                    orow = (int64_t)m * seg.ldo;
                }
                store8(outp, odt, out_z + orow + (n - colb), v, nvalid, out_vec);
            }
        }
    } else {
        // lanes run along m (the time axis of the transposed output): coalesced element stores
        const int col = tid & 127;                     // m_local
        const int m = m0 + col;
        if (m < M) {
            int64_t obase;
            if (rpb > 0) {
                const int b = m / rpb, t = m - b * rpb;
                obase = (int64_t)b * seg.out_batch_stride + t;
            } else {
                obase = m;
            }
            const bool zero = a.out_row_mask != nullptr && a.out_row_mask[m] == 0.f;
            const int rr = rmod > 0 ? m % rmod : m;
            for (int i = 0; i < 64; ++i) {
                const int row = (tid >> 7) + 2 * i;    // n_local
                const int n = n0 + row;
                if (n >= N) break;
                float v = Ct[row * CT_LD + col];
                if (a.bias) v += a.bias[n];
                v = apply_act(v, a.act);
                if (a.R) v += load_as_f32(a.R, a.r_dtype, (int64_t)rr * a.ldr + n);
                if (zero) v = 0.f;
                store_from_f32(outp, odt, out_z + obase + (int64_t)(n - colb) * seg.ldo, v);
            }
        }
    }
}


// =================================================================================================
// Fast path: bf16 activations and weights, no register prologue (no row mask, no additive A2).
// Operands go global -> LDS directly (global_load_lds, 16 B per lane, no VGPR staging), the LDS image
// is unpadded 128-byte rows with an XOR swizzle applied on the SOURCE side (lane picks which 16-byte
// chunk of the row it fetches) and undone on the fragment reads, so ds_read_b128 stays conflict-free.
// One 32 KB stage, two barriers per slab; latency is hidden by running 4 workgroups per CU (about 100
// VGPRs, 34 KB LDS).  The epilogue streams the tile out in two 64-row halves through the same LDS.

constexpr int G_CT_LD = BN + 4;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// NST = 1: one 32 KB stage, two barriers per slab, latency hidden by 4 workgroups per CU (grids that fill the chip).
// NST = 3: three stages, slabs kt+1 and kt+2 in flight while slab kt is multiplied, one barrier per slab and counted
//          vmcnt waits -- for grids of at most one workgroup per CU (the decoder's 64-row Linears), where nothing else
//          hides the ~2 us a slab takes to arrive.
// TRAIN: the epilogue carries the training-path options (Zout / gate / dropout); the eval instantiation stays lean.
// BMT = 128: 128 x 128 tiles, waves 2 x 2 (64 x 64 each).
// BMT = 64 : 64 x 128 tiles, waves 1 x 4 (64 x 32 each), 24 KB of LDS, SIX workgroups per CU (80 VGPRs, no spills; five until the end of
//            round 3: 5.185 -> 5.145 ms per training step, A/B of the two builds on one box): for grids that would leave the CUs with only one or
//            two 128-row workgroups each (a padded batch gathered down to its valid rows) -- with NST = 1 the only thing that
//            hides a slab's latency is the OTHER workgroups of the CU, so twice as many, half as tall, run faster.
template <int NST, bool TRAIN, int BMT, typename TC = bf16_t, bool X3 = false>      // X3 (TC = float only): split-bf16 products
__global__ __launch_bounds__(NTHREADS, NST == 1 ? (BMT == 64 ? 6 : (TRAIN ? 3 : 4)) : 1) void linear_glds_kernel(const MadeLinearArgs a) {   // (128-row tiles with the training epilogue: three per CU -- at four the 128-register budget spilled 56 bytes)
    // TC = float (round 4): the f32 parity mode's large Linears on the same loop -- a 128-byte slab row is 32 f32, a 16-byte fragment four
    // consecutive k of which v_mfma_f32_32x32x2_f32 takes one per instruction (the same k from both operands, so any k order is a valid sum)
    constexpr int PER16 = 16 / (int)sizeof(TC), KE = KB / (int)sizeof(TC);
    constexpr int STAGE = (BMT + BN) * KB;
    constexpr int NT = BMT == 128 ? 2 : 1;                 // 32-column accumulator tiles per wave
    constexpr int AP = BMT / 32;                           // 1 KB A pieces per wave per slab
    constexpr int EROWS = BMT == 128 ? 64 : 32;            // rows per epilogue pass
    constexpr int CT_BYTES = EROWS * G_CT_LD * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST == 1 ? (CT_BYTES > STAGE ? CT_BYTES : STAGE) : NST * STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = BMT == 128 ? wave >> 1 : 0, wn = BMT == 128 ? wave & 1 : wave;

    const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
    const int n_tiles = (N + BN - 1) / BN;
    // XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its L2); give each XCD a contiguous run of
    // tiles so the n-tiles of one 128-row activation panel hit the same L2 (bijective for any tile count)
    int Mv = M;                                            // row gather (see linear_kernel): only the first ceil(Mv/128) row tiles exist
    if (a.n_rows) { const int nv = *a.n_rows; Mv = nv < M ? nv : M; }
    const int nwg = ((Mv + BMT - 1) / BMT) * n_tiles;        // live tiles; workgroups are dispatched round-robin over the XCDs in
    if ((int)blockIdx.x >= nwg) return;                    // blockIdx order, so the first nwg of them spread evenly
    int tile_id;
    {
        const int xcd = blockIdx.x & 7, q = nwg >> 3, rem = nwg & 7;
        tile_id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
    }
    const int tile_m = tile_id / n_tiles, tile_n = tile_id % n_tiles;
    const int m0 = tile_m * BMT, n0 = tile_n * BN;
    const int64_t z = blockIdx.z;

    int si = 0;
#pragma unroll
    for (int s = 1; s < 4; ++s)
        if (s < a.nseg && n0 >= a.seg[s].col_begin) si = s;
    const MadeLinearSeg seg = a.seg[si];


    // ---- padded tiles: if every row of this tile is padding, there is nothing to compute.  Every wave looks at all 128
    // rows itself (2 per lane), so the answer is wave-uniform and identical in the 4 waves: no barrier on this path.
    if (a.tile_skip_mask) {
        const int g0 = m0 + lane, g1 = m0 + 64 + lane;
        const float v0 = g0 < M ? a.tile_skip_mask[g0] : 0.f, v1 = (BMT == 128 && g1 < M) ? a.tile_skip_mask[g1] : 0.f;
        if (!__any(v0 != 0.f || v1 != 0.f)) {
            if (a.out_row_mask && !seg.transposed && (a.split_k <= 1)) {       // consumers expect zeros in masked rows
                const int rpb0 = (int)seg.rows_per_batch;
                for (int idx = tid; idx < BMT * (BN / 8); idx += NTHREADS) {
                    const int row = idx / (BN / 8), c8 = idx % (BN / 8);
                    const int m = m0 + row, n = n0 + c8 * 8;
                    if (m >= M || n >= N) continue;
                    int64_t orow;
                    if (rpb0 > 0) { const int b = m / rpb0, t = m - b * rpb0; orow = (int64_t)b * seg.out_batch_stride + (int64_t)t * seg.ldo; }
                    else orow = (int64_t)m * seg.ldo;
                    const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    int nv = N - n; nv = nv > 8 ? 8 : nv;
                    store8(seg.out, seg.out_dtype, blockIdx.z * seg.out_z_stride + orow + (n - (int)seg.col_begin), zero8, nv, false);
                }
            }
            return;
        }
    }

    // ---- per-lane source pointers: wave w issues 1 KB pieces j = 4w+i (i = 0..3) of A and of W; piece j = rows
    // 8j..8j+7; lane l -> row 8j + l/8, LDS slot l%8 holding global chunk (l%8) ^ swz(row)
    const TC* Abase = ((seg.use_a2 && a.A2 && a.a2_replace) ? (const TC*)a.A2 : (const TC*)a.A) + z * a.a_z_stride;
    const int64_t lda = (seg.use_a2 && a.A2 && a.a2_replace) ? a.lda2 : a.lda;
    const TC* Wbase = (const TC*)a.W + z * a.w_z_stride;
    const TC* pa[AP];
    const TC* pw[4];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int row = 8 * (AP * wave + i) + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        int gm = m0 + row; gm = gm < Mv ? gm : Mv - 1;        // rows past the edge are fetched from a valid row and never stored
        if (a.row_index) gm = a.row_index[gm];
        pa[i] = Abase + (int64_t)gm * lda + chunk * PER16;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (4 * wave + i) + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
        pw[i] = Wbase + (int64_t)gn * a.ldw + chunk * PER16;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment read offsets: row rr, 16-byte chunk c = 2*ks + hh -> rr*128 + ((c ^ swz(rr)) * 16)
    int offa[2], offw[NT], sa[2], sw[NT];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = wm * 64 + t * 32 + r;
        offa[t] = ra * KB; sa[t] = swz(ra);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int rw = wn * (NT * 32) + t * 32 + r;
        offw[t] = BMT * KB + rw * KB; sw[t] = swz(rw);
    }

    const int nk = K / KE;
    auto issue = [&](int kt, unsigned char* st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AP; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pa[i] + kt * KE), (lds_ptr_t)(st + (AP * wave + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pw[i] + kt * KE), (lds_ptr_t)(st + BMT * KB + (4 * wave + i) * 1024), 16, 0, 0);
    };
    auto multiply = [&](const unsigned char* st) __attribute__((always_inline)) {
        if constexpr (sizeof(TC) == 4 && X3) {
            // split-bf16 products (common.h): two 8-deep steps per product, every fragment split once
#pragma unroll
            for (int kq = 0; kq < 4; kq += 2) {
                SplitF32x4 sa2[2][2], sw2[2][NT];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = 2 * (kq + u) + hh;
#pragma unroll
                    for (int t = 0; t < 2; ++t) sa2[u][t] = made_split4(*(const f32x4*)(st + offa[t] + ((c ^ sa[t]) << 4)));
#pragma unroll
                    for (int t = 0; t < NT; ++t) sw2[u][t] = made_split4(*(const f32x4*)(st + offw[t] + ((c ^ sw[t]) << 4)));
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = made_mfma_x3_16(sa2[0][mt], sa2[1][mt], sw2[0][nt], sw2[1][nt], acc[mt][nt]);
            }
            return;
        }
        if constexpr (sizeof(TC) == 4) {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                f32x4 fa[2], fw[NT];
                const int c = 2 * kq + hh;
#pragma unroll
                for (int t = 0; t < 2; ++t) fa[t] = *(const f32x4*)(st + offa[t] + ((c ^ sa[t]) << 4));
#pragma unroll
                for (int t = 0; t < NT; ++t) fw[t] = *(const f32x4*)(st + offw[t] + ((c ^ sw[t]) << 4));
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][j], fw[nt][j], acc[mt][nt], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[2], fw[NT];
            const int c = 2 * ks + hh;
#pragma unroll
            for (int t = 0; t < 2; ++t) fa[t] = *(const bf16x8*)(st + offa[t] + ((c ^ sa[t]) << 4));
#pragma unroll
            for (int t = 0; t < NT; ++t) fw[t] = *(const bf16x8*)(st + offw[t] + ((c ^ sw[t]) << 4));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt], fw[nt], acc[mt][nt], 0, 0, 0);
        }
    };
    if constexpr (NST == 1) {
        for (int kt = 0; kt < nk; ++kt) {
            issue(kt, lds);
            __syncthreads();                               // waits for the LDS-DMA (vmcnt(0)) of every wave
            multiply(lds);
            __syncthreads();                               // all fragment reads done before the stage is overwritten
        }
    } else {
        static_assert(BMT == 128, "the multi-stage ring counts 8 loads per wave per slab");
        // slab kt lives in stage kt % NST; 8 LDS-DMA loads per wave per slab, so "slab kt landed" == at most 8 newer loads
        // outstanding.  The barrier also proves every wave finished reading stage (kt - 1) % NST, which slab kt + 2 reuses.
        issue(0, lds);
        if (nk > 1) issue(1, lds + STAGE);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 2 < nk) issue(kt + 2, lds + ((kt + 2) % NST) * STAGE);
            multiply(lds + (kt % NST) * STAGE);
        }
        __syncthreads();                                   // the epilogue reuses the staging LDS
    }

    // ---- epilogue, two passes of EROWS rows through LDS ------------------------------------------------------
    float* Ct = (float*)lds;
    unsigned char* outp = (unsigned char*)seg.out;
    const int64_t out_z = z * seg.out_z_stride;
    const int rpb = (int)seg.rows_per_batch, rmod = (int)a.r_row_mod;
    const int colb = (int)seg.col_begin;
    const int odt = seg.out_dtype;
    const int cc = tid & 15;
    const int n = n0 + cc * 8;
    int nvalid = N - n; nvalid = nvalid > 8 ? 8 : nvalid;
    float bv[8];
    load_bias8(a.bias, n, N, bv);
    const bool out_vec = (seg.ldo % 8 == 0) && (seg.out_batch_stride % 8 == 0) && (seg.out_z_stride % 8 == 0) &&
                         (((uintptr_t)outp & 15) == 0) && (colb % 8 == 0);
    const bool r_vec = a.R && (a.ldr % 8 == 0) && (((uintptr_t)a.R & 15) == 0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (m0 + half * EROWS >= Mv) break;                // block-uniform: the decoder's 64-row problems have no second half
        if constexpr (BMT == 128) {
            if (wm == half) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            Ct[(mt * 32 + acc_row(e, hh)) * G_CT_LD + wn * 64 + nt * 32 + r] = acc[mt][nt][e];
            }
        } else {                                           // every wave holds rows 0..63 of its 32 columns: pass h = accumulator tile h
#pragma unroll
            for (int e = 0; e < 16; ++e)
                Ct[acc_row(e, hh) * G_CT_LD + wn * 32 + r] = half == 0 ? acc[0][0][e] : acc[1][0][e];
        }
        __syncthreads();
        if (nvalid > 0) {
#pragma unroll 2
            for (int i = 0; i < EROWS / 16; ++i) {
                const int row = (tid >> 4) + 16 * i;
                const int ml = m0 + half * EROWS + row;
                if (ml >= Mv) break;
                const int m = a.row_index ? a.row_index[ml] : ml;
                const float* cp = Ct + row * G_CT_LD + cc * 8;
                f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
                float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                epilogue8<TRAIN>(a, m, n, nvalid, v, bv, rmod, r_vec);
                int64_t orow;
                if (rpb > 0) {
                    const int b = m / rpb, t = m - b * rpb;
                    orow = (int64_t)b * seg.out_batch_stride + (int64_t)t * seg.ldo;
                } else {
                    orow = (int64_t)m * seg.ldo;
                }
                store8(outp, odt, out_z + orow + (n - colb), v, nvalid, out_vec);
            }
        }
        __syncthreads();
    }
}

// =================================================================================================
// Big tiles, persistent: BMB x 256 outputs per 512-thread workgroup (8 waves as 2 x 4, each BMB / 2 x 64), one workgroup per CU walking the
// launch's tiles.  For launches whose tiles fill the chip many times over (the retrieval path's per-pair Linear: millions of rows) and,
// with BMB = 128, the encoder-sized ones.  What it does differently from the single-stage kernels above:
//   * 256 x 256 tiles take in 7.8 bytes of operands per kFLOP (128 x 256: 11.7; the 64 x 128 tiles: 23) -- these launches are bound by the
//     LDS-DMA intake of a CU, not by the matrix pipe;
//   * two 64 KB (48 KB) LDS stages, slab k + 1 in flight while slab k is multiplied, ONE barrier per slab (lgkmcnt(0) in front of it: the
//     round-3 rule for raw barriers beside LDS-DMA), and the ring does not stop at a tile's end: the next tile's first slab is issued
//     before the current tile's epilogue, so the epilogue's stores run under it;
//   * operands swapped (acc = W-fragment x A-fragment): a lane holds one output ROW, its registers the columns; v_permlane32_swap pairs
//     the two lane halves' 4-column groups into 8 consecutive columns, and the tile leaves through the shared 8-column epilogue as
//     16-byte stores straight from registers -- no LDS pass, no barrier in the epilogue;
//   * the bias row of the whole problem (N <= 2048) sits in LDS for the launch.
// XCD-aware persistent order: in every round the workgroups of one XCD take consecutive tile numbers (column tile fastest), so the
// column tiles of an activation panel read it through one L2.
constexpr int BIG_BN = 256, BIG_THREADS = 512, BIG_NMAX = 2048;

// FAST: bias (+ ReLU) only, plain row-major output with 16-byte-aligned rows, N a multiple of 8 -- the epilogue is 64 straight-line groups
// of (swap, add, convert, store); the general one runs the shared 8-column epilogue with all its options per group.
// 8 consecutive values of a bf16 / f32 row at a 16-byte aligned offset (the straight-line epilogues of linear_big_kernel)
__device__ __forceinline__ void big_load8(const void* p, int dt, int64_t off, float* v) {
    if (dt == MADE_BF16) {
        const bf16x8 t = *(const bf16x8*)((const bf16_t*)p + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
    } else {
        const f32x4 t0 = *(const f32x4*)((const float*)p + off), t1 = *(const f32x4*)((const float*)p + off + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = t0[j]; v[4 + j] = t1[j]; }
    }
}
__device__ __forceinline__ void big_store8(void* p, int dt, int64_t off, const float* v) {
    if (dt == MADE_BF16) {
        bf16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (bf16_t)v[j];
        *(bf16x8*)((bf16_t*)p + off) = t;
    } else {
        *(f32x4*)((float*)p + off) = (f32x4){v[0], v[1], v[2], v[3]};
        *(f32x4*)((float*)p + off + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    }
}

template <int BMB, bool TRAIN, bool FAST>
__global__ __launch_bounds__(BIG_THREADS, 1) void linear_big_kernel(const MadeLinearArgs a) {
    constexpr int STAGE = (BMB + BIG_BN) * KB;
    constexpr int MT = BMB / 64;                           // 32-row tiles per wave
    constexpr int PA = BMB / 64, PW = BIG_BN / 64;         // 1 KB pieces (8 rows x 128 B) per wave per slab
    extern __shared__ __attribute__((aligned(16))) unsigned char blds[];
    float* bias_l = (float*)(blds + 2 * STAGE);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
    const int n_tiles = (N + BIG_BN - 1) / BIG_BN;
    int Mv = M;
    if (a.n_rows) { const int nv = *a.n_rows; Mv = nv < M ? nv : M; }
    const int nwg = ((Mv + BMB - 1) / BMB) * n_tiles;
    const int G = (int)gridDim.x;                          // a multiple of 8
    const int bperm = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    if (bperm >= nwg) return;
    for (int i = tid; i < BIG_NMAX; i += BIG_THREADS) bias_l[i] = (a.bias && i < N) ? a.bias[i] : 0.f;

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    const int nk = K / 64;
    const int rmod = (int)a.r_row_mod;
    const bool r_vec = a.R && (a.ldr % 8 == 0) && (((uintptr_t)a.R & 15) == 0);
    const uint32_t drop_thr = made_drop_threshold(a.drop.p);
    const float drop_sc = 1.f / (1.f - a.drop.p);
    const uint64_t drop_seed = (TRAIN && FAST && a.drop.p > 0.f) ? made_drop_seed(a.drop) : 0;

    // fragment read offsets (see linear_glds_kernel): row rr, 16-byte chunk c = 2 ks + hh at rr * 128 + ((c ^ swz(rr)) << 4)
    int offa[MT], offw[2], sa[MT], sw[2];
#pragma unroll
    for (int t = 0; t < MT; ++t) { const int ra = wm * (BMB / 2) + t * 32 + r; offa[t] = ra * KB; sa[t] = swz(ra); }
#pragma unroll
    for (int t = 0; t < 2; ++t) { const int rw = wn * 64 + t * 32 + r; offw[t] = BMB * KB + rw * KB; sw[t] = swz(rw); }

    // per-lane LDS-DMA sources of a tile: piece j = rows 8j .. 8j + 7 of the slab; lane l -> row 8j + l / 8, LDS slot l % 8 <- chunk (l % 8) ^ swz(row)
    const bf16_t* pa[PA];
    const bf16_t* pw[PW];
    auto tile_of = [&](int L, int& m0, int& n0) __attribute__((always_inline)) { m0 = (L / n_tiles) * BMB; n0 = (L % n_tiles) * BIG_BN; };
    auto seg_of = [&](int n0) __attribute__((always_inline)) {
        int si = 0;
#pragma unroll
        for (int s = 1; s < 4; ++s)
            if (s < a.nseg && n0 >= a.seg[s].col_begin) si = s;
        return si;
    };
    auto set_sources = [&](int m0, int n0) __attribute__((always_inline)) {
        const MadeLinearSeg& sg = a.seg[seg_of(n0)];
        const bool repl = sg.use_a2 && a.A2 && a.a2_replace;
        const bf16_t* Abase = repl ? (const bf16_t*)a.A2 : (const bf16_t*)a.A;
        const int64_t lda = repl ? a.lda2 : a.lda;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = 8 * (PA * wave + i) + (lane >> 3);
            int gm = m0 + row; gm = gm < Mv ? gm : Mv - 1;    // rows past the edge are fetched from a valid row and never stored
            if (a.row_index) gm = a.row_index[gm];
            pa[i] = Abase + (int64_t)gm * lda + (((lane & 7) ^ swz(row)) * 8);
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int row = 8 * (PW * wave + i) + (lane >> 3);
            int gn = n0 + row; gn = gn < N ? gn : N - 1;
            pw[i] = (const bf16_t*)a.W + (int64_t)gn * a.ldw + (((lane & 7) ^ swz(row)) * 8);
        }
    };
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
        unsigned char* st = blds + stage * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pa[i] + kt * 64), (lds_ptr_t)(st + (PA * wave + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pw[i] + kt * 64), (lds_ptr_t)(st + BMB * KB + (PW * wave + i) * 1024), 16, 0, 0);
    };

    int L = bperm, stage = 0;
    int m0, n0;
    tile_of(L, m0, n0);
    set_sources(m0, n0);
    issue(0, 0);
    while (true) {
        f32x16 acc[MT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        // the rows this lane finishes (one per 32-row tile): physical row numbers, requested under the flight of the first slab
        int mrow[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { const int ml = m0 + wm * (BMB / 2) + mt * 32 + r; mrow[mt] = ml < Mv ? ml : Mv - 1; }
        if (a.row_index) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) mrow[mt] = a.row_index[mrow[mt]];
        }
        const int cur_m0 = m0, cur_n0 = n0;
        const int Ln = L + G;
        const bool more = Ln < nwg;
        for (int kt = 0; kt < nk; ++kt) {
            // slab kt (the only LDS-DMA in flight) has landed; everyone is done reading the other stage
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 1 < nk) issue(kt + 1, stage ^ 1);
            else if (more) { tile_of(Ln, m0, n0); set_sources(m0, n0); issue(0, stage ^ 1); }   // the next tile's first slab flies under this tile's tail
            const unsigned char* st = blds + stage * STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 fa[MT], fw[2];
                const int c = 2 * ks + hh;
#pragma unroll
                for (int t = 0; t < MT; ++t) fa[t] = *(const bf16x8*)(st + offa[t] + ((c ^ sa[t]) << 4));
#pragma unroll
                for (int t = 0; t < 2; ++t) fw[t] = *(const bf16x8*)(st + offw[t] + ((c ^ sw[t]) << 4));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nt], fa[mt], acc[mt][nt], 0, 0, 0);
            }
            stage ^= 1;
        }
        // ---- epilogue from registers: lane (r, hh) holds row r of each 32-row tile, columns {0-3, 8-11, 16-19, 24-27} + 4 hh of each 32-column
        // tile; the lane halves swap 4-column groups (hh = 0 ends with columns 0-7 and 16-23, hh = 1 with 8-15 and 24-31)
        {
            const MadeLinearSeg& seg = a.seg[seg_of(cur_n0)];
            unsigned char* outp = (unsigned char*)seg.out;
            const int rpb = (int)seg.rows_per_batch, colb = (int)seg.col_begin, odt = seg.out_dtype;
            const bool out_vec = (seg.ldo % 8 == 0) && (seg.out_batch_stride % 8 == 0) && (((uintptr_t)outp & 15) == 0) && (colb % 8 == 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ml = cur_m0 + wm * (BMB / 2) + mt * 32 + r;
                const int m = mrow[mt];
                int64_t orow;
                if (rpb > 0) { const int b = m / rpb, t = m - b * rpb; orow = (int64_t)b * seg.out_batch_stride + (int64_t)t * seg.ldo; }
                else orow = (int64_t)m * seg.ldo;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        float v[8];
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) {
                            // (through scalar temporaries: __builtin_bit_cast applied DIRECTLY to an element of an ext_vector_type value --
                            // `__builtin_bit_cast(unsigned, acc[i])` -- reads element 0 whatever i is with this hipcc (ROCm 7.2, clang 22))
                            const float f0 = acc[mt][nt][8 * half + w2], f1 = acc[mt][nt][8 * half + 4 + w2];
                            const auto swp = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, f0), __builtin_bit_cast(unsigned, f1), false, false);
                            const unsigned s0 = swp[0], s1 = swp[1];
                            v[w2] = __builtin_bit_cast(float, s0); v[4 + w2] = __builtin_bit_cast(float, s1);
                        }
                        const int n = cur_n0 + wn * 64 + nt * 32 + 16 * half + 8 * hh;
                        int nvalid = N - n; nvalid = nvalid > 8 ? 8 : nvalid;
                        if constexpr (FAST) {
                            const f32x4 b0 = *(const f32x4*)(bias_l + (n < BIG_NMAX - 8 ? n : BIG_NMAX - 8)), b1 = *(const f32x4*)(bias_l + (n < BIG_NMAX - 8 ? n : BIG_NMAX - 8) + 4);
                            v[0] += b0[0]; v[1] += b0[1]; v[2] += b0[2]; v[3] += b0[3]; v[4] += b1[0]; v[5] += b1[1]; v[6] += b1[2]; v[7] += b1[3];
                            if constexpr (TRAIN) {
                                if (a.Zout && ml < Mv && nvalid == 8) big_store8(a.Zout, a.z_dtype, (int64_t)m * a.ldz + n, v);
                            }
                            if (a.act == MADE_ACT_RELU) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                            }
                            if constexpr (TRAIN) {
                                // the training forms the step uses on its large launches, straight-line and on whole 16-byte groups (the launcher
                                // checks the alignments): ReLU-output gate, dropout (one draw per element), residual, output row mask -- in
                                // epilogue8's order.  Rows past the edge compute on row Mv - 1's operands and are not stored.
                                const int nl = nvalid == 8 ? n : 0;           // (a column group past N: operands of group 0, nothing stored)
                                if (a.gate == MADE_GATE_RELU_OUT) {
                                    float g[8];
                                    big_load8(a.G, a.g_dtype, (int64_t)m * a.ldg + nl, g);
#pragma unroll
                                    for (int j = 0; j < 8; ++j) v[j] *= (g[j] != 0.f ? 1.f : 0.f) * a.gate_scale;
                                }
                                if (a.drop.p > 0.f) {
                                    const uint32_t kb = made_keep_bits<8>(drop_seed, a.drop.site, drop_thr, (uint64_t)m * (uint64_t)a.drop_ld + (uint64_t)n);
#pragma unroll
                                    for (int j = 0; j < 8; ++j) v[j] = ((kb >> j) & 1u) ? v[j] * drop_sc : 0.f;
                                }
                                if (a.R) {
                                    float rv[8];
                                    big_load8(a.R, a.r_dtype, (int64_t)(rmod > 0 ? m % rmod : m) * a.ldr + nl, rv);
#pragma unroll
                                    for (int j = 0; j < 8; ++j) v[j] += rv[j];
                                }
                                if (a.out_row_mask && a.out_row_mask[m] == 0.f) {
#pragma unroll
                                    for (int j = 0; j < 8; ++j) v[j] = 0.f;
                                }
                            }
                            if (ml < Mv && nvalid == 8) {
                                if (odt == MADE_BF16) {
                                    bf16x8 t8;
#pragma unroll
                                    for (int j = 0; j < 8; ++j) t8[j] = (bf16_t)v[j];
                                    *(bf16x8*)((bf16_t*)outp + orow + (n - colb)) = t8;
                                } else {
                                    f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                                    *(f32x4*)((float*)outp + orow + (n - colb)) = o0;
                                    *(f32x4*)((float*)outp + orow + (n - colb) + 4) = o1;
                                }
                            }
                            continue;
                        }
                        if (ml < Mv && nvalid > 0) {
                            const f32x4 b0 = *(const f32x4*)(bias_l + (n < BIG_NMAX - 8 ? n : BIG_NMAX - 8)), b1 = *(const f32x4*)(bias_l + (n < BIG_NMAX - 8 ? n : BIG_NMAX - 8) + 4);
                            const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                            epilogue8<TRAIN>(a, m, n, nvalid, v, bv, rmod, r_vec);
                            store8(outp, odt, orow + (n - colb), v, nvalid, out_vec);
                        }
                    }
                }
            }
        }
        if (!more) break;
        L = Ln;
    }
}

// =================================================================================================
// Skinny problems (the decoder's B*Q = 64 rows, the heads): a launch of a handful of workgroups is bound by the latency
// of its K loop, not by bandwidth or MFMA rate.  64 x 64 tiles (twice the workgroups of the 128-wide tiling) and an
// 8-stage LDS ring of 16 KB stages: for K <= 512 every slab of the problem is in flight before the first MFMA, so the
// loop costs one memory round trip plus the stream time instead of one round trip per two slabs.
constexpr int S_BM = 64, S_BN = 64, S_NST = 8;
constexpr int S_STAGE = (S_BM + S_BN) * KB;           // 16 KB
constexpr int S_CT_LD = S_BN + 4;

template <bool TRAIN>
__global__ __launch_bounds__(NTHREADS, 1) void linear_skinny_kernel(const MadeLinearArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[S_NST * S_STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
    const int n_tiles = (N + S_BN - 1) / S_BN;
    const int tile_m = blockIdx.x / n_tiles, tile_n = blockIdx.x % n_tiles;
    const int m0 = tile_m * S_BM, n0 = tile_n * S_BN;
    const int64_t z = blockIdx.z;
    int Mv = M;
    if (a.n_rows) { const int nv = *a.n_rows; Mv = nv < M ? nv : M; }
    if (m0 >= Mv) return;
    int si = 0;
#pragma unroll
    for (int s = 1; s < 4; ++s)
        if (s < a.nseg && n0 >= a.seg[s].col_begin) si = s;
    const MadeLinearSeg seg = a.seg[si];

    // wave w issues 1 KB pieces 2w, 2w+1 of A and of W per slab; piece j = rows 8j..8j+7; lane l -> row 8j + l/8, slot l%8
    // holding global chunk (l%8) ^ swz(row)
    const bool repl = seg.use_a2 && a.A2 && a.a2_replace;
    const bf16_t* Abase = (repl ? (const bf16_t*)a.A2 : (const bf16_t*)a.A) + z * a.a_z_stride;
    const int64_t lda = repl ? a.lda2 : a.lda;
    const bf16_t* Wbase = (const bf16_t*)a.W + z * a.w_z_stride;
    const bf16_t* pa[2];
    const bf16_t* pw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (2 * wave + i) + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        int gm = m0 + row; gm = gm < Mv ? gm : Mv - 1;
        if (a.row_index) gm = a.row_index[gm];
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
        pa[i] = Abase + (int64_t)gm * lda + chunk * 8;
        pw[i] = Wbase + (int64_t)gn * a.ldw + chunk * 8;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    auto issue = [&](int kt) __attribute__((always_inline)) {
        unsigned char* st = lds + (kt % S_NST) * S_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = 2 * wave + i;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pa[i] + kt * 64), (lds_ptr_t)(st + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pw[i] + kt * 64), (lds_ptr_t)(st + S_BM * KB + piece * 1024), 16, 0, 0);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int ra = wm * 32 + r, rw = wn * 32 + r;
    const int offa = ra * KB, sa = swz(ra), offw = S_BM * KB + rw * KB, sw = swz(rw);

    const int nk = K / 64;
    const int pre = nk < S_NST - 1 ? nk : S_NST - 1;
    for (int kt = 0; kt < pre; ++kt) issue(kt);
    for (int kt = 0; kt < nk; ++kt) {
        // slab kt has landed once at most 4 * (slabs issued after it) loads of this wave are outstanding
        int ahead = (kt + S_NST - 1 < nk ? kt + S_NST - 1 : nk) - (kt + 1);
        switch (ahead) {
            case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // also: every wave is done reading stage (kt - 1) % S_NST
        if (kt + S_NST - 1 < nk) issue(kt + S_NST - 1);
        const unsigned char* st = lds + (kt % S_NST) * S_STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = 2 * ks + hh;
            const bf16x8 fa = *(const bf16x8*)(st + offa + ((c ^ sa) << 4));
            const bf16x8 fw = *(const bf16x8*)(st + offw + ((c ^ sw) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fw, acc, 0, 0, 0);
        }
    }
    __syncthreads();

    float* Ct = (float*)lds;
#pragma unroll
    for (int e = 0; e < 16; ++e) Ct[(wm * 32 + acc_row(e, hh)) * S_CT_LD + wn * 32 + r] = acc[e];
    __syncthreads();
    unsigned char* outp = (unsigned char*)seg.out;
    const int64_t out_z = z * seg.out_z_stride;
    const int rpb = (int)seg.rows_per_batch, rmod = (int)a.r_row_mod;
    const int colb = (int)seg.col_begin;
    const int cc = tid & 7;
    const int n = n0 + cc * 8;
    int nvalid = N - n; nvalid = nvalid > 8 ? 8 : nvalid;
    if (nvalid <= 0) return;
    float bv[8];
    load_bias8(a.bias, n, N, bv);
    const bool out_vec = (seg.ldo % 8 == 0) && (seg.out_batch_stride % 8 == 0) && (seg.out_z_stride % 8 == 0) &&
                         (((uintptr_t)outp & 15) == 0) && (colb % 8 == 0);
    const bool r_vec = a.R && (a.ldr % 8 == 0) && (((uintptr_t)a.R & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const int ml = m0 + row;
        if (ml >= Mv) break;
        const int m = a.row_index ? a.row_index[ml] : ml;
        const float* cp = Ct + row * S_CT_LD + cc * 8;
        f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
        float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        epilogue8<TRAIN>(a, m, n, nvalid, v, bv, rmod, r_vec);
        int64_t orow;
        if (rpb > 0) {
            const int b = m / rpb, t = m - b * rpb;
            orow = (int64_t)b * seg.out_batch_stride + (int64_t)t * seg.ldo;
        } else {
            orow = (int64_t)m * seg.ldo;
        }
        store8(outp, seg.out_dtype, out_z + orow + (n - colb), v, nvalid, out_vec);
    }
}

// =================================================================================================
// Tiny-M problems, second form: no LDS staging at all.  A 64-row x 32-column output tile per workgroup; its four waves split
// K four ways and load their MFMA fragments straight from global memory (a 32x32x16 fragment is 16 contiguous bytes of one
// row per lane), four K steps of fragments in flight while the previous four are multiplied; the four partial tiles meet in
// LDS and the epilogue runs on all 256 threads (one row x 8 columns each).  Against the 64 x 64 ring kernel above: twice the
// workgroups, a quarter of the serial K steps per wave, no barrier inside the K loop -- a K = 512 Linear on 64 rows is one
// memory round trip long.
constexpr int T_BN = 32, T_BM = 64, T_CH = 4;
constexpr int T_CT_LD = T_BN + 4;

template <bool TRAIN>
__device__ __forceinline__ void linear_tiny_body(const MadeLinearArgs& a, const int bx, const int bz, float* Ct /* [4 * T_BM * T_CT_LD] LDS */) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
    const int n_tiles = (N + T_BN - 1) / T_BN;
    const int tile_m = bx / n_tiles, tile_n = bx % n_tiles;
    const int m0 = tile_m * T_BM, n0 = tile_n * T_BN;
    const int64_t z = bz;
    int Mv = M;
    if (a.n_rows) { const int nv = *a.n_rows; Mv = nv < M ? nv : M; }
    if (m0 >= Mv) return;
    int si = 0;
#pragma unroll
    for (int s = 1; s < 4; ++s)
        if (s < a.nseg && n0 >= a.seg[s].col_begin) si = s;
    const MadeLinearSeg seg = a.seg[si];

    const bool repl = seg.use_a2 && a.A2 && a.a2_replace;
    const bf16_t* Abase = (repl ? (const bf16_t*)a.A2 : (const bf16_t*)a.A) + z * a.a_z_stride;
    const int64_t lda = repl ? a.lda2 : a.lda;
    const int steps = K / 64;                              // 16-deep K steps per wave (K is a multiple of 64 on this path)
    const int kw = wave * steps * 16 + hh * 8;             // this lane's first K index
    const bf16_t* pa[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int gm = m0 + t * 32 + r; gm = gm < Mv ? gm : Mv - 1;   // rows past the edge are fetched from a valid row and never stored
        if (a.row_index) gm = a.row_index[gm];
        pa[t] = Abase + (int64_t)gm * lda + kw;
    }
    int gn = n0 + r; gn = gn < N ? gn : N - 1;
    const bf16_t* pw = (const bf16_t*)a.W + z * a.w_z_stride + (int64_t)gn * a.ldw + kw;

    // the epilogue's inputs (this thread finishes row tid / 4, 8 columns): physical row, bias, residual, gate, row mask -- all
    // requested now, so they travel with the operand fragments instead of costing a round trip each after the K loop
    const int e_cc = tid & 3, e_row = tid >> 2;
    const int e_n = n0 + e_cc * 8;
    int e_nvalid = N - e_n; e_nvalid = e_nvalid > 8 ? 8 : e_nvalid;
    const int e_ml = m0 + e_row;
    int e_m = e_ml < Mv ? e_ml : Mv - 1;
    if (a.row_index) e_m = a.row_index[e_m];
    float e_bv[8];
    load_bias8(a.bias_row_scale ? a.bias + z * a.bias_z_stride : a.bias, e_n, N, e_bv);
    float e_bsc = 1.f;                                      // per-(row, problem) bias scale (MadeLinearArgs.bias_row_scale)
    if (a.bias_row_scale) e_bsc = a.bias_row_scale[(int64_t)e_m * a.batch + z];
    const int e_rmod = (int)a.r_row_mod;
    const bool e_rpref = a.R && a.r_dtype == MADE_BF16 && (a.ldr % 8 == 0) && (((uintptr_t)a.R & 15) == 0) && e_nvalid == 8;
    const bool e_gpref = TRAIN && a.gate != MADE_GATE_NONE && a.g_dtype == MADE_BF16 && (a.ldg % 8 == 0) && (((uintptr_t)a.G & 15) == 0) && e_nvalid == 8;
    bf16x8 e_r = {}, e_g = {};
    if (e_rpref) e_r = *(const bf16x8*)((const bf16_t*)a.R + (int64_t)(e_rmod > 0 ? e_m % e_rmod : e_m) * a.ldr + e_n);
    if constexpr (TRAIN) { if (e_gpref) e_g = *(const bf16x8*)((const bf16_t*)a.G + (int64_t)e_m * a.ldg + e_n); }
    float e_om = 1.f;
    if (a.out_row_mask) e_om = a.out_row_mask[e_m];

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    bf16x8 fa[2][T_CH][2], fw[2][T_CH];
    auto load = [&](int buf, int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < T_CH; ++i) {
            const int s = s0 + i < steps ? s0 + i : steps - 1;       // past the end: a harmless re-read, never multiplied
            fw[buf][i] = *(const bf16x8*)(pw + s * 16);
            fa[buf][i][0] = *(const bf16x8*)(pa[0] + s * 16);
            fa[buf][i][1] = *(const bf16x8*)(pa[1] + s * 16);
        }
    };
    auto mul = [&](int buf, int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < T_CH; ++i) {
            if (s0 + i < steps) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][i][0], fw[buf][i], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][i][1], fw[buf][i], acc[1], 0, 0, 0);
            }
        }
    };
    load(0, 0);
    for (int s0 = 0; s0 < steps; s0 += 2 * T_CH) {
        if (s0 + T_CH < steps) load(1, s0 + T_CH);
        mul(0, s0);
        if (s0 + 2 * T_CH < steps) load(0, s0 + 2 * T_CH);
        if (s0 + T_CH < steps) mul(1, s0 + T_CH);
    }

    float* mine = Ct + wave * (T_BM * T_CT_LD);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[(t * 32 + acc_row(e, hh)) * T_CT_LD + r] = acc[t][e];
    __syncthreads();

    unsigned char* outp = (unsigned char*)seg.out;
    const int64_t out_z = z * seg.out_z_stride;
    const int rpb = (int)seg.rows_per_batch, rmod = (int)a.r_row_mod;
    const int colb = (int)seg.col_begin;
    const int cc = e_cc, row = e_row, n = e_n, nvalid = e_nvalid, ml = e_ml;
    if (nvalid <= 0 || ml >= Mv) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float* cp = Ct + w * (T_BM * T_CT_LD) + row * T_CT_LD + cc * 8;
        const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
        v[0] += c0[0]; v[1] += c0[1]; v[2] += c0[2]; v[3] += c0[3];
        v[4] += c1[0]; v[5] += c1[1]; v[6] += c1[2]; v[7] += c1[3];
    }
    const bool out_vec = (seg.ldo % 8 == 0) && (seg.out_batch_stride % 8 == 0) && (seg.out_z_stride % 8 == 0) &&
                         (((uintptr_t)outp & 15) == 0) && (colb % 8 == 0);
    const bool r_vec = a.R && (a.ldr % 8 == 0) && (((uintptr_t)a.R & 15) == 0);
    const int m = e_m;
    if (a.bias_row_scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) e_bv[j] *= e_bsc;
    }
    EpiPre pre;
    pre.has_r = e_rpref; pre.r = e_r; pre.has_g = e_gpref; pre.g = e_g; pre.has_om = true; pre.om = e_om;
    epilogue8<TRAIN>(a, m, n, nvalid, v, e_bv, rmod, r_vec, pre);
    int64_t orow;
    if (rpb > 0) {
        const int b = m / rpb, t = m - b * rpb;
        orow = (int64_t)b * seg.out_batch_stride + (int64_t)t * seg.ldo;
    } else {
        orow = (int64_t)m * seg.ldo;
    }
    store8(outp, seg.out_dtype, out_z + orow + (n - colb), v, nvalid, out_vec);
}

template <bool TRAIN>
__global__ __launch_bounds__(NTHREADS) void linear_tiny_kernel(const MadeLinearArgs a) {
    __shared__ __attribute__((aligned(16))) float Ct[4 * T_BM * T_CT_LD];
    linear_tiny_body<TRAIN>(a, blockIdx.x, blockIdx.z, Ct);
}

// =================================================================================================
// Tiny-M problems, third form (round 3): 16-row x 16-column output tiles.  What a 64-row stage of the decoder's chain costs is
// the bytes ONE workgroup pulls through its CU's memory queue -- measured (tools/tiny_model_probe.py, chains of dependent launches):
// t = 3.4 us + bytes / ~33 GB/s, i.e. 6.2 us for the 64 x 32 tiles above at K = 512 (64 KB of rows + 32 KB of weights per
// workgroup) -- not the arithmetic.  Here a workgroup reads 16 rows + 16 weight rows (32 KB at K = 512) and there are eight times as
// many of them; v_mfma_f32_16x16x32_bf16, the four waves take every fourth 32-deep K step, all of a lane's fragments (16 bytes of
// one row each, up to 16 loads) are requested before the first wait, the four partial tiles meet in LDS and 32 lanes finish one row
// x 8 columns each with epilogue8 (the element order of the dropout draws etc. does not depend on the tiling).
// Workgroups b and b + 8 share an XCD (observed dispatch order; speed only): the row tiles of one weight slice are put there.
constexpr int U_BM = 16, U_BN = 16, U_CT_LD = U_BN + 4;

template <bool TRAIN, int SPW>                             // SPW: 32-deep K steps per wave (K <= 128 * SPW)
__global__ __launch_bounds__(NTHREADS) void linear_t16_kernel(const MadeLinearArgs a) {
    __shared__ __attribute__((aligned(16))) float Ct[4 * U_BM * U_CT_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
    const int n_tiles = (N + U_BN - 1) / U_BN, m_tiles = (M + U_BM - 1) / U_BM;
    int tile_m, tile_n;
    {
        const int bx = blockIdx.x;
        if ((n_tiles & 7) == 0) { const int j = bx >> 3; tile_m = j % m_tiles; tile_n = (j / m_tiles) * 8 + (bx & 7); }
        else { tile_m = bx % m_tiles; tile_n = bx / m_tiles; }
    }
    const int m0 = tile_m * U_BM, n0 = tile_n * U_BN;
    const int64_t z = blockIdx.z;
    int Mv = M;
    if (a.n_rows) { const int nv = *a.n_rows; Mv = nv < M ? nv : M; }
    if (m0 >= Mv) return;
    int si = 0;
#pragma unroll
    for (int s = 1; s < 4; ++s)
        if (s < a.nseg && n0 >= a.seg[s].col_begin) si = s;
    const MadeLinearSeg seg = a.seg[si];
    const bool repl = seg.use_a2 && a.A2 && a.a2_replace;
    const bf16_t* Abase = (repl ? (const bf16_t*)a.A2 : (const bf16_t*)a.A) + z * a.a_z_stride;
    const int64_t lda = repl ? a.lda2 : a.lda;
    const int steps = K / 32;                              // (K is a multiple of 32 on this path)
    int gm = m0 + r16; gm = gm < Mv ? gm : Mv - 1;         // rows past the edge: a valid row, never stored
    if (a.row_index) gm = a.row_index[gm];
    int gn = n0 + r16; gn = gn < N ? gn : N - 1;
    const bf16_t* pa = Abase + (int64_t)gm * lda + kq * 8;
    const bf16_t* pw = (const bf16_t*)a.W + z * a.w_z_stride + (int64_t)gn * a.ldw + kq * 8;
    bf16x8 fa[SPW], fw[SPW];
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
        int s = wave + 4 * i; s = s < steps ? s : steps - 1;      // past the end: a harmless re-read, never multiplied
        fw[i] = *(const bf16x8*)(pw + s * 32);
        fa[i] = *(const bf16x8*)(pa + s * 32);
    }
    // the epilogue's inputs (lanes 0..31 of wave 0 finish row lane / 2, 8 columns each), requested with the fragments
    const int e_cc = tid & 1, e_row = (tid >> 1) & 15;
    const int e_n = n0 + e_cc * 8;
    int e_nvalid = N - e_n; e_nvalid = e_nvalid > 8 ? 8 : e_nvalid;
    const int e_ml = m0 + e_row;
    int e_m = e_ml < Mv ? e_ml : Mv - 1;
    if (a.row_index) e_m = a.row_index[e_m];
    float e_bv[8];
    load_bias8(a.bias_row_scale ? a.bias + z * a.bias_z_stride : a.bias, e_n, N, e_bv);
    float e_bsc = 1.f;
    if (a.bias_row_scale) e_bsc = a.bias_row_scale[(int64_t)e_m * a.batch + z];
    const int e_rmod = (int)a.r_row_mod;
    const bool e_rpref = a.R && a.r_dtype == MADE_BF16 && (a.ldr % 8 == 0) && (((uintptr_t)a.R & 15) == 0) && e_nvalid == 8;
    const bool e_gpref = TRAIN && a.gate != MADE_GATE_NONE && a.g_dtype == MADE_BF16 && (a.ldg % 8 == 0) && (((uintptr_t)a.G & 15) == 0) && e_nvalid == 8;
    bf16x8 e_r = {}, e_g = {};
    if (e_rpref) e_r = *(const bf16x8*)((const bf16_t*)a.R + (int64_t)(e_rmod > 0 ? e_m % e_rmod : e_m) * a.ldr + e_n);
    if constexpr (TRAIN) { if (e_gpref) e_g = *(const bf16x8*)((const bf16_t*)a.G + (int64_t)e_m * a.ldg + e_n); }
    float e_om = 1.f;
    if (a.out_row_mask) e_om = a.out_row_mask[e_m];
    uint64_t e_seed = 0;
    if constexpr (TRAIN) { if (a.drop.p > 0.f) e_seed = made_drop_seed(a.drop); }

    f32x4 acc;
    acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
#pragma unroll
    for (int i = 0; i < SPW; ++i)
        if (wave + 4 * i < steps) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fw[i], acc, 0, 0, 0);
    float* mine = Ct + wave * (U_BM * U_CT_LD);
#pragma unroll
    for (int e = 0; e < 4; ++e) mine[(kq * 4 + e) * U_CT_LD + r16] = acc[e];
    __syncthreads();
    if (tid >= 32) return;
    const int n = e_n, nvalid = e_nvalid;
    if (nvalid <= 0 || e_ml >= Mv) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float* cp = Ct + w * (U_BM * U_CT_LD) + e_row * U_CT_LD + e_cc * 8;
        const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
        v[0] += c0[0]; v[1] += c0[1]; v[2] += c0[2]; v[3] += c0[3];
        v[4] += c1[0]; v[5] += c1[1]; v[6] += c1[2]; v[7] += c1[3];
    }
    unsigned char* outp = (unsigned char*)seg.out;
    const int64_t out_z = z * seg.out_z_stride;
    const int rpb = (int)seg.rows_per_batch, rmod = (int)a.r_row_mod;
    const int colb = (int)seg.col_begin;
    const bool out_vec = (seg.ldo % 8 == 0) && (seg.out_batch_stride % 8 == 0) && (seg.out_z_stride % 8 == 0) &&
                         (((uintptr_t)outp & 15) == 0) && (colb % 8 == 0);
    const bool r_vec = a.R && (a.ldr % 8 == 0) && (((uintptr_t)a.R & 15) == 0);
    const int m = e_m;
    if (a.bias_row_scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) e_bv[j] *= e_bsc;
    }
    EpiPre pre;
    pre.has_r = e_rpref; pre.r = e_r; pre.has_g = e_gpref; pre.g = e_g; pre.has_om = true; pre.om = e_om;
    pre.has_seed = TRAIN && a.drop.p > 0.f; pre.seed = e_seed;
    epilogue8<TRAIN>(a, m, n, nvalid, v, e_bv, rmod, r_vec, pre);
    int64_t orow;
    if (rpb > 0) {
        const int b = m / rpb, t = m - b * rpb;
        orow = (int64_t)b * seg.out_batch_stride + (int64_t)t * seg.ldo;
    } else {
        orow = (int64_t)m * seg.ldo;
    }
    store8(outp, seg.out_dtype, out_z + orow + (n - colb), v, nvalid, out_vec);
}

template <bool TRAIN>
static void launch_t16(const MadeLinearArgs& a, hipStream_t st) {
    const int steps = (int)(a.K / 32), spw = (steps + 3) / 4;
    dim3 g((unsigned)(((a.M + U_BM - 1) / U_BM) * ((a.N + U_BN - 1) / U_BN)), 1, (unsigned)a.batch), block(NTHREADS);
    if (spw <= 1) hipLaunchKernelGGL((linear_t16_kernel<TRAIN, 1>), g, block, 0, st, a);
    else if (spw <= 2) hipLaunchKernelGGL((linear_t16_kernel<TRAIN, 2>), g, block, 0, st, a);
    else if (spw <= 4) hipLaunchKernelGGL((linear_t16_kernel<TRAIN, 4>), g, block, 0, st, a);
    else hipLaunchKernelGGL((linear_t16_kernel<TRAIN, 8>), g, block, 0, st, a);
}

}  // namespace

// tuning knob for the micro-benchmarks and tests: MADE_LINEAR_TILE=64|128 forces the single-stage direct-to-LDS kernels of round 1,
// 2128 the ring kernel (opt-in, see pick_variant), 1000 round 1's choice between its two kernels, 32 the 64 x 32-tile kernel where the
// 16 x 16-tile one is the default (at most 64 rows)

static int tile_pref() {                                   // read on every call: the tests switch kernels inside one process
    const char* e = made_variant_env("MADE_LINEAR_TILE");
    return e ? atoi(e) : 0;
}

static int64_t tiny_max() {                                // most 64 x 32 tiles for which the fragments-from-global kernel is chosen (MADE_TINY_MAX: knob for measurements)
    const char* e = made_variant_env("MADE_TINY_MAX");
    return e ? (int64_t)atoll(e) : 1024;
}

static int64_t f32_glds_min() {                            // least number of 64 x 128 tiles for the f32 LDS-DMA kernel (MADE_LINEAR_F32_GLDS_MIN: knob for measurements)
    const char* e = made_variant_env("MADE_LINEAR_F32_GLDS_MIN");
    return e ? (int64_t)atoll(e) : 1;                      // measured (f32 training step): 32 -> 27.8 ms, 8 -> 25.4 ms, 1 -> 19.5 ms (the general kernel: 35.2 ms)
}

static int64_t big_train_min() {                           // read on every call (A/B inside one process)
    const char* e = made_variant_env("MADE_LINEAR_BIG_TRAIN");
    return e ? (int64_t)atoll(e) : 0;
}

static int64_t t16_max() {                                 // (MADE_T16_MAX: knob for measurements)
    static const int64_t v = [] { const char* e = made_variant_env("MADE_T16_MAX"); return e ? (int64_t)atoll(e) : (int64_t)4096; }();
    return v;
}

// linear_big_kernel's straight-line epilogues serve: bias, ReLU, and (training) pre-activation copy, ReLU-output gate, per-element dropout,
// residual, output row mask -- on plain row-major rows of whole, 16-byte aligned groups of 8 outputs
static bool big_fast_epilogue(const MadeLinearArgs& a) {
    auto al = [](const void* p, int64_t ld, int dt) { return (((uintptr_t)p & 15) == 0) && ld % (dt == MADE_BF16 ? 8 : 4) == 0; };
    bool ok = (a.act == MADE_ACT_NONE || a.act == MADE_ACT_RELU) && a.N % 8 == 0 && (a.gate == MADE_GATE_NONE || a.gate == MADE_GATE_RELU_OUT);
    if (a.gate != MADE_GATE_NONE) ok = ok && al(a.G, a.ldg, a.g_dtype);
    if (a.Zout) ok = ok && al(a.Zout, a.ldz, a.z_dtype);
    if (a.R) ok = ok && al(a.R, a.ldr, a.r_dtype);
    if (a.drop.p > 0.f) ok = ok && a.drop_col_div <= 1;
    for (int s = 0; s < a.nseg; ++s)
        ok = ok && a.seg[s].rows_per_batch == 0 && a.seg[s].col_begin % 8 == 0 && al(a.seg[s].out, a.seg[s].ldo, a.seg[s].out_dtype);
    return ok;
}

// which kernel made_linear runs for these arguments (one place: the launcher and made_linear_variant both ask here)
static int pick_variant(const MadeLinearArgs& a) {
    if (a.w_dtype != MADE_BF16) {
        // f32 weights (the parity mode).  Its large launches take the LDS-DMA loop (round 4: the register-staged general kernel ran them at
        // 0.11 of the f32 MFMA peak); anything with an option that loop does not serve, and every small problem, stays on the general kernel.
        bool fast = a.a_dtype == MADE_F32 && a.K % 32 == 0 && a.a_row_mask == nullptr && a.split_k <= 1 && (a.A2 == nullptr || a.a2_replace) &&
                    ((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.W % 16 == 0) && (a.lda % 4 == 0) && (a.ldw % 4 == 0) && a.bias_row_scale == nullptr &&
                    made_variant_env("MADE_LINEAR_F32_GLDS_OFF") == nullptr;
        for (int s = 0; s < a.nseg; ++s) fast = fast && !a.seg[s].transposed;
        if (a.A2 && a.a2_replace) fast = fast && (a.a2_row_mod == 0) && (a.lda2 % 4 == 0) && ((uintptr_t)a.A2 % 16 == 0);
        if (a.batch > 1) fast = fast && (a.a_z_stride % 4 == 0) && (a.w_z_stride % 4 == 0);
        const int64_t tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
        const int64_t tiles64 = ((a.M + 63) / 64) * ((a.N + BN - 1) / BN);
        if (fast && tiles64 * a.batch >= f32_glds_min()) {    // (every such launch: even the decoder chain's four-workgroup problems run three times
            // faster on the LDS-DMA loop than on the register-staged 128 x 128 tiles)
            const int64_t live = a.row_index ? (tiles * a.batch * 9) / 16 : tiles * a.batch;
            return live > 1280 ? MADE_LINEAR_GLDS128_F32 : MADE_LINEAR_GLDS64_F32;
        }
        return MADE_LINEAR_GENERAL_F32;
    }
    bool fast = a.a_dtype == MADE_BF16 && a.K % 64 == 0 && a.a_row_mask == nullptr && a.split_k <= 1 &&
                (a.A2 == nullptr || a.a2_replace) && ((uintptr_t)a.A % 16 == 0) && (a.lda % 8 == 0);
    for (int s = 0; s < a.nseg; ++s) fast = fast && !a.seg[s].transposed;
    if (a.A2 && a.a2_replace) fast = fast && (a.a2_row_mod == 0) && (a.lda2 % 8 == 0);
    if (!fast) return a.a_dtype == MADE_F32 ? MADE_LINEAR_GENERAL_F32IN : MADE_LINEAR_GENERAL_BF16;
    const int64_t tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const int64_t tiles64 = ((a.M + S_BM - 1) / S_BM) * ((a.N + S_BN - 1) / S_BN);
    const int64_t tiles32 = ((a.M + T_BM - 1) / T_BM) * ((a.N + T_BN - 1) / T_BN);
    // one 64-row tile (the decoder's chain, its per-head batches): 16 x 16 tiles, a third of the bytes per workgroup
    // one 64-row tile (the decoder's chain, its per-head batches): 16 x 16 tiles, a third of the bytes per workgroup
    // (MADE_LINEAR_TILE=32: the 64 x 32-tile kernel instead, for A/B measurements)
    if (a.M <= 64 && a.K >= 128 && a.K <= 1024 && a.K % 32 == 0 && a.tile_skip_mask == nullptr && tile_pref() != 1 && tile_pref() != 32 &&
        ((a.M + 15) / 16) * ((a.N + 15) / 16) * a.batch <= t16_max())
        return MADE_LINEAR_TINY16;
    if (tiles32 * a.batch <= tiny_max() && a.K <= 1024 && a.tile_skip_mask == nullptr && tile_pref() != 1) return MADE_LINEAR_TINY;
    if (tiles64 * a.batch <= 256 && a.tile_skip_mask == nullptr) return MADE_LINEAR_SKINNY;
    if (tiles * a.batch <= 256) return MADE_LINEAR_GLDS3;                  // at most one workgroup per CU
    // Persistent big-tile kernel (round 4; the ring / W-stationary kernels of rounds 2-3 lost inside the step and were removed):
    // 256 x 256 tiles when the tiles fill the chip at least twice over and nothing is gathered (the retrieval path's per-pair Linear),
    // 128 x 256 tiles on request (MADE_LINEAR_TILE=256; 512 forces the 256-row tiles).
    {
        const bool train_like = a.gate != MADE_GATE_NONE || a.Zout != nullptr || a.drop.p > 0.f;
        bool big_ok = a.batch == 1 && a.tile_skip_mask == nullptr && a.N <= BIG_NMAX && a.ldw % 8 == 0 && (!train_like || big_fast_epilogue(a));   // (training forms: the straight-line epilogue only)
        for (int s = 0; s < a.nseg; ++s) big_ok = big_ok && (a.seg[s].col_begin % BIG_BN == 0) && a.seg[s].out_z_stride == 0;
        if (big_ok) {
            if (tile_pref() == 512) return MADE_LINEAR_BIG256;
            if (tile_pref() == 256) return MADE_LINEAR_BIG128;
            const int64_t tiles256 = ((a.M + 255) / 256) * ((a.N + BIG_BN - 1) / BIG_BN);
            if (tile_pref() == 0 && !a.row_index && tiles256 >= 2 * 256) return MADE_LINEAR_BIG256;
            // the training step's large launches (MADE_LINEAR_BIG_TRAIN=<least number of live 128 x 256 tiles>, 0 = off): the straight-line
            // training epilogue only -- the general one costs the big tiles more than they gain
            const int64_t tiles128 = ((a.M + 127) / 128) * ((a.N + BIG_BN - 1) / BIG_BN);
            const int64_t live128 = a.row_index ? (tiles128 * 9) / 16 : tiles128;
            if (tile_pref() == 0 && big_train_min() > 0 && a.N % BIG_BN == 0 && live128 >= big_train_min() && big_fast_epilogue(a)) return MADE_LINEAR_BIG128;
        }
    }
    // workgroups that will really run: a gathered batch keeps about half of its rows (the host does not know *n_rows)
    const int64_t live = a.row_index ? (tiles * a.batch * 9) / 16 : tiles * a.batch;
    if (tile_pref() == 64) return MADE_LINEAR_GLDS64;
    if (tile_pref() == 128) return MADE_LINEAR_GLDS128;
    if (live > 1280) return MADE_LINEAR_GLDS128;
    // Between one and five tall workgroups per CU.  Measured (tools/linear_tiles_bench.py, K = 512): 128-row tiles win by 6-8 % when they
    // fill whole rounds of the 256 x 4 resident workgroups (32768 x 512: 33.6 vs 36.0 us; x 1024: 58 vs 63) and lose by 10 %
    // when a last partial round idles most of the chip (34688 x 512: 42 vs 38) -- only knowable when the row count is (no gather)
    if (!a.row_index) {
        const int64_t slots = 256 * 4, rounds = (live + slots - 1) / slots;
        if (live * 10 >= rounds * slots * 9) return MADE_LINEAR_GLDS128;
    }
    return MADE_LINEAR_GLDS64;
}

// argument checks of made_linear
static int linear_validate(const MadeLinearArgs& a) {
    MADE_REQUIRE(a.A && a.W, "made_linear: null A or W");
    MADE_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0, "made_linear: bad dims M=%lld N=%lld K=%lld",
                 (long long)a.M, (long long)a.N, (long long)a.K);
    MADE_REQUIRE(a.nseg >= 1 && a.nseg <= 4, "made_linear: nseg=%d out of range", a.nseg);
    MADE_REQUIRE(a.batch >= 1 && a.batch <= 65535, "made_linear: batch=%lld out of range", (long long)a.batch);
    MADE_REQUIRE(a.w_dtype == MADE_F32 || a.w_dtype == MADE_BF16, "made_linear: bad w_dtype %d", a.w_dtype);
    MADE_UNSUPPORTED(!(a.a_dtype == MADE_BF16 && a.w_dtype == MADE_F32),
                     "made_linear: bf16 activations with f32 weights are not supported");
    const int per16 = a.w_dtype == MADE_F32 ? 4 : 8;
    const int a_align = a.a_dtype == MADE_F32 ? 4 : 8;
    MADE_UNSUPPORTED(a.K % per16 == 0, "made_linear: K=%lld must be a multiple of %d", (long long)a.K, per16);
    MADE_UNSUPPORTED(a.lda % a_align == 0 && a.ldw % per16 == 0 && ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0,
                     "made_linear: A/W rows must be 16-byte aligned (lda=%lld ldw=%lld)", (long long)a.lda, (long long)a.ldw);
    if (a.A2)
        MADE_UNSUPPORTED(a.lda2 % a_align == 0 && ((uintptr_t)a.A2 % 16) == 0, "made_linear: A2 rows must be 16-byte aligned");
    if (a.batch > 1)
        MADE_UNSUPPORTED(a.a_z_stride % a_align == 0 && a.w_z_stride % per16 == 0, "made_linear: batch strides must keep 16-byte alignment");
    for (int s = 0; s < a.nseg; ++s) {
        MADE_REQUIRE(a.seg[s].out != nullptr, "made_linear: segment %d has null out", s);
        MADE_REQUIRE(a.seg[s].col_begin >= 0 && a.seg[s].col_begin < a.N, "made_linear: segment %d col_begin out of range", s);
        if (s > 0) {
            MADE_REQUIRE(a.seg[s].col_begin > a.seg[s - 1].col_begin, "made_linear: segments must be ascending");
            MADE_UNSUPPORTED(a.seg[s].col_begin % BN == 0, "made_linear: segment boundaries must be multiples of %d", BN);
        } else {
            MADE_REQUIRE(a.seg[0].col_begin == 0, "made_linear: first segment must start at column 0");
        }
    }
    MADE_REQUIRE((a.row_index == nullptr) == (a.n_rows == nullptr), "made_linear: row_index and n_rows come together");
    if (a.row_index) {
        MADE_UNSUPPORTED(a.split_k <= 1 && a.batch == 1, "made_linear: row gather is not available with split-K or batches");
        for (int s = 0; s < a.nseg; ++s)
            MADE_UNSUPPORTED(!a.seg[s].transposed, "made_linear: row gather is not available on transposed segments");
    }
    if (a.bias_row_scale) {
        MADE_REQUIRE(a.bias != nullptr, "made_linear: bias_row_scale without bias");
        MADE_UNSUPPORTED(pick_variant(a) == MADE_LINEAR_TINY || pick_variant(a) == MADE_LINEAR_TINY16, "made_linear: bias_row_scale is served by the tiny-M kernels only");
    }
    if (a.gate != MADE_GATE_NONE) MADE_REQUIRE(a.G != nullptr, "made_linear: gate without G");
    if (a.gate != MADE_GATE_NONE || a.Zout || a.drop.p > 0.f) {
        MADE_REQUIRE(a.drop.p >= 0.f && a.drop.p < 1.f, "made_linear: dropout p=%f out of [0,1)", (double)a.drop.p);
        MADE_UNSUPPORTED(a.split_k <= 1, "made_linear: gate / Zout / dropout are not available with split-K");
        for (int s = 0; s < a.nseg; ++s)
            MADE_UNSUPPORTED(!a.seg[s].transposed, "made_linear: gate / Zout / dropout are not available on transposed segments");
    }
    return MADE_OK;
}

extern "C" int made_linear(const MadeLinearArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_linear: null args");
    const MadeLinearArgs& a = *args;
    { const int rc = linear_validate(a); if (rc != MADE_OK) return rc; }
    if (a.M == 0) return MADE_OK;
    const int64_t tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    MADE_UNSUPPORTED(tiles < (1LL << 31), "made_linear: too many tiles");
    if (a.split_k > 1) {
        MADE_REQUIRE(a.split_ws != nullptr, "made_linear: split_k > 1 needs split_ws");
        MADE_UNSUPPORTED(a.batch == 1 && a.nseg == 1 && !a.seg[0].transposed && a.split_k <= 256,
                         "made_linear: split-K needs batch == 1, one plain segment and split_k <= 256");
    }
    dim3 grid((unsigned)tiles, 1, (unsigned)(a.split_k > 1 ? a.split_k : a.batch)), block(NTHREADS);
    hipStream_t st = (hipStream_t)stream;
    const bool train = a.gate != MADE_GATE_NONE || a.Zout != nullptr || a.drop.p > 0.f;
    switch (pick_variant(a)) {
        case MADE_LINEAR_TINY16:
            if (train) launch_t16<true>(a, st); else launch_t16<false>(a, st);
            break;
        case MADE_LINEAR_TINY: {
            // latency-bound and short in K: fragments straight from global memory, K split over the four waves
            dim3 g32((unsigned)(((a.M + T_BM - 1) / T_BM) * ((a.N + T_BN - 1) / T_BN)), 1, (unsigned)a.batch);
            if (train) hipLaunchKernelGGL((linear_tiny_kernel<true>), g32, block, 0, st, a);
            else hipLaunchKernelGGL((linear_tiny_kernel<false>), g32, block, 0, st, a);
            break;
        }
        case MADE_LINEAR_SKINNY: {                         // latency-bound, long K: 64 x 64 tiles, all slabs in flight
            dim3 g64((unsigned)(((a.M + S_BM - 1) / S_BM) * ((a.N + S_BN - 1) / S_BN)), 1, (unsigned)a.batch);
            if (train) hipLaunchKernelGGL((linear_skinny_kernel<true>), g64, block, 0, st, a);
            else hipLaunchKernelGGL((linear_skinny_kernel<false>), g64, block, 0, st, a);
            break;
        }
        case MADE_LINEAR_GLDS3:
            if (train) hipLaunchKernelGGL((linear_glds_kernel<3, true, 128>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((linear_glds_kernel<3, false, 128>), grid, block, 0, st, a);
            break;
        case MADE_LINEAR_GLDS64: {                         // fewer than ~4 tall workgroups per CU: 64-row tiles
            dim3 g((unsigned)(((a.M + 63) / 64) * ((a.N + BN - 1) / BN)), 1, (unsigned)a.batch);
            if (train) hipLaunchKernelGGL((linear_glds_kernel<1, true, 64>), g, block, 0, st, a);
            else hipLaunchKernelGGL((linear_glds_kernel<1, false, 64>), g, block, 0, st, a);
            break;
        }
        case MADE_LINEAR_GLDS128:
            if (train) hipLaunchKernelGGL((linear_glds_kernel<1, true, 128>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((linear_glds_kernel<1, false, 128>), grid, block, 0, st, a);
            break;
        case MADE_LINEAR_GLDS64_F32: {
            dim3 g((unsigned)(((a.M + 63) / 64) * ((a.N + BN - 1) / BN)), 1, (unsigned)a.batch);
            if (train) { if (g_made_f32_products) hipLaunchKernelGGL((linear_glds_kernel<1, true, 64, float, true>), g, block, 0, st, a); else hipLaunchKernelGGL((linear_glds_kernel<1, true, 64, float>), g, block, 0, st, a); }
            else { if (g_made_f32_products) hipLaunchKernelGGL((linear_glds_kernel<1, false, 64, float, true>), g, block, 0, st, a); else hipLaunchKernelGGL((linear_glds_kernel<1, false, 64, float>), g, block, 0, st, a); }
            break;
        }
        case MADE_LINEAR_GLDS128_F32:
            if (train) { if (g_made_f32_products) hipLaunchKernelGGL((linear_glds_kernel<1, true, 128, float, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((linear_glds_kernel<1, true, 128, float>), grid, block, 0, st, a); }
            else { if (g_made_f32_products) hipLaunchKernelGGL((linear_glds_kernel<1, false, 128, float, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((linear_glds_kernel<1, false, 128, float>), grid, block, 0, st, a); }
            break;
        case MADE_LINEAR_BIG256:
        case MADE_LINEAR_BIG128: {                         // persistent: one workgroup per CU (a multiple of 8: one eighth per XCD)
            const bool b256 = pick_variant(a) == MADE_LINEAR_BIG256;
            const int bmb = b256 ? 256 : 128;
            const int ldsb = 2 * (bmb + BIG_BN) * KB + BIG_NMAX * 4;
            static const int n_cu = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 8 ? n : 256; }();
            static const bool once = [] {
                constexpr int l256 = 2 * (256 + BIG_BN) * KB + BIG_NMAX * 4, l128 = 2 * (128 + BIG_BN) * KB + BIG_NMAX * 4;
                bool ok = true;
                ok = ok && hipFuncSetAttribute((const void*)linear_big_kernel<256, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, l256) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)linear_big_kernel<256, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l256) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)linear_big_kernel<128, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, l128) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)linear_big_kernel<128, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l128) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)linear_big_kernel<256, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l256) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)linear_big_kernel<128, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l128) == hipSuccess;
                return ok;
            }();
            (void)once;
            const int64_t nt = ((a.M + bmb - 1) / bmb) * ((a.N + BIG_BN - 1) / BIG_BN);
            int64_t g = (n_cu / 8) * 8;
            if (nt < g) g = ((nt + 7) / 8) * 8;
            dim3 gb((unsigned)g), bb(BIG_THREADS);
            // the straight-line epilogue: bias (+ ReLU) only, plain row-major output rows of whole 16-byte groups
            const bool fastep = big_fast_epilogue(a);
            const bool lite = fastep && (train || a.R != nullptr || a.out_row_mask != nullptr);   // the straight-line training epilogue
            if (b256) {
                if (lite) hipLaunchKernelGGL((linear_big_kernel<256, true, true>), gb, bb, ldsb, st, a);
                else if (fastep) hipLaunchKernelGGL((linear_big_kernel<256, false, true>), gb, bb, ldsb, st, a);
                else hipLaunchKernelGGL((linear_big_kernel<256, false, false>), gb, bb, ldsb, st, a);
            } else {
                if (lite) hipLaunchKernelGGL((linear_big_kernel<128, true, true>), gb, bb, ldsb, st, a);
                else if (fastep) hipLaunchKernelGGL((linear_big_kernel<128, false, true>), gb, bb, ldsb, st, a);
                else hipLaunchKernelGGL((linear_big_kernel<128, false, false>), gb, bb, ldsb, st, a);
            }
            break;
        }
        case MADE_LINEAR_GENERAL_F32IN: hipLaunchKernelGGL((linear_kernel<float, bf16_t>), grid, block, 0, st, a); break;
        case MADE_LINEAR_GENERAL_BF16: hipLaunchKernelGGL((linear_kernel<bf16_t, bf16_t>), grid, block, 0, st, a); break;
        default: if (g_made_f32_products) hipLaunchKernelGGL((linear_kernel<float, float, true>), grid, block, 0, st, a); else hipLaunchKernelGGL((linear_kernel<float, float>), grid, block, 0, st, a); break;
    }
    return made_check_launch("made_linear");
}

extern "C" int made_linear_variant(const MadeLinearArgs* args) {
    return args ? pick_variant(*args) : -1;
}
