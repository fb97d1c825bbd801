// made_attention_bwd: flash-style backward of made_attention on MFMA, gfx950.
//
// Three kernels, none of which writes anything of size Lq x Lk:
//   delta   : delta[b,h,i] = dO_i . O_i                                  (one wave per query row)
//   dq      : one workgroup = 128 queries of one (batch, head), loop over key tiles of 64.  Same SWAPPED layout as the
//             forward (S^T = K Q^T: the query sits on the lane), so lse / delta are per-lane scalars and dS^T is directly
//             the B operand of dQ^T += K^T dS^T (K^T fragments through the transposing LDS read).
//   dkv     : one workgroup = 128 keys of one (batch, head) (32 per wave, K and V fragments stay in registers), loop over
//             query tiles of 64 staged in LDS.  NON-swapped layout (S = Q K^T: the key sits on the lane), so Pd and dS are
//             directly the B operands of dV^T += dO^T Pd and dK^T += Q^T dS (dO^T / Q^T through the transposing read).
// Probabilities are recomputed as exp(scale * s + mask - lse) from the log-sum-exp the forward saved; the dropout mask is
// regenerated from (seed, site, element index).  bf16: v_mfma_f32_32x32x16_bf16, f32: v_mfma_f32_32x32x2_f32.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int NTH = 256;
constexpr int BKEY = 64;       // keys per tile (dq kernel)
constexpr int BQT = 64;        // queries per tile (dkv kernel)

template <typename TC> struct Frag;
template <> struct Frag<float>  { typedef f32x4  type; };
template <> struct Frag<bf16_t> { typedef bf16x8 type; };

__device__ __forceinline__ bool drop_keep(uint64_t seed, uint32_t site, uint32_t thr, uint64_t idx) {
    return (made_rng_mix(seed, site, idx) >> 8) >= thr;
}

// ---------------------------------------------------------------------------------------------- delta
// delta[b,h,i] = dO_i,h . O_i,h : one wave per token row (all heads at once, 16-byte loads), head sums by lane-group shuffles;
// rows whose q_skip_mask is 0 are skipped (their delta is never read)
__global__ __launch_bounds__(NTH) void attn_delta_kernel(const MadeAttnBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // (b, i)
    if (row >= a.B * a.Lq) return;
    const int64_t i = row % a.Lq, b = row / a.Lq;
    if (a.q_skip_mask && a.q_skip_mask[row] == 0.f) return;
    const int D = (int)(a.H * a.hd);
    const int per = a.dtype == MADE_F32 ? 4 : 8;                           // elements per 16-byte chunk
    const int lanes_per_head = a.hd / per;                                 // 4, 8, 16 (bf16) / 8, 16, 32 (f32): powers of two
    for (int c0 = 0; c0 < D; c0 += 64 * per) {
        const int c = c0 + lane * per;
        float acc = 0.f;
        if (c < D) {
            if (a.dtype == MADE_F32) {
                const f32x4 o = *(const f32x4*)((const float*)a.O + b * a.o_bs + i * a.ldo + c);
                const f32x4 g = *(const f32x4*)((const float*)a.dO + b * a.do_bs + i * a.lddo + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += o[j] * g[j];
            } else {
                const bf16x8 o = *(const bf16x8*)((const bf16_t*)a.O + b * a.o_bs + i * a.ldo + c);
                const bf16x8 g = *(const bf16x8*)((const bf16_t*)a.dO + b * a.do_bs + i * a.lddo + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (float)o[j] * (float)g[j];
            }
        }
        for (int o2 = lanes_per_head >> 1; o2 > 0; o2 >>= 1) acc += __shfl_xor(acc, o2);
        if (c < D && (lane % lanes_per_head) == 0) a.delta[(b * a.H + c / a.hd) * a.Lq + i] = acc;
    }
}

// transposing fragment: A operand X^T (row index on the lane, reduction rows kb.. of an LDS tile stored [row][col], pitch P)
// slots j = 0..7 of lane half hh <-> tile row kb16 + 8 (j >> 2) + 4 hh + (j & 3); col block = c0 .. c0 + 31
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int P, int kb16, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const unsigned char* p = tile + (kb16 + 4 * (g >> 1) + (i >> 2)) * P + (c0 + (g & 1) * 16 + 4 * (i & 3)) * 2;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 8 * P));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---------------------------------------------------------------------------------------------- dQ
template <typename TC, int HD, bool X3 = false>        // X3 (f32 only): split-bf16 products (common.h, made_set_f32_products)
__global__ __launch_bounds__(NTH, (sizeof(TC) == 2 && HD <= 64) ? 2 : 1) void attn_bwd_dq_kernel(const MadeAttnBwdArgs a) {
    typedef typename Frag<TC>::type frag_t;
    constexpr int SZ = (int)sizeof(TC);
    constexpr int PER16 = 16 / SZ;
    constexpr int P = HD * SZ + 16;              // LDS row pitch (16-byte row reads conflict-free)
    constexpr int CPR = HD * SZ / 16;
    constexpr int NCH = BKEY * CPR / NTH;
    static_assert(BKEY * CPR % NTH == 0, "staging split");
    constexpr int NQF = HD * SZ / 32;
    constexpr int NDT = HD / 32;
    constexpr bool IS_BF16 = SZ == 2;
    constexpr int BQ = 128;

    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BKEY * P + 32];
    extern __shared__ __attribute__((aligned(16))) float lds_row[];      // the sample's key bias row (0 / -inf), whole tiles
    unsigned char* lds_k = lds;
    unsigned char* lds_v = lds + BKEY * P;
    float* kbias = lds_row;
    int* lds_flag = (int*)(lds + 2 * BKEY * P);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // XCD-aware, balanced placement as in the forward kernel: (batch, head) pair p -> XCD p % 8 with all its q-tiles
    const int qtiles = (int)((a.Lq + BQ - 1) / BQ);
    const int64_t pair = (int64_t)(blockIdx.x & 7) + 8 * (int64_t)((blockIdx.x >> 3) / qtiles);
    if (pair >= a.B * a.H) return;
    const int qt = (int)((blockIdx.x >> 3) % qtiles);
    const int64_t h = pair % a.H;
    const int64_t b = a.batch_order ? (int64_t)a.batch_order[pair / a.H] : pair / a.H;   // issue order: longest sample first
    const int64_t q0 = (int64_t)qt * BQ + wave * 32;
    const int64_t q = q0 + r;
    const int64_t qc = q < a.Lq ? q : a.Lq - 1;
    bool my_valid = q < a.Lq && (a.q_skip_mask == nullptr || a.q_skip_mask[b * a.Lq + qc] != 0.f);
    const bool wave_active = __any(my_valid);
    if (!__syncthreads_or(wave_active ? 1 : 0)) {
        // every query of this workgroup is padding: its dQ rows are zeros
        if (q < a.Lq) {
            TC* dst = (TC*)a.dQ + b * a.dq_bs + q * a.lddq + h * HD;
            for (int d = hh; d < HD; d += 2) dst[d] = (TC)0.f;
        }
        return;
    }

    const TC* Kg = (const TC*)a.K + b * a.k_bs + h * HD;
    const TC* Vg = (const TC*)a.V + b * a.v_bs + h * HD;
    const float* maskg = a.key_mask ? a.key_mask + b * a.Lk : nullptr;

    frag_t qf[NQF], dof[NQF];
    {
        const TC* qp = (const TC*)a.Q + b * a.q_bs + qc * a.ldq + h * HD;
        const TC* gp = (const TC*)a.dO + b * a.do_bs + qc * a.lddo + h * HD;
#pragma unroll
        for (int ks = 0; ks < NQF; ++ks) {
            qf[ks] = *(const frag_t*)(qp + ks * 2 * PER16 + hh * PER16);
            dof[ks] = keep_or_zero(*(const frag_t*)(gp + ks * 2 * PER16 + hh * PER16), my_valid);
        }
    }
    const float lse_q = my_valid ? a.lse[(b * a.H + h) * a.Lq + qc] : INFINITY;     // +inf: every probability is 0
    // delta_q = dO_q . O_q: the bf16 path makes it here, from the dO fragments this lane holds anyway (its partner lane has the other
    // half of the head's columns), and writes it for the dK / dV kernel behind it -- the separate delta launch (13 us per attention
    // layer on the step's main stream) is gone.  The f32 parity path keeps that launch and its summation order.
    float delta_q;
    if constexpr (IS_BF16) {
        const TC* op_ = (const TC*)a.O + b * a.o_bs + qc * a.ldo + h * HD;
        float dacc = 0.f;
#pragma unroll
        for (int ks = 0; ks < NQF; ++ks) {
            const frag_t of = *(const frag_t*)(op_ + ks * 2 * PER16 + hh * PER16);
#pragma unroll
            for (int j = 0; j < PER16; ++j) dacc += (float)dof[ks][j] * (float)of[j];
        }
        dacc += __shfl_xor(dacc, 32);
        delta_q = my_valid ? dacc : 0.f;
        if (my_valid && hh == 0) a.delta[(b * a.H + h) * a.Lq + qc] = dacc;
    } else {
        delta_q = my_valid ? a.delta[(b * a.H + h) * a.Lq + qc] : 0.f;
    }

    // load_tile only ISSUES the next tile's loads (a use right behind them would make the wave wait before multiplying the current
    // tile); masked keys are zeroed in store_tile, one iteration later, from the bias row staged in LDS at the start
    frag_t rk[NCH], rv[NCH];
    int64_t rkey0 = 0;
    // the forward's dropout decisions of this lane's query (MadeAttnBwdArgs.keep_bits): two words per 64-key tile, loaded with the tile
    const bool use_bits = IS_BF16 && a.keep_bits != nullptr && a.drop.p > 0.f;
    const int64_t nkt32 = (a.Lk + 31) / 32;
    const uint32_t* bits_q = use_bits ? a.keep_bits + (b * a.H + h) * nkt32 * a.ld_bits + (qc & ~(int64_t)31) + made_keep_slot((int)(qc & 31)) : nullptr;
    uint2 rbits = make_uint2(0u, 0u), cbits = make_uint2(0u, 0u);
    auto load_tile = [&](int64_t key0) __attribute__((always_inline)) {
        rkey0 = key0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTH;
            const int64_t key = key0 + c / CPR;
            const int64_t kcl = key < a.Lk ? key : a.Lk - 1;
            rk[i] = *(const frag_t*)(Kg + kcl * a.ldk + (c % CPR) * PER16);
            rv[i] = *(const frag_t*)(Vg + kcl * a.ldv + (c % CPR) * PER16);
        }
        if (use_bits) {
            const int64_t kt0 = (key0 / BKEY) * 2, kt1 = kt0 + 1 < nkt32 ? kt0 + 1 : kt0;
            rbits = make_uint2(bits_q[kt0 * a.ld_bits], bits_q[kt1 * a.ld_bits]);
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTH;
            const bool keep = kbias[rkey0 + c / CPR] == 0.f;
            *(frag_t*)(lds_k + (c / CPR) * P + (c % CPR) * 16) = keep_or_zero(rk[i], keep);
            *(frag_t*)(lds_v + (c / CPR) * P + (c % CPR) * 16) = keep_or_zero(rv[i], keep);
        }
    };

    f32x16 dq[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[d][e] = 0.f;

    // the sample's key bias row -> LDS (whole tiles; keys beyond Lk are masked), and the last valid key
    int64_t lk_eff = a.Lk;
    {
        const int lkp = (int)((a.Lk + BKEY - 1) / BKEY) * BKEY;
        int last = -1;
        for (int j = tid; j < lkp; j += NTH) {
            const bool valid = j < (int)a.Lk && (maskg == nullptr || maskg[j] != 0.f);
            kbias[j] = valid ? 0.f : -INFINITY;
            if (valid) last = j;
        }
        if (maskg) {
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) last = max(last, __shfl_xor(last, o2));
            if (lane == 0) lds_flag[1 + wave] = last;
        }
        __syncthreads();
        if (maskg) lk_eff = max(max(lds_flag[1], lds_flag[2]), max(lds_flag[3], lds_flag[4])) + 1;
    }
    const int64_t ntiles = (lk_eff + BKEY - 1) / BKEY;
    const uint32_t thr = made_drop_threshold(a.drop.p);
    const uint64_t drop_seed = a.drop.p > 0.f ? made_drop_seed(a.drop) : 0;
    const float dsc = a.drop.p > 0.f ? 1.f / (1.f - a.drop.p) : 1.f;
    const uint64_t rowbase = (uint64_t)((b * a.H + h) * a.Lq + qc) * (uint64_t)a.Lk;

    if (ntiles > 0) load_tile(0);
    for (int64_t t = 0; t < ntiles; ++t) {
        __syncthreads();
        store_tile();
        cbits = rbits;
        __syncthreads();
        if (t + 1 < ntiles) load_tile((t + 1) * BKEY);
        if (!wave_active) continue;
        const float* tb = kbias + t * BKEY;                 // this tile's bias row
        const bool tile_has_masked = __any(tb[lane] != 0.f);

        f32x16 s[2], dp[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[kt][e] = 0.f; dp[kt][e] = 0.f; }
        if constexpr (X3) {
#pragma unroll
            for (int ks = 0; ks < NQF; ks += 2) {
                const SplitF32x4 q0 = made_split4(qf[ks]), q1 = made_split4(qf[ks + 1]), g0 = made_split4(dof[ks]), g1 = made_split4(dof[ks + 1]);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const unsigned char* pk = lds_k + (kt * 32 + r) * P + ks * 32 + hh * 16;
                    const unsigned char* pv = lds_v + (kt * 32 + r) * P + ks * 32 + hh * 16;
                    s[kt] = made_mfma_x3_16(made_split4(*(const f32x4*)pk), made_split4(*(const f32x4*)(pk + 32)), q0, q1, s[kt]);
                    dp[kt] = made_mfma_x3_16(made_split4(*(const f32x4*)pv), made_split4(*(const f32x4*)(pv + 32)), g0, g1, dp[kt]);
                }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < NQF; ++ks) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                frag_t kf = *(const frag_t*)(lds_k + (kt * 32 + r) * P + ks * 32 + hh * 16);
                frag_t vf = *(const frag_t*)(lds_v + (kt * 32 + r) * P + ks * 32 + hh * 16);
                if constexpr (IS_BF16) {
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kt], 0, 0, 0);
                    dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp[kt], 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[ks][e], s[kt], 0, 0, 0);
                        dp[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], dof[ks][e], dp[kt], 0, 0, 0);
                    }
                }
            }
        }
        }
        // dS^T = P^T * (dP^T - delta) * scale, in place of s
        const uint64_t tbase = rowbase + (uint64_t)(t * BKEY);
        const uint32_t tlo = (uint32_t)tbase;
        const bool fast_idx = __all(tlo <= 0xFFFFFFFFu - BKEY);
        const uint32_t kk = made_rng_key(drop_seed, a.drop.site, (uint32_t)(tbase >> 32));
        if constexpr (IS_BF16) {
            // VALU diet (the elementwise part, not the MFMAs, bounds this kernel): exp2 with the scale and -lse folded into
            // one FMA, the key-mask bias only on tiles that contain an invalid key, delta pre-multiplied by the scale
            const float c2 = a.scale * 1.4426950408889634f;
            const float nl = -lse_q * 1.4426950408889634f;
            const float dsq = delta_q * a.scale;
            if (tile_has_masked) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[kt][e] += tb[kt * 32 + acc_row(e, hh)];   // -inf * anything stays -inf below
            }
            if (use_bits) {
                // one bit test per score: bit acc_row(e, hh) of the tile's word kt (the forward stored its decisions in key order)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const uint32_t wq = (kt == 0 ? cbits.x : cbits.y) >> (4 * hh);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][e], c2, nl));
                        const float g = (wq & (1u << ((e & 3) + 8 * (e >> 2)))) ? dp[kt][e] * dsc : 0.f;
                        s[kt][e] = p * __builtin_fmaf(g, a.scale, -dsq);
                    }
                }
            } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kl = kt * 32 + acc_row(e, hh);
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][e], c2, nl));
                    float g = dp[kt][e];
                    if (a.drop.p > 0.f) {
                        const uint32_t hsh = fast_idx ? made_rng_fmix32((tlo + (uint32_t)kl) ^ kk) : made_rng_mix(drop_seed, a.drop.site, tbase + (uint64_t)kl);
                        g = (hsh >> 8) >= thr ? g * dsc : 0.f;
                    }
                    s[kt][e] = p * __builtin_fmaf(g, a.scale, -dsq);
                }
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kl = kt * 32 + acc_row(e, hh);
                    const float p = expf(s[kt][e] * a.scale + tb[kl] - lse_q);
                    float g = dp[kt][e];
                    if (a.drop.p > 0.f) g = drop_keep(drop_seed, a.drop.site, thr, tbase + (uint64_t)kl) ? g * dsc : 0.f;
                    s[kt][e] = p * (g - delta_q) * a.scale;
                }
        }
        // dQ^T += K^T dS^T
        if constexpr (IS_BF16) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[kt][8 * s2 + j];
#pragma unroll
                    for (int d = 0; d < NDT; ++d)
                        dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(lds_k, P, kt * 32 + 16 * s2, d * 32, lane), pf, dq[d], 0, 0, 0);
                }
        } else if constexpr (X3) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4 += 2) {            // two register quads = two 8-deep steps per product
                    const SplitF32x4 b0 = made_split4(f32x4{s[kt][4 * g4], s[kt][4 * g4 + 1], s[kt][4 * g4 + 2], s[kt][4 * g4 + 3]});
                    const SplitF32x4 b1 = made_split4(f32x4{s[kt][4 * g4 + 4], s[kt][4 * g4 + 5], s[kt][4 * g4 + 6], s[kt][4 * g4 + 7]});
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        f32x4 v0, v1;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v0[j] = *(const float*)(lds_k + (kt * 32 + 8 * g4 + 4 * hh + j) * P + (d * 32 + r) * 4);
                            v1[j] = *(const float*)(lds_k + (kt * 32 + 8 * g4 + 8 + 4 * hh + j) * P + (d * 32 + r) * 4);
                        }
                        dq[d] = made_mfma_x3_16(made_split4(v0), made_split4(v1), b0, b1, dq[d]);
                    }
                }
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * 32 + acc_row(e, hh);
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        const float kv = *(const float*)(lds_k + key * P + (d * 32 + r) * 4);
                        dq[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, s[kt][e], dq[d], 0, 0, 0);
                    }
                }
        }
    }


    if (q >= a.Lq) return;
    TC* op = (TC*)a.dQ + b * a.dq_bs + q * a.lddq + h * HD;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            TC* dst = op + d * 32 + 8 * g + 4 * hh;
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = from_f32<TC>(my_valid ? dq[d][4 * g + j] : 0.f);
        }
}

// ---------------------------------------------------------------------------------------------- dK, dV
template <typename TC, int HD, bool X3 = false>
__global__ __launch_bounds__(NTH, (sizeof(TC) == 2 && HD <= 64) ? 2 : 1) void attn_bwd_dkv_kernel(const MadeAttnBwdArgs a) {
    typedef typename Frag<TC>::type frag_t;
    constexpr int SZ = (int)sizeof(TC);
    constexpr int PER16 = 16 / SZ;
    constexpr int P = HD * SZ + 16;
    constexpr int CPR = HD * SZ / 16;
    constexpr int NCH = BQT * CPR / NTH;
    static_assert(BQT * CPR % NTH == 0, "staging split");
    constexpr int NQF = HD * SZ / 32;
    constexpr int NDT = HD / 32;
    constexpr bool IS_BF16 = SZ == 2;
    constexpr int BK = 128;                       // keys per workgroup

    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BQT * P + 32];
    __shared__ __attribute__((aligned(16))) uint32_t lds_bits[4 * BQT];  // the forward's dropout decisions of the tile: [32-key block of this workgroup][query]
    extern __shared__ __attribute__((aligned(16))) float lds_row[];      // per query of this (batch, head), whole tiles: see below
    unsigned char* lds_q = lds;
    unsigned char* lds_do = lds + BQT * P;
    int* lds_flag = (int*)(lds + 2 * BQT * P);
    const int lqp = (int)((a.Lq + BQT - 1) / BQT) * BQT;
    float* lse_all = lds_row;                       // log-sum-exp (pre-folded for the exp2 form in bf16; +-inf for padded queries: p = 0)
    float* delta_all = lds_row + lqp;               // delta (pre-multiplied by the scale in bf16; 0 for padded queries)
    float* live_all = lds_row + 2 * lqp;            // 1 / 0: the query is computed

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int ktiles = (int)((a.Lk + BK - 1) / BK);
    const int64_t pair = (int64_t)(blockIdx.x & 7) + 8 * (int64_t)((blockIdx.x >> 3) / ktiles);      // pair p -> XCD p % 8 (see dq)
    if (pair >= a.B * a.H) return;
    const int kt_blk = (int)((blockIdx.x >> 3) % ktiles);
    const int64_t h = pair % a.H;
    const int64_t b = a.batch_order ? (int64_t)a.batch_order[pair / a.H] : pair / a.H;   // issue order: longest sample first
    const int64_t key = (int64_t)kt_blk * BK + wave * 32 + r;
    const int64_t keyc = key < a.Lk ? key : a.Lk - 1;
    const float* maskg = a.key_mask ? a.key_mask + b * a.Lk : nullptr;
    const bool key_valid = key < a.Lk && (maskg == nullptr || maskg[keyc] != 0.f);
    const float bias_key = key_valid ? 0.f : -INFINITY;
    if (!__syncthreads_or(key_valid ? 1 : 0)) {
        // every key of this workgroup is padding (a ragged batch: about 40 % of the key blocks at the step's shapes): its dK / dV rows are zeros,
        // and the query tiles need not be walked at all
        if (key < a.Lk) {
            TC* kz = (TC*)a.dK + b * a.dk_bs + key * a.lddk + h * HD;
            TC* vz = (TC*)a.dV + b * a.dv_bs + key * a.lddv + h * HD;
            for (int d = hh; d < HD; d += 2) { kz[d] = (TC)0.f; vz[d] = (TC)0.f; }
        }
        return;
    }

    // K / V fragments of this lane's key (B operands): K[key][ks*2*PER16 + hh*PER16 ..]; invalid keys read as zero rows
    frag_t kf[NQF], vf[NQF];
    {
        const TC* kp = (const TC*)a.K + b * a.k_bs + keyc * a.ldk + h * HD;
        const TC* vp = (const TC*)a.V + b * a.v_bs + keyc * a.ldv + h * HD;
#pragma unroll
        for (int ks = 0; ks < NQF; ++ks) {
            kf[ks] = keep_or_zero(*(const frag_t*)(kp + ks * 2 * PER16 + hh * PER16), key_valid);
            vf[ks] = keep_or_zero(*(const frag_t*)(vp + ks * 2 * PER16 + hh * PER16), key_valid);
        }
    }

    const TC* Qg = (const TC*)a.Q + b * a.q_bs + h * HD;
    const TC* Gg = (const TC*)a.dO + b * a.do_bs + h * HD;
    const float* skipg = a.q_skip_mask ? a.q_skip_mask + b * a.Lq : nullptr;
    const float* lseg = a.lse + (b * a.H + h) * a.Lq;
    const float* delg = a.delta + (b * a.H + h) * a.Lq;

    // (as in the dq kernel: the next tile's loads are only issued here; padded queries are zeroed in store_tile from the rows in LDS)
    frag_t rq[NCH], rg[NCH];
    int64_t rqbase = 0;
    // thread t stages the word of query t % 64 x key block t / 64 of this workgroup's 128 keys (MadeAttnBwdArgs.keep_bits)
    const bool use_bits = IS_BF16 && a.keep_bits != nullptr && a.drop.p > 0.f;
    const int64_t nkt32 = (a.Lk + 31) / 32;
    const int64_t bword = (int64_t)kt_blk * 4 + (tid >> 6);
    const uint32_t* bits_bh = use_bits ? a.keep_bits + ((b * a.H + h) * nkt32 + (bword < nkt32 ? bword : nkt32 - 1)) * a.ld_bits : nullptr;
    uint32_t rb = 0u;
    auto load_tile = [&](int64_t qbase) __attribute__((always_inline)) {
        rqbase = qbase;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTH;
            const int64_t qq = qbase + c / CPR;
            const int64_t qcl = qq < a.Lq ? qq : a.Lq - 1;
            rq[i] = *(const frag_t*)(Qg + qcl * a.ldq + (c % CPR) * PER16);
            rg[i] = *(const frag_t*)(Gg + qcl * a.lddo + (c % CPR) * PER16);
        }
        if (use_bits) {
            const int64_t qq = qbase + (tid & 63);
            const int64_t qb = qq < a.Lq ? qq : a.Lq - 1;
            rb = bits_bh[(qb & ~(int64_t)31) + made_keep_slot((int)(qb & 31))];
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTH;
            const bool keep = live_all[rqbase + c / CPR] != 0.f;
            *(frag_t*)(lds_q + (c / CPR) * P + (c % CPR) * 16) = keep_or_zero(rq[i], keep);
            *(frag_t*)(lds_do + (c / CPR) * P + (c % CPR) * 16) = keep_or_zero(rg[i], keep);
        }
        if (use_bits) lds_bits[tid] = rb;
    };

    f32x16 dk[NDT], dv[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[d][e] = 0.f; dv[d][e] = 0.f; }

    // the per-query scalars of this (batch, head) -> LDS once; queries after the last computed one contribute nothing
    int64_t lq_eff = a.Lq;
    {
        int last = -1;
        for (int j = tid; j < lqp; j += NTH) {
            const int jc = j < (int)a.Lq ? j : (int)a.Lq - 1;
            const bool ok = j < (int)a.Lq && (skipg == nullptr || skipg[jc] != 0.f);
            float l = ok ? lseg[jc] : INFINITY;            // +inf: the whole probability row is 0
            float dl = ok ? delg[jc] : 0.f;
            if constexpr (IS_BF16) { l = -l * 1.4426950408889634f; dl = dl * a.scale; }   // pre-folded for the exp2 / FMA form
            lse_all[j] = l; delta_all[j] = dl; live_all[j] = ok ? 1.f : 0.f;
            if (ok) last = j;
        }
        if (skipg) {
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) last = max(last, __shfl_xor(last, o2));
            if (lane == 0) lds_flag[1 + wave] = last;
        }
        __syncthreads();
        if (skipg) lq_eff = max(max(lds_flag[1], lds_flag[2]), max(lds_flag[3], lds_flag[4])) + 1;
    }
    const int64_t ntiles = (lq_eff + BQT - 1) / BQT;
    const uint32_t thr = made_drop_threshold(a.drop.p);
    const uint64_t drop_seed = a.drop.p > 0.f ? made_drop_seed(a.drop) : 0;
    const float dsc = a.drop.p > 0.f ? 1.f / (1.f - a.drop.p) : 1.f;
    const uint64_t bhbase = (uint64_t)(b * a.H + h) * (uint64_t)a.Lq;
    const bool wave_active = __any(key_valid);

    if (ntiles > 0) load_tile(0);
    for (int64_t t = 0; t < ntiles; ++t) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile((t + 1) * BQT);
        if (!wave_active) continue;
        const float* lds_lse = lse_all + t * BQT;
        const float* lds_delta = delta_all + t * BQT;

        const uint64_t tfirst = (bhbase + (uint64_t)(t * BQT)) * (uint64_t)a.Lk;
        const uint64_t tlast = tfirst + (uint64_t)BQT * (uint64_t)a.Lk;
        const bool fast_idx = (tfirst >> 32) == (tlast >> 32);            // wave-uniform: the tile's indices share their high word
        const uint32_t kk = made_rng_key(drop_seed, a.drop.site, (uint32_t)(tfirst >> 32));
        const uint32_t tlo = (uint32_t)tfirst + (uint32_t)keyc;
        // the two 32-query halves of the tile one after the other (not unrolled): one score / dP tile pair live at a time keeps
        // the kernel at 3 waves per SIMD
#pragma nounroll
        for (int qi = 0; qi < 2; ++qi) {
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
            if constexpr (X3) {
#pragma unroll
                for (int ks = 0; ks < NQF; ks += 2) {
                    const unsigned char* pq = lds_q + (qi * 32 + r) * P + ks * 32 + hh * 16;
                    const unsigned char* pg = lds_do + (qi * 32 + r) * P + ks * 32 + hh * 16;
                    s = made_mfma_x3_16(made_split4(*(const f32x4*)pq), made_split4(*(const f32x4*)(pq + 32)), made_split4(kf[ks]), made_split4(kf[ks + 1]), s);
                    dp = made_mfma_x3_16(made_split4(*(const f32x4*)pg), made_split4(*(const f32x4*)(pg + 32)), made_split4(vf[ks]), made_split4(vf[ks + 1]), dp);
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < NQF; ++ks) {
                frag_t qa = *(const frag_t*)(lds_q + (qi * 32 + r) * P + ks * 32 + hh * 16);
                frag_t ga = *(const frag_t*)(lds_do + (qi * 32 + r) * P + ks * 32 + hh * 16);
                if constexpr (IS_BF16) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, vf[ks], dp, 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[e], kf[ks][e], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[e], vf[ks][e], dp, 0, 0, 0);
                    }
                }
            }
            }
            // s <- Pd (dropped probabilities), dp <- dS
            if constexpr (IS_BF16) {
                const float c2 = a.scale * 1.4426950408889634f;
                if (use_bits) {
                    // The keep-bit path as ONE straight-line block (round 4).  With the path chosen per element (the loop below: `if (use_bits) ..
                    // else if (p > 0) ..` inside the unrolled loop) every score was a basic block of its own: its FMA -> exp2 -> select -> FMA
                    // chain ran alone, 125 cycles per score (2 000 per half; the dQ kernel, whose paths are separate loops, needs 50).  Here the
                    // sixteen chains of a half interleave.  The arithmetic is the loop's, operation for operation: the same bits.
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const f32x4 nl4 = *(const f32x4*)(lds_lse + qi * 32 + 8 * e4 + 4 * hh);
                        const f32x4 dl4 = *(const f32x4*)(lds_delta + qi * 32 + 8 * e4 + 4 * hh);
                        const u32x4 bw4 = *(const u32x4*)(lds_bits + wave * BQT + qi * 32 + 8 * e4 + 4 * hh);
#pragma unroll
                        for (int ej = 0; ej < 4; ++ej) {
                            const int e = 4 * e4 + ej;
                            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c2, nl4[ej]) + bias_key);
                            const bool keep = (bw4[ej] >> r) & 1u;
                            const float pd = keep ? p * dsc : 0.f;
                            const float g = keep ? dp[e] * dsc : 0.f;
                            s[e] = pd;
                            dp[e] = p * __builtin_fmaf(g, a.scale, -dl4[ej]);
                        }
                    }
                } else {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    // the 4 accumulator rows of a register quad are consecutive queries: one 16-byte LDS read each for -lse and delta
                    const f32x4 nl4 = *(const f32x4*)(lds_lse + qi * 32 + 8 * e4 + 4 * hh);
                    const f32x4 dl4 = *(const f32x4*)(lds_delta + qi * 32 + 8 * e4 + 4 * hh);
                    u32x4 bw4 = {0u, 0u, 0u, 0u};
                    if (use_bits) bw4 = *(const u32x4*)(lds_bits + wave * BQT + qi * 32 + 8 * e4 + 4 * hh);   // the words of these four queries, this wave's 32 keys
#pragma unroll
                    for (int ej = 0; ej < 4; ++ej) {
                        const int e = 4 * e4 + ej;
                        const int ql = qi * 32 + acc_row(e, hh);
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c2, nl4[ej]) + bias_key);
                        float pd = p, g = dp[e];
                        if (use_bits) {
                            const bool keep = (bw4[ej] >> r) & 1u;
                            pd = keep ? p * dsc : 0.f;
                            g = keep ? g * dsc : 0.f;
                        } else if (a.drop.p > 0.f) {
                            const uint32_t hsh = fast_idx ? made_rng_fmix32((tlo + (uint32_t)ql * (uint32_t)a.Lk) ^ kk)
                                                          : made_rng_mix(drop_seed, a.drop.site, tfirst + (uint64_t)ql * (uint64_t)a.Lk + (uint64_t)keyc);
                            const bool keep = (hsh >> 8) >= thr;
                            pd = keep ? p * dsc : 0.f;
                            g = keep ? g * dsc : 0.f;
                        }
                        s[e] = pd;
                        dp[e] = p * __builtin_fmaf(g, a.scale, -dl4[ej]);
                    }
                }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ql = qi * 32 + acc_row(e, hh);
                    const float p = expf(s[e] * a.scale + bias_key - lds_lse[ql]);
                    float pd = p, g = dp[e];
                    if (a.drop.p > 0.f) {
                        const bool keep = drop_keep(drop_seed, a.drop.site, thr, tfirst + (uint64_t)ql * (uint64_t)a.Lk + (uint64_t)keyc);
                        pd = keep ? p * dsc : 0.f;
                        g = keep ? g * dsc : 0.f;
                    }
                    s[e] = pd;
                    dp[e] = p * (g - lds_delta[ql]) * a.scale;
                }
            }
            // dV^T += dO^T Pd,  dK^T += Q^T dS
            if constexpr (IS_BF16) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf, sf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { pf[j] = (bf16_t)s[8 * s2 + j]; sf[j] = (bf16_t)dp[8 * s2 + j]; }
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(lds_do, P, qi * 32 + 16 * s2, d * 32, lane), pf, dv[d], 0, 0, 0);
                        dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(lds_q, P, qi * 32 + 16 * s2, d * 32, lane), sf, dk[d], 0, 0, 0);
                    }
                }
            } else if constexpr (X3) {
#pragma unroll
                for (int g4 = 0; g4 < 4; g4 += 2) {            // two register quads = two 8-deep steps per product
                    const SplitF32x4 p0 = made_split4(f32x4{s[4 * g4], s[4 * g4 + 1], s[4 * g4 + 2], s[4 * g4 + 3]});
                    const SplitF32x4 p1 = made_split4(f32x4{s[4 * g4 + 4], s[4 * g4 + 5], s[4 * g4 + 6], s[4 * g4 + 7]});
                    const SplitF32x4 d0 = made_split4(f32x4{dp[4 * g4], dp[4 * g4 + 1], dp[4 * g4 + 2], dp[4 * g4 + 3]});
                    const SplitF32x4 d1 = made_split4(f32x4{dp[4 * g4 + 4], dp[4 * g4 + 5], dp[4 * g4 + 6], dp[4 * g4 + 7]});
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        f32x4 g0, g1, q0, q1;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int r0 = qi * 32 + 8 * g4 + 4 * hh + j, r1 = r0 + 8;
                            g0[j] = *(const float*)(lds_do + r0 * P + (d * 32 + r) * 4); g1[j] = *(const float*)(lds_do + r1 * P + (d * 32 + r) * 4);
                            q0[j] = *(const float*)(lds_q + r0 * P + (d * 32 + r) * 4);  q1[j] = *(const float*)(lds_q + r1 * P + (d * 32 + r) * 4);
                        }
                        dv[d] = made_mfma_x3_16(made_split4(g0), made_split4(g1), p0, p1, dv[d]);
                        dk[d] = made_mfma_x3_16(made_split4(q0), made_split4(q1), d0, d1, dk[d]);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ql = qi * 32 + acc_row(e, hh);
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        const float gv = *(const float*)(lds_do + ql * P + (d * 32 + r) * 4);
                        const float qv = *(const float*)(lds_q + ql * P + (d * 32 + r) * 4);
                        dv[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, s[e], dv[d], 0, 0, 0);
                        dk[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv, dp[e], dk[d], 0, 0, 0);
                    }
                }
            }
        }
    }


    if (key >= a.Lk) return;
    TC* kp = (TC*)a.dK + b * a.dk_bs + key * a.lddk + h * HD;
    TC* vp = (TC*)a.dV + b * a.dv_bs + key * a.lddv + h * HD;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dd = d * 32 + 8 * g + 4 * hh + j;
                kp[dd] = from_f32<TC>(key_valid ? dk[d][4 * g + j] : 0.f);
                vp[dd] = from_f32<TC>(key_valid ? dv[d][4 * g + j] : 0.f);
            }
}

constexpr int BWD_ROW_LDS_MAX = 96 * 1024;        // per-sample rows staged in LDS by the dq (4 B per key) and dkv (12 B per query) kernels

template <typename TC, int HD>
int launch_bwd_hd(const MadeAttnBwdArgs& a, dim3 gq, dim3 gk, size_t lds_q, size_t lds_k, hipStream_t st) {
    // the kernels' static tiles (2 x 64 rows) + the per-sample rows must fit the CU's 160 KB
    constexpr int kStatic = 2 * 64 * (HD * (int)sizeof(TC) + 16) + 32;
    constexpr int kMaxDyn = (160 * 1024 - kStatic - 256 < BWD_ROW_LDS_MAX) ? (160 * 1024 - kStatic - 256) / 256 * 256 : BWD_ROW_LDS_MAX;
    if (lds_q > (size_t)kMaxDyn || lds_k > (size_t)kMaxDyn) {
        made_set_error("made_attention_bwd: Lq=%lld / Lk=%lld too long for the per-sample rows kept in LDS at head dim %d (at most %d queries, %d keys)",
                       (long long)a.Lq, (long long)a.Lk, HD, kMaxDyn / 12, kMaxDyn / 4);
        return MADE_ERR_UNSUPPORTED;
    }
    static bool attr_done = false;                // (per instantiation)
    if (!attr_done) {
        hipError_t e1 = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<TC, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDyn);
        hipError_t e2 = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<TC, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDyn);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            made_set_error("made_attention_bwd: cannot reserve %d bytes of LDS", kMaxDyn);
            return MADE_ERR_HIP;
        }
        attr_done = true;
    }
    if (sizeof(TC) == 4 && g_made_f32_products) {
        static bool attr3 = false;
        if (!attr3) {
            hipError_t e1 = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<TC, HD, sizeof(TC) == 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDyn);
            hipError_t e2 = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<TC, HD, sizeof(TC) == 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDyn);
            if (e1 != hipSuccess || e2 != hipSuccess) { made_set_error("made_attention_bwd: cannot reserve %d bytes of LDS", kMaxDyn); return MADE_ERR_HIP; }
            attr3 = true;
        }
        hipLaunchKernelGGL((attn_bwd_dq_kernel<TC, HD, sizeof(TC) == 4>), gq, dim3(NTH), lds_q, st, a);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<TC, HD, sizeof(TC) == 4>), gk, dim3(NTH), lds_k, st, a);
        return MADE_OK;
    }
    hipLaunchKernelGGL((attn_bwd_dq_kernel<TC, HD>), gq, dim3(NTH), lds_q, st, a);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<TC, HD>), gk, dim3(NTH), lds_k, st, a);
    return MADE_OK;
}

template <typename TC>
int launch_bwd(const MadeAttnBwdArgs& a, hipStream_t st) {
    const int64_t rows = a.B * a.Lq;
    if (a.dtype != MADE_BF16)                               // (bf16: the dQ kernel makes delta itself and hands it to the dK / dV kernel)
        hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(NTH), 0, st, a);
    const int64_t pairs8 = 8 * ((a.H * a.B + 7) / 8);
    dim3 gq((unsigned)(((a.Lq + 127) / 128) * pairs8)), gk((unsigned)(((a.Lk + 127) / 128) * pairs8));
    const size_t lds_q = (size_t)((a.Lk + BKEY - 1) / BKEY) * BKEY * 4, lds_k = (size_t)((a.Lq + BQT - 1) / BQT) * BQT * 12;
    int rc;
    switch (a.hd) {
        case 32: rc = launch_bwd_hd<TC, 32>(a, gq, gk, lds_q, lds_k, st); break;
        case 64: rc = launch_bwd_hd<TC, 64>(a, gq, gk, lds_q, lds_k, st); break;
        case 128: rc = launch_bwd_hd<TC, 128>(a, gq, gk, lds_q, lds_k, st); break;
        default:
            made_set_error("made_attention_bwd: head dim %d not in {32,64,128}", a.hd);
            return MADE_ERR_UNSUPPORTED;
    }
    if (rc != MADE_OK) return rc;
    return made_check_launch("made_attention_bwd");
}

}  // namespace

extern "C" int made_attention_bwd(const MadeAttnBwdArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_attention_bwd: null args");
    const MadeAttnBwdArgs& a = *args;
    MADE_REQUIRE(a.Q && a.K && a.V && a.O && a.dO && a.dQ && a.dK && a.dV && a.lse && a.delta, "made_attention_bwd: null tensor");
    MADE_REQUIRE(a.B >= 0 && a.H > 0 && a.Lq >= 0 && a.Lk > 0, "made_attention_bwd: bad dims");
    MADE_REQUIRE(a.dtype == MADE_F32 || a.dtype == MADE_BF16, "made_attention_bwd: bad dtype %d", a.dtype);
    MADE_REQUIRE(a.drop.p >= 0.f && a.drop.p < 1.f, "made_attention_bwd: dropout p out of [0,1)");
    if (a.keep_bits)
        MADE_REQUIRE(a.ld_bits >= 32 * ((a.Lq + 31) / 32) && a.ld_bits % 32 == 0 && ((uintptr_t)a.keep_bits % 8) == 0,
                     "made_attention_bwd: keep_bits rows need ld_bits = a multiple of 32 >= Lq and 8-byte alignment");
    const int per16 = a.dtype == MADE_F32 ? 4 : 8;
    MADE_UNSUPPORTED(a.ldq % per16 == 0 && a.ldk % per16 == 0 && a.ldv % per16 == 0 && a.lddo % per16 == 0 && a.ldo % per16 == 0 &&
                     a.q_bs % per16 == 0 && a.k_bs % per16 == 0 && a.v_bs % per16 == 0 && a.do_bs % per16 == 0 && a.o_bs % per16 == 0 &&
                     ((uintptr_t)a.O % 16) == 0,
                     "made_attention_bwd: strides must keep 16-byte alignment");
    MADE_UNSUPPORTED(((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.K % 16) == 0 && ((uintptr_t)a.V % 16) == 0 && ((uintptr_t)a.dO % 16) == 0,
                     "made_attention_bwd: base pointers must be 16-byte aligned");
    MADE_UNSUPPORTED(((a.Lq + 127) / 128) * a.H * a.B < (1LL << 31) && ((a.Lk + 127) / 128) * a.H * a.B < (1LL << 31),
                     "made_attention_bwd: too many workgroups");
    if (a.B == 0 || a.Lq == 0) return MADE_OK;
    hipStream_t st = (hipStream_t)stream;
    // bf16, head dim 64: the single-pass kernel (attention_bwd_fused.hip).  MADE_ATTN_BWD=split keeps the two-kernel form (A/B measurements).
    static const bool force_split = [] { const char* e = made_variant_env("MADE_ATTN_BWD"); return e && e[0] == 's'; }();
    if (!force_split) {
        const int rc = made_attention_bwd_fused_try(a, st);
        if (rc >= 0 || rc == MADE_ERR_HIP) return rc;
    }
    return a.dtype == MADE_BF16 ? launch_bwd<bf16_t>(a, st) : launch_bwd<float>(a, st);
}
