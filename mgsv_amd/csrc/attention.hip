// made_attention: fused multi-head attention core (flash style) on MFMA, gfx950.
//
// One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.  Keys are
// consumed in tiles of 64.  The score product is computed SWAPPED, S^T = K Q^T, so that in the
// 32x32 accumulator layout the query sits on the lane and the keys of a tile sit in the lane's
// registers: the softmax row reductions are in-lane plus one cross-half exchange, the running
// (max, sum) are per-lane scalars, and P^T is already the B operand of the second product
// O^T += V^T P^T (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's
// operand").  V stays ROW-major ([key][d], as the projection GEMM writes it): its A-operand fragments
// (d on the lane, keys along k) come out of LDS through ds_read_b64_tr_b16, the hardware transposing
// read (bf16), or plain ds_read_b32 with the lane on d (f32).  K and V tiles are staged
// global -> registers -> LDS with the next tile's loads in flight during the current tile's MFMAs;
// scores never touch HBM.
//   bf16 : v_mfma_f32_32x32x16_bf16,  f32 : v_mfma_f32_32x32x2_f32 (exact f32, used for parity)
#include "common.h"

#include <type_traits>

namespace {

constexpr int BQ = 128;       // queries per workgroup
constexpr int BKEY = 64;      // keys per tile
constexpr int NTHREADS = 256;

template <typename TC> struct Frag;
template <> struct Frag<float>  { typedef f32x4  type; };
template <> struct Frag<bf16_t> { typedef bf16x8 type; };

template <typename TC, int HD, bool X3 = false>        // X3 (f32 only): split-bf16 products (common.h, made_set_f32_products)
__global__ __launch_bounds__(NTHREADS, (HD <= 64 ? 2 : 1)) void attention_kernel(const MadeAttnArgs a) {
    typedef typename Frag<TC>::type frag_t;
    constexpr int SZ = (int)sizeof(TC);
    constexpr int PER16 = 16 / SZ;
    constexpr int K_ROW = HD * SZ + 16;          // bytes, padded (conflict-free 16-byte row reads)
    // V rows: bf16 -> 4 consecutive rows must fall on disjoint quarters of the 64 banks (transposing reads)
    constexpr int V_ROW = SZ == 2 ? HD * 2 + (HD == 32 ? 0 : 64) : HD * 4 + 16;
    constexpr int K_CPR = HD * SZ / 16;          // 16-byte chunks per K / V row
    constexpr int NCH = BKEY * K_CPR / NTHREADS; // chunks per thread (same count for K and V)
    static_assert(BKEY * K_CPR % NTHREADS == 0, "staging split");
    constexpr int NQF = HD * SZ / 32;            // Q fragments (k-steps of the score product)
    constexpr int NDT = HD / 32;                 // 32-row tiles of O^T
    constexpr bool IS_BF16 = SZ == 2;
    const uint64_t drop_seed = a.drop.p > 0.f ? made_drop_seed(a.drop) : 0;

    __shared__ __attribute__((aligned(16))) unsigned char lds[BKEY * K_ROW + BKEY * V_ROW + BKEY * 4 + 32];
    unsigned char* lds_k = lds;
    unsigned char* lds_v = lds + BKEY * K_ROW;
    float* lds_bias = (float*)(lds + BKEY * K_ROW + BKEY * V_ROW);
    int* lds_flag = (int*)(lds + BKEY * K_ROW + BKEY * V_ROW + BKEY * 4);     // != 0: this tile has a masked / out-of-range key

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // 1-D grid, XCD-aware order: workgroups b, b+8, ... share an XCD; give each XCD a contiguous run of (batch, head,
    // q-tile) triples so the q-tiles of one (batch, head) re-read its K / V from that XCD's L2 instead of HBM
    // 1-D grid, XCD-aware and balanced: workgroup i runs on XCD i % 8.  (batch, head) pair p goes to XCD p % 8 with all its
    // q-tiles (they re-read its K / V from that XCD's L2); with H a multiple of 8 every XCD gets the same heads of EVERY sample,
    // so ragged sequence lengths load the 8 XCDs equally (contiguous runs of samples per XCD left whole XCDs idle)
    const int qtiles = (int)((a.Lq + BQ - 1) / BQ);
    const int64_t pair = (int64_t)(blockIdx.x & 7) + 8 * (int64_t)((blockIdx.x >> 3) / qtiles);
    if (pair >= a.B * a.H) return;
    const int qt = (int)((blockIdx.x >> 3) % qtiles);
    const int64_t h = pair % a.H;
    const int64_t b = a.batch_order ? (int64_t)a.batch_order[pair / a.H] : pair / a.H;   // issue order: longest sample first
    const int64_t q0 = (int64_t)qt * BQ + wave * 32;
    bool wave_active = q0 < a.Lq;
    if (a.q_skip_mask) {
        // 32-query groups made only of padded queries are not computed (nobody reads their rows); a workgroup made only
        // of such groups exits before touching K / V
        const int64_t q = q0 + (lane & 31);
        const bool mine = wave_active && q < a.Lq && a.q_skip_mask[b * a.Lq + q] != 0.f;
        wave_active = __any(mine);
        if (!__syncthreads_or(wave_active ? 1 : 0)) return;
    }

    const TC* Kg = (const TC*)a.K + b * a.k_bs + h * HD;
    const TC* Vg = (const TC*)a.V + b * a.v_bs + h * HD;
    const float* maskg = a.key_mask ? a.key_mask + b * a.Lk : nullptr;

    // ---- Q fragments (B operand of S^T = K Q^T): lane (r, hh) holds Q[q0+r][ks*2*PER16 + hh*PER16 ..]
    frag_t qf[NQF];
    {
        int64_t q = q0 + r;
        if (q >= a.Lq) q = a.Lq - 1;
        const TC* qp = (const TC*)a.Q + b * a.q_bs + q * a.ldq + h * HD;
#pragma unroll
        for (int ks = 0; ks < NQF; ++ks) qf[ks] = *(const frag_t*)(qp + ks * 2 * PER16 + hh * PER16);
    }

    // ---- staging registers.  load_tile only ISSUES the loads of the next tile (K / V rows, their mask values); everything that
    // consumes a loaded value -- zeroing the rows of masked keys, the tile's bias row -- happens in store_tile, one iteration later,
    // after the current tile has been multiplied: a use right behind the loads would make the wave wait for them before its MFMAs.
    frag_t rk[NCH], rv[NCH];
    float rmk[NCH], rmkb = 1.f;
    int64_t rkey0 = 0;
    auto load_tile_impl = [&](int64_t key0, auto has_mask) __attribute__((always_inline)) {
        // branch-free: every lane always loads (row index clamped into the tensor), masking happens on the registers
        rkey0 = key0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + i * NTHREADS;
            int krow = c / K_CPR, kc = c % K_CPR;
            int64_t key = key0 + krow;
            const int64_t kcl = key < a.Lk ? key : a.Lk - 1;
            rk[i] = *(const frag_t*)(Kg + kcl * a.ldk + kc * PER16);
            rv[i] = *(const frag_t*)(Vg + kcl * a.ldv + kc * PER16);
            if constexpr (decltype(has_mask)::value) rmk[i] = maskg[kcl]; else rmk[i] = 1.f;
        }
        if constexpr (decltype(has_mask)::value) {       // (every wave loads, wave 0 stores: a load under `if (tid < 64)` would be
            const int64_t key = key0 + lane;               // joined with the other waves' untouched register, and the join waits)
            rmkb = maskg[key < a.Lk ? key : a.Lk - 1];
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + i * NTHREADS;
            const bool keep = (rkey0 + c / K_CPR) < a.Lk && rmk[i] != 0.f;   // masked keys read as zero rows: 0*garbage must stay 0
            *(frag_t*)(lds_k + (c / K_CPR) * K_ROW + (c % K_CPR) * 16) = keep_or_zero(rk[i], keep);
            *(frag_t*)(lds_v + (c / K_CPR) * V_ROW + (c % K_CPR) * 16) = keep_or_zero(rv[i], keep);
        }
        if (tid < BKEY) {                                  // tid < 64 is exactly wave 0
            const bool valid = (rkey0 + tid) < a.Lk && rmkb != 0.f;
            lds_bias[tid] = valid ? 0.f : -INFINITY;
            const int flag = __any(!valid) ? 1 : 0;
            if (tid == 0) lds_flag[0] = flag;
        }
    };

    f32x16 o[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // keys after the last valid one contribute exactly 0: stop there (padding is a suffix in the dataset's masks)
    int64_t lk_eff = a.Lk;
    if (maskg) {
        int last = -1;
        for (int j = tid; j < (int)a.Lk; j += NTHREADS)
            if (maskg[j] != 0.f) last = j;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) last = max(last, __shfl_xor(last, o2));
        if (lane == 0) lds_flag[1 + wave] = last;
        __syncthreads();
        lk_eff = max(max(lds_flag[1], lds_flag[2]), max(lds_flag[3], lds_flag[4])) + 1;
    }
    const int64_t ntiles = (lk_eff + BKEY - 1) / BKEY;
    // (the whole tile loop exists twice, with and without a mask: selecting per load would join two register-loading paths in
    // front of the MFMAs, and the join waits for the loads)
    auto tile_loop = [&](auto has_mask) __attribute__((always_inline)) {
    if (ntiles > 0) load_tile_impl(0, has_mask);
    for (int64_t t = 0; t < ntiles; ++t) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile_impl((t + 1) * BKEY, has_mask);
        if (!wave_active) continue;

        // ---- S^T tile [64 keys x 32 queries] = two 32x32 accumulators
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
        if constexpr (X3) {
            // split-bf16 products (common.h): two 8-deep steps per product (NQF is even: head dims 32 / 64 / 128)
#pragma unroll
            for (int ks = 0; ks < NQF; ks += 2) {
                const SplitF32x4 q0 = made_split4(qf[ks]), q1 = made_split4(qf[ks + 1]);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const SplitF32x4 k0 = made_split4(*(const f32x4*)(lds_k + (kt * 32 + r) * K_ROW + ks * 32 + hh * 16));
                    const SplitF32x4 k1 = made_split4(*(const f32x4*)(lds_k + (kt * 32 + r) * K_ROW + (ks + 1) * 32 + hh * 16));
                    s[kt] = made_mfma_x3_16(k0, k1, q0, q1, s[kt]);
                }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < NQF; ++ks) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                frag_t kf = *(const frag_t*)(lds_k + (kt * 32 + r) * K_ROW + ks * 32 + hh * 16);
                if constexpr (IS_BF16) {
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kt], 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[ks][e], s[kt], 0, 0, 0);
                }
            }
        }
        }

        // ---- online softmax (per query = per lane column; the two lane halves hold different keys)
        if constexpr (IS_BF16) {
            // VALU diet: scores stay raw; exp2 with one FMA per element (scale*log2e folded), the key-mask bias only on
            // tiles that contain a masked key, the O / l rescale only when some lane's running max moved (wave-uniform)
            const float c = a.scale * 1.4426950408889634f;
            if (lds_flag[0] != 0) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[kt][e] += lds_bias[kt * 32 + acc_row(e, hh)];
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[kt][e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            if (!__all(m_new == m_run)) {
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * c);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < NDT; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
                m_run = m_new;
            }
            const float mc = m_use * c;
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][e], c, -mc));
                    s[kt][e] = p;
                    psum += p;
                }
            l_run += psum;
        } else {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = s[kt][e] * a.scale + lds_bias[kt * 32 + acc_row(e, hh)];
                    s[kt][e] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = expf(m_run - m_use);
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float p = expf(s[kt][e] - m_use);
                    s[kt][e] = p;
                    psum += p;
                }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
        }

        // ---- training: dropout on the attention weights (the row sum l keeps the undropped probabilities)
        if (a.drop.p > 0.f) {
            const uint32_t thr = made_drop_threshold(a.drop.p);
            const float sc = 1.f / (1.f - a.drop.p);
            int64_t qq = q0 + r; qq = qq < a.Lq ? qq : a.Lq - 1;
            const uint64_t base = (uint64_t)((b * a.H + h) * a.Lq + qq) * (uint64_t)a.Lk + (uint64_t)(t * BKEY);
            const uint32_t lo = (uint32_t)base;
            // kb[kt]: this lane's 16 decisions of key block kt in register order (bit e = element e kept), for the backward's bit cache
            uint32_t kb[2] = {0u, 0u};
            if (__all(lo <= 0xFFFFFFFFu - BKEY)) {           // the tile's indices share their high word: hoist the key
                const uint32_t kk = made_rng_key(drop_seed, a.drop.site, (uint32_t)(base >> 32));
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t hsh = made_rng_fmix32((lo + (uint32_t)(kt * 32 + acc_row(e, hh))) ^ kk);
                        const bool keep = (hsh >> 8) >= thr;
                        s[kt][e] = keep ? s[kt][e] * sc : 0.f;
                        kb[kt] |= keep ? (1u << e) : 0u;
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t hsh = made_rng_mix(drop_seed, a.drop.site, base + (uint64_t)(kt * 32 + acc_row(e, hh)));
                        const bool keep = (hsh >> 8) >= thr;
                        s[kt][e] = keep ? s[kt][e] * sc : 0.f;
                        kb[kt] |= keep ? (1u << e) : 0u;
                    }
            }
            if (a.keep_bits) {
                // register order -> key order: nibble j (elements 4j .. 4j + 3) holds keys 8j + 4 hh .. + 3; the two lane halves hold
                // disjoint keys of the block, the lower half stores the word pair of the tile
                uint32_t w2[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    uint32_t x = kb[kt];
                    x = (x | (x << 8)) & 0x00FF00FFu;
                    x = (x | (x << 4)) & 0x0F0F0F0Fu;
                    x <<= 4 * hh;
                    w2[kt] = x | (uint32_t)__shfl_xor((int)x, 32);
                }
                // layout (made_hip.h, MadeAttnArgs.keep_bits): [pair][32-key tile][query slot], the 32 slots of a query group permuted so
                // that the group's words are the 16 lane masks of a 32 x 32 accumulator tile (made_attention_bwd's single-pass kernel)
                const int64_t qs = q0 + r;
                if (hh == 0 && qs < a.Lq) {
                    const int64_t nkt = (a.Lk + 31) / 32;
                    uint32_t* bp = a.keep_bits + ((b * a.H + h) * nkt + 2 * t) * a.ld_bits + q0 + made_keep_slot(r);
                    bp[0] = w2[0];
                    if (2 * t + 1 < nkt) bp[a.ld_bits] = w2[1];
                }
            }
        }

        // ---- O^T += V^T P^T
        if constexpr (IS_BF16) {
            const int g = lane >> 4, i = lane & 15;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[kt][8 * s2 + j];
                    // transposing read: the 16 lanes of group g fetch a 4-key x 16-d block; lane 4q+p supplies the
                    // address of key row kb+q, d columns 4p..4p+3 and receives column (lane & 15) of the 4 rows
                    const int kb = kt * 32 + 16 * s2 + 4 * (g >> 1);
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        const unsigned char* vp = lds_v + (kb + (i >> 2)) * V_ROW + (d * 32 + (g & 1) * 16 + 4 * (i & 3)) * 2;
                        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)vp);
                        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + 8 * V_ROW));
                        bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
                    }
                }
        } else if constexpr (X3) {
            // split-bf16 products: the four keys of a register quad (rows 8 g + 4 hh .. + 3 of a key block) are one 8-deep step
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4 += 2) {            // two register quads = two 8-deep steps per product
                    const SplitF32x4 pb0 = made_split4(f32x4{s[kt][4 * g4], s[kt][4 * g4 + 1], s[kt][4 * g4 + 2], s[kt][4 * g4 + 3]});
                    const SplitF32x4 pb1 = made_split4(f32x4{s[kt][4 * g4 + 4], s[kt][4 * g4 + 5], s[kt][4 * g4 + 6], s[kt][4 * g4 + 7]});
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        f32x4 v0, v1;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v0[j] = *(const float*)(lds_v + (kt * 32 + 8 * g4 + 4 * hh + j) * V_ROW + (d * 32 + r) * 4);
                            v1[j] = *(const float*)(lds_v + (kt * 32 + 8 * g4 + 8 + 4 * hh + j) * V_ROW + (d * 32 + r) * 4);
                        }
                        o[d] = made_mfma_x3_16(made_split4(v0), made_split4(v1), pb0, pb1, o[d]);
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
                        float vv = *(const float*)(lds_v + key * V_ROW + (d * 32 + r) * 4);
                        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, s[kt][e], o[d], 0, 0, 0);
                    }
                }
        }
    }
    };
    if (maskg) tile_loop(std::true_type{}); else tile_loop(std::false_type{});

    if (!wave_active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    float inv = 1.f / l_tot;
    const int64_t q = q0 + r;
    if (q >= a.Lq) return;
    if (a.lse && hh == 0) {
        // natural-log units of the scaled scores; the bf16 path keeps its running max on the RAW scores
        const float m_nat = IS_BF16 ? m_run * a.scale : m_run;
        a.lse[(b * a.H + h) * a.Lq + q] = l_tot > 0.f ? m_nat + logf(l_tot) : INFINITY;
    }
    bool zero_row = a.q_mask != nullptr && a.q_mask[b * a.Lq + q] == 0.f;
    TC* op = (TC*)a.O + b * a.o_bs + q * a.ldo + h * HD;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v0 = o[d][4 * g + 0] * inv, v1 = o[d][4 * g + 1] * inv;
            float v2 = o[d][4 * g + 2] * inv, v3 = o[d][4 * g + 3] * inv;
            if (zero_row) { v0 = v1 = v2 = v3 = 0.f; }
            TC* dst = op + d * 32 + 8 * g + 4 * hh;
            if constexpr (IS_BF16) {
                bf16x4 pk;
                pk[0] = (bf16_t)v0; pk[1] = (bf16_t)v1; pk[2] = (bf16_t)v2; pk[3] = (bf16_t)v3;
                *(bf16x4*)dst = pk;
            } else {
                f32x4 pk;
                pk[0] = v0; pk[1] = v1; pk[2] = v2; pk[3] = v3;
                *(f32x4*)dst = pk;
            }
        }
}

template <typename TC>
int launch_attention(const MadeAttnArgs& a, hipStream_t st) {
    dim3 grid((unsigned)(((a.Lq + BQ - 1) / BQ) * 8 * ((a.H * a.B + 7) / 8))), block(NTHREADS);
    switch (a.hd) {
        case 32:
            if (sizeof(TC) == 4 && g_made_f32_products) hipLaunchKernelGGL((attention_kernel<TC, 32, sizeof(TC) == 4>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((attention_kernel<TC, 32>), grid, block, 0, st, a);
            break;
        case 64:
            if (sizeof(TC) == 4 && g_made_f32_products) hipLaunchKernelGGL((attention_kernel<TC, 64, sizeof(TC) == 4>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((attention_kernel<TC, 64>), grid, block, 0, st, a);
            break;
        case 128:
            if (sizeof(TC) == 4 && g_made_f32_products) hipLaunchKernelGGL((attention_kernel<TC, 128, sizeof(TC) == 4>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((attention_kernel<TC, 128>), grid, block, 0, st, a);
            break;
        default:
            made_set_error("made_attention: head dim %d not in {32,64,128}", a.hd);
            return MADE_ERR_UNSUPPORTED;
    }
    return made_check_launch("made_attention");
}

}  // namespace

extern "C" int made_attention(const MadeAttnArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_attention: null args");
    const MadeAttnArgs& a = *args;
    MADE_REQUIRE(a.Q && a.K && a.V && a.O, "made_attention: null tensor");
    MADE_REQUIRE(a.B >= 0 && a.H > 0 && a.Lq >= 0 && a.Lk > 0, "made_attention: bad dims");
    MADE_REQUIRE(a.dtype == MADE_F32 || a.dtype == MADE_BF16, "made_attention: bad dtype %d", a.dtype);
    if (a.keep_bits)
        MADE_REQUIRE(a.ld_bits >= 32 * ((a.Lq + 31) / 32) && a.ld_bits % 32 == 0 && ((uintptr_t)a.keep_bits % 8) == 0,
                     "made_attention: keep_bits rows need ld_bits = a multiple of 32 >= Lq (%lld words per key tile) and 8-byte alignment", (long long)(32 * ((a.Lq + 31) / 32)));
    MADE_UNSUPPORTED(((a.Lq + BQ - 1) / BQ) * a.H * a.B < (1LL << 31), "made_attention: too many workgroups");
    const int per16 = a.dtype == MADE_F32 ? 4 : 8;
    MADE_UNSUPPORTED(a.ldq % per16 == 0 && a.ldk % per16 == 0 && a.ldv % per16 == 0 && a.ldo % 4 == 0 &&
                     a.q_bs % per16 == 0 && a.k_bs % per16 == 0 && a.v_bs % per16 == 0 && a.o_bs % 4 == 0,
                     "made_attention: strides must keep 16-byte alignment");
    MADE_UNSUPPORTED(((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.K % 16) == 0 && ((uintptr_t)a.V % 16) == 0 && ((uintptr_t)a.O % 16) == 0,
                     "made_attention: base pointers must be 16-byte aligned");
    if (a.B == 0 || a.Lq == 0) return MADE_OK;
    hipStream_t st = (hipStream_t)stream;
    return a.dtype == MADE_BF16 ? launch_attention<bf16_t>(a, st) : launch_attention<float>(a, st);
}
