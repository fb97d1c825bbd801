// made_attention_wide: single-head attention whose head dimension is the whole model width D
// (256 or 512), keys / values row-major, scores never in HBM.  gfx950.
//
// Two users on the MaDe path:
//   * the X-Pool block (reference modules/transformer.py:87-123): queries = all videos (shared by
//     every batch entry = music track), keys/values = that track's projected segments;
//   * the DETR decoder's cross-attention evaluated in MEMORY SPACE: for few queries per sample it is
//     cheaper to move W_k onto the query (q' = W_k,h^T q_h, one row of width D per head) and W_v onto
//     the pooled result than to project all L memory rows for all 6 layers -- then keys = memory + pos,
//     values = memory, "queries" = H*Q rows of width D per sample.
//
// One workgroup = 4 waves = 32 queries of one batch entry.  The four waves SPLIT D: wave w contracts
// its quarter of D in S^T = K Q^T (swapped product: query on the lane, keys in the registers), the four
// partial 32x32 tiles are summed through LDS so every wave holds the full scores, does the online
// softmax redundantly (cheap) and accumulates ITS quarter of O^T += V^T P^T, reading the row-major V
// tile as an MFMA A operand with ds_read_b64_tr_b16 (bf16) or plain ds_read_b32 (f32: lane = d).
// Keys are consumed 32 at a time; K (+Kadd) and V tiles are staged global -> registers -> LDS.
#include "common.h"

#include <type_traits>

namespace {

constexpr int WQ = 32;        // queries per workgroup
constexpr int WKEY = 32;      // keys per tile
constexpr int NTHREADS = 256;

template <typename TC> struct Frag;
template <> struct Frag<float>  { typedef f32x4  type; };
template <> struct Frag<bf16_t> { typedef bf16x8 type; };


// NSL = D slices per query tile: 4 -> one tile of 32 queries per workgroup, its four waves split D four ways; 2 -> two query
// tiles per workgroup (waves 0-1 and 2-3), each split two ways: the K / V tiles staged through LDS then serve 64 queries, which
// halves the staging per flop where a batch entry has 64 or more query rows (X-Pool: all videos against one track).
template <typename TC, int D, bool DB, int NSL, bool X3 = false>      // X3 (f32 only): split-bf16 products (common.h, made_set_f32_products)
__global__ __launch_bounds__(NTHREADS) void attention_wide_kernel(const MadeWideAttnArgs a) {
    typedef typename Frag<TC>::type frag_t;
    constexpr int SZ = (int)sizeof(TC);
    constexpr bool IS_BF16 = SZ == 2;
    constexpr int PER16 = 16 / SZ;
    constexpr int DS = D / NSL;                     // this wave's slice of D
    constexpr int WQB = WQ * (4 / NSL);             // queries per workgroup
    // DMA: the two-stage bf16 variants (few, latency-bound workgroups: decoder cross-attention, in-batch X-Pool) stage K / V tiles
    // global -> LDS directly (global_load_lds), so a whole tile is in flight under the current one without holding registers;
    // rows are unpadded and XOR-swizzled on the source side (16-byte chunks by row for the K row reads, 64-byte groups by row for
    // the transposing V reads); masked / out-of-range rows are fetched from the entry's first valid key instead of being zeroed.
    constexpr bool DMA = IS_BF16 && DB;
    constexpr int K_ROW = DMA ? D * SZ : D * SZ + 16;              // padded: conflict-free 16-byte row reads
    constexpr int V_ROW = DMA ? D * SZ : D * SZ + (IS_BF16 ? 64 : 16);   // bf16: 4 consecutive rows land on disjoint bank quarters (tr reads)
    constexpr int CPR = D * SZ / 16;                // 16-byte chunks per row
    constexpr int NCH = WKEY * CPR / NTHREADS;      // chunks per thread per tensor
    static_assert(WKEY * CPR % NTHREADS == 0, "staging split");
    constexpr int NQF = DS * SZ / 32;               // k-steps (16-byte fragment pairs) of this wave's QK slice
    constexpr int NDT = DS / 32;                    // 32-row tiles of this wave's O^T slice
    constexpr int KV_STAGE = WKEY * K_ROW + WKEY * V_ROW;
    constexpr int NSTAGE = DB ? 2 : 1;              // two K/V stages: one barrier less per tile and the loads overlap the MFMAs (few, latency-bound
                                                    // workgroups: the decoder); one stage leaves room for 3 workgroups per CU (many query tiles: X-Pool)

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* lds_s = (float*)(lds + NSTAGE * KV_STAGE);                 // [4][32*32] partial score tiles
    float* lds_bias_all = lds_s + 4 * 1024;                           // [NSTAGE][32]
    uint32_t* lds_mbits = (uint32_t*)(lds_bias_all + NSTAGE * 32);    // one bit per key of this batch entry: 1 = attended to

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int sl = wave % NSL, qt = wave / NSL;     // D slice and query tile of this wave
    // few query tiles per batch entry (in-batch X-Pool: 2): the batch index runs fastest, so the query tiles of one entry are
    // gridDim.x apart in dispatch order = on the same XCD when the batch is a multiple of 8, and share its K / V in that L2;
    // many query tiles (retrieval scale): the tiles of an entry are consecutive workgroups and share it in time instead
    const bool batch_fast = gridDim.x == (unsigned)a.B && (a.NQ1 * a.NQ2 + WQB - 1) / WQB != (int64_t)a.B;
    const int64_t b = batch_fast ? blockIdx.x : blockIdx.y;
    const int64_t nq_total = a.NQ1 * a.NQ2;
    const int64_t nq0 = (int64_t)(batch_fast ? blockIdx.y : blockIdx.x) * WQB + qt * WQ;

    const TC* Kg = (const TC*)a.K + b * a.k_bs;
    const TC* Ag = a.Kadd ? (const TC*)a.Kadd + b * a.kadd_bs : nullptr;
    const TC* Vg = (const TC*)a.V + b * a.v_bs;
    const float* maskg = a.key_mask ? a.key_mask + b * a.L : nullptr;

    // ---- Q fragments of this wave's D slice: lane (r, hh) holds Q[nq0 + r][w*DS + ks*2*PER16 + hh*PER16 ..]
    frag_t qf[NQF];
    int64_t my_q = nq0 + r;
    {
        int64_t q = my_q < nq_total ? my_q : nq_total - 1;
        const TC* qp = (const TC*)a.Q + b * a.q_bs + (q / a.NQ2) * a.q_s1 + (q % a.NQ2) * a.q_s2 + sl * DS;
#pragma unroll
        for (int ks = 0; ks < NQF; ++ks) qf[ks] = *(const frag_t*)(qp + ks * 2 * PER16 + hh * PER16);
    }

    // load_tile only ISSUES the next tile's loads; what consumes them (zeroing the rows of masked keys) runs in store_tile, after
    // the current tile's MFMAs: a use right behind the loads would make the wave wait for them first.  Which keys are masked comes
    // from a bit row staged in LDS at the start, so the loop loads nothing but K / V rows.
    frag_t rk[NCH], rv[NCH];
    int64_t rkey0 = 0;
    auto load_tile = [&](int64_t key0) __attribute__((always_inline)) {
        rkey0 = key0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + i * NTHREADS;
            int row = c / CPR, cc = c % CPR;
            int64_t key = key0 + row;
            const int64_t kcl = key < a.L ? key : a.L - 1;
            rk[i] = *(const frag_t*)(Kg + kcl * a.ldk + cc * PER16);
            rv[i] = *(const frag_t*)(Vg + kcl * a.ldv + cc * PER16);
        }
        if (Ag) {                                            // (K + Kadd formed here, at the price of waiting for both: not on the model's path,
#pragma unroll                                               // which hands in the precomputed sum)
            for (int i = 0; i < NCH; ++i) {
                int c = tid + i * NTHREADS;
                int64_t key = key0 + c / CPR;
                const int64_t kcl = key < a.L ? key : a.L - 1;
                const frag_t av = *(const frag_t*)(Ag + kcl * a.ldkadd + (c % CPR) * PER16);
#pragma unroll
                for (int j = 0; j < PER16; ++j) rk[i][j] = from_f32<TC>(to_f32(rk[i][j]) + to_f32(av[j]));
            }
        }
    };
    auto store_tile = [&](int stage) __attribute__((always_inline)) {
        unsigned char* sk = lds + stage * KV_STAGE;
        unsigned char* sv = sk + WKEY * K_ROW;
        const uint32_t bits = lds_mbits[rkey0 / WKEY];       // (WKEY = 32 keys = one word; tiles start at multiples of 32)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int c = tid + i * NTHREADS;
            const bool keep = (bits >> (c / CPR)) & 1u;      // masked / out-of-range keys read as zero rows
            *(frag_t*)(sk + (c / CPR) * K_ROW + (c % CPR) * 16) = keep_or_zero(rk[i], keep);
            *(frag_t*)(sv + (c / CPR) * V_ROW + (c % CPR) * 16) = keep_or_zero(rv[i], keep);
        }
        if (tid < WKEY) lds_bias_all[stage * WKEY + tid] = ((bits >> tid) & 1u) ? 0.f : -INFINITY;
    };

    f32x16 o[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    float sd_run = 0.f;                             // training: running sum of the DROPPED probabilities (they no longer sum to l)
    const uint32_t drop_thr = made_drop_threshold(a.drop.p);
    const uint64_t drop_seed = a.drop.p > 0.f ? made_drop_seed(a.drop) : 0;
    const float drop_sc = a.drop.p > 0.f ? 1.f / (1.f - a.drop.p) : 1.f;
    const uint64_t drop_base = (uint64_t)(b * nq_total + (my_q < nq_total ? my_q : nq_total - 1)) * (uint64_t)a.L;
    // (the mix's key depends on the index's high word only: made once when the row's indices do not cross a 2^32 boundary)
    const bool drop_fast = (uint32_t)drop_base <= 0xFFFFFFFFu - (uint32_t)(a.L + 64);
    const uint32_t drop_key = made_rng_key(drop_seed, a.drop.site, (uint32_t)(drop_base >> 32));
    const uint32_t drop_lo = (uint32_t)drop_base;

    // keys of this workgroup: all of them, or the blockIdx.z-th slice when the keys are split over workgroups
    // keys after the last valid one contribute exactly 0: stop there (padding is a suffix in the dataset's masks)
    int64_t l_eff = a.L;
    int first_valid = 0;
    {
        const int lpad = (int)((a.L + 63) / 64) * 64;
        int last = -1, first = 0x7fffffff;
        for (int j = tid; j < lpad; j += NTHREADS) {
            const bool valid = j < (int)a.L && (maskg == nullptr || maskg[j] != 0.f);
            const unsigned long long bal = __ballot(valid);
            if (lane == 0) { lds_mbits[j / 32] = (uint32_t)bal; lds_mbits[j / 32 + 1] = (uint32_t)(bal >> 32); }
            if (valid) { last = j; first = min(first, j); }
        }
        if (tid == 0) lds_mbits[lpad / 32] = 0u;
        if (maskg) {
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) { last = max(last, __shfl_xor(last, o2)); first = min(first, __shfl_xor(first, o2)); }
            int* red = (int*)lds_s;
            if (lane == 0) { red[wave] = last; red[4 + wave] = first; }
            __syncthreads();
            l_eff = max(max(red[0], red[1]), max(red[2], red[3])) + 1;
            first_valid = min(min(red[4], red[5]), min(red[6], red[7]));
            if (first_valid == 0x7fffffff) first_valid = 0;
        }
        __syncthreads();
    }
    const int64_t nsplit = a.n_split > 1 ? a.n_split : 1;
    const int64_t tiles_all = (l_eff + WKEY - 1) / WKEY;
    const int64_t tiles_per = (tiles_all + nsplit - 1) / nsplit;
    const int64_t tile0 = (int64_t)blockIdx.z * tiles_per;
    const int64_t ntiles = tile0 >= tiles_all ? 0 : (tile0 + tiles_per <= tiles_all ? tiles_per : tiles_all - tile0);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    // (DMA) tile `key0` -> stage: this wave moves pieces wave, wave + 4, ... of both tiles (1 KB each: D = 512 one row, D = 256 two)
    auto issue_tile = [&](int64_t key0, int stage) __attribute__((always_inline)) {
        constexpr int ROWS_PER_PIECE = DMA ? 1024 / (D * SZ) : 1;
        constexpr int NPIECE = WKEY / ROWS_PER_PIECE;
        constexpr int CPRW = D * SZ / 16;                          // chunks per row
        const uint32_t bits = lds_mbits[key0 / WKEY];
        unsigned char* st = lds + stage * KV_STAGE;
        const unsigned char* Kb = (const unsigned char*)Kg;
        const unsigned char* Vb = (const unsigned char*)Vg;
        const uint32_t ldk_b = (uint32_t)a.ldk * SZ, ldv_b = (uint32_t)a.ldv * SZ;
#pragma unroll
        for (int i = 0; i < NPIECE / 4; ++i) {
            const int jp = wave + 4 * i;
            const int row = ROWS_PER_PIECE == 1 ? jp : 2 * jp + (lane >> 5);
            const uint32_t cl = ROWS_PER_PIECE == 1 ? (uint32_t)lane : (uint32_t)(lane & 31);
            const uint32_t srow = ((bits >> row) & 1u) ? (uint32_t)(key0 + row) : (uint32_t)first_valid;
            const uint32_t ck = cl ^ ((uint32_t)(row & 31) & (uint32_t)(CPRW - 1));
            const uint32_t cu = ((((cl >> 2) ^ (uint32_t)(row & 7)) << 2) | (cl & 3));
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(Kb + (size_t)(srow * ldk_b + ck * 16u)), (lds_ptr_t)(st + jp * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(Vb + (size_t)(srow * ldv_b + cu * 16u)), (lds_ptr_t)(st + WKEY * K_ROW + jp * 1024), 16, 0, 0);
        }
    };
    if constexpr (DMA) {
        __builtin_amdgcn_s_waitcnt(0x0070);                         // (the Q fragment loads: see csrc/xpool_fused.hip on why a builtin)
        if (ntiles > 0) issue_tile(tile0 * WKEY, 0);
    } else if (NSTAGE == 2 && ntiles > 0) {
        load_tile(tile0 * WKEY);
        store_tile(0);
        __syncthreads();
    }
    for (int64_t tt = 0; tt < ntiles; ++tt) {
        const int64_t t = tile0 + tt;
        const int cur = NSTAGE == 2 ? (int)(tt & 1) : 0;
        uint32_t tbits = 0xffffffffu;
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile t landed; the other stage is free
            tbits = lds_mbits[t];
        } else if (NSTAGE == 2) {
            if (tt + 1 < ntiles) load_tile((t + 1) * WKEY);   // in flight during this tile's MFMAs; stored at the end of the tile
        } else {
            load_tile(t * WKEY);
            __syncthreads();                                  // previous tile fully consumed
            store_tile(0);
            __syncthreads();
        }
        const unsigned char* lds_k = lds + cur * KV_STAGE;
        const unsigned char* lds_v = lds_k + WKEY * K_ROW;
        const float* lds_bias = lds_bias_all + cur * WKEY;

        // ---- partial S^T [32 keys x 32 queries] over this wave's D slice
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        if constexpr (X3) {
            // split-bf16 products (common.h): two 8-deep steps per product (f32: never the DMA layout; NQF is even)
            static_assert(!X3 || (NQF % 2 == 0 && !DMA), "f32 slices of 16 columns or more");
#pragma unroll
            for (int ks = 0; ks < NQF; ks += 2) {
                const SplitF32x4 k0 = made_split4(*(const f32x4*)(lds_k + r * K_ROW + sl * DS * SZ + ks * 32 + hh * 16));
                const SplitF32x4 k1 = made_split4(*(const f32x4*)(lds_k + r * K_ROW + sl * DS * SZ + (ks + 1) * 32 + hh * 16));
                s = made_mfma_x3_16(k0, k1, made_split4(qf[ks]), made_split4(qf[ks + 1]), s);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < NQF; ++ks) {
            frag_t kf;
            if constexpr (DMA) kf = *(const frag_t*)(lds_k + r * K_ROW + ((((sl * DS * SZ) / 16 + ks * 2 + hh) ^ r) << 4));   // (chunk ^ row)
            else kf = *(const frag_t*)(lds_k + r * K_ROW + sl * DS * SZ + ks * 32 + hh * 16);
            if constexpr (IS_BF16) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[ks][e], s, 0, 0, 0);
            }
        }
        }
        if constexpr (DMA) {
            // the next tile's pieces are issued behind this tile's score MFMAs (a piece costs its wave 60-180 cycles of issue)
            if (tt + 1 < ntiles) issue_tile((t + 1) * WKEY, cur ^ 1);
        }
        // ---- sum the four partial tiles through LDS; every wave ends with the full tile
#pragma unroll
        for (int e = 0; e < 16; ++e) lds_s[wave * 1024 + acc_row(e, hh) * 32 + r] = s[e];
        // LDS-only wait + raw barrier: __syncthreads() would also drain vmcnt and stall on the prefetched tile
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int idx = acc_row(e, hh) * 32 + r;
            float v;
            if constexpr (NSL == 4) v = (lds_s[idx] + lds_s[1024 + idx]) + (lds_s[2048 + idx] + lds_s[3072 + idx]);
            else v = lds_s[qt * 2048 + idx] + lds_s[qt * 2048 + 1024 + idx];
            if constexpr (DMA) v = ((tbits >> acc_row(e, hh)) & 1u) ? v * a.scale : -INFINITY;
            else v = v * a.scale + lds_bias[acc_row(e, hh)];
            s[e] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = IS_BF16 ? __expf(m_run - m_use) : expf(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float p = IS_BF16 ? __expf(s[e] - m_use) : expf(s[e] - m_use);
            s[e] = p;
            psum += p;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (a.drop.p > 0.f) {                       // dropout on the attention weights; l keeps the undropped sum
            float dsum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t kidx = (uint32_t)(t * WKEY + acc_row(e, hh));
                const uint32_t hsh = drop_fast ? made_rng_fmix32((drop_lo + kidx) ^ drop_key) : made_rng_mix(drop_seed, a.drop.site, drop_base + (uint64_t)kidx);
                const bool kp = (hsh >> 8) >= drop_thr;
                s[e] = kp ? s[e] * drop_sc : 0.f;
                dsum += s[e];
            }
            sd_run = sd_run * alpha + dsum;
        }
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[d][e] *= alpha;

        // ---- O^T[slice] += V^T[slice x keys] P^T[keys x queries]
        if constexpr (IS_BF16) {
            const int g = lane >> 4, i = lane & 15;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[8 * s2 + j];
                // block rows = keys kb + (0..3) [+8 for the second read], block cols = 16 d's of this lane group
                const int kb = 16 * s2 + 4 * (g >> 1);
                if constexpr (DMA) {
                    // transposing reads as inline assembly with a hand-placed wait: as builtins behind an LDS-DMA the compiler guards them
                    // with s_waitcnt vmcnt(0), i.e. with a wait for the tile in flight (csrc/xpool_fused.hip)
                    const int row = kb + (i >> 2);
                    const uint32_t vb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_v + row * V_ROW + (g & 1) * 32 + (i & 3) * 8;
                    bf16x4 lo[NDT], hi[NDT];
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        const uint32_t va = vb + ((((sl * DS) / 32 + d) ^ (row & 7)) << 6);
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[d]) : "v"(va));
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[d]) : "v"(va), "n"(8 * V_ROW));
                    }
#pragma unroll
                    for (int d = 0; d < NDT; ++d) asm volatile("" : "+v"(lo[d]), "+v"(hi[d]));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        asm volatile("" : "+v"(lo[d]), "+v"(hi[d]));
                        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(lo[d], hi[d], 0, 1, 2, 3, 4, 5, 6, 7), pf, o[d], 0, 0, 0);
                    }
                } else {
#pragma unroll
                for (int d = 0; d < NDT; ++d) {
                    const int dcol = sl * DS + d * 32 + (g & 1) * 16 + 4 * (i & 3);
                    const unsigned char* vp = lds_v + (kb + (i >> 2)) * V_ROW + dcol * 2;
                    // (the _v4i16 flavour + per-element casts miscompiles on ROCm 7.2: keep whole-vector bf16 types)
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)vp);
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + 8 * V_ROW));
                    bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
                }
                }
            }
        } else if constexpr (X3) {
            // split-bf16 products: the four keys of a register quad (rows 8 g + 4 hh .. + 3) are one 8-deep step
#pragma unroll
            for (int g4 = 0; g4 < 4; g4 += 2) {                // two register quads = two 8-deep steps per product
                const SplitF32x4 pb0 = made_split4(f32x4{s[4 * g4], s[4 * g4 + 1], s[4 * g4 + 2], s[4 * g4 + 3]});
                const SplitF32x4 pb1 = made_split4(f32x4{s[4 * g4 + 4], s[4 * g4 + 5], s[4 * g4 + 6], s[4 * g4 + 7]});
#pragma unroll
                for (int d = 0; d < NDT; ++d) {
                    f32x4 v0, v1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v0[j] = *(const float*)(lds_v + (8 * g4 + 4 * hh + j) * V_ROW + (sl * DS + d * 32 + r) * 4);
                        v1[j] = *(const float*)(lds_v + (8 * g4 + 8 + 4 * hh + j) * V_ROW + (sl * DS + d * 32 + r) * 4);
                    }
                    o[d] = made_mfma_x3_16(made_split4(v0), made_split4(v1), pb0, pb1, o[d]);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = acc_row(e, hh);
#pragma unroll
                for (int d = 0; d < NDT; ++d) {
                    float vv = *(const float*)(lds_v + key * V_ROW + (sl * DS + d * 32 + r) * 4);
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, s[e], o[d], 0, 0, 0);
                }
            }
        }
        if (NSTAGE == 2 && !DMA) {
            if (tt + 1 < ntiles) store_tile(cur ^ 1);          // the other stage: nobody reads it during this tile
            __syncthreads();
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    const float sd_tot = sd_run + __shfl_xor(sd_run, 32);
    const bool live_q = my_q < nq_total;
    if (nsplit == 1 && !live_q) return;
    if (nsplit == 1 && sl == 0 && hh == 0) {
        if (a.sum_out) a.sum_out[b * nq_total + my_q] = a.drop.p > 0.f ? sd_tot * inv : 1.f;
        if (a.lse_out) a.lse_out[b * nq_total + my_q] = m_run + logf(l_tot);
    }
    if (nsplit > 1) {
        // un-normalised partial result of this key slice
        const int64_t prow = (b * nsplit + blockIdx.z) * nq_total + (live_q ? my_q : 0);
        if (live_q) {
            float* po = a.part_o + prow * D + sl * DS;
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 pk; pk[0] = o[d][4 * g4]; pk[1] = o[d][4 * g4 + 1]; pk[2] = o[d][4 * g4 + 2]; pk[3] = o[d][4 * g4 + 3];
                    *(f32x4*)(po + d * 32 + 8 * g4 + 4 * hh) = pk;
                }
            if (sl == 0 && hh == 0) {
                a.part_ml[prow * 4] = m_run; a.part_ml[prow * 4 + 1] = l_tot; a.part_ml[prow * 4 + 2] = a.drop.p > 0.f ? sd_tot : l_tot;
            }
        }
        // the slices are merged by made_attention_wide_combine, a second launch (a merge inside the launch behind a ticket -- the last
        // workgroup of an entry to sign in merges -- was built and measured in round 3: 15-35 us SLOWER per call, the agent-scope
        // release / acquire pair costs more than the kernel boundary; profiles/r03_micro_merge_in_launch_vs_second_launch.txt)
        return;
    }
    const int64_t obase = b * a.o_bs + (my_q / a.NQ2) * a.o_s1 + (my_q % a.NQ2) * a.o_s2 + sl * DS;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int64_t off = obase + d * 32 + 8 * g4 + 4 * hh;
            float v0 = o[d][4 * g4] * inv, v1 = o[d][4 * g4 + 1] * inv, v2 = o[d][4 * g4 + 2] * inv, v3 = o[d][4 * g4 + 3] * inv;
            if (a.o_dtype == MADE_F32) {
                f32x4 pk; pk[0] = v0; pk[1] = v1; pk[2] = v2; pk[3] = v3;
                *(f32x4*)((float*)a.O + off) = pk;
            } else {
                bf16x4 pk; pk[0] = (bf16_t)v0; pk[1] = (bf16_t)v1; pk[2] = (bf16_t)v2; pk[3] = (bf16_t)v3;
                *(bf16x4*)((bf16_t*)a.O + off) = pk;
            }
        }
}

// merge the key slices: O = sum_s O_s e^{m_s - M} / sum_s l_s e^{m_s - M}; one wave per (batch, query).  Up to four slices (the usual
// case) are requested together before the first use: slice after slice the row is four dependent round trips (29.6 us for the in-batch
// X-Pool's 4 096 rows x 4 slices against 33 MB of partial rows).
__global__ __launch_bounds__(NTHREADS) void attention_wide_combine_kernel(const MadeWideAttnArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t nq_total = a.NQ1 * a.NQ2;
    const int64_t row = (int64_t)blockIdx.x * (NTHREADS / 64) + (threadIdx.x >> 6);
    if (row >= a.B * nq_total) return;
    const int64_t b = row / nq_total, q = row % nq_total;
    const int D = (int)a.D;
    const int ns = (int)a.n_split;
    float Lsum = 0.f, Dsum = 0.f, Muse = 0.f;
    f32x4 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f; }
    if (ns <= 4) {
        f32x4 t[4][2];
        float ml[4][3];
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            const int64_t pr = (b * ns + (sp < ns ? sp : ns - 1)) * nq_total + q;
            ml[sp][0] = a.part_ml[pr * 4]; ml[sp][1] = a.part_ml[pr * 4 + 1]; ml[sp][2] = a.part_ml[pr * 4 + 2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = (i * 64 + lane) * 4;
                t[sp][i] = *(const f32x4*)(a.part_o + pr * D + (c < D ? c : 0));
            }
        }
        float M = -INFINITY;
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) if (sp < ns) M = fmaxf(M, ml[sp][0]);
        Muse = (M == -INFINITY) ? 0.f : M;
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            if (sp < ns) {                                      // (slice order: the same sums as the loop below)
                const float w = expf(ml[sp][0] - Muse);
                Lsum += ml[sp][1] * w;
                Dsum += ml[sp][2] * w;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += t[sp][i][j] * w;
            }
        }
    } else {
        float M = -INFINITY;
        for (int64_t s = 0; s < ns; ++s) M = fmaxf(M, a.part_ml[((b * ns + s) * nq_total + q) * 4]);
        Muse = (M == -INFINITY) ? 0.f : M;
        for (int64_t s = 0; s < ns; ++s) {
            const int64_t prow = (b * ns + s) * nq_total + q;
            const float w = expf(a.part_ml[prow * 4] - Muse);
            Lsum += a.part_ml[prow * 4 + 1] * w;
            Dsum += a.part_ml[prow * 4 + 2] * w;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int c = (i * 64 + lane) * 4;
                if (c < D) {
                    f32x4 tt = *(const f32x4*)(a.part_o + prow * D + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += tt[j] * w;
                }
            }
        }
    }
    const float inv = 1.f / Lsum;
    if (a.sum_out && lane == 0) a.sum_out[b * nq_total + q] = Dsum * inv;     // sum of the dropped weights (1 without dropout)
    if (a.lse_out && lane == 0) a.lse_out[b * nq_total + q] = Muse + logf(Lsum);
    const int64_t obase = b * a.o_bs + (q / a.NQ2) * a.o_s1 + (q % a.NQ2) * a.o_s2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < D) {
            if (a.o_dtype == MADE_F32) {
                f32x4 pk; pk[0] = acc[i][0] * inv; pk[1] = acc[i][1] * inv; pk[2] = acc[i][2] * inv; pk[3] = acc[i][3] * inv;
                *(f32x4*)((float*)a.O + obase + c) = pk;
            } else {
                bf16x4 pk; pk[0] = (bf16_t)(acc[i][0] * inv); pk[1] = (bf16_t)(acc[i][1] * inv);
                pk[2] = (bf16_t)(acc[i][2] * inv); pk[3] = (bf16_t)(acc[i][3] * inv);
                *(bf16x4*)((bf16_t*)a.O + obase + c) = pk;
            }
        }
    }
}

template <typename TC, int D, bool DB, int NSL>
int launch_wide(const MadeWideAttnArgs& a, hipStream_t st) {
    constexpr int SZ = (int)sizeof(TC);
    constexpr bool DMA = SZ == 2 && DB;                // (as in the kernel)
    constexpr int K_ROW = DMA ? D * SZ : D * SZ + 16, V_ROW = DMA ? D * SZ : D * SZ + (SZ == 2 ? 64 : 16);
    constexpr int NSTAGE = DB ? 2 : 1;
    constexpr size_t kBase = (size_t)NSTAGE * (WKEY * K_ROW + WKEY * V_ROW) + 4 * 1024 * 4 + NSTAGE * 32 * 4;
    constexpr size_t kCap = kBase + 8192 < 160 * 1024 ? kBase + 8192 : 160 * 1024;      // + the mask bit row (one bit per key)
    const size_t lds_bytes = kBase + (size_t)((a.L + 63) / 64 * 2 + 2) * 4;
    if (lds_bytes > kCap) {
        made_set_error("made_attention_wide: L=%lld keys: the mask bit row does not fit in LDS beside the K / V stages", (long long)a.L);
        return MADE_ERR_UNSUPPORTED;
    }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)attention_wide_kernel<TC, D, DB, NSL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCap);
        if (e == hipSuccess && SZ == 4)
            e = hipFuncSetAttribute((const void*)attention_wide_kernel<TC, D, DB, NSL, SZ == 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCap);
        if (e != hipSuccess) {
            made_set_error("made_attention_wide: cannot reserve %zu bytes of LDS: %s", kCap, hipGetErrorString(e));
            return MADE_ERR_HIP;
        }
        attr_done = true;
    }
    const int64_t nq = a.NQ1 * a.NQ2;
    const int64_t nsplit = a.n_split > 1 ? a.n_split : 1;
    constexpr int WQB = WQ * (4 / NSL);
    const int64_t qtiles = (nq + WQB - 1) / WQB;
    const bool batch_fast = qtiles > 1 && qtiles <= 8 && qtiles != a.B;       // see the kernel: which index runs fastest
    dim3 grid((unsigned)(batch_fast ? a.B : qtiles), (unsigned)(batch_fast ? qtiles : a.B), (unsigned)nsplit), block(NTHREADS);
    if (SZ == 4 && g_made_f32_products) hipLaunchKernelGGL((attention_wide_kernel<TC, D, DB, NSL, SZ == 4>), grid, block, lds_bytes, st, a);
    else hipLaunchKernelGGL((attention_wide_kernel<TC, D, DB, NSL>), grid, block, lds_bytes, st, a);
    int rc = made_check_launch("made_attention_wide");
    if (rc != MADE_OK || nsplit == 1) return rc;
    const int64_t rows = a.B * nq;
    hipLaunchKernelGGL(attention_wide_combine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(NTHREADS), 0, st, a);
    return made_check_launch("made_attention_wide(combine)");
}

}  // namespace

extern "C" int made_attention_wide(const MadeWideAttnArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_attention_wide: null args");
    const MadeWideAttnArgs& a = *args;
    MADE_REQUIRE(a.Q && a.K && a.V && a.O, "made_attention_wide: null tensor");
    MADE_REQUIRE(a.B >= 0 && a.NQ1 >= 0 && a.NQ2 > 0 && a.L > 0, "made_attention_wide: bad dims");
    MADE_REQUIRE(a.dtype == MADE_F32 || a.dtype == MADE_BF16, "made_attention_wide: bad dtype %d", a.dtype);
    MADE_REQUIRE(a.o_dtype == MADE_F32 || a.o_dtype == MADE_BF16, "made_attention_wide: bad o_dtype %d", a.o_dtype);
    MADE_UNSUPPORTED(a.D == 128 || a.D == 256 || a.D == 512, "made_attention_wide: D=%lld not in {128, 256, 512}", (long long)a.D);
    MADE_UNSUPPORTED(a.B <= 65535, "made_attention_wide: B too large for the grid");
    const int per16 = a.dtype == MADE_F32 ? 4 : 8;
    MADE_UNSUPPORTED(a.q_bs % per16 == 0 && a.q_s1 % per16 == 0 && a.q_s2 % per16 == 0 && a.k_bs % per16 == 0 && a.ldk % per16 == 0 &&
                     a.v_bs % per16 == 0 && a.ldv % per16 == 0 && a.kadd_bs % per16 == 0 && a.ldkadd % per16 == 0 &&
                     a.o_bs % 4 == 0 && a.o_s1 % 4 == 0 && a.o_s2 % 4 == 0,
                     "made_attention_wide: strides must keep 16-byte alignment");
    MADE_UNSUPPORTED(((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.K % 16) == 0 && ((uintptr_t)a.V % 16) == 0 && ((uintptr_t)a.O % 16) == 0 &&
                     ((uintptr_t)a.Kadd % 16) == 0, "made_attention_wide: base pointers must be 16-byte aligned");
    MADE_REQUIRE(a.drop.p >= 0.f && a.drop.p < 1.f, "made_attention_wide: dropout p out of [0,1)");
    if (a.n_split > 1) {
        MADE_REQUIRE(a.part_o != nullptr && a.part_ml != nullptr, "made_attention_wide: n_split > 1 needs part_o / part_ml");
        MADE_UNSUPPORTED(a.n_split <= 64, "made_attention_wide: n_split <= 64");
    }
    if (a.B == 0 || a.NQ1 == 0) return MADE_OK;
    hipStream_t st = (hipStream_t)stream;
    if (a.dtype == MADE_BF16) {
        const int64_t nq = a.NQ1 * a.NQ2;
        // few query tiles per batch entry: latency-bound, use the two-stage (LDS-DMA) variant -- which cannot add Kadd on the way
        const bool few = nq <= 64 && a.Kadd == nullptr;
        const bool pair = nq > 32 && nq <= 64 && a.Kadd == nullptr;   // exactly two query tiles per entry (in-batch X-Pool at B = 64): one workgroup
        if (a.D == 128) return launch_wide<bf16_t, 128, false, 4>(a, st);     // (narrow models: the register-staged variant only -- correct, not tuned)
        if (a.D == 512) {
            if (pair) return launch_wide<bf16_t, 512, true, 2>(a, st);
            return few ? launch_wide<bf16_t, 512, true, 4>(a, st) : launch_wide<bf16_t, 512, false, 4>(a, st);
        }
        if (pair) return launch_wide<bf16_t, 256, true, 2>(a, st);
        return few ? launch_wide<bf16_t, 256, true, 4>(a, st) : launch_wide<bf16_t, 256, false, 4>(a, st);
    }
    if (a.D == 128) return launch_wide<float, 128, false, 4>(a, st);
    return a.D == 512 ? launch_wide<float, 512, false, 4>(a, st) : launch_wide<float, 256, false, 4>(a, st);
}
