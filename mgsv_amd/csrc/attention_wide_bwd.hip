// made_attention_wide_bwd: backward of the moment-DETR decoder's cross-attention evaluated in MEMORY SPACE (made_attention_wide
// with few "query" rows q'_h = W_k,h^T q_h per sample; reference music_detr/transformer.py:293-296 + nn.MultiheadAttention's
// softmax / dropout, under model.train()).  bf16, gfx950.
//
// Round 2 ran this as four dependent launches per decoder layer on the backward's critical path -- dP = dO V^T (batched Linear),
// the row softmax backward over materialised f32 scores (which a fifth launch had recomputed for all layers), dq' = dS K (batched
// A^T B product) and the value-bias reduction in front of them -- 62 us per layer against ~18 us for streaming memory and
// memory + pos once.  Here one launch per layer does all of it, flash style:
//   * the row statistics the softmax backward needs come from the forward: lse (log-sum-exp of the scaled scores) and
//     delta = sum_j Pd_j dPd_j = dO . O + extra * ssum  (O = the saved pooled rows, ssum = the saved sum of the dropped weights,
//     extra = the gradient of that sum = <d attc_h, b_v,h>, reduced here from d attc), so the keys can be split over workgroups freely;
//   * a workgroup = one sample x one key slice, its four waves split D: partial S^T = K Q'^T and dP^T = V dO^T tiles of 32 keys meet
//     in LDS (only the <= 8 live query columns travel), every wave forms P, Pd, dS in registers, waves 0 / 1 store Pd / dS (the
//     operands of the layer-batched memory-gradient product that follows the decoder), and dQ'^T += K^T dS^T is accumulated per
//     D slice with K read through the transposing LDS read;
//   * K and V tiles are staged global -> LDS directly (global_load_lds), two stages; K carries the 64-byte-group swizzle the
//     transposing reads want (its row reads are then 2-way bank conflicted, which this latency-bound kernel does not notice),
//     V the 16-byte-chunk swizzle of the row reads;
//   * the key slices' partial dQ' meet in an f32 workspace and a small second launch sums them in slice order (summing them inside
//     the launch behind a ticket was built and measured in round 3: +15-20 us per launch, the agent-scope fences cost more than the
//     kernel boundary; profiles/r03_micro_merge_in_launch_vs_second_launch.txt).
#include "common.h"

namespace {

constexpr int BK = 32;        // keys per tile
constexpr int NT = 256;
constexpr int MAXQ = 8;       // query rows per sample (H * Q with one moment query)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int D>
__global__ __launch_bounds__(NT) void attention_wide_bwd_kernel(const MadeWideAttnBwdArgs a) {
    constexpr int DS = D / 4;                 // this wave's slice of D
    constexpr int NQF = DS / 16;              // 16-deep k-steps of the slice
    constexpr int NDT = DS / 32;              // 32-row tiles of the dQ'^T slice
    constexpr int ROWB = D * 2;               // bytes per (unpadded, swizzled) LDS row
    constexpr int CPR = ROWB / 16;            // 16-byte chunks per row
    constexpr int STAGE = 2 * BK * ROWB;      // K tile + V tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* lds_x = (float*)(lds + 2 * STAGE);                       // [4 waves][2 tensors][32 keys][8 queries]
    float* lds_q = lds_x + 4 * 2 * BK * MAXQ;                       // [8] extra, [4][8] partial dO.O
    uint32_t* lds_mbits = (uint32_t*)(lds_q + 8 + 4 * MAXQ);        // one bit per key: 1 = attended to

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int sl = wave;
    const int64_t b = blockIdx.y;
    const int NQ = (int)a.NQ;
    const int rq = (r & 7) < NQ ? (r & 7) : NQ - 1;                 // this lane's query row (lanes r >= 8 duplicate, never stored)
    const bool live = r < NQ;

    const bf16_t* Kg = (const bf16_t*)a.K + b * a.k_bs;
    const bf16_t* Vg = (const bf16_t*)a.V + b * a.v_bs;
    const float* maskg = a.key_mask ? a.key_mask + b * a.L : nullptr;

    // ---- fragments of this wave's D slice: lane (r, hh) holds X[rq][sl*DS + ks*16 + hh*8 ..]
    bf16x8 qf[NQF], dof[NQF];
    float dpart = 0.f;
    {
        const bf16_t* qp = (const bf16_t*)a.Q + b * a.q_bs + (int64_t)rq * a.ld_q + sl * DS + hh * 8;
        const bf16_t* gp = (const bf16_t*)a.dO + b * a.do_bs + (int64_t)rq * a.ld_do + sl * DS + hh * 8;
        const bf16_t* op = (const bf16_t*)a.O + b * a.o_bs + (int64_t)rq * a.ld_o + sl * DS + hh * 8;
        bf16x8 of[NQF];
#pragma unroll
        for (int ks = 0; ks < NQF; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 16); dof[ks] = *(const bf16x8*)(gp + ks * 16); of[ks] = *(const bf16x8*)(op + ks * 16); }
#pragma unroll
        for (int ks = 0; ks < NQF; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) dpart += (float)dof[ks][j] * (float)of[ks][j];
    }
    dpart += __shfl_xor(dpart, 32);
    if (r < MAXQ && hh == 0) lds_q[8 + wave * MAXQ + r] = dpart;
    const float lse = a.lse[b * NQ + rq];
    const float ssum = a.ssum ? a.ssum[b * NQ + rq] : 1.f;
    if (wave == 0) {
        // extra[q] = gradient of the sum of the dropped weights of row q: given, or <d attc_h, b_v,h> over head q's hd columns
        float ex = 0.f;
        if (a.extra) {
            ex = a.extra[b * NQ + ((lane >> 3) < NQ ? (lane >> 3) : NQ - 1)];
        } else if (a.dattc) {
            const int h = lane >> 3, per = (int)a.hd / 8;
            const bf16_t* dp_ = (const bf16_t*)a.dattc + b * a.ld_dattc + h * a.hd + (lane & 7) * per;
            const float* bp = a.vbias + h * a.hd + (lane & 7) * per;
            if (h < NQ)
                for (int e = 0; e < per; ++e) ex += (float)dp_[e] * bp[e];
            ex += __shfl_xor(ex, 1); ex += __shfl_xor(ex, 2); ex += __shfl_xor(ex, 4);
        }
        if ((lane & 7) == 0) lds_q[lane >> 3] = ex;
    }

    // ---- mask bits; keys after the last valid one contribute nothing (padding is a suffix in the dataset's masks)
    int64_t l_eff = a.L;
    int first_valid = 0;
    {
        const int lpad = (int)((a.L + 63) / 64) * 64;
        int last = -1, first = 0x7fffffff;
        for (int j = tid; j < lpad; j += NT) {
            const bool valid = j < (int)a.L && (maskg == nullptr || maskg[j] != 0.f);
            const unsigned long long bal = __ballot(valid);
            if (lane == 0) { lds_mbits[j / 32] = (uint32_t)bal; lds_mbits[j / 32 + 1] = (uint32_t)(bal >> 32); }
            if (valid) { last = j; first = min(first, j); }
        }
        if (tid == 0) lds_mbits[lpad / 32] = 0u;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { last = max(last, __shfl_xor(last, o2)); first = min(first, __shfl_xor(first, o2)); }
        int* red = (int*)lds_x;
        if (lane == 0) { red[wave] = last; red[4 + wave] = first; }
        __syncthreads();
        l_eff = max(max(red[0], red[1]), max(red[2], red[3])) + 1;
        first_valid = min(min(red[4], red[5]), min(red[6], red[7]));
        if (first_valid == 0x7fffffff) first_valid = 0;
        __syncthreads();
    }
    const float extra = lds_q[rq];
    const float delta = ((lds_q[8 + rq] + lds_q[8 + MAXQ + rq]) + (lds_q[8 + 2 * MAXQ + rq] + lds_q[8 + 3 * MAXQ + rq])) + extra * ssum;

    const int64_t nsplit = a.n_split > 1 ? a.n_split : 1;
    const int64_t tiles_all = (l_eff + BK - 1) / BK;
    const int64_t tiles_L = (a.L + BK - 1) / BK;
    const int64_t tiles_per = (tiles_all + nsplit - 1) / nsplit;
    const int64_t tile0 = (int64_t)blockIdx.z * tiles_per;
    const int64_t ntiles = tile0 >= tiles_all ? 0 : (tile0 + tiles_per <= tiles_all ? tiles_per : tiles_all - tile0);

    // tile `key0` -> stage: this wave moves pieces wave, wave + 4, ... of both tiles (1 KB each: D = 512 one row, D = 256 two)
    auto issue_tile = [&](int64_t key0, int stage) __attribute__((always_inline)) {
        constexpr int ROWS_PER_PIECE = 1024 / ROWB;
        constexpr int NPIECE = BK / ROWS_PER_PIECE;
        const uint32_t bits = lds_mbits[key0 / BK];
        unsigned char* st = lds + stage * STAGE;
        const unsigned char* Kb = (const unsigned char*)Kg;
        const unsigned char* Vb = (const unsigned char*)Vg;
        const uint32_t ldk_b = (uint32_t)a.ldk * 2, ldv_b = (uint32_t)a.ldv * 2;
#pragma unroll
        for (int i = 0; i < NPIECE / 4; ++i) {
            const int jp = wave + 4 * i;
            const int row = ROWS_PER_PIECE == 1 ? jp : 2 * jp + (lane >> 5);
            const uint32_t cl = ROWS_PER_PIECE == 1 ? (uint32_t)lane : (uint32_t)(lane & 31);
            const uint32_t srow = ((bits >> row) & 1u) ? (uint32_t)(key0 + row) : (uint32_t)first_valid;   // masked rows: any finite row
            const uint32_t c_row = cl ^ ((uint32_t)(row & 31) & (uint32_t)(CPR - 1));                      // V: 16-byte chunks by row (row reads)
            const uint32_t c_tr = ((((cl >> 2) ^ (uint32_t)(row & 7)) << 2) | (cl & 3));                    // K: 64-byte groups by row (transposing reads)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(Kb + (size_t)(srow * ldk_b + c_tr * 16u)), (lds_ptr_t)(st + jp * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(Vb + (size_t)(srow * ldv_b + c_row * 16u)), (lds_ptr_t)(st + BK * ROWB + jp * 1024), 16, 0, 0);
        }
    };

    f32x16 o[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    const uint32_t drop_thr = made_drop_threshold(a.drop.p);
    const uint64_t drop_seed = a.drop.p > 0.f ? made_drop_seed(a.drop) : 0;
    const float drop_sc = a.drop.p > 0.f ? 1.f / (1.f - a.drop.p) : 1.f;
    const uint64_t drop_base = (uint64_t)(b * NQ + rq) * (uint64_t)a.L;
    const bool drop_fast = (uint32_t)drop_base <= 0xFFFFFFFFu - (uint32_t)(a.L + 64);      // (the mix's key made once: see made_keep_bits)
    const uint32_t drop_key = made_rng_key(drop_seed, a.drop.site, (uint32_t)(drop_base >> 32));
    const uint32_t drop_lo = (uint32_t)drop_base;
    bf16_t* out_row = (wave == 0 ? (bf16_t*)a.Pd : (bf16_t*)a.dS) + b * a.p_bs + (int64_t)rq * a.ld_p;

    __builtin_amdgcn_s_waitcnt(0x0070);                             // (the fragment loads: hipcc does not see waits inside inline asm)
    if (ntiles > 0) issue_tile(tile0 * BK, 0);
    for (int64_t tt = 0; tt < ntiles; ++tt) {
        const int64_t t = tile0 + tt;
        const int cur = (int)(tt & 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile t landed; the other stage is free
        const uint32_t tbits = lds_mbits[t];
        const unsigned char* lds_k = lds + cur * STAGE;
        const unsigned char* lds_v = lds_k + BK * ROWB;

        // ---- partial S^T = K Q'^T and dP^T = V dO^T [32 keys x 32 query columns] over this wave's D slice
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < NQF; ++ks) {
            const int c = (sl * DS * 2) / 16 + ks * 2 + hh;
            const bf16x8 kf = *(const bf16x8*)(lds_k + r * ROWB + ((((c >> 2) ^ (r & 7)) << 6) | ((c & 3) << 4)));
            const bf16x8 vf = *(const bf16x8*)(lds_v + r * ROWB + (((c ^ r) & (CPR - 1)) << 4));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp, 0, 0, 0);
        }
        if (tt + 1 < ntiles) issue_tile((t + 1) * BK, cur ^ 1);
        // ---- the four waves' partial tiles meet in LDS (live query columns only)
        if (r < MAXQ) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                lds_x[((wave * 2 + 0) * BK + acc_row(e, hh)) * MAXQ + r] = s[e];
                lds_x[((wave * 2 + 1) * BK + acc_row(e, hh)) * MAXQ + r] = dp[e];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float pd[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int kr = acc_row(e, hh);
            const float* xs = lds_x + kr * MAXQ + (r & 7);
            const float S = (xs[0] + xs[2 * BK * MAXQ]) + (xs[4 * BK * MAXQ] + xs[6 * BK * MAXQ]);
            const float dP = (xs[BK * MAXQ] + xs[3 * BK * MAXQ]) + (xs[5 * BK * MAXQ] + xs[7 * BK * MAXQ]);
            const bool valid = (tbits >> kr) & 1u;
            const float p = valid ? __expf(S * a.scale - lse) : 0.f;
            bool kp = true;
            if (a.drop.p > 0.f) {
                const uint32_t kidx = (uint32_t)(t * BK + kr);
                const uint32_t hsh = drop_fast ? made_rng_fmix32((drop_lo + kidx) ^ drop_key) : made_rng_mix(drop_seed, a.drop.site, drop_base + (uint64_t)kidx);
                kp = (hsh >> 8) >= drop_thr;
            }
            pd[e] = kp ? p * drop_sc : 0.f;
            const float dpd = valid ? dP + extra : 0.f;
            const float dpu = kp ? dpd * drop_sc : 0.f;
            s[e] = p * (dpu - delta) * a.scale;
        }
        // ---- Pd (wave 0) and dS (wave 1) rows for the memory-gradient product
        if (wave < 2 && live) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int64_t key0 = t * BK + 8 * g4 + 4 * hh;
                if (key0 < a.ld_p) {
                    bf16x4 pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pk[j] = (bf16_t)(wave == 0 ? pd[4 * g4 + j] : s[4 * g4 + j]);
                    *(bf16x4*)(out_row + key0) = pk;
                }
            }
        }
        // ---- dQ'^T[slice] += K^T[slice x keys] dS^T[keys x queries]
        {
            const int g = lane >> 4, i = lane & 15;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[8 * s2 + j];
                const int kb = 16 * s2 + 4 * (g >> 1);
                const int row = kb + (i >> 2);
                const uint32_t vb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_k + row * ROWB + (g & 1) * 32 + (i & 3) * 8;
                bf16x4 lo[NDT], hi[NDT];
#pragma unroll
                for (int d = 0; d < NDT; ++d) {
                    const uint32_t va = vb + ((((sl * DS) / 32 + d) ^ (row & 7)) << 6);
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[d]) : "v"(va));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[d]) : "v"(va), "n"(8 * ROWB));
                }
#pragma unroll
                for (int d = 0; d < NDT; ++d) asm volatile("" : "+v"(lo[d]), "+v"(hi[d]));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int d = 0; d < NDT; ++d) {
                    asm volatile("" : "+v"(lo[d]), "+v"(hi[d]));
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(lo[d], hi[d], 0, 1, 2, 3, 4, 5, 6, 7), pf, o[d], 0, 0, 0);
                }
            }
        }
    }
    // ---- key tiles behind the last valid key: their Pd / dS columns are zero (the buffers are reused from batch to batch)
    if (wave < 2 && live) {
        for (int64_t tz = tiles_all + blockIdx.z; tz < tiles_L; tz += nsplit) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int64_t key0 = tz * BK + 8 * g4 + 4 * hh;
                if (key0 < a.ld_p) {
                    bf16x4 z4; z4[0] = z4[1] = z4[2] = z4[3] = (bf16_t)0.f;
                    *(bf16x4*)(out_row + key0) = z4;
                }
            }
        }
    }

    // ---- dQ': this slice's partial rows, merged over the key slices by the workgroup that finishes the sample last
    if (nsplit == 1) {
        if (live) {
            bf16_t* dq = (bf16_t*)a.dQ + b * a.dq_bs + (int64_t)r * a.ld_dq + sl * DS;
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    bf16x4 pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pk[j] = (bf16_t)o[d][4 * g4 + j];
                    *(bf16x4*)(dq + d * 32 + 8 * g4 + 4 * hh) = pk;
                }
        }
        return;
    }
    if (live) {
        float* po = a.part_dq + ((b * nsplit + blockIdx.z) * NQ + r) * D + sl * DS;
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 pk; pk[0] = o[d][4 * g4]; pk[1] = o[d][4 * g4 + 1]; pk[2] = o[d][4 * g4 + 2]; pk[3] = o[d][4 * g4 + 3];
                *(f32x4*)(po + d * 32 + 8 * g4 + 4 * hh) = pk;
            }
    }
    // (the slices are summed by wide_bwd_merge_kernel, a second launch)
}

// dQ[b, q, :] = sum over the key slices of part_dq[b, slice, q, :] (slice order), one wave per row
__global__ __launch_bounds__(NT) void wide_bwd_merge_kernel(const MadeWideAttnBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.B * a.NQ) return;
    const int64_t b = row / a.NQ, q = row % a.NQ;
    const int D = (int)a.D;
    const int per = D / 64;                                         // 8 or 4 columns per lane
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int64_t sp = 0; sp < a.n_split; ++sp) {
        const float* pp = a.part_dq + ((b * a.n_split + sp) * a.NQ + q) * D + lane * per;
        const f32x4 t0 = *(const f32x4*)pp;
        acc[0] += t0[0]; acc[1] += t0[1]; acc[2] += t0[2]; acc[3] += t0[3];
        if (per == 8) {
            const f32x4 t1 = *(const f32x4*)(pp + 4);
            acc[4] += t1[0]; acc[5] += t1[1]; acc[6] += t1[2]; acc[7] += t1[3];
        }
    }
    bf16_t* dq = (bf16_t*)a.dQ + b * a.dq_bs + q * a.ld_dq + lane * per;
    bf16x4 p0; p0[0] = (bf16_t)acc[0]; p0[1] = (bf16_t)acc[1]; p0[2] = (bf16_t)acc[2]; p0[3] = (bf16_t)acc[3];
    *(bf16x4*)dq = p0;
    if (per == 8) {
        bf16x4 p1; p1[0] = (bf16_t)acc[4]; p1[1] = (bf16_t)acc[5]; p1[2] = (bf16_t)acc[6]; p1[3] = (bf16_t)acc[7];
        *(bf16x4*)(dq + 4) = p1;
    }
}

template <int D>
int launch_wide_bwd(const MadeWideAttnBwdArgs& a, hipStream_t st) {
    constexpr size_t kBase = (size_t)2 * 2 * BK * D * 2 + (size_t)(4 * 2 * BK * MAXQ + 8 + 4 * MAXQ) * 4;
    const size_t lds_bytes = kBase + (size_t)((a.L + 63) / 64 * 2 + 2) * 4;
    if (lds_bytes > 160 * 1024) {
        made_set_error("made_attention_wide_bwd: L=%lld keys: the mask bit row does not fit in LDS beside the K / V stages", (long long)a.L);
        return MADE_ERR_UNSUPPORTED;
    }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)attention_wide_bwd_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            made_set_error("made_attention_wide_bwd: cannot reserve LDS: %s", hipGetErrorString(e));
            return MADE_ERR_HIP;
        }
        attr_done = true;
    }
    const int64_t nsplit = a.n_split > 1 ? a.n_split : 1;
    hipLaunchKernelGGL((attention_wide_bwd_kernel<D>), dim3(1, (unsigned)a.B, (unsigned)nsplit), dim3(NT), lds_bytes, st, a);
    int rc = made_check_launch("made_attention_wide_bwd");
    if (rc != MADE_OK || nsplit == 1) return rc;
    hipLaunchKernelGGL(wide_bwd_merge_kernel, dim3((unsigned)((a.B * a.NQ + 3) / 4)), dim3(NT), 0, st, a);
    return made_check_launch("made_attention_wide_bwd(merge)");
}

}  // namespace

extern "C" int made_attention_wide_bwd(const MadeWideAttnBwdArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_attention_wide_bwd: null args");
    const MadeWideAttnBwdArgs& a = *args;
    MADE_REQUIRE(a.Q && a.dO && a.O && a.K && a.V && a.lse && a.Pd && a.dS && a.dQ, "made_attention_wide_bwd: null tensor");
    MADE_REQUIRE(a.B >= 0 && a.NQ > 0 && a.L > 0, "made_attention_wide_bwd: bad dims");
    MADE_UNSUPPORTED(a.NQ <= MAXQ, "made_attention_wide_bwd: NQ=%lld query rows per sample (at most %d: one moment query)", (long long)a.NQ, MAXQ);
    MADE_UNSUPPORTED(a.D == 256 || a.D == 512, "made_attention_wide_bwd: D=%lld not in {256, 512}", (long long)a.D);
    MADE_UNSUPPORTED(a.B <= 65535, "made_attention_wide_bwd: B too large for the grid");
    MADE_UNSUPPORTED(a.q_bs % 8 == 0 && a.ld_q % 8 == 0 && a.do_bs % 8 == 0 && a.ld_do % 8 == 0 && a.o_bs % 8 == 0 && a.ld_o % 8 == 0 &&
                     a.k_bs % 8 == 0 && a.ldk % 8 == 0 && a.v_bs % 8 == 0 && a.ldv % 8 == 0 && a.p_bs % 4 == 0 && a.ld_p % 4 == 0 &&
                     a.dq_bs % 4 == 0 && a.ld_dq % 4 == 0, "made_attention_wide_bwd: strides must keep the rows aligned (16 bytes in, 8 bytes out)");
    MADE_UNSUPPORTED(((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.dO % 16) == 0 && ((uintptr_t)a.O % 16) == 0 && ((uintptr_t)a.K % 16) == 0 &&
                     ((uintptr_t)a.V % 16) == 0 && ((uintptr_t)a.Pd % 8) == 0 && ((uintptr_t)a.dS % 8) == 0 && ((uintptr_t)a.dQ % 8) == 0,
                     "made_attention_wide_bwd: base pointers must be aligned");
    MADE_REQUIRE(a.ld_p >= a.L, "made_attention_wide_bwd: ld_p < L");
    MADE_REQUIRE(a.drop.p >= 0.f && a.drop.p < 1.f, "made_attention_wide_bwd: dropout p out of [0,1)");
    if (a.dattc && !a.extra) {
        MADE_REQUIRE(a.vbias != nullptr && a.hd > 0 && a.hd % 8 == 0, "made_attention_wide_bwd: dattc needs vbias and hd (a multiple of 8)");
        // the value-bias term maps query ROW q to HEAD q (row q of a sample is head q of its one moment query): with several moment
        // queries per sample (rows = heads x queries) the caller passes `extra` instead
        MADE_UNSUPPORTED(a.hd * a.NQ == a.D, "made_attention_wide_bwd: dattc needs the one-query layout NQ * hd == D (NQ=%lld hd=%lld D=%lld); pass `extra` otherwise",
                         (long long)a.NQ, (long long)a.hd, (long long)a.D);
    }
    if (a.n_split > 1) {
        MADE_REQUIRE(a.part_dq != nullptr, "made_attention_wide_bwd: n_split > 1 needs part_dq");
        MADE_UNSUPPORTED(a.n_split <= 64, "made_attention_wide_bwd: n_split <= 64");
    }
    if (a.B == 0) return MADE_OK;
    hipStream_t st = (hipStream_t)stream;
    return a.D == 512 ? launch_wide_bwd<512>(a, st) : launch_wide_bwd<256>(a, st);
}
