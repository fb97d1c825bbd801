// EXPERIMENT, not built into libmade_hip.so (see DESIGN.md section 6 and profiles/r01_f_wide_streaming_experiment.txt).
// A second made_attention_wide kernel for <= 64 query rows: K fragments straight from global, V by LDS-DMA into wave-private
// stages, no barriers at D = 256.  Parity-green (tests/test_ops_gpu.py::test_attention_wide_few_queries passed through it), but
// measured 86.7 / 50.2 / 33.0 us against the general kernel's 74.9 / 46.2 / 33.4 us at key splits 1 / 2 / 4 on the north_star
// shape: the shape is bound by how many bytes one CU keeps in flight and by the per-launch fixed cost, not by the instruction
// count this kernel removes.  Kept for the next attempt (deeper prefetch needs more LDS than a CU has at these tile sizes).
// made_attention_wide, second kernel: few query rows per batch entry (<= 64: the decoder's H*Q rows in memory space, the in-batch
// X-Pool block at B = 64), long key sequences, bf16, no dropout.  gfx950.
//
// The first kernel (attention_wide.hip) stages K and V tiles through registers into LDS, splits D over its four waves and
// exchanges partial score tiles through LDS: three barriers and ~750 VALU instructions per 32-key tile, 42 % of the wave cycles
// parked.  At these shapes the job is to STREAM K and V once, so here
//   * a wave-group (1 wave for D = 256, 2 for D = 512: 256 columns of D per wave) owns whole 32-key tiles: its K fragments come
//     straight from global memory as MFMA A operands (16 contiguous bytes of one key row per lane and K step) -- no staging, no
//     address arithmetic beyond a pointer bump, the next tile's fragments in flight under this tile's MFMAs;
//   * its V sub-tile goes global -> LDS by LDS-DMA (one instruction per key row and wave, no registers) into a double-buffered
//     region PRIVATE to the wave, and is read back transposed (ds_read_b64_tr_b16) as the A operand of O^T += V^T P^T: no
//     barrier guards it, only the wave's own vmcnt;
//   * the groups of a workgroup take different query tiles and/or different key subsets; partial (m, l, O) of key subsets are
//     merged through LDS at the end; the only barrier in the loop is the score exchange of the two D halves at D = 512.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int W2_T = 256;                 // threads
constexpr int W2_KEY = 32;                // keys per tile
constexpr int W2_DS = 256;                // columns of D per wave
constexpr int W2_VROW = W2_DS * 2;        // LDS row of a V sub-tile slice: 32 chunks of 16 bytes, chunk index ^= (row & 3) << 2
constexpr int W2_VSTAGE = W2_KEY * W2_VROW;           // 16 384 B
static_assert(8 * W2_VROW == 4096, "the transposed reads hard-code the 8-row offset");
constexpr int W2_LMAX = 2048;             // keys per batch entry this kernel accepts (one mask bit per key, 32 per lane)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int D>
__global__ __launch_bounds__(W2_T, 1) void attention_wide2_kernel(const MadeWideAttnArgs a) {
    constexpr int NSL = D / W2_DS;        // waves per group (D slices)
    constexpr int NG = 4 / NSL;           // groups per workgroup
    constexpr int NQF = W2_DS / 16;       // K steps of a wave's score product
    constexpr int NDT = W2_DS / 32;       // 32-row tiles of a wave's O^T slice
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* vbase = lds;                                        // [4 waves][2 stages][32][W2_VROW]
    float* lds_s = (float*)(lds + 4 * 2 * W2_VSTAGE);                  // [2 slots][4 waves][32*32] partial scores (D = 512)
    float* lds_ml = lds_s;                                             // [4 groups][32 queries][2] for the final merge (after the loop)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int sl = wave % NSL, grp = wave / NSL;
    const int64_t b = blockIdx.y;
    const int64_t nq_total = a.NQ1 * a.NQ2;
    const int nqt = (int)((nq_total + 31) / 32) < NG ? (int)((nq_total + 31) / 32) : NG;      // query tiles of this workgroup
    const int nks = NG / nqt;                                                               // key subsets (groups per query tile)
    const int qt = grp % nqt, ks = grp / nqt;
    const int64_t my_q = (int64_t)blockIdx.x * (32 * nqt) + qt * 32 + r;
    const int64_t qc = my_q < nq_total ? my_q : nq_total - 1;
    const int64_t L = a.L;

    const bf16_t* Kg = (const bf16_t*)a.K + b * a.k_bs + sl * W2_DS;
    const bf16_t* Vg = (const bf16_t*)a.V + b * a.v_bs + sl * W2_DS;
    const float* maskg = a.key_mask ? a.key_mask + b * L : nullptr;

    // ---- the batch entry's mask row as bits: lane t holds keys 32 t .. 32 t + 31 (tile t)
    unsigned mword = 0;
    if ((int64_t)lane * 32 < L) {
        if (maskg) {
            f32x4 mv[8];                                               // L % 4 == 0 (checked by the launcher): whole vectors only
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t key = (int64_t)lane * 32 + 4 * j;
                mv[j] = *(const f32x4*)(maskg + (key < L ? key : 0));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if ((int64_t)lane * 32 + 4 * j < L && mv[j][i] != 0.f) mword |= 1u << (4 * j + i);
        } else {
            const int64_t left = L - (int64_t)lane * 32;
            mword = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
        }
    }
    bf16x8 qf[NQF];
    {
        const bf16_t* qp = (const bf16_t*)a.Q + b * a.q_bs + (qc / a.NQ2) * a.q_s1 + (qc % a.NQ2) * a.q_s2 + sl * W2_DS + hh * 8;
#pragma unroll
        for (int k = 0; k < NQF; ++k) qf[k] = *(const bf16x8*)(qp + k * 16);
    }
    // keys after the last valid one contribute exactly 0: stop there
    int64_t l_eff;
    {
        int last = mword ? lane * 32 + 31 - __builtin_clz(mword) : -1;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) last = max(last, __shfl_xor(last, o2));
        l_eff = __builtin_amdgcn_readfirstlane(last + 1);
    }
    const int64_t nsplit = a.n_split > 1 ? a.n_split : 1;
    const int64_t tiles_all = (l_eff + W2_KEY - 1) / W2_KEY;
    const int64_t tiles_per = (tiles_all + nsplit - 1) / nsplit;
    const int64_t tile0 = (int64_t)blockIdx.z * tiles_per;
    const int64_t ntiles = tile0 >= tiles_all ? 0 : (tile0 + tiles_per <= tiles_all ? tiles_per : tiles_all - tile0);
    const int niter = (int)((ntiles + nks - 1) / nks);                 // the same for every group: the loop holds barriers
    const int64_t last_row = l_eff > 0 ? l_eff - 1 : 0;                // rows past it are never fetched (they may hold anything)

    unsigned char* vst = vbase + wave * 2 * W2_VSTAGE;                 // this wave's two V stages
    bf16x8 kf[NQF];
    // request tile `it` of this group.  V rows (this wave's 256 columns) by LDS-DMA, one 512-byte row per instruction on the lower
    // half-wave, 16-byte chunks permuted by the row (see W2_VROW); K fragments: lane (r, hh) = key row r, 16 bytes per K step.
    const unsigned ldv_u = (unsigned)a.ldv, ldk_u = (unsigned)a.ldk;  // 32 rows x ld fits 32 bits (checked by the launcher)
    auto issue_v = [&](int it) __attribute__((always_inline)) {
        const int64_t key0 = (tile0 + ks + (int64_t)it * nks) * W2_KEY;
        const int64_t kbase = key0 < last_row ? key0 : last_row;       // rows past the last valid key are never fetched
        const unsigned kmax = (unsigned)(last_row - kbase);
        const bf16_t* vb = Vg + kbase * a.ldv;                         // wave-uniform
        unsigned char* st = vst + (it & 1) * W2_VSTAGE;
        if (lane < 32) {
#pragma unroll
            for (unsigned k = 0; k < W2_KEY; ++k) {
                const bf16_t* row = vb + (k < kmax ? k : kmax) * ldv_u;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(row + (lane ^ ((k & 3) << 2)) * 8), (lds_ptr_t)(st + k * W2_VROW), 16, 0, 0);
            }
        }
    };
    auto issue_k = [&](int it) __attribute__((always_inline)) {
        const int64_t key0 = (tile0 + ks + (int64_t)it * nks) * W2_KEY;
        const int64_t kbase = key0 < last_row ? key0 : last_row;
        const unsigned kmax = (unsigned)(last_row - kbase);
        const bf16_t* kp = Kg + kbase * a.ldk + ((unsigned)r < kmax ? (unsigned)r : kmax) * ldk_u + hh * 8;
#pragma unroll
        for (int k = 0; k < NQF; ++k) kf[k] = *(const bf16x8*)(kp + k * 16);
    };

    f32x16 o[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c = a.scale * 1.4426950408889634f;
    const int g = lane >> 4, i16 = lane & 15;
    // transposed-read address of this lane inside a V stage (row part and permuted chunk part; d tile and key block added later)
    const int v_row = i16 >> 2;
    const int v_chunk = (g & 1) * 2 + ((i16 & 3) >> 1), v_byte = (i16 & 1) * 8, v_xor = v_row << 2;

    // One tile's requests (32 DMA + 16 loads per wave) are issued right after the previous tile's score product has consumed the
    // fragment registers; they land under that tile's softmax and P V.  The stage a request overwrites was read (and the reads
    // consumed by MFMAs) one tile earlier by this same wave; nobody else touches it.
    if (niter > 0) { issue_k(0); issue_v(0); }
#pragma unroll 1
    for (int it = 0; it < niter; ++it) {
        const unsigned char* lds_v = vst + (it & 1) * W2_VSTAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int64_t tl = tile0 + ks + (int64_t)it * nks;
        const bool tile_live = ks + (int64_t)it * nks < ntiles;        // (group-uniform) a subset may run out of tiles one early
        const unsigned mw = tile_live ? (unsigned)__builtin_amdgcn_readlane((int)mword, __builtin_amdgcn_readfirstlane((int)(tl & 63))) : 0u;

        // ---- S^T [32 keys x 32 queries] over this wave's D slice
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int k = 0; k < NQF; ++k) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[k], qf[k], s, 0, 0, 0);
        if constexpr (NSL == 2) {                                      // the two D halves meet through LDS (slot alternates per tile)
            float* slot = lds_s + (it & 1) * 4 * 1024;
#pragma unroll
            for (int e = 0; e < 16; ++e) slot[wave * 1024 + acc_row(e, hh) * 32 + r] = s[e];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const float* other = slot + (wave ^ 1) * 1024;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] += other[acc_row(e, hh) * 32 + r];
        }
        if (it + 1 < niter) { issue_k(it + 1); issue_v(it + 1); }
        // ---- online softmax per query (lane column); masked keys are selected out, never multiplied
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool ok = (mw >> acc_row(e, hh)) & 1u;
            s[e] = ok ? s[e] * c : -INFINITY;
            mx = fmaxf(mx, s[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = __builtin_amdgcn_exp2f(s[e] - m_use);
            psum += s[e];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (!__all(alpha == 1.f)) {
            // O^T lives in accumulation registers (MFMA-only); the (rare, after the first tiles) rescale goes through one
            // scratch VGPR per value in assembly so that the compiler does not shuttle all 128 of them every tile
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float t;
                    asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1"
                                 : "+a"(o[d][e]), "=&v"(t) : "v"(alpha));
                }
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        }
        // ---- O^T[slice] += V^T[slice x keys] P^T[keys x queries]: V rows read transposed out of the wave's own stage.
        // The reads are inline assembly: the compiler would otherwise hold every LDS read until the LDS-DMA requests of the NEXT
        // tile (just issued) have landed.  Each half (16 keys) issues its 16 reads at once and consumes them in two groups.
        const unsigned va = (unsigned)(uintptr_t)(lds_v - lds) + v_row * W2_VROW + v_byte;      // LDS byte address (dynamic LDS at 0)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[8 * s2 + j];
            const unsigned vk = va + (16 * s2 + 4 * (g >> 1)) * W2_VROW;
            unsigned ad[NDT];
#pragma unroll
            for (int d = 0; d < NDT; ++d) ad[d] = vk + (((d * 4 + v_chunk) ^ v_xor) * 16);
            bf16x4 lo[NDT], hi[NDT];
            asm volatile(
                "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:4096\n\t"
                "ds_read_b64_tr_b16 %2, %17\n\tds_read_b64_tr_b16 %3, %17 offset:4096\n\t"
                "ds_read_b64_tr_b16 %4, %18\n\tds_read_b64_tr_b16 %5, %18 offset:4096\n\t"
                "ds_read_b64_tr_b16 %6, %19\n\tds_read_b64_tr_b16 %7, %19 offset:4096\n\t"
                "ds_read_b64_tr_b16 %8, %20\n\tds_read_b64_tr_b16 %9, %20 offset:4096\n\t"
                "ds_read_b64_tr_b16 %10, %21\n\tds_read_b64_tr_b16 %11, %21 offset:4096\n\t"
                "ds_read_b64_tr_b16 %12, %22\n\tds_read_b64_tr_b16 %13, %22 offset:4096\n\t"
                "ds_read_b64_tr_b16 %14, %23\n\tds_read_b64_tr_b16 %15, %23 offset:4096\n\t"
                "s_waitcnt lgkmcnt(8)"
                : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2]), "=&v"(lo[3]), "=&v"(hi[3]),
                  "=&v"(lo[4]), "=&v"(hi[4]), "=&v"(lo[5]), "=&v"(hi[5]), "=&v"(lo[6]), "=&v"(hi[6]), "=&v"(lo[7]), "=&v"(hi[7])
                : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
                : "memory");
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bf16x8 vf = __builtin_shufflevector(lo[d], hi[d], 0, 1, 2, 3, 4, 5, 6, 7);
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(lo[4]), "+v"(hi[4]), "+v"(lo[5]), "+v"(hi[5]), "+v"(lo[6]), "+v"(hi[6]), "+v"(lo[7]), "+v"(hi[7])
                         :: "memory");
#pragma unroll
            for (int d = 4; d < NDT; ++d) {
                const bf16x8 vf = __builtin_shufflevector(lo[d], hi[d], 0, 1, 2, 3, 4, 5, 6, 7);
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
            }
        }
    }

    float l_tot = l_run + __shfl_xor(l_run, 32);
    // ---- merge the key subsets of a query tile: subset ks > 0 hands (m, l, O^T) to subset 0 through the V stages
    if (nks > 1) {
        __syncthreads();                                               // every wave is done with its stages
        float* mine = (float*)vst;                                     // two stages = 32 768 B = 256 x 32 x 4
        if (ks > 0) {
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) mine[(d * 32 + acc_row(e, hh)) * 32 + r] = o[d][e];
            if (sl == 0 && hh == 0) { lds_ml[(grp * 32 + r) * 2] = m_run; lds_ml[(grp * 32 + r) * 2 + 1] = l_tot; }
        }
        __syncthreads();
        if (ks > 0) return;
        for (int k2 = 1; k2 < nks; ++k2) {
            const int og = qt + k2 * nqt;                              // the group holding subset k2 of this query tile
            const float* theirs = (const float*)(vbase + (og * NSL + sl) * 2 * W2_VSTAGE);
            const float m_o = lds_ml[(og * 32 + r) * 2], l_o = lds_ml[(og * 32 + r) * 2 + 1];
            const float m_new = fmaxf(m_run, m_o);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float wa = __builtin_amdgcn_exp2f(m_run - m_use), wb = __builtin_amdgcn_exp2f(m_o - m_use);
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[d][e] = o[d][e] * wa + theirs[(d * 32 + acc_row(e, hh)) * 32 + r] * wb;
            l_tot = l_tot * wa + l_o * wb;
            m_run = m_new;
        }
    }
    if (my_q >= nq_total) return;
    if (nsplit > 1) {
        // un-normalised partial result of this key slice (running max in the log2 domain -> natural log for the merge kernel)
        const int64_t prow = (b * nsplit + blockIdx.z) * nq_total + my_q;
        float* po = a.part_o + prow * D + sl * W2_DS;
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 pk; pk[0] = o[d][4 * g4]; pk[1] = o[d][4 * g4 + 1]; pk[2] = o[d][4 * g4 + 2]; pk[3] = o[d][4 * g4 + 3];
                *(f32x4*)(po + d * 32 + 8 * g4 + 4 * hh) = pk;
            }
        if (sl == 0 && hh == 0) {
            a.part_ml[prow * 4] = m_run * 0.6931471805599453f; a.part_ml[prow * 4 + 1] = l_tot; a.part_ml[prow * 4 + 2] = l_tot;
        }
        return;
    }
    const float inv = 1.f / l_tot;
    const int64_t obase = b * a.o_bs + (my_q / a.NQ2) * a.o_s1 + (my_q % a.NQ2) * a.o_s2 + sl * W2_DS;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int64_t off = obase + d * 32 + 8 * g4 + 4 * hh;
            const float v0 = o[d][4 * g4] * inv, v1 = o[d][4 * g4 + 1] * inv, v2 = o[d][4 * g4 + 2] * inv, v3 = o[d][4 * g4 + 3] * inv;
            if (a.o_dtype == MADE_F32) {
                f32x4 pk; pk[0] = v0; pk[1] = v1; pk[2] = v2; pk[3] = v3;
                *(f32x4*)((float*)a.O + off) = pk;
            } else {
                bf16x4 pk; pk[0] = (bf16_t)v0; pk[1] = (bf16_t)v1; pk[2] = (bf16_t)v2; pk[3] = (bf16_t)v3;
                *(bf16x4*)((bf16_t*)a.O + off) = pk;
            }
        }
}

constexpr int W2_LDS = 4 * 2 * W2_VSTAGE + 2 * 4 * 1024 * 4;       // 160 KB: the whole CU

template <int D>
int launch_wide2(const MadeWideAttnArgs& a, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)attention_wide2_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS);
        if (e != hipSuccess) {
            made_set_error("made_attention_wide: cannot reserve %d bytes of LDS: %s", W2_LDS, hipGetErrorString(e));
            return MADE_ERR_HIP;
        }
        attr_done = true;
    }
    constexpr int NG = 4 / (D / W2_DS);
    const int64_t nq = a.NQ1 * a.NQ2;
    const int64_t nqt = (nq + 31) / 32 < NG ? (nq + 31) / 32 : NG;
    const int64_t nsplit = a.n_split > 1 ? a.n_split : 1;
    dim3 grid((unsigned)((nq + 32 * nqt - 1) / (32 * nqt)), (unsigned)a.B, (unsigned)nsplit), block(W2_T);
    hipLaunchKernelGGL((attention_wide2_kernel<D>), grid, block, W2_LDS, st, a);
    return made_check_launch("made_attention_wide");
}

}  // namespace

// returns MADE_OK after launching, or 1 when the arguments are not this kernel's (the caller then uses the general kernel)
int made_attention_wide2_try(const MadeWideAttnArgs& a, hipStream_t st) {
    const int64_t nq = a.NQ1 * a.NQ2;
    const bool mine = a.dtype == MADE_BF16 && (a.D == 256 || a.D == 512) && a.Kadd == nullptr && a.drop.p == 0.f && a.sum_out == nullptr &&
                      nq <= 64 && a.L <= W2_LMAX && a.L >= 64 && a.L % 4 == 0 && a.ldk < (1 << 24) && a.ldv < (1 << 24) &&
                      (a.key_mask == nullptr || (uintptr_t)a.key_mask % 16 == 0);
    static const bool off = getenv("MADE_WIDE_GENERAL") != nullptr;          // measurement knob: keep everything on the general kernel
    if (!mine || off) return 1;
    return a.D == 512 ? launch_wide2<512>(a, st) : launch_wide2<256>(a, st);
}
