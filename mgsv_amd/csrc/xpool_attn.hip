// made_xpool_attention: the attention of the X-Pool block at retrieval scale for head dim = model width D = 256 or 512
// (reference modules/transformer.py:87-123 inside Transformer_XA.forward :156-180: every video attends to the segments of every track,
// ONE head of width D), followed by the normalisation half of LayerNorm2 (:172).  gfx950, bf16.
//
//   xhat[m, n, :] = LN(softmax_s(<Q[n], K[m, s]> * scale + mask[m, s]) . U[m, s, :])         (no affine: the caller folds gamma / beta of
//                                                                                             LayerNorm2 into the Linear behind it)
//
// Why not the flash form (made_xpool_fused, D = 256): with the output of 32 videos in registers a wave needs 32 x D f32 = 256 registers at
// D = 512 before it holds a single operand, and splitting the width over waves makes every wave recompute (or exchange) the scores.
// Here the contraction is done in TWO PASSES over a track with the probabilities parked in LDS in between:
//   pass 1  S^T = K Q^T for all S <= 512 segments of the track: a wave keeps the Q rows of 32 videos in registers for the whole chunk of
//           tracks (its B operand), takes every fourth 32-segment tile of K, and keeps its <= 4 score tiles (64 registers) until the
//           track's last tile is in -- then ONE exchange of the per-video maxima (4 floats per video through LDS), exp2, and the
//           probabilities go to LDS as bf16 in exactly the register order the second product wants them in (the accumulator layout of
//           a 32 x 32 tile IS the K-slot order of the next MFMA's B operand, see made_xpool_fused);
//   pass 2  O^T = U^T P^T: wave w owns D / 8 output rows (width columns) for all 64 videos (64 accumulator registers at D = 512),
//           U^T fragments through the transposing LDS read, P^T fragments as 16-byte reads;
//   tail    O / l, the row statistics over D summed across the 8 waves through LDS, (x - mean) * rstd as bf16, 16-byte stores
//           (v_permlane32_swap pairs the two lane halves' 4-column groups).
// K and U move global -> LDS directly (global_load_lds, 1 KB per wave instruction, rows XOR-swizzled on the source side): pass 1 has
// four 32-segment stages (two batches of two tiles: one batch is multiplied -- by the four waves that own its tiles, one per SIMD --
// while the other is in flight), pass 2 a ring of four 16-segment half tiles in the first two stages; the probabilities overlay
// stages 2 and 3, which pass 2 does not use.  One workgroup = 64 videos x a chunk of tracks; per track at S = D = 512 a workgroup takes
// in 1 MB of K / U for 100 MFLOP, so the launch is bound by the CU's LDS-DMA intake, not by the matrix pipe.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int XA_PQ = 64;                          // videos per workgroup
constexpr int XA_T = 512;                          // threads
constexpr int XA_SMAX = 512;                       // segments per track
constexpr int XA_PP = XA_SMAX * 2 + 16;            // pitch of a video's probability row in LDS (bytes): 16 mod 256 -> conflict-free 16-byte reads
constexpr int XA_INFO = 32;                        // ints per track in the info table: [0] last valid + 1, [1] first valid, [16 .. 31] valid bits

template <int D> struct XaCfg {
    static constexpr int ROWB = D * 2;             // bytes of a K / U row
    static constexpr int CPR = ROWB / 16;          // 16-byte chunks per row
    static constexpr int RPP = 64 / CPR;           // rows per 1 KB LDS-DMA piece (1 at D = 512, 2 at D = 256)
    static constexpr int STG = 32 * ROWB;          // one 32-segment tile
    static constexpr int HSTG = 16 * ROWB;         // one 16-segment half tile (pass 2)
    static constexpr int PT = STG / 1024;          // pieces per tile
    static constexpr int PB = 2 * PT / 8;          // pieces per wave per batch of two tiles
    static constexpr int PH = HSTG / 1024 / 8;     // pieces per wave per half tile
    static constexpr int P_OFF = 2 * STG;          // the probabilities overlay stages 2 and 3
    static constexpr int MAX_OFF = P_OFF + XA_PQ * XA_PP;        // [4][64] f32: per-video maxima of the four tile classes
    static constexpr int SUM_OFF = MAX_OFF + 4 * XA_PQ * 4;      // [4][64] f32: their sums of exp2
    static constexpr int STAT_OFF = SUM_OFF + 4 * XA_PQ * 4;     // [8][64][2] f32: sum x, sum x^2 of a wave's columns
    static constexpr int LDS_BYTES = STAT_OFF + 8 * XA_PQ * 8;
    static constexpr int NQF = D / 16;             // Q fragments (16 columns each) per lane
    static constexpr int NDT = D / 8 / 32;         // 32-row output tiles per wave in pass 2
    static_assert(P_OFF + XA_PQ * XA_PP >= 4 * STG, "the probability rows must cover stages 2 and 3");
    static_assert(PH >= 1 && PB >= 1, "piece split");
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((address_space(3))) unsigned char* lds3_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float xa_other_half(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? sw[0] : sw[1]);
}
template <typename T> __device__ __forceinline__ T xa_rd(uint32_t addr) { return *(const __attribute__((address_space(3))) T*)(uintptr_t)addr; }
template <typename T> __device__ __forceinline__ void xa_wr(uint32_t addr, T v) { *(__attribute__((address_space(3))) T*)(uintptr_t)addr = v; }
#define XA_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// counted wait for this wave's own LDS-DMA pieces (the count is an immediate: one case per value used)
__device__ __forceinline__ void xa_wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}
// fragment reads as inline assembly: the compiler cannot tell an LDS read from a read of the stage an LDS-DMA is filling and would wait
// for vmcnt(0) in front of every one of them; completion is awaited by hand (the registers are tied to the wait statement, so nothing
// that uses them can be scheduled above it)
__device__ __forceinline__ bf16x8 xa_read128(uint32_t addr) { bf16x8 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); return v; }
template <int OFF> __device__ __forceinline__ bf16x8 xa_read128_off(uint32_t addr) {
    bf16x8 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v;
}
template <int N> __device__ __forceinline__ void xa_wait_lgkm(bf16x8& v) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N)); }
__device__ __forceinline__ uint32_t xa_opaque(uint32_t x) { asm volatile("" : "+v"(x)); return x; }

template <int D>
__global__ __launch_bounds__(XA_T, 1) void xpool_attn_kernel(const MadeXpoolAttnArgs a, const int* __restrict__ info, int tracks_per_chunk) {
    using C = XaCfg<D>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int vh = wave & 1, jw = wave >> 1, grp = jw >> 1;       // pass 1: video half, tile class (tiles t = jw mod 4), batch parity
    const int64_t n0 = (int64_t)blockIdx.x * XA_PQ;
    const int64_t m_begin = (int64_t)blockIdx.y * tracks_per_chunk;
    const int64_t m_end = (m_begin + tracks_per_chunk < a.Nm) ? m_begin + tracks_per_chunk : a.Nm;
    if (m_begin >= m_end) return;
    const int T = (int)(m_end - m_begin);
    const uint32_t lbase = (uint32_t)(uintptr_t)(lds3_t)lds;

    // ---- Q rows of this wave's 32 videos: B operand of S^T = K Q^T, lane (r, hh) holds Q[n][16 ks + 8 hh ..]
    bf16x8 qf[C::NQF];
    {
        const int64_t n = n0 + 32 * vh + r;
        const bf16_t* qp = (const bf16_t*)a.Q + (n < a.Nv ? n : a.Nv - 1) * a.ldq + hh * 8;
#pragma unroll
        for (int ks = 0; ks < C::NQF; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }
    const float c = a.scale * 1.4426950408889634f;

    // ---- per-lane constants of the LDS-DMA pieces: lane -> (row within the piece, 16-byte slot of the row)
    const uint32_t rip = (uint32_t)lane / C::CPR, slot = (uint32_t)lane % C::CPR;
    const uint32_t ldk_b = (uint32_t)a.ldk * 2u, ldu_b = (uint32_t)a.ldu * 2u;

    // K tiles 2b, 2b + 1 of track m -> stages (2b) & 3, (2b + 1) & 3.  Rows of masked / missing segments are fetched from the track's first
    // valid row (their probability is exactly 0; whatever a skipped projection tile left in memory must not reach the product)
    auto issue_k = [&](int64_t m, int b, int s_eff, int first) __attribute__((always_inline)) {
        const unsigned char* Kb = (const unsigned char*)a.K + m * a.k_bs * 2;
        const unsigned* ib = (const unsigned*)info + m * XA_INFO;
#pragma unroll
        for (int i = 0; i < C::PB; ++i) {
            const int p = wave * C::PB + i;                       // piece of the batch (wave-uniform)
            const int t = 2 * b + p / C::PT, pit = p % C::PT;
            const unsigned word = ib[16 + t];
            const uint32_t row = (uint32_t)pit * C::RPP + rip;    // row of the tile
            const int seg = t * 32 + (int)row;
            const bool valid = seg < s_eff && ((word >> row) & 1u);
            const uint32_t srow = (uint32_t)(valid ? seg : first);
            const uint32_t chunk = slot ^ (row & 15u);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(Kb + (size_t)(srow * ldk_b + chunk * 16u)), (lds_ptr_t)(lds + (t & 3) * C::STG + pit * 1024), 16, 0, 0);
        }
    };
    // U half tile h (16 segments) of track m -> ring slot h & 3 (64-byte groups swizzled by row for the transposing reads)
    auto issue_u = [&](int64_t m, int h, int s_eff, int first) __attribute__((always_inline)) {
        const unsigned char* Ub = (const unsigned char*)a.U + m * a.u_bs * 2;
        const unsigned* ib = (const unsigned*)info + m * XA_INFO;
        const unsigned word = ib[16 + (h >> 1)];
#pragma unroll
        for (int i = 0; i < C::PH; ++i) {
            const int p = wave * C::PH + i;
            const uint32_t row = (uint32_t)p * C::RPP + rip;      // row of the half tile
            const int seg = h * 16 + (int)row;
            const bool valid = seg < s_eff && ((word >> ((h & 1) * 16 + row)) & 1u);
            const uint32_t srow = (uint32_t)(valid ? seg : first);
            const uint32_t chunk = (((slot >> 2) ^ (row & 7u)) << 2) | (slot & 3u);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(Ub + (size_t)(srow * ldu_b + chunk * 16u)), (lds_ptr_t)(lds + (h & 3) * C::HSTG + p * 1024), 16, 0, 0);
        }
    };
    auto track_info = [&](int64_t m, int& s_eff, int& first) __attribute__((always_inline)) {
        const int* ip = info + m * XA_INFO;
        s_eff = __builtin_amdgcn_readfirstlane(ip[0]); first = __builtin_amdgcn_readfirstlane(ip[1]);
    };

    // ---- LDS addresses of this lane's fragment reads
    const uint32_t k_rd = lbase + (uint32_t)r * C::ROWB;                      // K row r of a stage; chunk c at ((c ^ (r & 15)) << 4)
    const uint32_t ksw = (uint32_t)(r & 15);
    const int g4 = lane >> 4, i16 = lane & 15;
    const uint32_t trow = 4 * (g4 >> 1) + (i16 >> 2);                          // row of the transposing read within a half tile (+ 8 for the upper half)
    const uint32_t u_rd = lbase + trow * C::ROWB + (g4 & 1) * 32 + (i16 & 3) * 8;   // + ((group ^ (trow & 7)) << 6), group = 64-byte column group
    const uint32_t p_wr = lbase + C::P_OFF + (uint32_t)(32 * vh + r) * XA_PP + hh * 16;   // 16-segment group G at + G * 32
    const uint32_t p_rd = lbase + C::P_OFF + (uint32_t)r * XA_PP + hh * 16;                // video tile vt at + vt * 32 * XA_PP

    __builtin_amdgcn_s_waitcnt(0x0070);                            // the Q rows have landed (a builtin: the compiler's own bookkeeping sees it)
    int s_eff, first;
    track_info(m_begin, s_eff, first);
    {
        const int nb0 = ((s_eff > 0 ? (s_eff + 31) / 32 : 1) + 1) / 2;
        issue_k(m_begin, 0, s_eff, first);
        if (nb0 > 1) issue_k(m_begin, 1, s_eff, first);
    }

    for (int jt = 0; jt < T; ++jt) {
        const int64_t m = m_begin + jt;
        const int ntiles = s_eff > 0 ? (s_eff + 31) / 32 : 1;
        const int NB = (ntiles + 1) / 2;
        const int NH = s_eff > 0 ? (s_eff + 15) / 16 : 1;
        const unsigned* ib = (const unsigned*)info + m * XA_INFO;

        // ================================================================================================ pass 1: scores
        f32x16 sacc[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[ti][e] = 0.f;
        // batch 0 has landed (batch 1, if there is one, may still be in flight)
        xa_wait_vm(NB > 1 ? C::PB : 0);
        XA_BARRIER();
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int b = 2 * ti + bb;
                if (b < NB) {
                    if (grp == bb) {
                        // this wave's tile of the batch: t = 2 b + (jw & 1), stage t & 3 = 2 bb + (jw & 1).  Fragment ks of K row r is
                        // 16-byte chunk (2 ks + hh) ^ (r & 15) of the row: its low four bits take one of eight per-lane values (one
                        // address register each, made here from an opaque base so that nothing is hoisted out of the track loop and
                        // spilled), the rest is the instruction's immediate offset; reads run four fragments ahead of their MFMAs
                        const uint32_t kx = xa_opaque(k_rd + (uint32_t)(2 * bb + (jw & 1)) * C::STG + (((uint32_t)hh ^ ksw) << 4));
                        uint32_t ka[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) ka[q] = kx ^ (uint32_t)(q << 5);
                        bf16x8 f[4];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) f[ks] = xa_read128(ka[ks]);
#pragma unroll
                        for (int ks = 0; ks < C::NQF; ++ks) {
                            if (ks + 3 < C::NQF) xa_wait_lgkm<3>(f[ks & 3]);
                            else if (ks + 2 < C::NQF) xa_wait_lgkm<2>(f[ks & 3]);
                            else if (ks + 1 < C::NQF) xa_wait_lgkm<1>(f[ks & 3]);
                            else xa_wait_lgkm<0>(f[ks & 3]);
                            sacc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks & 3], qf[ks], sacc[ti], 0, 0, 0);
                            if (ks + 4 < C::NQF) {
                                switch ((ks + 4) >> 3) {
                                    case 0: f[ks & 3] = xa_read128_off<0>(ka[(ks + 4) & 7]); break;
                                    case 1: f[ks & 3] = xa_read128_off<256>(ka[(ks + 4) & 7]); break;
                                    case 2: f[ks & 3] = xa_read128_off<512>(ka[(ks + 4) & 7]); break;
                                    default: f[ks & 3] = xa_read128_off<768>(ka[(ks + 4) & 7]); break;
                                }
                            }
                        }
                    }
                    // batch b + 1 has landed; everyone is done with batch b: its stages take batch b + 2
                    if (b + 1 < NB) xa_wait_vm(0);
                    XA_BARRIER();
                    if (b + 2 < NB) issue_k(m, b + 2, s_eff, first);
                }
            }
        }
        // every K tile is consumed: the first half tiles of U go out now and fly under the softmax
        issue_u(m, 0, s_eff, first);
        if (NH > 1) issue_u(m, 1, s_eff, first);
        if (NH > 2) issue_u(m, 2, s_eff, first);

        // ---- scores -> scaled, masked; maximum of this wave's tiles per video
        float mx = -INFINITY;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const int t = jw + 4 * ti;
            if (t < ntiles) {
                const unsigned word = ib[16 + t] >> (4 * hh);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float s = ((word >> ((e & 3) + 8 * (e >> 2))) & 1u) ? sacc[ti][e] * c : -INFINITY;
                    sacc[ti][e] = s;
                    mx = fmaxf(mx, s);
                }
            }
        }
        mx = fmaxf(mx, xa_other_half(mx));
        if (hh == 0) xa_wr<float>(lbase + C::MAX_OFF + (uint32_t)(jw * XA_PQ + 32 * vh + r) * 4, mx);
        XA_BARRIER();
        float M = xa_rd<float>(lbase + C::MAX_OFF + (uint32_t)(32 * vh + r) * 4);
#pragma unroll
        for (int q = 1; q < 4; ++q) M = fmaxf(M, xa_rd<float>(lbase + C::MAX_OFF + (uint32_t)(q * XA_PQ + 32 * vh + r) * 4));
        // (a track without a valid segment: M = -inf, exp2(-inf - -inf) = NaN, like the reference's softmax over -inf)
        float psum = 0.f;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const int t = jw + 4 * ti;
            if (t < ntiles) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const float p = __builtin_amdgcn_exp2f(sacc[ti][8 * s2 + jj] - M);
                        psum += p;
                        pf[jj] = (bf16_t)p;
                    }
                    xa_wr<bf16x8>(p_wr + (uint32_t)(2 * t + s2) * 32, pf);
                }
            }
        }
        psum += xa_other_half(psum);
        if (hh == 0) xa_wr<float>(lbase + C::SUM_OFF + (uint32_t)(jw * XA_PQ + 32 * vh + r) * 4, psum);

        // ================================================================================================ pass 2: O^T = U^T P^T
        f32x16 oacc[C::NDT][2];
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
            for (int vt = 0; vt < 2; ++vt)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[dt][vt][e] = 0.f;
        for (int h = 0; h < NH; ++h) {
            // half tile h has landed (h + 1 and h + 2 may be in flight); everyone is done with h - 1: its slot takes h + 3
            const int ahead = NH - 1 - h;
            xa_wait_vm((ahead >= 2 ? 2 : ahead) * C::PH);
            XA_BARRIER();                                           // (the first one also publishes the probabilities and their sums)
            if (h + 3 < NH) issue_u(m, h + 3, s_eff, first);
            const uint32_t ub = u_rd + (uint32_t)(h & 3) * C::HSTG;
            bf16x8 pb0 = xa_read128(p_rd + (uint32_t)h * 32), pb1 = xa_read128(p_rd + 32 * XA_PP + (uint32_t)h * 32);
            bf16x4 lo[C::NDT], hi[C::NDT];
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt) {
                const uint32_t va = ub + ((((uint32_t)(wave * C::NDT + dt)) ^ (trow & 7u)) << 6);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[dt]) : "v"(va));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[dt]) : "v"(va), "n"(8 * C::ROWB));
            }
            if constexpr (C::NDT == 2)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pb0), "+v"(pb1), "+v"(lo[0]), "+v"(lo[1]), "+v"(hi[0]), "+v"(hi[1]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pb0), "+v"(pb1), "+v"(lo[0]), "+v"(hi[0]));
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt) {
                const bf16x8 uf = __builtin_shufflevector(lo[dt], hi[dt], 0, 1, 2, 3, 4, 5, 6, 7);
                oacc[dt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uf, pb0, oacc[dt][0], 0, 0, 0);
                oacc[dt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uf, pb1, oacc[dt][1], 0, 0, 0);
            }
        }
        XA_BARRIER();                                               // U ring and probabilities are free

        // ---- the next track's first K batches go out before the tail (stores first would be simpler to count, but the tail's stores are
        // never waited for: they are older than every piece issued here, so a counted wait for a piece covers them)
        int nx_seff = 0, nx_first = 0;
        if (jt + 1 < T) {
            track_info(m + 1, nx_seff, nx_first);
            const int nbn = ((nx_seff > 0 ? (nx_seff + 31) / 32 : 1) + 1) / 2;
            issue_k(m + 1, 0, nx_seff, nx_first);
            if (nbn > 1) issue_k(m + 1, 1, nx_seff, nx_first);
        }

        // ================================================================================================ tail: O / l, row statistics, store
        float inv_l[2];
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            float l = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) l += xa_rd<float>(lbase + C::SUM_OFF + (uint32_t)(q * XA_PQ + 32 * vt + r) * 4);
            inv_l[vt] = 1.f / l;
        }
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float x = oacc[dt][vt][e] * inv_l[vt];
                    oacc[dt][vt][e] = x;
                    su += x; sq = __builtin_fmaf(x, x, sq);
                }
            su += xa_other_half(su); sq += xa_other_half(sq);
            if (hh == 0) xa_wr<f32x2_t>(lbase + C::STAT_OFF + (uint32_t)((wave * XA_PQ + 32 * vt + r) * 8), (f32x2_t){su, sq});
        }
        XA_BARRIER();
        // (32-bit row offsets from an opaque video index, made per track: hoisted out of the track loop the compiler spills them, and a
        // reload from scratch in front of a store is a vmcnt(0) -- which would wait for the K batches issued just above)
        unsigned char* ob = (unsigned char*)a.out + (size_t)m * (size_t)a.Nv * (size_t)a.ldo * 2;
        const uint32_t nlane = xa_opaque((uint32_t)n0 + (uint32_t)r);
        const uint32_t ldo_b = (uint32_t)a.ldo * 2u;
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            float mean = 0.f, rstd = 1.f;
            if (a.normalize) {
                f32x2_t st = xa_rd<f32x2_t>(lbase + C::STAT_OFF + (uint32_t)((32 * vt + r) * 8));
#pragma unroll
                for (int q = 1; q < 8; ++q) st += xa_rd<f32x2_t>(lbase + C::STAT_OFF + (uint32_t)((q * XA_PQ + 32 * vt + r) * 8));
                mean = st[0] * (1.f / D);
                const float var = fmaxf(st[1] * (1.f / D) - mean * mean, 0.f);
                rstd = __builtin_amdgcn_rsqf(var + a.eps);
            }
            const uint32_t n = nlane + 32u * vt;
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt) {
                // the lane holds rows {0-3, 8-11, 16-19, 24-27} + 4 hh of the 32-row tile: the halves swap 4-row groups so that each ends
                // with 8 consecutive rows twice (hh = 0: rows 0-7 and 16-23; hh = 1: rows 8-15 and 24-31) -> two 16-byte stores
                uint32_t pk[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float x0 = (oacc[dt][vt][2 * q] - mean) * rstd, x1 = (oacc[dt][vt][2 * q + 1] - mean) * rstd;
                    const bf16_t b0 = (bf16_t)x0, b1 = (bf16_t)x1;
                    pk[q] = (uint32_t)__builtin_bit_cast(unsigned short, b0) | ((uint32_t)__builtin_bit_cast(unsigned short, b1) << 16);
                }
                // pk[0..1] = rows 0-3 (+4 hh), pk[2..3] = rows 8-11, pk[4..5] = rows 16-19, pk[6..7] = rows 24-27
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    u32x4 v;
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(pk[4 * half + w2], pk[4 * half + 2 + w2], false, false);
                        // after the swap: lanes 0-31 hold (own first group, partner's first group), lanes 32-63 (partner's second, own second)
                        v[w2] = sw[0]; v[2 + w2] = sw[1];
                    }
                    const uint32_t drow = (uint32_t)(wave * (D / 8) + 32 * dt + 16 * half + 8 * hh);      // first of the 8 consecutive width columns
                    if (n < (uint32_t)a.Nv) *(u32x4*)(ob + (n * ldo_b + drow * 2u)) = v;
                }
            }
        }
        s_eff = nx_seff; first = nx_first;
    }
}

// per track: last valid segment + 1, first valid segment, one bit per segment (1 = attended to).  One wave per track.
__global__ __launch_bounds__(256) void xpool_attn_info_kernel(const float* key_mask, int64_t S, int64_t Nm, int* info) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= Nm) return;
    int last = -1, first = 0x7fffffff;
    int* ip = info + m * XA_INFO;
    for (int j0 = 0; j0 < XA_SMAX; j0 += 64) {
        const int j = j0 + lane;
        const bool v = j < (int)S && (key_mask == nullptr || key_mask[m * S + j] != 0.f);
        const unsigned long long bal = __ballot(v);
        if (lane == 0) { ip[16 + j0 / 32] = (int)(unsigned)bal; ip[16 + j0 / 32 + 1] = (int)(unsigned)(bal >> 32); }
        if (v) { last = j; first = min(first, j); }
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) { last = max(last, __shfl_xor(last, o2)); first = min(first, __shfl_xor(first, o2)); }
    if (lane == 0) { ip[0] = last + 1; ip[1] = last < 0 ? 0 : first; }
}

template <int D>
int launch_xpool_attn(const MadeXpoolAttnArgs& a, const int* info, dim3 grid, int per, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)xpool_attn_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, XaCfg<D>::LDS_BYTES);
        if (e != hipSuccess) {
            made_set_error("made_xpool_attention: cannot reserve %d bytes of LDS: %s", XaCfg<D>::LDS_BYTES, hipGetErrorString(e));
            return MADE_ERR_HIP;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(xpool_attn_kernel<D>, grid, dim3(XA_T), XaCfg<D>::LDS_BYTES, st, a, info, per);
    return MADE_OK;
}


// ================================================================================================================================
// made_xpool_sims (round 4): the two passes above with value rows of width 2 D -- u_s | u''_s, u'' = W'' u -- and the whole rest of the pair
// chain in the tail, so that the per-pair Linear (2 D^2 flops, 70 % of made_xpool_fused's work at S = 96) is a second P.V product (2 S D)
// and nothing per pair leaves the chip but its similarity.  D = 256, tracks of at most 96 segments (the retrieval set): K tiles are XaCfg<256>'s,
// value half tiles XaCfg<512>'s (one row = one 1 KB LDS-DMA piece).  Longer tracks stay on made_xpool_fused: a 64-video / eight-wave form of these
// passes for up to 512 segments existed (commit 4819965) and was not faster there.

// workspace (floats): gv = g3 * vn [Nv][D]; per video (sum gv, sum b3 vn, sum gv Bv, sum gv Av) [Nv][4]; sixteen model constants (XsConst); gv as bf16 [Nv][D] (round 5); 32 ints per track
__host__ __device__ inline int64_t xs_ws_pp(int64_t Nv, int64_t D) { return Nv * D; }
__host__ __device__ inline int64_t xs_ws_c(int64_t Nv, int64_t D) { return Nv * (D + 4); }
__host__ __device__ inline int64_t xs_ws_gvb(int64_t Nv, int64_t D) { return Nv * (D + 4) + 16; }       // gv once more as bf16, per video in the order the 64-video kernels' lanes read it: [32-row tile][hh][g][4]
__host__ __device__ inline int64_t xs_ws_info(int64_t Nv, int64_t D) { return Nv * (D + 4) + 16 + Nv * D / 2; }
// model constants: with g2 = g3^2, gb = g3 b3 and the folded Linear's vectors Av, Bv (y = k1 z + k2 Bv + Av)
enum XsConst { XC_G2 = 0, XC_GB, XC_B2, XC_BV, XC_AV, XC_BV2, XC_BVAV, XC_AV2, XC_G2BV, XC_G2AV, XC_G2BV2, XC_G2BVAV, XC_G2AV2, XC_GBBV, XC_GBAV };

// One 1 KB LDS-DMA piece issued from inline assembly (lane l brings 16 bytes from base + voff to LDS byte lds_dst + 16 l).  The builtin makes every
// later plain LDS access of the kernel wait for vmcnt(0) -- the compiler cannot know that the track table, the softmax exchange or the tail's sums
// never overlap a stage that is being filled -- which puts full DMA round trips into every track.  Here the compiler does not see the transfer at all; the kernel waits by hand (counted s_waitcnt vmcnt).
// (M0 is written behind the compiler's back -- it refuses M0 in a clobber list: the kernel below has no other user of M0: no builtin LDS-DMA, no s_movrel, no GWS / sendmsg)
__device__ __forceinline__ void xs_dma16(const unsigned char* base, uint32_t voff, uint32_t lds_dst) {
    const uint32_t d = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(d), "v"(voff), "s"(base) : "memory");
}
// DBG: workgroup (0, 0) writes s_memtime stamps of its phases into the sims buffer ([wave][track < 32][point < 16] int64) instead of similarities
#define XS_STAMP(pt) do { if (DBG && stamp_on && jt < 32 && lane == 0) ((long long*)a.sims)[(wave * 32 + jt) * 16 + (pt)] = (long long)__builtin_readcyclecounter(); } while (0)
// ---- made_xpool_sims for tracks of at most 96 segments (the retrieval set's S): 32 videos and FOUR waves per workgroup, TWO workgroups per CU.
// A first short-track kernel (64 videos, eight waves, one workgroup per CU, K stages and value ring with LDS of their own; commit 179a265, stamps in
// profiles/r04_ab_*) ran at 72 ms on the 53 k x 4 k set against this one's 63 and was removed.  Its stamps say what binds a track: the vector instructions of the softmax and of the tail (5 600 cycles of a SIMD's
// vector issue per 64 videos and track, both waves of the SIMD in lockstep), then the matrix pipe (2 300) and the CU's vector-memory intake (2 300), one
// after the other between nine workgroup barriers.  Two INDEPENDENT workgroups per CU drift apart and overlap these phases -- one is in its sums while
// the other multiplies or waits for a barrier -- at the price of streaming K and the value rows once per 32 videos instead of once per 64 (80 KB of LDS and
// 256 registers per wave are what a half-CU workgroup may hold, so the video tile halves: 4 output tiles of 32 x 32 per wave).
//   LDS: K tiles (3 x 16 KB) and the ring of three value half tiles share 48 KB; probabilities 6.5 KB; the chunk's track table behind the exchange arrays.
//   wave w: scores of K tile w (all 32 videos); rows [32 w, +32) and [128 + 32 w, +32) of o and the same rows of z in the second product and the tail.
template <int D> struct Xs32 {
    using CK = XaCfg<D>;
    using CU = XaCfg<2 * D>;
    static constexpr int NW = 4, PQ = 32, T = 256;
    static constexpr int PP = 96 * 2 + 16;
    static constexpr int P_OFF = 3 * CK::STG;
    static constexpr int MAX_OFF = P_OFF + PQ * PP;
    static constexpr int SUM_OFF = MAX_OFF + NW * PQ * 4;
    static constexpr int PART_OFF = SUM_OFF + NW * PQ * 4;       // [4][32][12] f32: a wave's partial sums of a video (see the tail)
    static constexpr int VEC_OFF = PART_OFF + NW * PQ * 48;      // [4][D] f32
    static constexpr int TBL_OFF = VEC_OFF + 4 * D * 4;
    static constexpr int MAX_TRACKS = (80 * 1024 - TBL_OFF) / 32;
    static_assert(CK::STG == CU::HSTG && CK::PT == 16 && D == 256, "tile split");
    static_assert(MAX_TRACKS >= 256, "made_xpool_sims: the LDS map does not fit half a CU");
};

template <int D, bool DBG>
__global__ __launch_bounds__(256, 2) void xpool_sims32_kernel(const MadeXpoolSimsArgs a, const int* __restrict__ info, int tracks_per_chunk, int nvt, int nchunks) {
    using X = Xs32<D>;
    using CK = typename X::CK;
    using CU = typename X::CU;
    constexpr int PP = X::PP, PQ = X::PQ;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    // 1-D grid, XCD-aware: workgroups b, b + 8, b + 16, ... share an XCD and its L2.  A chunk of tracks belongs to ONE XCD -- chunk c to XCD c % 8 --
    // which takes it through all its video tiles: the chunk's K / value rows are fetched into one L2 instead of eight (with fewer than eight chunks, or
    // a count that is not a multiple of eight, the plain order: video tile fastest)
    int vtile, chunk;
    if ((nchunks & 7) == 0) {
        const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
        chunk = (jx / nvt) * 8 + xcd; vtile = jx % nvt;
    } else { chunk = blockIdx.x / nvt; vtile = blockIdx.x % nvt; }
    const int64_t n0 = (int64_t)vtile * PQ;
    const int64_t m_begin = (int64_t)chunk * tracks_per_chunk;
    const int64_t m_end = (m_begin + tracks_per_chunk < a.Nm) ? m_begin + tracks_per_chunk : a.Nm;
    if (m_begin >= m_end) return;
    const int T = (int)(m_end - m_begin);
    const uint32_t lbase = (uint32_t)(uintptr_t)(lds3_t)lds;
    const float* wsf = (const float*)a.ws;
    const bool stamp_on = blockIdx.x == 0;
    const int64_t nv = n0 + r < a.Nv ? n0 + r : a.Nv - 1;            // this lane's video (both lane halves)

    bf16x8 qf[CK::NQF];
    {
        const bf16_t* qp = (const bf16_t*)a.Q + nv * a.ldq + hh * 8;
#pragma unroll
        for (int ks = 0; ks < CK::NQF; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }
    // g3 * vn of this lane's video at the rows its two z tiles hold: 32 (w + 4 j) + 8 g + 4 hh + (0..3)
    bf16x4 gv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float* gp = wsf + nv * D + 32 * (wave + 4 * j) + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 x = *(const f32x4*)(gp + 8 * g);
            gv[j][g] = (bf16x4){(bf16_t)x[0], (bf16_t)x[1], (bf16_t)x[2], (bf16_t)x[3]};
        }
    }
    const f32x4 pv4 = *(const f32x4*)(wsf + xs_ws_pp(a.Nv, D) + nv * 4);     // sum gv, sum b3 vn, sum gv Bv, sum gv Av of this lane's video
    // the model's constants (XsConst) in scalar registers: read through the pointer inside the track loop they were re-loaded from global memory for every
    // track (the LDS-DMA statements clobber memory), a round trip -- and a vmcnt(0) that also covered the next track's K pieces -- in front of every store
    float mc[16];
#pragma unroll
    for (int q = 0; q < 15; ++q) mc[q] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wsf[xs_ws_c(a.Nv, D) + q])));
    {
        float* vec = (float*)(lds + X::VEC_OFF);
        const float g3 = a.ln3_g[tid], b3 = a.ln3_b[tid];             // (256 threads = D)
        vec[tid] = a.av[tid]; vec[D + tid] = a.bv[tid]; vec[2 * D + tid] = g3 * g3; vec[3 * D + tid] = g3 * b3;
    }
    for (int i = tid; i < T; i += X::T) {
        const int* ip = info + (m_begin + i) * XA_INFO;
        *(u32x4*)(lds + X::TBL_OFF + i * 32) = (u32x4){(unsigned)ip[0], (unsigned)ip[1], 0u, 0u};
        *(u32x4*)(lds + X::TBL_OFF + i * 32 + 16) = *(const u32x4*)(ip + 16);
    }
    const float c = a.scale * 1.4426950408889634f;

    struct Trk { int s_eff, first; unsigned w0, w1, w2; };       // (three scalars, not an array: the compiler turned the array behind a select chain into a scratch lookup -- a VMEM load whose wait also covered the LDS-DMA pieces just issued)
    auto load_track = [&](int jt_, Trk& t) __attribute__((always_inline)) {
        const u32x4 h = xa_rd<u32x4>(lbase + X::TBL_OFF + (uint32_t)jt_ * 32), wv = xa_rd<u32x4>(lbase + X::TBL_OFF + (uint32_t)jt_ * 32 + 16);
        t.s_eff = __builtin_amdgcn_readfirstlane((int)h[0]); t.first = __builtin_amdgcn_readfirstlane((int)h[1]);
        t.w0 = __builtin_amdgcn_readfirstlane(wv[0]); t.w1 = __builtin_amdgcn_readfirstlane(wv[1]); t.w2 = __builtin_amdgcn_readfirstlane(wv[2]);
    };
    auto tile_word = [&](const Trk& t, int tile) __attribute__((always_inline)) -> unsigned {       // (masks, not selects: see Trk)
        return (t.w0 & (tile == 0 ? ~0u : 0u)) | (t.w1 & (tile == 1 ? ~0u : 0u)) | (t.w2 & (tile >= 2 ? ~0u : 0u));
    };

    const uint32_t ldk_b = (uint32_t)a.ldk * 2u, ldu_b = (uint32_t)a.ldu * 2u;
    // every K tile of the track: wave w brings pieces 4 w .. 4 w + 3 (rows 8 w .. 8 w + 7) of each
    auto issue_k = [&](int64_t m, const Trk& tk) __attribute__((always_inline)) {
        const uint32_t rip = xa_opaque((uint32_t)lane / CK::CPR), slot = xa_opaque((uint32_t)lane % CK::CPR);
        const unsigned char* Kb = (const unsigned char*)a.K + m * a.k_bs * 2;
        const int ntl = tk.s_eff > 0 ? (tk.s_eff + 31) / 32 : 1;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (t < ntl) {
                const unsigned word = tile_word(tk, t);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pit = wave * 4 + i;
                    const uint32_t row = (uint32_t)pit * CK::RPP + rip;
                    const int seg = t * 32 + (int)row;
                    const bool valid = seg < tk.s_eff && ((word >> row) & 1u);
                    const uint32_t srow = (uint32_t)(valid ? seg : tk.first);
                    const uint32_t chunk = slot ^ (row & 15u);
                    xs_dma16(Kb, __umul24(srow, ldk_b) + chunk * 16u, lbase + (uint32_t)(t * CK::STG + pit * 1024));
                }
            }
        }
    };
    // value half tile h (16 segments x 1 KB) -> ring slot h % 3: wave w brings rows 4 w .. 4 w + 3
    auto issue_u = [&](int64_t m, int h, const Trk& tk) __attribute__((always_inline)) {
        const uint32_t slot = xa_opaque((uint32_t)lane);
        const unsigned char* Ub = (const unsigned char*)a.UU + m * a.u_bs * 2;
        const unsigned bits = (tile_word(tk, h >> 1) >> ((h & 1) * 16)) & 0xFFFFu;
        const uint32_t dst = (uint32_t)((h % 3) * CU::HSTG);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t row = (uint32_t)(wave * 4 + i);
            const int seg = h * 16 + (int)row;
            const bool valid = seg < tk.s_eff && ((bits >> row) & 1u);
            const uint32_t srow = (uint32_t)(valid ? seg : tk.first);
            const uint32_t chunk = (((slot >> 2) ^ (row & 7u)) << 2) | (slot & 3u);
            xs_dma16(Ub, __umul24(srow, ldu_b) + chunk * 16u, lbase + dst + row * 1024u);
        }
    };

    const uint32_t k_rd = lbase + (uint32_t)r * CK::ROWB;
    const uint32_t ksw = (uint32_t)(r & 15);
    const int g4 = lane >> 4, i16 = lane & 15;
    const uint32_t trow = 4 * (g4 >> 1) + (i16 >> 2);
    const uint32_t u_rd = lbase + trow * CU::ROWB + (g4 & 1) * 32 + (i16 & 3) * 8;
    uint32_t ug[4];                                                // column groups of this wave's tiles: o rows 32 w, 128 + 32 w; z the same + 256
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ug[dt] = (((uint32_t)(wave + 4 * dt)) ^ (trow & 7u)) << 6;
    const uint32_t p_wr = lbase + X::P_OFF + (uint32_t)r * PP + hh * 16;
    const uint32_t p_rd = p_wr;

    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();                                               // the track table and the constants are in LDS
    // ---- the sums of the tail that are LINEAR in z -- sum z c_d for c = 1, Bv, Av, g2, g2 Bv, g2 Av, gb -- go through the matrix pipe: a z tile in
    // accumulator order is the B operand of the next MFMA (k-slot (hh, jj) of k-step s2 = accumulator row 8 (2 s2 + (jj >> 2)) + 4 hh + (jj & 3)), the
    // constants of this wave's rows in the same order are the A operand: row m of the product is functional m (rows 7 .. 31: zero)
    bf16x8 afr[2][2];
    {
        const int mf = lane & 31;
        const float* vec = (const float*)(lds + X::VEC_OFF);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int d = 32 * (wave + 4 * j) + 8 * (2 * s2 + (jj >> 2)) + 4 * hh + (jj & 3);
                    const float Av = vec[d], Bv = vec[D + d], g2 = vec[2 * D + d], gb = vec[3 * D + d];
                    const float val = mf == 0 ? 1.f : mf == 1 ? Bv : mf == 2 ? Av : mf == 3 ? g2 : mf == 4 ? g2 * Bv : mf == 5 ? g2 * Av : mf == 6 ? gb : 0.f;
                    afr[j][s2][jj] = (bf16_t)val;
                }
    }
    Trk tk;
    load_track(0, tk);
    issue_k(m_begin, tk);

    for (int jt = 0; jt < T; ++jt) {
        const int64_t m = m_begin + jt;
        const int s_eff = tk.s_eff;
        const int ntiles = s_eff > 0 ? (s_eff + 31) / 32 : 1;
        const int NH = s_eff > 0 ? (s_eff + 15) / 16 : 1;

        // ================================================================================================ pass 1: scores of K tile `wave`
        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
        XS_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's K pieces
        XS_STAMP(1);
        XA_BARRIER();
        XS_STAMP(2);
        if (wave < ntiles) {
            const uint32_t kx = xa_opaque(k_rd + (uint32_t)wave * CK::STG + (((uint32_t)hh ^ ksw) << 4));
            uint32_t ka[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ka[q] = kx ^ (uint32_t)(q << 5);
            bf16x8 f[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) f[ks] = xa_read128(ka[ks]);
#pragma unroll
            for (int ks = 0; ks < CK::NQF; ++ks) {
                if (ks + 3 < CK::NQF) xa_wait_lgkm<3>(f[ks & 3]);
                else if (ks + 2 < CK::NQF) xa_wait_lgkm<2>(f[ks & 3]);
                else if (ks + 1 < CK::NQF) xa_wait_lgkm<1>(f[ks & 3]);
                else xa_wait_lgkm<0>(f[ks & 3]);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks & 3], qf[ks], sacc, 0, 0, 0);
                if (ks + 4 < CK::NQF) {
                    if (((ks + 4) >> 3) == 0) f[ks & 3] = xa_read128_off<0>(ka[(ks + 4) & 7]);
                    else f[ks & 3] = xa_read128_off<256>(ka[(ks + 4) & 7]);
                }
            }
        }
        XS_STAMP(3);
        XA_BARRIER();                                               // the K tiles are consumed: their LDS is the value ring now
        XS_STAMP(4);
        issue_u(m, 0, tk);
        if (NH > 1) issue_u(m, 1, tk);
        XS_STAMP(5);

        float mx = -INFINITY;
        if (wave < ntiles) {
            const unsigned word = tile_word(tk, wave) >> (4 * hh);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float sc = ((word >> ((e & 3) + 8 * (e >> 2))) & 1u) ? sacc[e] * c : -INFINITY;
                sacc[e] = sc;
                mx = fmaxf(mx, sc);
            }
        }
        mx = fmaxf(mx, xa_other_half(mx));
        if (hh == 0) xa_wr<float>(lbase + X::MAX_OFF + (uint32_t)(wave * PQ + r) * 4, mx);
        XS_STAMP(6);
        XA_BARRIER();
        XS_STAMP(7);
        float M = xa_rd<float>(lbase + X::MAX_OFF + (uint32_t)r * 4);
#pragma unroll
        for (int q = 1; q < 4; ++q) M = fmaxf(M, xa_rd<float>(lbase + X::MAX_OFF + (uint32_t)(q * PQ + r) * 4));
        float psum = 0.f;
        if (wave < ntiles) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const float pr_ = __builtin_amdgcn_exp2f(sacc[8 * s2 + jj] - M);
                    psum += pr_;
                    pf[jj] = (bf16_t)pr_;
                }
                xa_wr<bf16x8>(p_wr + (uint32_t)(2 * wave + s2) * 32, pf);
            }
        }
        psum += xa_other_half(psum);
        if (hh == 0) xa_wr<float>(lbase + X::SUM_OFF + (uint32_t)(wave * PQ + r) * 4, psum);

        // ================================================================================================ pass 2: [O | Z]^T = [U | U'']^T P^T, 16 segments per step
        XS_STAMP(8);
        f32x16 oacc[4];                                             // [0], [1]: rows 32 w .., 128 + 32 w .. of o; [2], [3]: the same rows of z
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
        for (int h = 0; h < NH; ++h) {
            if (h + 1 < NH) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // half tile h has landed (h + 1 may be in flight)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            XA_BARRIER();                                           // (the first one also publishes the probabilities and their sums)
            if (h + 2 < NH) issue_u(m, h + 2, tk);                  // into the slot of half tile h - 1, which everyone has left
            const uint32_t ub = u_rd + (uint32_t)((h % 3) * CU::HSTG);
            bf16x8 pbf = xa_read128(p_rd + (uint32_t)h * 32);
            bf16x4 lo[4], hi[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const uint32_t va = ub + ug[dt];
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[dt]) : "v"(va));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[dt]) : "v"(va), "n"(8 * CU::ROWB));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pbf), "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(lo[dt], hi[dt], 0, 1, 2, 3, 4, 5, 6, 7), pbf, oacc[dt], 0, 0, 0);
        }
        XS_STAMP(9);
        XA_BARRIER();                                               // value ring and probabilities are free
        XS_STAMP(10);
        Trk nx; nx.s_eff = 0; nx.first = 0; nx.w0 = nx.w1 = nx.w2 = 0u;
        if (jt + 1 < T) { load_track(jt + 1, nx); issue_k(m + 1, nx); }
        XS_STAMP(11);

        // ================================================================================================ tail: this wave's partial sums of video r
        // Over its 64 rows of o: sum o, sum o^2 (LayerNorm2's statistics; on the unnormalised accumulators, 1 / l applied to the sums).  Over its 64 rows
        // of z = P~.U'' (unnormalised too: 1 / l rides on k1): with y = k1 z + k2 Bv + Av the six sums of LayerNorm3 + cosine expand into
        //   vector work: Q1 = sum z^2, Q2 = sum g2 z^2, B1 = sum gv z                                 (4 packed instructions per pair of elements, not 9)
        //   matrix work: F1..F7 = sum z, z Bv, z Av, g2 z, g2 z Bv, g2 z Av, gb z                       (4 MFMAs per track)
        // none of which needs k1 / k2: ONE exchange and ONE barrier, then a single wave puts the pair together.
        float inv_l;
        {
            float l = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) l += xa_rd<float>(lbase + X::SUM_OFF + (uint32_t)(q * PQ + r) * 4);
            inv_l = 1.f / l;
            f32x2_t su2 = {0.f, 0.f}, sq2 = {0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2_t x2 = {oacc[dt][e], oacc[dt][e + 1]};
                    su2 += x2; sq2 += x2 * x2;
                }
            float su = (su2[0] + su2[1]) * inv_l, sq = (sq2[0] + sq2[1]) * (inv_l * inv_l);
            su += xa_other_half(su); sq += xa_other_half(sq);
            f32x2_t q1 = {0.f, 0.f}, q2 = {0.f, 0.f}, b1 = {0.f, 0.f};
            f32x16 facc;
#pragma unroll
            for (int e = 0; e < 16; ++e) facc[e] = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t g2_r = xa_opaque(lbase + X::VEC_OFF + (uint32_t)(2 * D + 32 * (wave + 4 * j) + 4 * hh) * 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 g2 = xa_rd<f32x4>(g2_r + g * 32);
#pragma unroll
                    for (int jj = 0; jj < 4; jj += 2) {
                        const f32x2_t z2 = {oacc[2 + j][4 * g + jj], oacc[2 + j][4 * g + jj + 1]};
                        const f32x2_t zz = z2 * z2;
                        q1 += zz;
                        q2 += zz * (f32x2_t){g2[jj], g2[jj + 1]};
                        b1 += z2 * (f32x2_t){(float)gv[j][g][jj], (float)gv[j][g][jj + 1]};
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 zb;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) zb[jj] = (bf16_t)oacc[2 + j][8 * s2 + jj];
                    facc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[j][s2], zb, facc, 0, 0, 0);
                }
            }
            float Q1 = q1[0] + q1[1], Q2 = q2[0] + q2[1], B1 = b1[0] + b1[1];
            Q1 += xa_other_half(Q1); Q2 += xa_other_half(Q2); B1 += xa_other_half(B1);
            // [0..3] sum o / l, sum (o / l)^2, Q1, Q2; [4..7] F1..F4 (rows 0-3 of the product: lower lane half); [8..11] F5, F6, F7 (upper half), B1
            const uint32_t pw = lbase + X::PART_OFF + (uint32_t)((wave * PQ + r) * 48);
            if (hh == 0) { xa_wr<f32x4>(pw, (f32x4){su, sq, Q1, Q2}); xa_wr<f32x4>(pw + 16, (f32x4){facc[0], facc[1], facc[2], facc[3]}); }
            else xa_wr<f32x4>(pw + 32, (f32x4){facc[0], facc[1], facc[2], B1});
        }
        XS_STAMP(12);
        XA_BARRIER();
        XS_STAMP(13);
        // ---- one wave (they take turns), lane = video: LayerNorm2's k1 / k2, the six sums, LayerNorm3 + cosine
        if (!DBG && wave == (jt & 3) && hh == 0) {
            const uint32_t pr = xa_opaque(lbase + X::PART_OFF + (uint32_t)r * 48);
            f32x4 A0 = xa_rd<f32x4>(pr), A1 = xa_rd<f32x4>(pr + 16), A2 = xa_rd<f32x4>(pr + 32);
#pragma unroll
            for (int q = 1; q < 4; ++q) { A0 += xa_rd<f32x4>(pr + q * (PQ * 48)); A1 += xa_rd<f32x4>(pr + q * (PQ * 48) + 16); A2 += xa_rd<f32x4>(pr + q * (PQ * 48) + 32); }
            const float mean = A0[0] * (1.f / D);
            const float var = fmaxf(A0[1] * (1.f / D) - mean * mean, 0.f);
            const float k1n = __builtin_amdgcn_rsqf(var + a.eps), k2 = -mean * k1n, k1 = k1n * inv_l;
            const float Q1 = A0[2], Q2 = A0[3], F1 = A1[0], F2 = A1[1], F3 = A1[2], F4 = A1[3], F5 = A2[0], F6 = A2[1], F7 = A2[2], B1 = A2[3];
            const float s1 = k1 * F1 + k2 * mc[XC_BV] + mc[XC_AV];
            const float s2 = k1 * k1 * Q1 + 2.f * k1 * (k2 * F2 + F3) + k2 * k2 * mc[XC_BV2] + 2.f * k2 * mc[XC_BVAV] + mc[XC_AV2];
            const float p1 = k1 * B1 + k2 * pv4[2] + pv4[3];
            const float c1 = k1 * F4 + k2 * mc[XC_G2BV] + mc[XC_G2AV];
            const float c2 = k1 * k1 * Q2 + 2.f * k1 * (k2 * F5 + F6) + k2 * k2 * mc[XC_G2BV2] + 2.f * k2 * mc[XC_G2BVAV] + mc[XC_G2AV2];
            const float e1 = k1 * F7 + k2 * mc[XC_GBBV] + mc[XC_GBAV];
            const float mu = s1 * (1.f / D);
            const float vy = fmaxf(s2 * (1.f / D) - mu * mu, 0.f);
            const float rs = __builtin_amdgcn_rsqf(vy + a.eps);
            const float dot = rs * (p1 - mu * pv4[0]) + pv4[1];
            const float zz = rs * rs * (c2 - 2.f * mu * c1 + mu * mu * mc[XC_G2]) + 2.f * rs * (e1 - mu * mc[XC_GB]) + mc[XC_B2];
            if (n0 + r < a.Nv) a.sims[(n0 + r) * a.ld_sims + m] = dot * __builtin_amdgcn_rsqf(zz);
        }
        XS_STAMP(14);
        XS_STAMP(15);
        tk = nx;
    }
}

// ---- made_xpool_sims, 64 videos per workgroup (round 5; opt-in: MADE_XPOOL_SIMS_PQ=64).  Built to test what binds the 32-video kernel above.
// Round 4 read its stamps as "the CU's LDS-DMA intake": a (32 videos, track) unit takes in 144 KB (K 48 KB, u | u'' 96 KB), two workgroups per CU,
// 6.63 M units = 955 GB through 256 CUs x ~70 GB/s (what LDS-DMA reads L2-resident rows at) = 53 ms of the 54 it takes.  Twice the videos per
// loaded tile halves those bytes per pair; what stood in the way was the register file: Q of the tile's videos as B operands of 32 x 32 score
// tiles is 64 registers per 32 videos IN EVERY WAVE (wave w = K tile w).  Here the score product is turned: wave w owns 16 VIDEOS (16 w ..) and
// all of the track's segments -- 16 x 16 x 32 MFMAs, A = K rows of a 16-segment tile out of LDS, B = its 16 videos' Q rows (32 registers) -- so
//   * the softmax of a video is wave-local (two lane-group shuffles for the maximum and the sum; no exchange arrays, one barrier less),
//   * all four waves share pass 1 evenly (the 32-video kernel idles wave 3 on tracks of at most 96 segments),
//   * pass 2 and the tail are the 32-video kernel's with a second video tile: 8 accumulator tiles per wave, 8 MFMAs per 16-segment step on the
//     same ten LDS reads, the same barriers per track for twice the pairs.
// 128 accumulator registers leave no room for Q and g3 vn across a track: Q is re-read for the next track under the tail, g3 vn (a bf16 copy in
// the workspace, in lane order) at the head of the tail.  LDS 68 KB: K tiles / value ring 48 KB, probabilities 13 KB (the tail's exchange on top
// of them), softmax denominators, constants.
// MEASURED (profiles/r05_xpool_sims64_*): bit-compatible with the 32-video kernel (9e-8), 53.1 against 53.8 ms on 53 k x 4 k, same box,
// alternating.  Half the LDS-DMA bytes per pair, a quarter fewer vector instructions per pair (the literal-zero accumulators below), ten barriers
// per 64 videos instead of eleven per 32 -- and the same time: neither the bytes nor the vector issue bind this loop.  Its stamps say what does:
// of 18 900 cycles per (64 videos, track) a wave spends 8 000 parked -- on the round trips of a chain of dependent transfers (K, then the value
// half tiles two at a time through a three-slot ring, then g3 vn: 560 + 2 500 + 1 640), on vector-memory ISSUE (100-135 cycles per 1 KB
// piece or load, 28-44 of them per track: a wave with more than ~20 in flight blocks at issue, so earlier prefetches only move the wait), and on
// the barriers between the phases (950).  What would shorten it is a deeper value ring (a fourth 16 KB slot does not fit 80 KB beside the
// probabilities) or a third workgroup per CU (170 registers per wave: the accumulators alone are 128).  The default stays the 32-video kernel.
template <int D> struct Xs64 {
    using CK = XaCfg<D>;
    using CU = XaCfg<2 * D>;
    static constexpr int NW = 4, PQ = 64, T = 256;
    static constexpr int PP = 96 * 2 + 16;
    static constexpr int P_OFF = 3 * CK::STG;
    static constexpr int PART_OFF = P_OFF;                       // [4][64][12] f32: a wave's partial sums of a video (see the tail) -- over the probabilities,
                                                                 // which are dead between pass 2's last barrier and the next track's pass 1 (two barriers later)
    static constexpr int L_OFF = P_OFF + PQ * PP;                // [64] f32: softmax denominators
    static constexpr int G2_OFF = L_OFF + PQ * 4;                // [D] f32: g3^2
    static constexpr int AFR_OFF = G2_OFF + D * 4;               // [4 waves][2][2][2 lane halves][8] 16-byte fragments: the tail's constant A operands (row 7: zeros)
    static constexpr int TBL_OFF = AFR_OFF + NW * 4 * 2 * 8 * 16;
    static constexpr int MAX_TRACKS = (80 * 1024 - TBL_OFF) / 32;
    static_assert(NW * PQ * 48 <= PQ * PP, "the tail's exchange overlays the probabilities");
    static_assert(CK::STG == CU::HSTG && CK::PT == 16 && D == 256, "tile split");
};

template <int ST_OFF> __device__ __forceinline__ bf16x8 xs_read_k(uint32_t addr) {
    bf16x8 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(ST_OFF)); return v;
}

template <int D, bool DBG>
__global__ __launch_bounds__(256, 2) void xpool_sims64_kernel(const MadeXpoolSimsArgs a, const int* __restrict__ info, int tracks_per_chunk, int nvt, int nchunks) {
    using X = Xs64<D>;
    using CK = typename X::CK;
    using CU = typename X::CU;
    constexpr int PP = X::PP, PQ = X::PQ;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int v16 = lane & 15, g4 = lane >> 4;
    int vtile, chunk;                                             // (workgroup order: see the 32-video kernel)
    if ((nchunks & 7) == 0) {
        const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
        chunk = (jx / nvt) * 8 + xcd; vtile = jx % nvt;
    } else { chunk = blockIdx.x / nvt; vtile = blockIdx.x % nvt; }
    const int64_t n0 = (int64_t)vtile * PQ;
    const int64_t m_begin = (int64_t)chunk * tracks_per_chunk;
    const int64_t m_end = (m_begin + tracks_per_chunk < a.Nm) ? m_begin + tracks_per_chunk : a.Nm;
    if (m_begin >= m_end) return;
    const int T = (int)(m_end - m_begin);
    const uint32_t lbase = (uint32_t)(uintptr_t)(lds3_t)lds;
    const float* wsf = (const float*)a.ws;
    const bool stamp_on = blockIdx.x == 0;                        // (DBG: XS_STAMP as in the 32-video kernel)
    auto vid = [&](int64_t i) __attribute__((always_inline)) -> int64_t { return i < a.Nv ? i : a.Nv - 1; };

    // pass 1's B operand: Q rows of this wave's 16 videos, k-slots 8 g4 .. 8 g4 + 7 of every 32-wide k-step
    // (re-read from L2 at the head of every track: 32 registers that pass 2 and the tail need more -- 8 KB per wave and track beside the 144 KB
    // of LDS-DMA pieces)
    // (byte offsets in 32 bits beside the uniform base pointers -- the launcher checks the ranges: a 64-bit per-lane pointer is two registers)
    const uint32_t q_off = (uint32_t)(vid(n0 + 16 * wave + v16) * a.ldq * 2 + g4 * 16);
    // g3 * vn of the lane's two videos (r and 32 + r) at the rows its z tiles hold -- 32 (w + 4 j) + 8 g + 4 hh + (0..3) -- is read from the
    // workspace's bf16 copy at the head of every tail (64 bytes per video and lane): 32 registers that pass 2 needs for its accumulators
    // (its byte offsets, like the combine's per-video sums', are recomputed per track from the lane number: three registers)
    float mc[16];
#pragma unroll
    for (int q = 0; q < 15; ++q) mc[q] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wsf[xs_ws_c(a.Nv, D) + q])));
    {
        const float g3 = a.ln3_g[tid];                                // (256 threads = D)
        ((float*)(lds + X::G2_OFF))[tid] = g3 * g3;
    }
    // the tail's constant A operands (see the 32-video kernel: functional m of the product = row m; rows 7 .. 31 are zero): built once and parked
    // in LDS -- 16 registers this kernel does not have.  Lane (m < 8, hh) of wave w writes the fragments of its rows; row 7 stays zero for m >= 7.
    if ((lane & 31) < 8) {
        const int mf = lane & 31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 fr;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int d = 32 * (wave + 4 * j) + 8 * (2 * s2 + (jj >> 2)) + 4 * hh + (jj & 3);
                    const float g3 = a.ln3_g[d], Av = a.av[d], Bv = a.bv[d], g2 = g3 * g3, gb = g3 * a.ln3_b[d];
                    const float val = mf == 0 ? 1.f : mf == 1 ? Bv : mf == 2 ? Av : mf == 3 ? g2 : mf == 4 ? g2 * Bv : mf == 5 ? g2 * Av : mf == 6 ? gb : 0.f;
                    fr[jj] = (bf16_t)val;
                }
                *(bf16x8*)(lds + X::AFR_OFF + ((((wave * 2 + j) * 2 + s2) * 2 + hh) * 8 + mf) * 16) = fr;
            }
    }
    for (int i = tid; i < T; i += X::T) {
        const int* ip = info + (m_begin + i) * XA_INFO;
        *(u32x4*)(lds + X::TBL_OFF + i * 32) = (u32x4){(unsigned)ip[0], (unsigned)ip[1], 0u, 0u};
        *(u32x4*)(lds + X::TBL_OFF + i * 32 + 16) = *(const u32x4*)(ip + 16);
    }
    const float c = a.scale * 1.4426950408889634f;

    struct Trk { int s_eff, first; unsigned w0, w1, w2; };
    auto load_track = [&](int jt_, Trk& t) __attribute__((always_inline)) {
        const u32x4 h = xa_rd<u32x4>(lbase + X::TBL_OFF + (uint32_t)jt_ * 32), wv = xa_rd<u32x4>(lbase + X::TBL_OFF + (uint32_t)jt_ * 32 + 16);
        t.s_eff = __builtin_amdgcn_readfirstlane((int)h[0]); t.first = __builtin_amdgcn_readfirstlane((int)h[1]);
        t.w0 = __builtin_amdgcn_readfirstlane(wv[0]); t.w1 = __builtin_amdgcn_readfirstlane(wv[1]); t.w2 = __builtin_amdgcn_readfirstlane(wv[2]);
    };
    auto tile_word = [&](const Trk& t, int tile) __attribute__((always_inline)) -> unsigned {
        return (t.w0 & (tile == 0 ? ~0u : 0u)) | (t.w1 & (tile == 1 ? ~0u : 0u)) | (t.w2 & (tile >= 2 ? ~0u : 0u));
    };
    const uint32_t ldk_b = (uint32_t)a.ldk * 2u, ldu_b = (uint32_t)a.ldu * 2u;
    // (the lane's part of a piece's address -- row in the piece, slot, swizzle -- is recomputed for every piece: kept in eight registers it was
    // the first thing the allocator spilled, and the scratch reloads then sat between the pieces)
    auto issue_k = [&](int64_t m, const Trk& tk) __attribute__((always_inline)) {
        const uint32_t rip = xa_opaque((uint32_t)lane / CK::CPR), slot = xa_opaque((uint32_t)lane % CK::CPR);
        const unsigned char* Kb = (const unsigned char*)a.K + m * a.k_bs * 2;
        const int ntl = tk.s_eff > 0 ? (tk.s_eff + 31) / 32 : 1;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (t < ntl) {
                const unsigned word = tile_word(tk, t);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pit = wave * 4 + i;
                    const uint32_t row = (uint32_t)pit * CK::RPP + rip;
                    const int seg = t * 32 + (int)row;
                    const bool valid = seg < tk.s_eff && ((word >> row) & 1u);
                    const uint32_t srow = (uint32_t)(valid ? seg : tk.first);
                    const uint32_t chunk = slot ^ (row & 15u);
                    xs_dma16(Kb, __umul24(srow, ldk_b) + chunk * 16u, lbase + (uint32_t)(t * CK::STG + pit * 1024));
                }
            }
        }
    };
    auto issue_u = [&](int64_t m, int h, const Trk& tk) __attribute__((always_inline)) {
        const uint32_t slot = xa_opaque((uint32_t)lane);
        const unsigned char* Ub = (const unsigned char*)a.UU + m * a.u_bs * 2;
        const unsigned bits = (tile_word(tk, h >> 1) >> ((h & 1) * 16)) & 0xFFFFu;
        const uint32_t dst = (uint32_t)((h % 3) * CU::HSTG);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t row = (uint32_t)(wave * 4 + i);
            const int seg = h * 16 + (int)row;
            const bool valid = seg < tk.s_eff && ((bits >> row) & 1u);
            const uint32_t srow = (uint32_t)(valid ? seg : tk.first);
            const uint32_t chunk = (((slot >> 2) ^ (row & 7u)) << 2) | (slot & 3u);
            xs_dma16(Ub, __umul24(srow, ldu_b) + chunk * 16u, lbase + dst + row * 1024u);
        }
    };

    // pass 1 reads: K row v16 of a 16-segment tile, 16-byte chunk 4 ks + g4, stored at slot chunk ^ (row & 15) (issue_k): the lane's address for
    // ks = 0 with the slot's bits 2-3 folded in; ks & 3 flips those bits (XOR 64 bytes each), ks >> 2 and the tile are immediates
    const uint32_t kx0 = lbase + (uint32_t)v16 * CK::ROWB + ((uint32_t)((g4 ^ (v16 & 3)) | (v16 & 12)) << 4);
    // pass 2 reads (the 32-video kernel's)
    const int i16 = lane & 15;
    const uint32_t trow = 4 * (g4 >> 1) + (i16 >> 2);
    const uint32_t u_rd = lbase + trow * CU::ROWB + (g4 & 1) * 32 + (i16 & 3) * 8;
    // video 16 w + v16, segments 4 g4 .. 4 g4 + 3 of a 16-segment tile -- in the K-SLOT order pass 2's operands use (the 32 x 32 accumulator order
    // the value fragments' transposing reads follow): slot (hh, jj) of a half tile is segment 8 (jj >> 2) + 4 hh + (jj & 3), so the four segments of
    // lane group g4 sit at byte 16 (g4 & 1) + 8 (g4 >> 1) of the half tile's 32
    const uint32_t p_wr = lbase + X::P_OFF + (uint32_t)(16 * wave + v16) * PP + (uint32_t)((g4 & 1) * 16 + (g4 >> 1) * 8);
    const uint32_t p_rd = lbase + X::P_OFF + (uint32_t)r * PP + hh * 16;                              // video r (+ 32 vt), k-slots 8 hh .. of a half tile

    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();                                               // the track table and the constants are in LDS
    Trk tk;
    load_track(0, tk);
    // pass 1's Q fragments (this wave's 16 videos): not kept through pass 2 (its accumulators need the registers) but re-read for the NEXT track
    // under the z tiles' sums, in front of that track's K pieces -- read at the head of the track they cost 2 400 cycles of it (stamps: the rows
    // leave L2 between two tracks)
    bf16x8 qf[D / 32];
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) qf[ks] = *(const bf16x8*)((const unsigned char*)a.Q + (q_off + (uint32_t)ks * 64u));
    issue_k(m_begin, tk);

    for (int jt = 0; jt < T; ++jt) {
        const int64_t m = m_begin + jt;
        const int s_eff = tk.s_eff;
        const int NH = s_eff > 0 ? (s_eff + 15) / 16 : 1;          // 16-segment tiles of the track: score tiles of pass 1, half tiles of pass 2
        XS_STAMP(0);

        // ================================================================================================ pass 1: scores of this wave's 16 videos
        f32x4 sacc[6];                                              // (a tile's first MFMA takes a literal zero accumulator: nothing to clear)
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        uint32_t ka[4];
        {
            const uint32_t kx = xa_opaque(kx0);                    // (recomputed per track: registers, not instructions, are what this kernel is short of)
#pragma unroll
            for (int q = 0; q < 4; ++q) ka[q] = kx ^ (uint32_t)(q << 6);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(qf[4]), "+v"(qf[5]), "+v"(qf[6]), "+v"(qf[7]) :: "memory");   // this wave's K pieces (and Q)
        XS_STAMP(1);
        XA_BARRIER();
        XS_STAMP(2);
#define XS64_TILE(ST)                                                                                                                   \
        if (ST < NH) {                                                                                                                  \
            constexpr int TO = (ST >> 1) * CK::STG + (ST & 1) * 16 * CK::ROWB;                                                          \
            bf16x8 f[4];                                                                                                                \
            f[0] = xs_read_k<TO>(ka[0]); f[1] = xs_read_k<TO>(ka[1]); f[2] = xs_read_k<TO>(ka[2]); f[3] = xs_read_k<TO>(ka[3]);          \
            xa_wait_lgkm<3>(f[0]); sacc[ST] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], qf[0], z4, 0, 0, 0); f[0] = xs_read_k<TO + 256>(ka[0]); \
            xa_wait_lgkm<3>(f[1]); sacc[ST] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], qf[1], sacc[ST], 0, 0, 0); f[1] = xs_read_k<TO + 256>(ka[1]); \
            xa_wait_lgkm<3>(f[2]); sacc[ST] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[2], qf[2], sacc[ST], 0, 0, 0); f[2] = xs_read_k<TO + 256>(ka[2]); \
            xa_wait_lgkm<3>(f[3]); sacc[ST] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[3], qf[3], sacc[ST], 0, 0, 0); f[3] = xs_read_k<TO + 256>(ka[3]); \
            xa_wait_lgkm<3>(f[0]); sacc[ST] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], qf[4], sacc[ST], 0, 0, 0);                  \
            xa_wait_lgkm<2>(f[1]); sacc[ST] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], qf[5], sacc[ST], 0, 0, 0);                  \
            xa_wait_lgkm<1>(f[2]); sacc[ST] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[2], qf[6], sacc[ST], 0, 0, 0);                  \
            xa_wait_lgkm<0>(f[3]); sacc[ST] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[3], qf[7], sacc[ST], 0, 0, 0);                  \
        }
        XS64_TILE(0) XS64_TILE(1) XS64_TILE(2) XS64_TILE(3) XS64_TILE(4) XS64_TILE(5)
#undef XS64_TILE
        XS_STAMP(3);
        XA_BARRIER();                                               // the K tiles are consumed: their LDS is the value ring now
        XS_STAMP(4);
        issue_u(m, 0, tk);
        if (NH > 1) issue_u(m, 1, tk);
        XS_STAMP(5);

        // the softmax of a video is this wave's own: lane (g4, v16) holds segments 16 st + 4 g4 + (0..3) of video 16 w + v16
        float mx = -INFINITY;
#pragma unroll
        for (int st = 0; st < 6; ++st)
            if (st < NH) {
                const unsigned wbits = tile_word(tk, st >> 1) >> (16 * (st & 1) + 4 * g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sc = ((wbits >> e) & 1u) ? sacc[st][e] * c : -INFINITY;
                    sacc[st][e] = sc;
                    mx = fmaxf(mx, sc);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, xa_other_half(mx));
        float psum = 0.f;
#pragma unroll
        for (int st = 0; st < 6; ++st)
            if (st < NH) {
                bf16x4 pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pr_ = __builtin_amdgcn_exp2f(sacc[st][e] - mx);
                    psum += pr_;
                    pf[e] = (bf16_t)pr_;
                }
                xa_wr<bf16x4>(p_wr + (uint32_t)st * 32, pf);
            }
        psum += __shfl_xor(psum, 16);
        psum += xa_other_half(psum);
        if (g4 == 0) xa_wr<float>(lbase + X::L_OFF + (uint32_t)(16 * wave + v16) * 4, psum);
        XS_STAMP(6);

        // ================================================================================================ pass 2: [O | Z]^T = [U | U'']^T P^T, 16 segments per step
        f32x16 oacc[4][2];                                          // [0], [1]: rows 32 w .., 128 + 32 w .. of o; [2], [3]: the same rows of z; x video tile
        uint32_t ug[4];
        {
            const uint32_t tr7 = xa_opaque(trow & 7u);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) ug[dt] = (((uint32_t)(wave + 4 * dt)) ^ tr7) << 6;
        }
        // one step: half tile h (its pieces awaited, the barrier passed) times the probabilities of both video tiles.  FIRST: the accumulators
        // start from a literal zero instead of being cleared (128 moves per track)
#define XS64_STEP(FIRST)                                                                                                                         \
        {                                                                                                                                        \
            if (h + 1 < NH) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      /* half tile h has landed (h + 1 may be in flight) */          \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                                \
            XA_BARRIER();                                           /* (the first one also publishes the probabilities and their sums) */       \
            if (h + 2 < NH) issue_u(m, h + 2, tk);                  /* into the slot of half tile h - 1, which everyone has left */              \
            const uint32_t ub = u_rd + (uint32_t)((h % 3) * CU::HSTG);                                                                           \
            bf16x8 pb0 = xa_read128(p_rd + (uint32_t)h * 32), pb1 = xa_read128_off<32 * PP>(p_rd + (uint32_t)h * 32);                            \
            bf16x4 lo[4], hi[4];                                                                                                                 \
            _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) {                                                                                   \
                const uint32_t va = ub + ug[dt];                                                                                                 \
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[dt]) : "v"(va));                                                              \
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[dt]) : "v"(va), "n"(8 * CU::ROWB));                                 \
            }                                                                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pb0), "+v"(pb1), "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])); \
            _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) {                                                                                   \
                const bf16x8 uf = __builtin_shufflevector(lo[dt], hi[dt], 0, 1, 2, 3, 4, 5, 6, 7);                                               \
                oacc[dt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uf, pb0, FIRST ? z16 : oacc[dt][0], 0, 0, 0);                              \
                oacc[dt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uf, pb1, FIRST ? z16 : oacc[dt][1], 0, 0, 0);                              \
            }                                                                                                                                    \
        }
        {
            const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            { const int h = 0; XS64_STEP(true) }
            for (int h = 1; h < NH; ++h) XS64_STEP(false)
        }
#undef XS64_STEP
        XS_STAMP(7);
        XA_BARRIER();                                               // value ring and probabilities are free
        XS_STAMP(8);
        // g3 * vn for the sums below ([video tile][j][g >> 1]: eight 16-byte loads, about 2 900 cycles from L2's far side by the stamps), in flight
        // under the o tiles' sums.  Plain loads: the same loads from inline assembly with a counted wait further down (so that the next track's K
        // pieces could go out in between) gave wrong pairs at scale -- between an assembly load and its wait the compiler is free to move the
        // registers it was given -- and did not pay either: a wave with 28 loads in flight blocks at issue (3 800 cycles for the 12 K pieces).
        bf16x8 gq[2][2][2];
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            const int rr = (int)xa_opaque((uint32_t)lane) & 31;
            const uint32_t gv_off = (uint32_t)(xs_ws_gvb(a.Nv, D) * 4 + vid(n0 + 32 * vt + rr) * (D * 2) + (wave * 2 + hh) * 32);   // tile wave (+ 4 j)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) gq[vt][j][q] = *(const bf16x8*)((const unsigned char*)a.ws + (gv_off + (uint32_t)(j * 256 + q * 16)));
        }
        // the o tiles first (LayerNorm2's statistics are all that is read of them): their 64 registers are free before the z tiles' sums begin
        float su_[2], sq_[2];
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            const float inv_l = 1.f / xa_rd<float>(lbase + X::L_OFF + (uint32_t)(32 * vt + r) * 4);
            f32x2_t su2 = {0.f, 0.f}, sq2 = {0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2_t x2 = {oacc[dt][vt][e], oacc[dt][vt][e + 1]};
                    su2 += x2; sq2 += x2 * x2;
                }
            float su = (su2[0] + su2[1]) * inv_l, sq = (sq2[0] + sq2[1]) * (inv_l * inv_l);
            su += xa_other_half(su); sq += xa_other_half(sq);
            su_[vt] = su; sq_[vt] = sq;
        }
        XS_STAMP(9);
        // gv has arrived (nothing else of this wave is in flight); only now the next track's Q rows and K pieces go out -- issued earlier they
        // would sit between these loads and their wait, which the compiler can only write as vmcnt(0)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(su_[0]), "+v"(su_[1]), "+v"(sq_[0]), "+v"(sq_[1]), "+v"(gq[0][0][0]), "+v"(gq[0][0][1]), "+v"(gq[0][1][0]), "+v"(gq[0][1][1]),
                     "+v"(gq[1][0][0]), "+v"(gq[1][0][1]), "+v"(gq[1][1][0]), "+v"(gq[1][1][1]) :: "memory");
        XS_STAMP(10);
        Trk nx; nx.s_eff = 0; nx.first = 0; nx.w0 = nx.w1 = nx.w2 = 0u;
        // (unconditionally: inside the `if` the old fragments would have to stay alive through pass 2 for the path that skips the load)
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) qf[ks] = *(const bf16x8*)((const unsigned char*)a.Q + (q_off + (uint32_t)ks * 64u));
        if (jt + 1 < T) { load_track(jt + 1, nx); issue_k(m + 1, nx); }
        XS_STAMP(11);
        float Q1_[2], Q2_[2];
        f32x16 facc[2];
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            f32x2_t q1 = {0.f, 0.f}, q2 = {0.f, 0.f};
            const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t g2_r = xa_opaque(lbase + X::G2_OFF + (uint32_t)(32 * (wave + 4 * j) + 4 * hh) * 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 g2 = xa_rd<f32x4>(g2_r + g * 32);
#pragma unroll
                    for (int jj = 0; jj < 4; jj += 2) {
                        const f32x2_t z2 = {oacc[2 + j][vt][4 * g + jj], oacc[2 + j][vt][4 * g + jj + 1]};
                        const f32x2_t zz = z2 * z2;
                        q1 += zz;
                        q2 += zz * (f32x2_t){g2[jj], g2[jj + 1]};
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 zb;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) zb[jj] = (bf16_t)oacc[2 + j][vt][8 * s2 + jj];
                    const bf16x8 af = xa_rd<bf16x8>(lbase + X::AFR_OFF + (uint32_t)((((wave * 2 + j) * 2 + s2) * 2 + hh) * 8 + (r < 7 ? r : 7)) * 16);
                    facc[vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, zb, (j == 0 && s2 == 0) ? z16 : facc[vt], 0, 0, 0);
                }
            }
            float Q1 = q1[0] + q1[1], Q2 = q2[0] + q2[1];
            Q1 += xa_other_half(Q1); Q2 += xa_other_half(Q2);
            Q1_[vt] = Q1; Q2_[vt] = Q2;
        }
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            f32x2_t b1 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int jj = 0; jj < 4; jj += 2) {
                        const f32x2_t z2 = {oacc[2 + j][vt][4 * g + jj], oacc[2 + j][vt][4 * g + jj + 1]};
                        b1 += z2 * (f32x2_t){(float)gq[vt][j][g >> 1][(g & 1) * 4 + jj], (float)gq[vt][j][g >> 1][(g & 1) * 4 + jj + 1]};
                    }
            float B1 = b1[0] + b1[1];
            B1 += xa_other_half(B1);
            const uint32_t pw = lbase + X::PART_OFF + (uint32_t)((wave * PQ + 32 * vt + r) * 48);
            if (hh == 0) { xa_wr<f32x4>(pw, (f32x4){su_[vt], sq_[vt], Q1_[vt], Q2_[vt]}); xa_wr<f32x4>(pw + 16, (f32x4){facc[vt][0], facc[vt][1], facc[vt][2], facc[vt][3]}); }
            else xa_wr<f32x4>(pw + 32, (f32x4){facc[vt][0], facc[vt][1], facc[vt][2], B1});
        }
        XS_STAMP(12);
        XA_BARRIER();
        XS_STAMP(13);
        // ---- one wave (they take turns), lane = video: LayerNorm2's k1 / k2, the six sums, LayerNorm3 + cosine
        if (!DBG && wave == (jt & 3)) {
            const uint32_t pr = xa_opaque(lbase + X::PART_OFF + (uint32_t)lane * 48);
            f32x4 A0 = xa_rd<f32x4>(pr), A1 = xa_rd<f32x4>(pr + 16), A2 = xa_rd<f32x4>(pr + 32);
#pragma unroll
            for (int q = 1; q < 4; ++q) { A0 += xa_rd<f32x4>(pr + q * (PQ * 48)); A1 += xa_rd<f32x4>(pr + q * (PQ * 48) + 16); A2 += xa_rd<f32x4>(pr + q * (PQ * 48) + 32); }
            const float inv_l = 1.f / xa_rd<float>(lbase + X::L_OFF + (uint32_t)lane * 4);
            const uint32_t pv_off = (uint32_t)((xs_ws_pp(a.Nv, D) + vid(n0 + (int64_t)xa_opaque((uint32_t)lane)) * 4) * 4);   // sum gv, sum b3 vn, sum gv Bv, sum gv Av of video `lane`
            const f32x4 pv4 = *(const f32x4*)((const unsigned char*)a.ws + pv_off);
            const float mean = A0[0] * (1.f / D);
            const float var = fmaxf(A0[1] * (1.f / D) - mean * mean, 0.f);
            const float k1n = __builtin_amdgcn_rsqf(var + a.eps), k2 = -mean * k1n, k1 = k1n * inv_l;
            const float Q1 = A0[2], Q2 = A0[3], F1 = A1[0], F2 = A1[1], F3 = A1[2], F4 = A1[3], F5 = A2[0], F6 = A2[1], F7 = A2[2], B1 = A2[3];
            const float s1 = k1 * F1 + k2 * mc[XC_BV] + mc[XC_AV];
            const float s2 = k1 * k1 * Q1 + 2.f * k1 * (k2 * F2 + F3) + k2 * k2 * mc[XC_BV2] + 2.f * k2 * mc[XC_BVAV] + mc[XC_AV2];
            const float p1 = k1 * B1 + k2 * pv4[2] + pv4[3];
            const float c1 = k1 * F4 + k2 * mc[XC_G2BV] + mc[XC_G2AV];
            const float c2 = k1 * k1 * Q2 + 2.f * k1 * (k2 * F5 + F6) + k2 * k2 * mc[XC_G2BV2] + 2.f * k2 * mc[XC_G2BVAV] + mc[XC_G2AV2];
            const float e1 = k1 * F7 + k2 * mc[XC_GBBV] + mc[XC_GBAV];
            const float mu = s1 * (1.f / D);
            const float vy = fmaxf(s2 * (1.f / D) - mu * mu, 0.f);
            const float rs = __builtin_amdgcn_rsqf(vy + a.eps);
            const float dot = rs * (p1 - mu * pv4[0]) + pv4[1];
            const float zz = rs * rs * (c2 - 2.f * mu * c1 + mu * mu * mc[XC_G2]) + 2.f * rs * (e1 - mu * mc[XC_GB]) + mc[XC_B2];
            const int64_t vrow = n0 + (int64_t)xa_opaque((uint32_t)lane);          // (not hoisted out of the track loop: it would live in two registers)
            if (vrow < a.Nv) a.sims[vrow * a.ld_sims + m] = dot * __builtin_amdgcn_rsqf(zz);
        }
        XS_STAMP(14);
        XS_STAMP(15);
        tk = nx;
    }
}

// (Round 5 also tried this kernel with EIGHT waves of <= 128 registers per workgroup -- four waves per SIMD from two workgroups, a wave per
//  32 rows of o and of z, the softmax of a video split over two waves: 120 registers, no spill, the same sums to 1.2e-7 -- and measured
//  60.1 ms against 53.9 ms on 53 000 x 4 000: the per-pair instruction work is what the loop costs, and the split adds an exchange barrier and
//  reads every probability row twice.  profiles/r05_xpool_sims_8wave.txt; the source is in the history, commit "made_xpool_sims: 8-wave variant".)

// per video: gv = g3 * vn, (sum gv, sum b3 vn, sum gv Bv, sum gv Av); per model: the XsConst sums.  One wave per video.
template <int D>
__global__ __launch_bounds__(256) void xpool_sims_prep_kernel(const float* vn, int64_t ldvn, const float* g3, const float* b3, const float* av, const float* bv,
                                                             float* ws, int64_t Nv) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    static_assert(D == 256, "one f32x4 per lane");
    const f32x4 g = *(const f32x4*)(g3 + lane * 4), b = *(const f32x4*)(b3 + lane * 4);
    const f32x4 A = *(const f32x4*)(av + lane * 4), B = *(const f32x4*)(bv + lane * 4);
    if (n < Nv) {
        const f32x4 v = *(const f32x4*)(vn + n * ldvn + lane * 4);
        f32x4 gvv;
        float sg = 0.f, sb = 0.f, sgb = 0.f, sga = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { gvv[j] = g[j] * v[j]; sg += gvv[j]; sb += b[j] * v[j]; sgb += gvv[j] * B[j]; sga += gvv[j] * A[j]; }
        *(f32x4*)(ws + n * D + lane * 4) = gvv;
        {   // d = 4 lane = 32 t + 8 g + 4 hh + i  ->  bf16 index ((t 2 + hh) 4 + g) 4 + i of the video's 256 (a lane's values of one accumulator tile: 32 contiguous bytes)
            const int d = 4 * lane, hh = (d >> 2) & 1, g = (d >> 3) & 3, t = d >> 5;      // (t: the 32-row tile)
            bf16_t* gb = (bf16_t*)(ws + xs_ws_gvb(Nv, D)) + n * D + (((t * 2 + hh) * 4 + g) * 4);
            *(bf16x4*)gb = (bf16x4){(bf16_t)gvv[0], (bf16_t)gvv[1], (bf16_t)gvv[2], (bf16_t)gvv[3]};
        }
        sg = wave_sum(sg); sb = wave_sum(sb); sgb = wave_sum(sgb); sga = wave_sum(sga);
        if (lane == 0) *(f32x4*)(ws + xs_ws_pp(Nv, D) + n * 4) = (f32x4){sg, sb, sgb, sga};
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        float c[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) c[q] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g2 = g[j] * g[j], gb = g[j] * b[j];
            c[XC_G2] += g2; c[XC_GB] += gb; c[XC_B2] += b[j] * b[j];
            c[XC_BV] += B[j]; c[XC_AV] += A[j]; c[XC_BV2] += B[j] * B[j]; c[XC_BVAV] += B[j] * A[j]; c[XC_AV2] += A[j] * A[j];
            c[XC_G2BV] += g2 * B[j]; c[XC_G2AV] += g2 * A[j]; c[XC_G2BV2] += g2 * B[j] * B[j]; c[XC_G2BVAV] += g2 * B[j] * A[j]; c[XC_G2AV2] += g2 * A[j] * A[j];
            c[XC_GBBV] += gb * B[j]; c[XC_GBAV] += gb * A[j];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float t = wave_sum(c[q]); if (lane == 0) ws[xs_ws_c(Nv, D) + q] = t; }
    }
}

}  // namespace

extern "C" int made_xpool_attention(const MadeXpoolAttnArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_xpool_attention: null args");
    const MadeXpoolAttnArgs& a = *args;
    MADE_REQUIRE(a.Q && a.K && a.U && a.out && a.ws, "made_xpool_attention: null pointer");
    MADE_REQUIRE(a.Nv >= 0 && a.Nm >= 0 && a.S > 0, "made_xpool_attention: bad dims");
    MADE_UNSUPPORTED(a.D == 256 || a.D == 512, "made_xpool_attention: D=%lld not in {256, 512}", (long long)a.D);
    MADE_UNSUPPORTED(a.S <= XA_SMAX, "made_xpool_attention: S=%lld segments per track (at most %d)", (long long)a.S, XA_SMAX);
    MADE_UNSUPPORTED(a.Nm <= 65535, "made_xpool_attention: more than 65535 tracks per call (chunk them)");
    MADE_UNSUPPORTED(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldu % 8 == 0 && a.k_bs % 8 == 0 && a.u_bs % 8 == 0 && a.ldo % 8 == 0 &&
                     ((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.K % 16) == 0 && ((uintptr_t)a.U % 16) == 0 && ((uintptr_t)a.out % 16) == 0 &&
                     ((uintptr_t)a.ws % 16) == 0, "made_xpool_attention: pointers / strides must keep 16-byte alignment");
    MADE_UNSUPPORTED((uint64_t)a.S * (uint64_t)a.ldk * 2 < (1ull << 32) && (uint64_t)a.S * (uint64_t)a.ldu * 2 < (1ull << 32) &&
                     (uint64_t)(a.Nv + 64) * (uint64_t)a.ldo * 2 < (1ull << 32),
                     "made_xpool_attention: a track's K / U rows and its Nv output rows must each span less than 4 GB");
    if (a.Nv == 0 || a.Nm == 0) return MADE_OK;
    hipStream_t st = (hipStream_t)stream;
    int* info = (int*)a.ws;
    hipLaunchKernelGGL(xpool_attn_info_kernel, dim3((unsigned)((a.Nm + 3) / 4)), dim3(256), 0, st, a.key_mask, a.S, a.Nm, info);
    // chunks of tracks per video tile: about four workgroups per CU over the launch, whole tracks
    const int64_t nvt = (a.Nv + XA_PQ - 1) / XA_PQ;
    int64_t chunks = (1024 + nvt - 1) / nvt;
    if (chunks > a.Nm) chunks = a.Nm;
    if (chunks < 1) chunks = 1;
    const int per = (int)((a.Nm + chunks - 1) / chunks);
    dim3 grid((unsigned)nvt, (unsigned)((a.Nm + per - 1) / per));
    const int rc = a.D == 512 ? launch_xpool_attn<512>(a, info, grid, per, st) : launch_xpool_attn<256>(a, info, grid, per, st);
    if (rc != MADE_OK) return rc;
    return made_check_launch("made_xpool_attention");
}

extern "C" int64_t made_xpool_sims_ws_bytes(int64_t Nv, int64_t Nm, int64_t D) { return (xs_ws_info(Nv, D) + Nm * XA_INFO) * 4; }

extern "C" int made_xpool_sims(const MadeXpoolSimsArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_xpool_sims: null args");
    const MadeXpoolSimsArgs& a = *args;
    MADE_REQUIRE(a.Q && a.K && a.UU && a.av && a.bv && a.ln3_g && a.ln3_b && a.vn && a.sims && a.ws, "made_xpool_sims: null pointer");
    MADE_REQUIRE(a.Nv >= 0 && a.Nm >= 0 && a.S > 0, "made_xpool_sims: bad dims");
    MADE_UNSUPPORTED(a.D == 256, "made_xpool_sims: D=%lld (built for 256)", (long long)a.D);
    MADE_UNSUPPORTED(a.S <= 96, "made_xpool_sims: S=%lld segments per track (at most 96: longer tracks take made_xpool_fused)", (long long)a.S);
    MADE_UNSUPPORTED(a.ldk * 2 < (1 << 24) && a.ldu * 2 < (1 << 24), "made_xpool_sims: row strides of at most 8 M elements (row offsets through 24-bit multiplies)");
    MADE_UNSUPPORTED(a.Nm <= 65535, "made_xpool_sims: more than 65535 tracks per call (chunk them)");
    MADE_UNSUPPORTED(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldu % 8 == 0 && a.k_bs % 8 == 0 && a.u_bs % 8 == 0 && a.ldvn % 4 == 0 &&
                     ((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.K % 16) == 0 && ((uintptr_t)a.UU % 16) == 0 && ((uintptr_t)a.vn % 16) == 0 &&
                     ((uintptr_t)a.ws % 16) == 0 && ((uintptr_t)a.ln3_g % 16) == 0 && ((uintptr_t)a.ln3_b % 16) == 0 && ((uintptr_t)a.av % 16) == 0 &&
                     ((uintptr_t)a.bv % 16) == 0,
                     "made_xpool_sims: pointers / strides must keep 16-byte alignment");
    MADE_UNSUPPORTED(a.ldu >= 2 * a.D && (uint64_t)a.S * (uint64_t)a.ldk * 2 < (1ull << 32) && (uint64_t)a.S * (uint64_t)a.ldu * 2 < (1ull << 32),
                     "made_xpool_sims: value rows hold u | u'' (ldu >= 2 D); a track's K / value rows must each span less than 4 GB");
    MADE_UNSUPPORTED((uint64_t)a.Nv * (uint64_t)a.ldq * 2 < (1ull << 32) && (uint64_t)a.Nv * (uint64_t)(a.D + 4 + a.D / 2) * 4 + 64 < (1ull << 32),
                     "made_xpool_sims: Q and the per-video workspace rows are addressed with 32-bit byte offsets (Nv * ldq * 2 and Nv * (1.5 D + 4) * 4 below 4 GB)");
    if (a.Nv == 0 || a.Nm == 0) return MADE_OK;
    hipStream_t st = (hipStream_t)stream;
    constexpr int D = 256;
    float* wsf = (float*)a.ws;
    if (a.prepare_ws)
        hipLaunchKernelGGL(xpool_sims_prep_kernel<D>, dim3((unsigned)((a.Nv + 3) / 4)), dim3(256), 0, st, a.vn, a.ldvn, a.ln3_g, a.ln3_b, a.av, a.bv, wsf, a.Nv);
    int* info = (int*)(wsf + xs_ws_info(a.Nv, D));
    hipLaunchKernelGGL(xpool_attn_info_kernel, dim3((unsigned)((a.Nm + 3) / 4)), dim3(256), 0, st, a.key_mask, a.S, a.Nm, info);
    // the retrieval set's tracks: 32 videos (MADE_XPOOL_SIMS_PQ=64: round 5's 64-video kernel -- half the LDS-DMA bytes per pair, the same speed
    // within 1.2 %: the comment in front of it) and four waves per workgroup, two workgroups per CU; at most MAX_TRACKS per chunk (the track table
    // in LDS), as few partial rounds of the chip as possible
    const bool pq64 = made_variant_env("MADE_XPOOL_SIMS_PQ") && atoi(made_variant_env("MADE_XPOOL_SIMS_PQ")) == 64;
    static bool attr32 = false;
    if (!attr32) {
        hipError_t e = hipFuncSetAttribute((const void*)xpool_sims32_kernel<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)xpool_sims32_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)xpool_sims64_kernel<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)xpool_sims64_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) { made_set_error("made_xpool_sims: cannot reserve 80 KB of LDS: %s", hipGetErrorString(e)); return MADE_ERR_HIP; }
        attr32 = true;
    }
    // MADE_XPOOL_DBG=32: the phase-stamp build of the 32-video kernel (tools/xpool_sims_stamps.py): workgroup (0, 0) writes cycle stamps into the sims buffer, no similarity is written
    const bool stamps32 = made_variant_env("MADE_XPOOL_DBG") && atoi(made_variant_env("MADE_XPOOL_DBG")) == 32;
    const bool stamps64 = made_variant_env("MADE_XPOOL_DBG") && atoi(made_variant_env("MADE_XPOOL_DBG")) == 64;      // the same for the 64-video kernel
    const bool use64 = (pq64 && !stamps32) || stamps64;
    const int64_t nvt32 = use64 ? (a.Nv + 63) / 64 : (a.Nv + 31) / 32;
    // Chunks of at most 64 tracks: the workgroups of an XCD walk a chunk together and share its rows through that XCD's L2 -- the shorter the
    // chunk, the less they drift apart.  53 k x 4 k (profiles/r04_ao_*): 464 tracks per chunk 56.4 ms with 66 GB of L2 misses per launch, 128: 56.2 ms /
    // 56 GB, 64: 57.2 ms / 31 GB, 32: 59.2 ms / 26 GB (a workgroup's prologue -- Q, g3 vn, the constant fragments -- costs about two tracks).
    int64_t max_per = 64;
    if (made_variant_env("MADE_XPOOL_SIMS_PER") && atoi(made_variant_env("MADE_XPOOL_SIMS_PER")) > 0) max_per = atoi(made_variant_env("MADE_XPOOL_SIMS_PER"));
    const int64_t max_tracks = use64 ? Xs64<D>::MAX_TRACKS : Xs32<D>::MAX_TRACKS;
    if (max_per > max_tracks) max_per = max_tracks;
    const int64_t c_lo = (a.Nm + max_per - 1) / max_per;
    double best = 1e30; int64_t bc = c_lo;
    for (int64_t cch = c_lo; cch <= c_lo + 24 && cch <= a.Nm; ++cch) {
        const int64_t pr = (a.Nm + cch - 1) / cch, rounds = (nvt32 * ((a.Nm + pr - 1) / pr) + 511) / 512;
        const double cost = (double)rounds * (double)(pr + 2);
        if (cost < best) { best = cost; bc = cch; }
    }
    if (bc >= 8) bc = (bc + 7) / 8 * 8;                              // whole chunks per XCD (see the kernel's workgroup order)
    const int per32 = (int)((a.Nm + bc - 1) / bc);
    int nch = (int)((a.Nm + per32 - 1) / per32);
    if (nch >= 8) nch = (nch + 7) / 8 * 8;                          // (chunks behind the last track exit at once)
    dim3 g32((unsigned)(nvt32 * nch));
    if (use64 && stamps64) hipLaunchKernelGGL((xpool_sims64_kernel<D, true>), g32, dim3(256), Xs64<D>::TBL_OFF + per32 * 32, st, a, (const int*)info, per32, (int)nvt32, nch);
    else if (use64) hipLaunchKernelGGL((xpool_sims64_kernel<D, false>), g32, dim3(256), Xs64<D>::TBL_OFF + per32 * 32, st, a, (const int*)info, per32, (int)nvt32, nch);
    else if (stamps32) hipLaunchKernelGGL((xpool_sims32_kernel<D, true>), g32, dim3(256), Xs32<D>::TBL_OFF + per32 * 32, st, a, (const int*)info, per32, (int)nvt32, nch);
    else hipLaunchKernelGGL((xpool_sims32_kernel<D, false>), g32, dim3(256), Xs32<D>::TBL_OFF + per32 * 32, st, a, (const int*)info, per32, (int)nvt32, nch);
    return made_check_launch("made_xpool_sims");
}
