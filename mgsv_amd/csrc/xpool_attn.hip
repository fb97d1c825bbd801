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
