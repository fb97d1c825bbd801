// made_xpool_fused: the whole per-pair chain of the X-Pool block in ONE kernel, for all-pairs retrieval scoring
// (reference test-MaDe.py:392-403 = modules/transformer.py:156-180 + modules/metrics.py:10-24).  gfx950, bf16, D = 256.
//
// For every (video n, track m):   scores over the track's segments -> softmax -> pooled U rows (out_proj hoisted onto the
// values: rows of the softmax sum to 1) -> LayerNorm2 -> + Linear (residual) -> LayerNorm3 -> cosine with the video.
// Done as separate launches this chain writes and re-reads three [Nm*Nv, D] tensors -- 0.65 TB of HBM traffic at 53 k x 4 k
// pairs.  Here nothing per-pair ever leaves the chip, and (second generation, below) nothing that does not depend on the
// track is fetched more than once per chunk of tracks.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int XD = 256;                 // model width
constexpr int XKEY = 32;                // segments per tile
constexpr int XT = 256;                 // threads of the small preparation kernels
constexpr int XS_MAX = 1024;            // segments per track this kernel accepts
// ---------------------------------------------------------------------------------------------- the kernel
// Round 1's kernel (one workgroup = 128 videos x ONE track, four waves, everything of a pair in one wave's registers) asked its CU
// for ~375 KB of operands per workgroup -- K / U tiles, the Linear's weight (128 KB), the Q fragments, the per-video cosine vectors
// (PMC: 337 GB of L2 misses per 53 k x 4 k launch against 1 GB of algorithmic bytes) -- and, with one wave per SIMD, exposed every
// LDS / barrier / memory latency to the matrix pipe: 17 us per workgroup.  Here a workgroup of EIGHT waves keeps
// 64 videos for a whole CHUNK OF TRACKS, everything that does not depend on the track stays on chip, and the two waves of a SIMD
// do different things at any time (one is in MFMAs while the other is in softmax / epilogue arithmetic or waiting for LDS):
//   * waves 0-3 (attention): video tile vt = w & 1, half dh = w >> 1 of the width of O^T (64 accumulator VGPRs); both waves of a
//     video tile compute the tile's scores; Q fragments (64 VGPRs) stay in registers for the chunk.  They publish o^ = O / l (bf16,
//     as ready-made B fragments, 32 KB) and their halves of the LayerNorm2 sums in LDS;
//   * waves 4-7 (linear): the Linear's weight lives in REGISTERS as A-operand fragments, 64 output rows per wave (128 VGPRs), so the
//     Linear is split by OUTPUT ROW (not by video): every wave multiplies its rows with o^ of all 64 videos; the LayerNorm3 / cosine
//     sums of a wave's 64 rows are partial sums per video, reduced through LDS.  They work on track j - 1 while the attention waves
//     work on track j;
//   * LayerNorm2 is folded into that product: with k1 = rstd, k2 = -mean rstd of o^,
//         y = W LN2(o^) + b + LN2(o^) = k1 (W'' o^) + k2 Bv + Av,   W'' = (W + I) diag(g2), Av = W b2 + b + b2, Bv = W g2 + g2
//     (W'', Av, Bv prepared once per call), so the residual costs nothing and no normalised copy of O is ever formed;
//   * per track only the K / U tiles move.  Rounds 1-3: global -> LDS directly (global_load_lds), two stages, ONE tile in flight.  Round 4: the
//     tile in flight lives in registers (32 per attention wave), two tiles ahead of the one being multiplied -- see the tile pipeline in the
//     kernel.  Rows are XOR-swizzled on the source side so both the row reads (K) and the transposing reads (U) are bank-conflict free
//     without padding; rows of masked segments are fetched from the track's first valid row instead (their probability is exactly 0, so the
//     product stays 0 whatever the masked rows hold);
//   * s_barrier is workgroup-wide, so both roles execute the same number of barriers per track: the tile barriers of the attention
//     waves are matched by barriers between the linear waves' work units.  Per iteration j (track j for the attention waves):
//         attention:  tile 0 | T | tile 1 | T | ... | X | publish o^(j) | Y
//         linear:     P0 S0 [T] P1 [T] S1 [T ..] | X | Y | finish track j-1      (P / S = product / sums of a video tile of track j-1;
//                                                                                  one-tile tracks: S1 behind X, beside the publishing)
//     X = "o^(j-1) has been consumed", Y = "o^(j) is published, the partial sums of track j-1 are complete".
//   * the launch runs at the socket's power limit (profiles/r04_o_retrieval_clock_power.txt: 1.3 kW, sclk 2.26 GHz, MFMA pipe 32 % busy):
//     cycles saved by better overlap come back as a lower clock; what would pay is less work per pair -- both waves of a video tile
//     compute the same scores (a third of the attention waves' MFMAs), every attention wave reads the whole K tile from LDS.
constexpr int PQ = 64;                             // videos per workgroup
constexpr int PT = 512;                            // threads
constexpr int PTILE_B = XKEY * XD * 2;             // one K (or U) tile: 32 rows x 512 B
constexpr int PSTAGE_B = 2 * PTILE_B;
constexpr int POFF_A3 = 2 * PSTAGE_B;              // [2 video tiles][16 fragments][64 lanes] x 16 B
constexpr int POFF_GV = POFF_A3 + 2 * 16 * 64 * 16;   // [4 waves][2 video tiles][2 row tiles][2][64 lanes] x 16 B
constexpr int POFF_VEC = POFF_GV + 4 * 2 * 2 * 2 * 64 * 16;   // [4][256] f32: Av, Bv, g3^2, g3 b3
constexpr int POFF_PART = POFF_VEC + 4 * XD * 4;   // [2 tracks][4 waves][2 video tiles][32 videos][6] f32
constexpr int PPART_B = 4 * 2 * 32 * 6 * 4;       // (two of them: tracks alternate)
constexpr int POFF_STAT = POFF_PART + 2 * PPART_B;                // [2 video tiles][2 halves][2 lane halves][32 videos][2] f32
constexpr int POFF_BIAS = POFF_STAT + 2 * 2 * 2 * 32 * 2 * 4;
constexpr int POFF_INFO = POFF_BIAS + 2 * XKEY * 4;   // per track of the chunk: last valid + 1 | first valid << 11 | leading valid << 21
#ifndef MADE_XPOOL_LATE_N
#define MADE_XPOOL_LATE_N 1
#endif
constexpr int PLATE_N = MADE_XPOOL_LATE_N;          // tracks of at most this many tiles: the second video tile's sums run behind X (see the linear waves)
constexpr int PMAX_TRACKS = 1024;                   // tracks per chunk (the table's size)
constexpr int PLDS = POFF_INFO + PMAX_TRACKS * 4;
static_assert(PLDS <= 160 * 1024, "made_xpool_fused: the LDS map does not fit a CU");
// workspace sections behind the per-video ones (floats): Av, Bv, then W'' (bf16), then the per-track info
constexpr int64_t WS_AV = 4, WS_BV = 4 + XD, WS_W2 = 4 + 2 * XD, WS_INFO = 4 + 2 * XD + XD * XD / 2;

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char* lds3_t;

// LDS accesses of the persistent kernel go through 32-bit LDS addresses of the form (opaque per-lane base) + (compile-time offset):
// the offset then sits in the instruction's immediate field.  Left to itself the compiler hoists every distinct (base + offset)
// of the track loop into a register of its own -- several hundred of them -- and spills the weight fragments to make room.
__device__ __forceinline__ uint32_t opaque(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
template <typename T> __device__ __forceinline__ T lds_rd(uint32_t addr) { return *(const __attribute__((address_space(3))) T*)(uintptr_t)addr; }
template <typename T> __device__ __forceinline__ void lds_wr(uint32_t addr, T v) { *(__attribute__((address_space(3))) T*)(uintptr_t)addr = v; }
#define XP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define XP_BARRIER_VM() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
// value of the other 32-lane half (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second)
__device__ __forceinline__ float other_half(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? sw[0] : sw[1]);
}

// DBG & 32: workgroup (0, 0) stamps s_memtime at its phase boundaries into the sims buffer ([wave][iteration < 16][point < 16] int64)
#define XP_STAMP(pt) do { if ((DBG & 32) && stamp_on && j < 16 && lane == 0) ((long long*)a.sims)[(wave * 16 + j) * 16 + (pt)] = (long long)__builtin_readcyclecounter(); } while (0)
template <int DBG>
__global__ __launch_bounds__(PT, 1) void xpool_fused_persist_kernel(const MadeXpoolFusedArgs a, const int* __restrict__ info, int tracks_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w4 = wave & 3;
    const bool linear_role = wave >= 4;
    const int r = lane & 31, hh = lane >> 5;
    const int64_t n0 = (int64_t)blockIdx.x * PQ;
    const int64_t m_begin = (int64_t)blockIdx.y * tracks_per_chunk;
    const int64_t m_end = (m_begin + tracks_per_chunk < a.Nm) ? m_begin + tracks_per_chunk : a.Nm;
    if (m_begin >= m_end) return;
    const int T = (int)(m_end - m_begin);
    const int S = (int)a.S;
    const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0;
    const uint32_t lbase = (uint32_t)(uintptr_t)(lds3_t)lds;
    const float* wsc = a.ws + a.Nv * (XD + 2);         // sum g^2, sum g b, sum b^2, -, Av[256], Bv[256], W''[256][256] bf16

    if (tid < XD) {
        float* vec = (float*)(lds + POFF_VEC);
        const float g3 = a.ln3_g[tid], b3 = a.ln3_b[tid];
        vec[tid] = wsc[WS_AV + tid]; vec[XD + tid] = wsc[WS_BV + tid]; vec[2 * XD + tid] = g3 * g3; vec[3 * XD + tid] = g3 * b3;
    }
    // the chunk's track table in LDS (a global load per track would sit on the attention waves' critical path)
    for (int i = tid; i < T; i += PT) {
        unsigned v = (unsigned)S | ((unsigned)S << 21);
        if (info) { const int* ip = info + 4 * (m_begin + i); v = (unsigned)ip[0] | ((unsigned)ip[1] << 11) | ((unsigned)ip[2] << 21); }
        ((unsigned*)(lds + POFF_INFO))[i] = v;
    }
    auto track_info = [&](int i, int& s_eff, int& first, int& nfull) __attribute__((always_inline)) {
        const unsigned v = __builtin_amdgcn_readfirstlane(lds_rd<unsigned>(lbase + POFF_INFO + 4 * i));
        s_eff = (int)(v & 2047u); first = (int)((v >> 11) & 1023u); nfull = (int)(v >> 21);
    };
    auto tiles_of = [&](int s_eff) __attribute__((always_inline)) { return s_eff > 0 ? (s_eff + XKEY - 1) / XKEY : 1; };
    const uint32_t ldk_b = (uint32_t)a.ldk * 2u, ldu_b = (uint32_t)a.ldu * 2u;

    if (!linear_role) {
        // ================================================================================================ attention waves
        const int g = lane >> 4, i16 = lane & 15;
        const int vt = w4 & 1, dh = w4 >> 1;
        const int64_t my_n = n0 + vt * 32 + r;
        const int64_t nc = my_n < a.Nv ? my_n : a.Nv - 1;
        bf16x8 qf[XD / 16];                            // B operand of S^T = K Q^T: lane (r, hh) holds Q[n][ks*16 + hh*8 ..]
        {
            const bf16_t* qp = (const bf16_t*)a.Q + nc * a.ldq + hh * 8;
#pragma unroll
            for (int ks = 0; ks < XD / 16; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
        }
        const float c = a.scale * 1.4426950408889634f;
        const uint32_t a3w0 = lbase + POFF_A3 + 16 * lane + (vt * 16 + dh * 8) * 1024;   // fragment (d, s2) of this half at + (d*2 + s2) * 1024
        const uint32_t statw0 = lbase + POFF_STAT + (((vt * 2 + dh) * 2 + hh) * 32 + r) * 8;
        const uint32_t bias_h0 = lbase + POFF_BIAS + 16 * hh;
        const uint32_t k_l0 = lbase + r * 512 + ((hh ^ r) << 4);                          // K fragment ks at ^ (ks << 5)
        const int trow = 4 * (g >> 1) + (i16 >> 2);                                       // row of the transposing read within a 16-row half
        const uint32_t u_l0 = lbase + PTILE_B + trow * 512 + (g & 1) * 32 + (i16 & 3) * 8 + (((trow & 7) ^ (4 * dh)) << 6);   // 64-byte group d at ^ (d << 6)

        // ---- the K / U tile pipeline (round 4).  Rounds 1-3 moved a tile global -> LDS directly (global_load_lds) into the stage the previous
        // tile had just left: ONE tile in flight, and the stamps (profiles/r04_m_*) show a tile step lasting as long as that transfer --
        // ~1300 cycles for the CU's texture path to take the four waves' 32 x 1 KB instructions, then the way back from L2 -- whoever issues
        // it.  A third 32 KB stage does not fit beside the o^ and g3 v fragments.  So the tile in flight lives in REGISTERS (32 per attention
        // wave: its four 2-row pieces of K and of U): at the top of step k the registers (tile k + 1, fetched a whole step ago) go to the stage
        // tile k - 1 has just left, the loads of tile k + 2 are issued into them, and tile k is computed from the other stage -- two tiles
        // ahead with two stages, plain loads (cheaper on the texture path than LDS-DMA), and nothing ever waited for.  The swizzle is applied on
        // the global side as before, the LDS write is lane-linear; masked rows fetch the track's first valid row (their probability is 0).
        bf16x8 kreg[4], ureg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) { kreg[i][jj] = (bf16_t)0.f; ureg[i][jj] = (bf16_t)0.f; }
        float pend_bias = 0.f;                         // wave 0, lanes < 32: the mask bias of the tile in the registers
        int lj = 0, lt = 0, l_seff = 0, l_first = 0, l_nfull = 0, l_nt = 1;   // the loader's cursor: tile lt of track lj is the next to fetch
        // fetch, part 1: byte offsets (from the track's K / U base) of this wave's pieces of the cursor's tile, the tile's mask bias, and the
        // cursor's advance.  Behind the chunk's last tile the offsets are those of that tile again (the loads are still issued, the data
        // lands in a stage nobody reads): one code path, so the loaded registers are plain values and not the merge of two branches, which
        // the compiler would wait for on the spot.
        uint32_t offk[4], offu[4];
        const unsigned char *Kb = (const unsigned char*)a.K, *Ub = (const unsigned char*)a.U;
        float next_bias = 0.f;
        auto fetch_prepare = [&]() __attribute__((always_inline)) {
            const bool live = lj < T;
            const int64_t m = m_begin + (live ? lj : T - 1);
            Kb = (const unsigned char*)a.K + m * a.k_bs * 2;
            Ub = (const unsigned char*)a.U + m * a.u_bs * 2;
            const uint32_t cl = lane & 31;
            if ((lt + 1) * XKEY <= l_nfull) {              // every segment of the tile is valid: nothing to compute per piece
                const uint32_t tk = (uint32_t)(lt * XKEY) * ldk_b, tu = (uint32_t)(lt * XKEY) * ldu_b;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t row = 2 * (i * 4 + w4) + hh;
                    offk[i] = tk + row * ldk_b + (cl ^ (row & 31)) * 16u;
                    offu[i] = tu + row * ldu_b + ((((cl >> 2) ^ (row & 7)) << 2) | (cl & 3)) * 16u;
                }
                next_bias = 0.f;
            } else {
                const float* maskg = a.key_mask ? a.key_mask + m * a.S : nullptr;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 2 * (i * 4 + w4) + hh, seg = lt * XKEY + row;
                    bool valid = seg < l_nfull;
                    if (!valid && seg < l_seff) valid = *(const float*)((const unsigned char*)maskg + (size_t)opaque((uint32_t)seg * 4u)) != 0.f;
                    const uint32_t srow = (uint32_t)(valid ? seg : l_first);
                    offk[i] = srow * ldk_b + (cl ^ (uint32_t)(row & 31)) * 16u;                                  // 16-byte chunks swizzled by row
                    offu[i] = srow * ldu_b + ((((cl >> 2) ^ (uint32_t)(row & 7)) << 2) | (cl & 3)) * 16u;        // 64-byte groups swizzled by row
                }
                if (wave == 0) {
                    const int seg = lt * XKEY + (lane & 31);
                    bool valid = seg < l_nfull;
                    if (!valid && seg < l_seff) valid = *(const float*)((const unsigned char*)maskg + (size_t)opaque((uint32_t)seg * 4u)) != 0.f;
                    next_bias = valid ? 0.f : -INFINITY;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { offk[i] = opaque(offk[i]); offu[i] = opaque(offu[i]); }   // (32-bit offsets from a uniform base: the scalar-base form of the load)
            if (live && ++lt == l_nt) {
                lt = 0; ++lj;
                if (lj < T) { track_info(lj, l_seff, l_first, l_nfull); l_nt = tiles_of(l_seff); }
            }
        };
        // fetch, part 2 (the loads) and the stash (registers -> stage p, lane-linear) are spelled out where they are used: inside the tile
        // step they sit BETWEEN the MFMAs of the score product, one per MFMA -- issued back to back they cost the wave ~70-90 cycles each
        // (the CU's one texture path takes the four waves' 32 KB at 64 B / cycle; the LDS pipe is shared with the fragment reads)
        const uint32_t st_w0 = lbase + w4 * 1024 + 16 * lane;                              // piece i of K at + i * 4096, of U at + PTILE_B + i * 4096
#define XP_LOADS() do { _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
            kreg[i] = *(const bf16x8*)(Kb + (size_t)offk[i]); ureg[i] = *(const bf16x8*)(Ub + (size_t)offu[i]); } } while (0)
#define XP_STASH(st) do { _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
            lds_wr<bf16x8>((st) + i * 4096, kreg[i]); lds_wr<bf16x8>((st) + PTILE_B + i * 4096, ureg[i]); } } while (0)
        auto stash_bias = [&](int p) __attribute__((always_inline)) {
            if (wave == 0 && lane < XKEY) lds_wr<float>(opaque(lbase + POFF_BIAS + 4 * lane + p * (XKEY * 4)), pend_bias);
        };

        int s_eff, first, nfull;
        int p = 0;
        // every register load of the prologue has landed (a builtin, so that the compiler's own wait-count bookkeeping knows it)
        __builtin_amdgcn_s_waitcnt(0x0070);
        XP_BARRIER();                                                    // B0: vec / gv / track table published
        track_info(0, l_seff, l_first, l_nfull); l_nt = tiles_of(l_seff);
        if (!(DBG & 16)) {
            fetch_prepare(); pend_bias = next_bias; XP_LOADS();
            stash_bias(0); XP_STASH(opaque(st_w0));                      // tile 0 -> stage 0
            fetch_prepare(); pend_bias = next_bias; XP_LOADS();          // tile 1 -> registers
        }
        XP_BARRIER();                                                    // B1: tile 0 of the first track is in LDS

        for (int j = 0; j <= T; ++j) {
            f32x16 o[4];
            float l_tot = 1.f;
            XP_STAMP(0);
            if (j < T) {
                track_info(j, s_eff, first, nfull);
                const int ntiles = tiles_of(s_eff);
                const int cur_nfull = nfull;
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
                float m_run = -INFINITY, l_run = 0.f;

                for (int t = 0; t < ntiles; ++t) {
                    if (t > 0) XP_BARRIER();                             // T: every wave has left stage p ^ 1 (tile k - 1); tile k is in stage p
                    if (!(DBG & 16)) { stash_bias(p ^ 1); fetch_prepare(); pend_bias = next_bias; }
                    if (t == 0) XP_STAMP(11);
                    const uint32_t k_l = opaque(k_l0 + p * PSTAGE_B), u_l = opaque(u_l0 + p * PSTAGE_B), bias_h = opaque(bias_h0 + p * (XKEY * 4));

                    // ---- S^T [32 segments x 32 videos] (both waves of the video tile); between its MFMAs: the registers (tile k + 1) -> stage
                    // p ^ 1, then the loads of tile k + 2 into them
                    f32x16 s;
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[e] = 0.f;
                    {
                        // (the fragment reads run three ahead of the MFMAs)
                        constexpr int NK = (DBG & 8) ? 1 : XD / 16;
                        bf16x8 kf[NK];
                        // program order IS the issue order here (a scheduling fence after every MFMA): fragment read three ahead, MFMA, then
                        // one piece of the stash (MFMAs 0-7) or one load of the next fetch (MFMAs 8-15)
                        const uint32_t st = opaque(st_w0 + (p ^ 1) * PSTAGE_B);
#pragma unroll
                        for (int ks = 0; ks < (NK < 3 ? NK : 3); ++ks) kf[ks] = lds_rd<bf16x8>(k_l ^ (ks << 5));
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ks = 0; ks < NK; ++ks) {
                            if (ks + 3 < NK) kf[ks + 3] = lds_rd<bf16x8>(k_l ^ ((ks + 3) << 5));
                            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
                            if (!(DBG & 16) && NK == 16) {
                                const int i = (ks & 7) >> 1;
                                if (ks < 8) { if (ks & 1) lds_wr<bf16x8>(st + PTILE_B + i * 4096, ureg[i]); else lds_wr<bf16x8>(st + i * 4096, kreg[i]); }
                                else { if (ks & 1) ureg[i] = *(const bf16x8*)(Ub + (size_t)offu[i]); else kreg[i] = *(const bf16x8*)(Kb + (size_t)offk[i]); }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (!(DBG & 16) && NK != 16) { XP_STASH(st); XP_LOADS(); }
                    }
                    if (t == 0) { asm volatile("" : "+v"(s[0])); XP_STAMP(12); }
                    // ---- online softmax (per video = per lane column; the two lane halves hold different segments)
                    float mx = -INFINITY;
                    if ((t + 1) * XKEY <= cur_nfull) {                   // every segment of the tile is valid
#pragma unroll
                        for (int e = 0; e < 16; ++e) { s[e] = s[e] * c; mx = fmaxf(mx, s[e]); }
                    } else {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const f32x4 b4 = lds_rd<f32x4>(bias_h + 32 * g4);
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) { s[4 * g4 + jj] = s[4 * g4 + jj] * c + b4[jj]; mx = fmaxf(mx, s[4 * g4 + jj]); }
                        }
                    }
                    mx = fmaxf(mx, other_half(mx));
                    // the running maximum moves only when a score beats it by more than 2^8 (round 1's rule)
                    const bool move = mx > m_run + 8.f || m_run == -INFINITY;
                    const float m_new = move ? fmaxf(m_run, mx) : m_run;
                    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
                    float psum = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { s[e] = __builtin_amdgcn_exp2f(s[e] - m_use); psum += s[e]; }
                    l_run = l_run * alpha + psum;
                    m_run = m_new;
                    if (t > 0 && __any(move)) {
#pragma unroll
                        for (int d = 0; d < 4; ++d)
#pragma unroll
                            for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
                    }
                    if (t == 0) { asm volatile("" : "+v"(s[0])); XP_STAMP(13); }
                    // ---- O^T (this wave's 128 rows) += U^T P^T, U^T read transposed out of the row-major tile.  The transposing reads
                    // are inline assembly: as the builtin the compiler cannot tell them from a read of the stage the next tile is being
                    // DMA'd into and puts s_waitcnt vmcnt(0) in front of them, which serialises the prefetch with the multiply.  Their
                    // completion is therefore awaited by hand (the partner wave of the SIMD covers the wait).
                    {
                        constexpr int ND = (DBG & 8) ? 1 : 4;
                        bf16x8 pf[2];
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int jj = 0; jj < 8; ++jj) pf[s2][jj] = (bf16_t)s[8 * s2 + jj];
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            bf16x4 lo[ND], hi[ND];
#pragma unroll
                            for (int d = 0; d < ND; ++d) {
                                const uint32_t va = (u_l ^ (d << 6)) + s2 * (16 * 512);
                                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[d]) : "v"(va));
                                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(hi[d]) : "v"(va));
                            }
                            if (ND == 4)
                                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
                            else
                                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]));
#pragma unroll
                            for (int d = 0; d < ND; ++d)
                                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(lo[d], hi[d], 0, 1, 2, 3, 4, 5, 6, 7), pf[s2], o[d], 0, 0, 0);
                        }
                    }
                    p ^= 1;
                    XP_STAMP(1 + (t < 3 ? t : 3));
                }
                l_tot = l_run + other_half(l_run);
            }
            XP_STAMP(5);
            XP_BARRIER();                                                // X: the linear waves have consumed o^ of the previous track
            XP_STAMP(6);
            if (j < T) {
                // ---- o^ = O / l as B fragments for the linear waves, and this half's (and lane half's) part of the LayerNorm2 sums
                const float inv_l = 1.f / l_tot;
                const uint32_t a3w = opaque(a3w0);
                float su = 0.f, sq = 0.f;
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        bf16x8 fr;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const float x = o[d][8 * s2 + jj] * inv_l;
                            su += x; sq = __builtin_fmaf(x, x, sq);
                            fr[jj] = (bf16_t)x;
                        }
                        lds_wr<bf16x8>(a3w + (d * 2 + s2) * 1024, fr);
                    }
                lds_wr<f32x2_t>(opaque(statw0), (f32x2_t){su, sq});
            }
            XP_STAMP(7);
            XP_BARRIER();                                                // Y: o^ of track j is published
            XP_STAMP(8);
        }
    } else {
        // ================================================================================================ linear waves
        // W'' rows 64 w4 + 32 rt + r as A fragments.  The B fragments are accumulator registers of the attention (a lane holds rows
        // {0-3, 8-11} + 4 hh of every 16-row group), so the K index of the product is permuted accordingly:
        // slot j of fragment f <-> column 16 f + 8 (j >> 2) + 4 hh + (j & 3)
        bf16x8 wf[2][16];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const bf16_t* wp = (const bf16_t*)(wsc + WS_W2) + (64 * w4 + 32 * rt + r) * XD + 4 * hh;
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                const bf16x4 lo = *(const bf16x4*)(wp + f * 16), hi = *(const bf16x4*)(wp + f * 16 + 8);
                wf[rt][f] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
        // g3 * v of video (v2, r) at this wave's output rows, bf16 in accumulator order: fragment (v2, rt, s2) of this wave
#pragma unroll
        for (int v2 = 0; v2 < 2; ++v2) {
            const int64_t nn = n0 + v2 * 32 + r;
            const float* gp = a.ws + (nn < a.Nv ? nn : a.Nv - 1) * XD + 64 * w4 + 4 * hh;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const f32x4 x0 = *(const f32x4*)(gp + 32 * rt + 16 * s2), x1 = *(const f32x4*)(gp + 32 * rt + 16 * s2 + 8);
                    bf16x8 fr;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) { fr[jj] = (bf16_t)x0[jj]; fr[4 + jj] = (bf16_t)x1[jj]; }
                    *(bf16x8*)(lds + POFF_GV + ((((w4 * 2 + v2) * 2 + rt) * 2 + s2) * 64 + lane) * 16) = fr;
                }
        }
        // waves 4 / 5 finish the videos of tile w4 (lanes of the lower half): their per-video constants
        const int64_t my_n = n0 + (w4 & 1) * 32 + r;
        const int64_t nc = my_n < a.Nv ? my_n : a.Nv - 1;
        const float* pvp = a.ws + a.Nv * XD + nc * 2;  // sum g v, sum b v
        const float p0 = pvp[0], pb = pvp[1], c0 = wsc[0], e0 = wsc[1], f0 = wsc[2];
        const uint32_t vec_w0 = lbase + POFF_VEC + 16 * hh + 4 * 64 * w4;                  // rows 64 w4 + 8 g4 + 4 hh ..
        const uint32_t a3r0 = lbase + POFF_A3 + 16 * lane;                                // fragment (v2, f) at + (v2*16 + f) * 1024
        const uint32_t gv0 = lbase + POFF_GV + w4 * 8 * 1024 + 16 * lane;                 // fragment (v2, rt, s2) at + ((v2*2 + rt)*2 + s2) * 1024
        const uint32_t partw0 = lbase + POFF_PART + ((w4 * 2) * 32 + r) * 24;              // [w4][v2][r][6]: v2 at + v2 * 768
        const uint32_t partr0 = lbase + POFF_PART + ((w4 & 1) * 32 + r) * 24;             // partial of wave w' at + w' * 1536
        const uint32_t statr0 = lbase + POFF_STAT + r * 8;                                // (v2, half, lane half) at + ((v2*2 + half)*2 + hh') * 256

        __builtin_amdgcn_s_waitcnt(0x0070);
        XP_BARRIER();                                                    // B0
        int s_eff, first, nfull;
        track_info(0, s_eff, first, nfull);
        XP_BARRIER();                                                    // B1

        for (int j = 0; j <= T; ++j) {
            int n = 1;                                                   // tile steps of the attention waves in this iteration
            if (j < T) {
                n = tiles_of(s_eff);
                if (j + 1 < T) track_info(j + 1, s_eff, first, nfull);
            }
            XP_STAMP(0);
            // Per track (j - 1) this wave has four work units: per video tile the product (reads o^), then the sums (read only registers and
            // constants).  The attention waves' barriers of this iteration -- n - 1 tile barriers, then X ("o^ may be overwritten") and Y
            // ("o^ of track j is published") -- fall between them where the attention waves are about to arrive:
            //   n <= 2 tiles:  P0 S0 [T] P1 | X | S1 | Y      the second sums run beside the attention waves' publishing of o^ (which this
            //                                                 wave used to sit out); with one or two tiles this wave is the longer chain
            //   n >= 3 tiles:  P0 S0 T P1 T S1 [T ..] | X | Y the attention waves are the longer chain
            // (bit 2 of the schedule = a barrier behind the first sums, bit 3 = behind the second product).  LayerNorm3 / cosine of a track are
            // finished behind Y from the partial sums of BOTH video tiles; the partial sums alternate between two buffers, so that the next
            // track's first sums cannot overtake that finish.  (A barrier inside the sums, between the row tiles, would fit three tiles
            // better still, but the compiler answers it with 164 bytes of spills.)
            const bool late = n <= PLATE_N;
            const unsigned bar_after = n >= 3 ? 0x0cu : n == 2 ? 0x04u : 0u;
            const uint32_t par_off = (uint32_t)(j & 1) * (uint32_t)PPART_B;
            const uint32_t vec_w = opaque(vec_w0), a3r = opaque(a3r0), gvr = opaque(gv0), partw = opaque(partw0 + par_off), statr = opaque(statr0);
            if (j == 0) for (int t = 1; t < n; ++t) XP_BARRIER();           // (nothing to multiply yet)
#pragma unroll
            for (int v2 = 0; v2 < 2; ++v2) {
                if (j >= 1) {
                    // ---- this wave's 64 rows of W'' o^ for the 32 videos of tile v2 (track j - 1)
                    f32x16 acc[2];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[rt][e] = 0.f;
                    {
                        constexpr int NF = (DBG & 4) ? 1 : 16;
                        bf16x8 bfr[NF];
#pragma unroll
                        for (int f = 0; f < NF; ++f) bfr[f] = lds_rd<bf16x8>(a3r + (v2 * 16 + f) * 1024);
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][f], bfr[f], acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][f], bfr[f], acc[1], 0, 0, 0);
                        }
                        if (NF == 16) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                            for (int i = 0; i < 13; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                        }
                    }
                    // LayerNorm2 of video (v2, r): the four partial sums (read before X: the attention waves rewrite them while publishing)
                    f32x2_t st = lds_rd<f32x2_t>(statr + (v2 * 4) * 256);
#pragma unroll
                    for (int q = 1; q < 4; ++q) st += lds_rd<f32x2_t>(statr + (v2 * 4 + q) * 256);
                    const float mean2 = st[0] * (1.f / XD);
                    const float var2 = fmaxf(st[1] * (1.f / XD) - mean2 * mean2, 0.f);
                    const float k1 = __builtin_amdgcn_rsqf(var2 + a.eps), k2 = -mean2 * k1;
                    XP_STAMP(1 + 3 * v2);
                    if (v2 == 1) {
                        if ((bar_after >> 3) & 1u) XP_BARRIER();
                        if (late) { XP_STAMP(7); XP_BARRIER(); XP_STAMP(8); }   // X: o^ of track j - 1 consumed
                    }
                    // the six sums of LayerNorm3 + cosine (round 1's rule) restricted to these rows (and this lane half's rows)
                    float S1 = 0.f, S2 = 0.f, P1 = 0.f, C2 = 0.f, C1 = 0.f, E1 = 0.f;
                    if (DBG & 2) { S1 = acc[0][0] + acc[1][5] + k1; S2 = k2 + acc[0][9]; }
                    else {
                        // eight groups of four rows; the constants of group i + 1 are requested before group i is summed (nobody else
                        // in this wave hides the LDS latency), two elements per instruction
                        f32x2_t s1 = {0.f, 0.f}, s2 = {0.f, 0.f}, p1 = {0.f, 0.f}, c2 = {0.f, 0.f}, c1 = {0.f, 0.f}, e1 = {0.f, 0.f};
                        const f32x2_t k1v = {k1, k1}, k2v = {k2, k2};
                        f32x4 av, bv, g2, gb; bf16x4 gv4;
                        auto load_consts = [&](int i) __attribute__((always_inline)) {
                            const int rt = i >> 2, g4 = i & 3;
                            const uint32_t ro = (32 * rt + 8 * g4) * 4;
                            av = lds_rd<f32x4>(vec_w + ro); bv = lds_rd<f32x4>(vec_w + XD * 4 + ro);
                            g2 = lds_rd<f32x4>(vec_w + 2 * XD * 4 + ro); gb = lds_rd<f32x4>(vec_w + 3 * XD * 4 + ro);
                            gv4 = lds_rd<bf16x4>(gvr + ((v2 * 2 + rt) * 2 + (g4 >> 1)) * 1024 + (g4 & 1) * 8);
                        };
                        load_consts(0);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const f32x4 av_ = av, bv_ = bv, g2_ = g2, gb_ = gb; const bf16x4 gv_ = gv4;
                            if (i + 1 < 8) load_consts(i + 1);
                            const int rt = i >> 2, g4 = i & 3;
#pragma unroll
                            for (int jj = 0; jj < 4; jj += 2) {
                                const f32x2_t a2 = {acc[rt][4 * g4 + jj], acc[rt][4 * g4 + jj + 1]};
                                const f32x2_t yv = a2 * k1v + ((f32x2_t){bv_[jj], bv_[jj + 1]} * k2v + (f32x2_t){av_[jj], av_[jj + 1]});
                                const f32x2_t yy = yv * yv;
                                const f32x2_t gg = {g2_[jj], g2_[jj + 1]};
                                s1 += yv; s2 += yy;
                                p1 += yv * (f32x2_t){(float)gv_[jj], (float)gv_[jj + 1]};
                                c2 += yy * gg; c1 += yv * gg;
                                e1 += yv * (f32x2_t){gb_[jj], gb_[jj + 1]};
                            }
                        }
                        S1 = s1[0] + s1[1]; S2 = s2[0] + s2[1]; P1 = p1[0] + p1[1]; C2 = c2[0] + c2[1]; C1 = c1[0] + c1[1]; E1 = e1[0] + e1[1];
                    }
                    S1 += other_half(S1); S2 += other_half(S2); P1 += other_half(P1); C2 += other_half(C2); C1 += other_half(C1); E1 += other_half(E1);
                    if (hh == 0) {
                        lds_wr<f32x2_t>(partw + v2 * 768, (f32x2_t){S1, S2}); lds_wr<f32x2_t>(partw + v2 * 768 + 8, (f32x2_t){P1, C2});
                        lds_wr<f32x2_t>(partw + v2 * 768 + 16, (f32x2_t){C1, E1});
                    }
                }
                XP_STAMP(2 + 3 * v2);
                if (j >= 1 && v2 == 0 && ((bar_after >> 2) & 1u)) XP_BARRIER();
                XP_STAMP(3 + 3 * v2);
            }
            if (j >= 1) for (int t = 3; t < n; ++t) XP_BARRIER();
            if (!(j >= 1 && late)) { XP_STAMP(7); XP_BARRIER(); XP_STAMP(8); }   // X: o^ of track j - 1 consumed
            XP_STAMP(9);
            XP_BARRIER();                                                // Y: every wave's partial sums of track j - 1 are in LDS (and o^ of track j)
            XP_STAMP(10);
            if (j >= 1 && w4 < 2 && hh == 0) {
                // ---- the four partial sums (one per row block = linear wave) of video (w4, r): LayerNorm3 + cosine
                const uint32_t partr = opaque(partr0 + par_off);
                float s1 = 0.f, s2 = 0.f, p1 = 0.f, c2 = 0.f, c1 = 0.f, e1 = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2_t x0 = lds_rd<f32x2_t>(partr + q * 1536), x1 = lds_rd<f32x2_t>(partr + q * 1536 + 8), x2 = lds_rd<f32x2_t>(partr + q * 1536 + 16);
                    s1 += x0[0]; s2 += x0[1]; p1 += x1[0]; c2 += x1[1]; c1 += x2[0]; e1 += x2[1];
                }
                const float mu = s1 * (1.f / XD);
                const float var = fmaxf(s2 * (1.f / XD) - mu * mu, 0.f);
                const float rs = __builtin_amdgcn_rsqf(var + a.eps);
                const float dot = rs * (p1 - mu * p0) + pb;
                const float zz = rs * rs * (c2 - 2.f * mu * c1 + mu * mu * c0) + 2.f * rs * (e1 - mu * e0) + f0;
                float out = dot * __builtin_amdgcn_rsqf(zz);
                // (a track without a valid segment: the attention waves produced 0 / 0 = NaN already, like the reference's softmax over -inf)
                if (!(DBG & 1) && my_n < a.Nv) a.sims[my_n * a.ld_sims + (m_begin + j - 1)] = out;
            }
        }
    }
}

// W'' = (W + I) diag(g2) (bf16), Av = W b2 + b + b2, Bv = W g2 + g2: one wave per output row (see the persistent kernel)
__global__ __launch_bounds__(XT) void xpool_fold_kernel(const bf16_t* W, int64_t ldw, const float* bl, const float* g2, const float* b2, float* wsc) {
    const int lane = threadIdx.x & 63;
    const int d = blockIdx.x * (XT / 64) + (threadIdx.x >> 6);
    const bf16x4 w4 = *(const bf16x4*)(W + (int64_t)d * ldw + lane * 4);
    const f32x4 g = *(const f32x4*)(g2 + lane * 4), b = *(const f32x4*)(b2 + lane * 4);
    float sa = 0.f, sb = 0.f;
    bf16x4 o4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float w = (float)w4[j];
        sa += w * b[j]; sb += w * g[j];
        o4[j] = (bf16_t)((w + (lane * 4 + j == d ? 1.f : 0.f)) * g[j]);
    }
    *(bf16x4*)((bf16_t*)(wsc + WS_W2) + d * XD + lane * 4) = o4;
    sa = wave_sum(sa); sb = wave_sum(sb);
    if (lane == 0) { wsc[WS_AV + d] = sa + bl[d] + b2[d]; wsc[WS_BV + d] = sb + g2[d]; }
}

// per track: (last valid segment + 1, first valid segment, number of leading valid segments, 0)
__global__ __launch_bounds__(XT) void xpool_track_info_kernel(const float* key_mask, int64_t S, int64_t Nm, int* info) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * (XT / 64) + (threadIdx.x >> 6);
    if (m >= Nm) return;
    int last = -1, first = 0x7fffffff, hole = 0x7fffffff;
    for (int j = lane; j < (int)S; j += 64) {
        const bool v = key_mask[m * S + j] != 0.f;
        if (v) { last = j; first = min(first, j); } else hole = min(hole, j);
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        last = max(last, __shfl_xor(last, o2)); first = min(first, __shfl_xor(first, o2)); hole = min(hole, __shfl_xor(hole, o2));
    }
    if (lane == 0) {
        info[4 * m] = last + 1; info[4 * m + 1] = last < 0 ? 0 : first; info[4 * m + 2] = hole == 0x7fffffff ? (int)S : hole; info[4 * m + 3] = 0;
    }
}


// per video: ws[n, :] = g3 * vn[n, :], then (sum g3 vn, sum b3 vn); per model: sum g3^2, sum g3 b3, sum b3^2.  One wave per video.
__global__ __launch_bounds__(XT) void xpool_prep_kernel(const float* vn, int64_t ldvn, const float* g3, const float* b3, float* ws, int64_t Nv) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * (XT / 64) + (threadIdx.x >> 6);
    const f32x4 g = *(const f32x4*)(g3 + lane * 4), b = *(const f32x4*)(b3 + lane * 4);
    if (n < Nv) {
        const f32x4 v = *(const f32x4*)(vn + n * ldvn + lane * 4);
        f32x4 gv;
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { gv[j] = g[j] * v[j]; sg += gv[j]; sb += b[j] * v[j]; }
        *(f32x4*)(ws + n * XD + lane * 4) = gv;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { sg += __shfl_xor(sg, o2); sb += __shfl_xor(sb, o2); }
        if (lane == 0) { ws[Nv * XD + n * 2] = sg; ws[Nv * XD + n * 2 + 1] = sb; }
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        float c0 = 0.f, e0 = 0.f, f0 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { c0 += g[j] * g[j]; e0 += g[j] * b[j]; f0 += b[j] * b[j]; }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { c0 += __shfl_xor(c0, o2); e0 += __shfl_xor(e0, o2); f0 += __shfl_xor(f0, o2); }
        if (lane == 0) { float* cs = ws + Nv * (XD + 2); cs[0] = c0; cs[1] = e0; cs[2] = f0; }
    }
}

}  // namespace

extern "C" int made_xpool_fused(const MadeXpoolFusedArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_xpool_fused: null args");
    const MadeXpoolFusedArgs& a = *args;
    MADE_REQUIRE(a.Q && a.K && a.U && a.ln2_g && a.ln2_b && a.Wl && a.bl && a.ln3_g && a.ln3_b && a.vn && a.sims && a.ws,
                 "made_xpool_fused: null pointer");
    MADE_REQUIRE(a.Nv >= 0 && a.Nm >= 0 && a.S > 0, "made_xpool_fused: bad dims");
    MADE_UNSUPPORTED(a.D == XD, "made_xpool_fused: D=%lld (built for %d)", (long long)a.D, XD);
    MADE_UNSUPPORTED(a.S <= XS_MAX, "made_xpool_fused: S=%lld segments per track (at most %d)", (long long)a.S, XS_MAX);
    MADE_UNSUPPORTED(a.Nm <= 65535, "made_xpool_fused: more than 65535 tracks per call (chunk them)");
    MADE_UNSUPPORTED(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldu % 8 == 0 && a.k_bs % 8 == 0 && a.u_bs % 8 == 0 && a.ldw % 8 == 0 &&
                     a.ldvn % 4 == 0 && ((uintptr_t)a.ws % 16) == 0 && ((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.K % 16) == 0 && ((uintptr_t)a.U % 16) == 0 &&
                     ((uintptr_t)a.Wl % 16) == 0 && ((uintptr_t)a.vn % 16) == 0,
                     "made_xpool_fused: pointers / strides must keep 16-byte alignment");
    if (a.Nv == 0 || a.Nm == 0) return MADE_OK;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)xpool_fused_persist_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
        if (e != hipSuccess) {
            made_set_error("made_xpool_fused: cannot reserve %d bytes of LDS: %s", PLDS, hipGetErrorString(e));
            return MADE_ERR_HIP;
        }
        attr_done = true;
    }
    if (a.prepare_ws)
        hipLaunchKernelGGL(xpool_prep_kernel, dim3((unsigned)((a.Nv + 3) / 4)), dim3(XT), 0, (hipStream_t)stream, a.vn, a.ldvn, a.ln3_g,
                           a.ln3_b, a.ws, a.Nv);
    float* wsc = a.ws + a.Nv * (XD + 2);
    if (a.prepare_ws)
        hipLaunchKernelGGL(xpool_fold_kernel, dim3(XD / 4), dim3(XT), 0, (hipStream_t)stream, (const bf16_t*)a.Wl, a.ldw, a.bl, a.ln2_g, a.ln2_b, wsc);
    int* info = nullptr;
    if (a.key_mask) {
        info = (int*)(wsc + WS_INFO);
        hipLaunchKernelGGL(xpool_track_info_kernel, dim3((unsigned)((a.Nm + 3) / 4)), dim3(XT), 0, (hipStream_t)stream, a.key_mask, a.S, a.Nm, info);
    }
    // chunks of tracks per video tile: enough workgroups to fill the chip a few times over, as few partial rounds as possible
    // (a workgroup's prologue -- weight, Q and cosine fragments -- costs about as much as three tracks)
    const int64_t nvt = (a.Nv + PQ - 1) / PQ;
    int64_t chunks = a.Nm;
    if (nvt * a.Nm > 2048) {
        const int64_t lo = (1024 + nvt - 1) / nvt, hi = lo + 24 < a.Nm ? lo + 24 : a.Nm;
        double best = 1e30;
        for (int64_t cch = lo; cch <= hi; ++cch) {
            const int64_t per = (a.Nm + cch - 1) / cch, rounds = (nvt * ((a.Nm + per - 1) / per) + 255) / 256;
            const double cost = (double)rounds * (double)(per + 3);
            if (cost < best) { best = cost; chunks = (a.Nm + per - 1) / per; }
        }
    }
    int per = (int)((a.Nm + chunks - 1) / chunks);
    if (per > PMAX_TRACKS) per = PMAX_TRACKS;                 // (the kernel's track table)
    dim3 grid((unsigned)nvt, (unsigned)((a.Nm + per - 1) / per)), block(PT);
    // MADE_XPOOL_DBG=33: the phase-stamp build of the kernel (tools/xpool_stamps.py): no similarity is written, workgroup (0, 0)
    // writes cycle stamps into the sims buffer instead
    static const bool stamps = made_variant_env("MADE_XPOOL_DBG") && atoi(made_variant_env("MADE_XPOOL_DBG")) == 33;
    if (stamps) {
        hipFuncSetAttribute((const void*)xpool_fused_persist_kernel<33>, hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
        hipLaunchKernelGGL(xpool_fused_persist_kernel<33>, grid, block, PLDS, (hipStream_t)stream, a, (const int*)info, per);
    } else {
        hipLaunchKernelGGL(xpool_fused_persist_kernel<0>, grid, block, PLDS, (hipStream_t)stream, a, (const int*)info, per);
    }
    return made_check_launch("made_xpool_fused");
}
