// made_attention_bwd, single-pass form (bf16, head dim 64): ONE workgroup per (batch, head) computes dQ, dK and dV from one
// recomputation of the probabilities -- the split form (attention_bwd.hip: a dQ kernel and a dK / dV kernel) does the per-score
// arithmetic twice and was bound by it (19 - 20 vector instructions per MFMA, MFMA pipe 8 - 12 % busy).
//
// Geometry.  Eight waves.  The sample's key tiles (32 keys) that hold a valid key and its query tiles (32 queries) that hold a computed
// query are listed first; tiles of padding are never walked.  Key tiles are taken eight at a time (a key block: one tile per wave), and
// for every key block the workgroup sweeps the query tiles:
//   * key on the lane: S = Q K^T and dP = dO V^T (A rows from the LDS images of the query tile, B = the wave's K / V fragments, held in
//     registers for the whole key block), so the accumulators of Pd and dS are directly the B operands of dV^T += dO^T Pd and
//     dK^T += Q^T dS (A through the transposing LDS read) -- dK^T / dV^T stay in the wave's registers over the sweep;
//   * dQ is summed over the keys, i.e. across waves: every wave writes its dS tile (bf16, transposed: [key][query]) into an exchange
//     image, and behind a barrier each wave of a group of four computes 16 (head dim) x 32 (query) of dQ^T = K^T dS^T over the group's
//     128 keys with v_mfma_f32_16x16x32_bf16 (both operands through the transposing read), accumulating into an f32 image of dQ that
//     lives in LDS in accumulator order (16-byte reads and writes) until every key block has been swept: no atomics, no second
//     recomputation, dQ leaves the chip once.
//   * PING-PONG: the step of a wave is an M phase (matrix products and LDS reads: dV / dK and dQ of the previous query tile, S / dP of
//     this one) and a V phase (the per-score arithmetic of this tile), a barrier behind each.  Waves 0..3 and waves 4..7 (the two waves
//     of every SIMD) run half a step apart, so that a SIMD always has one wave on the matrix pipe and one on the vector pipe; with all
//     eight in the same phase (the first form of this kernel) a step took 5 000 cycles, every phase's latency in line.
// (Measured and dropped, round 6: the chunks of a pair as SEPARATE workgroups handing their dK / dV rows on through sc1 stores and a counter
// per key block -- a long sample no longer sets the launch's duration, but every workgroup pays the prologue and every key block its round
// trips again: 196 against 178 us ragged, 600 against 380 us dense.  docs/EXPERIMENTS.md.)
// The f32 dQ image bounds the queries per sweep (8 KB per query tile); longer samples are swept in chunks of query tiles, and the
// dK / dV of the second and later chunks are added to the rows the first one stored (bf16: one extra rounding, only on those samples).
//
// Per score: t = S * (scale * log2 e) - lse * log2 e, p = exp2(t), and with dropout (keep mask m, 1 / (1 - p_drop) = c):
//   Pd = m ? p c : 0,   dS / scale = p c * (m ? dP - delta / c : -delta / c)      (dP's accumulator starts at -delta / c)
// = 7 vector instructions besides the two conversions.  The keep decisions come from the forward's bit cache, whose layout
// (MadeAttnArgs.keep_bits) was chosen for this kernel: the 32 words of a (key tile, query tile) pair are 16 consecutive 64-bit lane
// masks -- register e of a 32 x 32 accumulator holds query row (e & 3) + 8 (e >> 2) + 4 hh at key column lane & 31, so its keep mask
// over the 64 lanes is {word of query row(e, 0), word of query row(e, 1)} -- fetched with scalar loads and used as the condition of
// v_cndmask_b32 as they stand.  `scale` is applied once per dK row and once per dQ row at the end.
#include "common.h"

namespace {

constexpr int FT = 512;                  // threads per workgroup: eight waves of at most 256 registers (with four waves of 512 hipcc parks the
                                         // accumulators in AGPRs and moves them every step: 780 v_accvgpr_* per step, measured)
constexpr int FHD = 64;
constexpr int FP = FHD * 2 + 16;         // pitch of the query-tile images (Q, dO): row reads of 16 bytes conflict-free
constexpr int KP = FHD * 2;              // pitch of the key-block image of K: read only through the transposing read; the four 32-byte
                                         // granules of row r sit at granule ^ (((r >> 1) & 1) | ((r >> 3) & 1) << 1): conflict-free
constexpr int DSP = 64;                  // pitch of a dS^T exchange tile: 32 queries x 2 bytes, 16-query halves swapped on rows with bit 3 set
constexpr int DQ_TILE_BYTES = 32 * FHD * 4;
constexpr int IMG_BUF = 32 * FP;         // one image of a query tile; Q and dO images are double-buffered
constexpr int NIMG = 2;
constexpr int DSX_BUF = 8 * 32 * DSP;    // one set of exchange tiles (double-buffered)
constexpr float LOG2E = 1.4426950408889634f;
#ifndef FUSED_STAMPS             // diagnostic build: cycles per section of a workgroup, written behind the B * H * Lq floats of `delta` (16 words per pair)
#define FUSED_STAMPS 0
#endif
#if FUSED_STAMPS
#define STAMP(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#ifndef FUSED_SKIP               // elimination builds (tools/attn_bwd_variants.sh): 1 S / dP products, 2 per-score arithmetic, 4 dV / dK products, 8 dQ phase
#define FUSED_SKIP 0
#endif

__device__ __forceinline__ float sel_m(float if0, float if1, uint64_t m) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(m));
    return r;
}
__device__ __forceinline__ float sel0_m(float if1, uint64_t m) {
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(if1), "s"(m));
    return r;
}

// transposing read of a 4-row x 16-column block of 16-bit elements: the 16 lanes of a group address row (i >> 2), columns 4 (i & 3) ..;
// lane i receives column i of the four rows
__device__ __forceinline__ bf16x4 tr4(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}
// A operand X^T of a 32 x 32 x 16 product out of an image stored [row][col] (see attention_bwd.hip tr_frag)
__device__ __forceinline__ bf16x8 tr_frag32(const unsigned char* tile, int P, int kb16, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const unsigned char* p = tile + (kb16 + 4 * (g >> 1) + (i >> 2)) * P + (c0 + (g & 1) * 16 + 4 * (i & 3)) * 2;
    bf16x4 lo = tr4(p), hi = tr4(p + 8 * P);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct FusedLayout {                     // byte offsets into the dynamic LDS block
    int dq, imq, imdo, imk, dsx, nl, nd, live, qflag, kflag, qlist, klist, misc, total;
};
__host__ __device__ inline FusedLayout fused_layout(int qc, int lqp, int lkp) {
    FusedLayout L;
    int o = 0;
    L.dq = o;    o += qc * DQ_TILE_BYTES;
    L.imq = o;   o += NIMG * IMG_BUF;
    L.imdo = o;  o += NIMG * IMG_BUF;
    L.imk = o;   o += 256 * KP;
    L.dsx = o;   o += 2 * DSX_BUF;
    L.nl = o;    o += (lqp + 32) * 4;            // (+ 32: the all-dead tile that pads a sweep to a multiple of three steps)
    L.nd = o;    o += (lqp + 32) * 4;
    L.live = o;  o += lqp + 32;
    L.qflag = o; o += (lqp / 32 + 3) / 4 * 4;
    L.kflag = o; o += (lkp / 32 + 3) / 4 * 4;
    L.qlist = o; o += (lqp / 32) * 2 + 2; o = (o + 3) / 4 * 4;
    L.klist = o; o += (lkp / 32) * 2 + 2; o = (o + 15) / 16 * 16;
    L.misc = o;  o += 32;
    L.total = o;
    return L;
}

// DROP: dropout on the attention weights, its decisions read from the forward's bit cache (BITS: always with DROP -- re-drawing the mask per
// score costs registers this kernel does not have; without the cache the caller uses the split kernels)
template <bool DROP, bool BITS>
__global__ __launch_bounds__(FT, 2) void attn_bwd_fused_kernel(const MadeAttnBwdArgs a, const uint32_t* __restrict__ kbits, const int qc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int Lq = (int)a.Lq, Lk = (int)a.Lk;
    const int nqt_all = (Lq + 31) / 32, nkt_all = (Lk + 31) / 32;           // (at most 64 each: the launcher)
    const int lqp = nqt_all * 32, lkp = nkt_all * 32;
    const FusedLayout L = fused_layout(qc, lqp, lkp);
    unsigned char* dq_acc = lds + L.dq;
    unsigned char* img_q = lds + L.imq;
    unsigned char* img_do = lds + L.imdo;
    unsigned char* img_k = lds + L.imk;
    unsigned char* ds_x = lds + L.dsx;
    float* nl = (float*)(lds + L.nl);
    float* nd = (float*)(lds + L.nd);
    unsigned char* live = lds + L.live;
    unsigned char* kflag = lds + L.kflag;
    unsigned short* qlist = (unsigned short*)(lds + L.qlist);
    unsigned short* klist = (unsigned short*)(lds + L.klist);

#if FUSED_STAMPS
    uint64_t st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t st_last = __builtin_amdgcn_s_memtime();
    const uint64_t st_first = st_last;
#endif
    const int64_t pair = blockIdx.x;                       // pairs of the longest sample first (batch_order)
    const int64_t h = pair % a.H;
    const int64_t b = a.batch_order ? (int64_t)__builtin_amdgcn_readfirstlane(a.batch_order[pair / a.H]) : pair / a.H;   // (uniform: scalar addressing below)
    const int64_t bh = b * a.H + h;

    const bf16_t* Qg = (const bf16_t*)a.Q + b * a.q_bs + h * FHD;
    const bf16_t* Gg = (const bf16_t*)a.dO + b * a.do_bs + h * FHD;
    const bf16_t* Og = (const bf16_t*)a.O + b * a.o_bs + h * FHD;
    const bf16_t* Kg = (const bf16_t*)a.K + b * a.k_bs + h * FHD;
    const bf16_t* Vg = (const bf16_t*)a.V + b * a.v_bs + h * FHD;
    const float* skipg = a.q_skip_mask ? a.q_skip_mask + b * a.Lq : nullptr;
    const float* maskg = a.key_mask ? a.key_mask + b * a.Lk : nullptr;
    const float dsc = DROP ? 1.f / (1.f - a.drop.p) : 1.f;
    const float inv_dsc = DROP ? (1.f - a.drop.p) : 1.f;

    // ---- per-query scalars: eight lanes per query row (16 bytes of O and of dO each), delta = dO . O of the head summed over the eight.
    // A global round trip costs several thousand cycles here (one workgroup per CU, nothing else to run meanwhile): every row group's
    // loads are requested before the first one is used (up to 64 x 8 = 512 queries per pass of the four waves).
    {
        const int sub = lane >> 3, ch = lane & 7;
        constexpr int NG = 9;
        for (int j00 = 0; j00 < lqp; j00 += NG * 64) {
            bf16x8 o[NG], g[NG];
            float l[NG];
            bool ok[NG];
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int j = j00 + 64 * i + w * 8 + sub;
                const int jc = j < Lq ? j : Lq - 1;
                ok[i] = j < Lq && (skipg == nullptr || skipg[jc] != 0.f);
                o[i] = *(const bf16x8*)(Og + (int64_t)jc * a.ldo + ch * 8);
                g[i] = *(const bf16x8*)(Gg + (int64_t)jc * a.lddo + ch * 8);
                l[i] = a.lse[bh * a.Lq + jc];
            }
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int j = j00 + 64 * i + w * 8 + sub;
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d += (float)o[i][e] * (float)g[i][e];
                d += dpp_f32<0xB1, 0xF>(d, d);                          // quad_perm [1,0,3,2]
                d += dpp_f32<0x4E, 0xF>(d, d);                          // quad_perm [2,3,0,1]
                d += __shfl_xor(d, 4);
                if (ch == 0 && j < lqp) {
                    nl[j] = ok[i] ? -l[i] * LOG2E : -INFINITY;
                    nd[j] = ok[i] ? -d * inv_dsc : 0.f;
                    live[j] = ok[i] ? 1 : 0;
                    if (ok[i]) a.delta[bh * a.Lq + j] = d;
                }
            }
        }
        if (tid < 32) { nl[lqp + tid] = -INFINITY; nd[lqp + tid] = 0.f; live[lqp + tid] = 0; }
        for (int j = tid; j < lkp; j += FT) {
            const int jc = j < Lk ? j : Lk - 1;
            const bool ok = j < Lk && (maskg == nullptr || maskg[jc] != 0.f);
            const uint64_t bal = __ballot(ok);
            if (r == 0) kflag[j >> 5] = (hh ? (uint32_t)(bal >> 32) : (uint32_t)bal) != 0u;
        }
    }
    __syncthreads();
    STAMP(0);
    // tile lists: every wave derives the two 64-bit tile masks (lane t: tile t), wave 0 writes the lists
    uint64_t qmask, kmask;
    {
        bool qf = false, kf_ = false;
        if (lane < nqt_all) {
            const u32x4 a0 = *(const u32x4*)(live + lane * 32), a1 = *(const u32x4*)(live + lane * 32 + 16);
            qf = (a0[0] | a0[1] | a0[2] | a0[3] | a1[0] | a1[1] | a1[2] | a1[3]) != 0u;
        }
        if (lane < nkt_all) kf_ = kflag[lane] != 0;
        qmask = __ballot(qf); kmask = __ballot(kf_);
        if (w == 0) {
            if (qf) qlist[__popcll(qmask & ((1ull << lane) - 1ull))] = (unsigned short)lane;
            if (kf_) klist[__popcll(kmask & ((1ull << lane) - 1ull))] = (unsigned short)lane;
        }
    }
    const int nqt = __popcll(qmask), nkt = __popcll(kmask);
    __syncthreads();
    STAMP(1);

    // rows of tiles that are never walked: zeros (every row of dQ / dK / dV is defined)
    {
        const int row = tid >> 3, ch = tid & 7;            // 64 rows per pass: two tiles
        const bf16x8 z = {};
        uint64_t dead = (nkt > 0 ? ~qmask : ~0ull) & (nqt_all >= 64 ? ~0ull : ((1ull << nqt_all) - 1ull));
        while (dead) {
            const int t0 = __builtin_ctzll(dead); dead &= dead - 1;
            int t1 = -1;
            if (dead) { t1 = __builtin_ctzll(dead); dead &= dead - 1; }
            const int t = row < 32 ? t0 : t1;
            const int q = t * 32 + (row & 31);
            if (t >= 0 && q < Lq) *(bf16x8*)((bf16_t*)a.dQ + b * a.dq_bs + (int64_t)q * a.lddq + h * FHD + ch * 8) = z;
        }
        dead = (nqt > 0 ? ~kmask : ~0ull) & (nkt_all >= 64 ? ~0ull : ((1ull << nkt_all) - 1ull));
        while (dead) {
            const int t0 = __builtin_ctzll(dead); dead &= dead - 1;
            int t1 = -1;
            if (dead) { t1 = __builtin_ctzll(dead); dead &= dead - 1; }
            const int t = row < 32 ? t0 : t1;
            const int k = t * 32 + (row & 31);
            if (t >= 0 && k < Lk) {
                *(bf16x8*)((bf16_t*)a.dK + b * a.dk_bs + (int64_t)k * a.lddk + h * FHD + ch * 8) = z;
                *(bf16x8*)((bf16_t*)a.dV + b * a.dv_bs + (int64_t)k * a.lddv + h * FHD + ch * 8) = z;
            }
        }
    }
    if (nqt == 0 || nkt == 0) return;
    STAMP(2);

    // ---- constants of the sweep
    const float c2 = a.scale * LOG2E;
    const int nkb = (nkt + 7) >> 3;
    // staging role of this thread: tensor (Q / dO), row and 16-byte chunk of a query tile
    const int st_t = tid >> 8, st_row = (tid & 255) >> 3, st_ch = tid & 7;
    const bf16_t* st_base = (st_t ? Gg : Qg) + st_ch * 8;
    const int64_t st_ld = st_t ? a.lddo : a.ldq;
    unsigned char* st_dst = lds + (st_t ? L.imdo : L.imq) + st_row * FP + st_ch * 16;        // (+ buffer * IMG_BUF)
    // dQ role of this wave: head-dim slice hs (16 columns), query half qh; operands of this lane: K^T rows and dS^T rows, all 256 keys
    const int g16 = lane >> 4, i16 = lane & 15;
    const int qh = w & 1, hs = w >> 1;
    const int skx = ((i16 >> 3) & 1) | ((g16 & 1) << 1);                                    // K image swizzle of the rows this lane addresses
    const unsigned char* ka0 = img_k + (8 * g16 + (i16 >> 2)) * KP + ((hs ^ skx) * 32) + (i16 & 3) * 8;
    const unsigned char* sb0 = ds_x + (8 * g16 + (i16 >> 2)) * DSP + (((qh * 16 + 4 * (i16 & 3)) ^ ((g16 & 1) << 4)) * 2);

    // dQ^T piece [16 head-dim columns x 16 queries] of a tile += K^T dS^T over the block's 256 keys (exchange buffer xb)
    auto dq_phase = [&](int tile_in_chunk, int xb) __attribute__((always_inline)) {
        unsigned char* cp = dq_acc + (tile_in_chunk * 8 + w) * 1024 + lane * 16;
        const unsigned char* sb = sb0 + xb * DSX_BUF;
        f32x4 c = *(const f32x4*)cp;
#pragma unroll
        for (int j = 0; j < ((FUSED_SKIP & 8) ? 0 : 8); ++j) {
            const bf16x4 alo = tr4(ka0 + j * 32 * KP), ahi = tr4(ka0 + j * 32 * KP + 4 * KP);
            const bf16x4 blo = tr4(sb + j * 32 * DSP), bhi = tr4(sb + j * 32 * DSP + 4 * DSP);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7),
                                                        __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7), c, 0, 0, 0);
        }
        *(f32x4*)cp = c;
    };

    const int qc3 = qc - qc % 2;
    const int nchunks = (nqt + qc3 - 1) / qc3;
    const int chunk_len = (nqt + nchunks - 1) / nchunks;                   // balanced chunks of query tiles
    for (int c0 = 0; c0 < nqt; c0 += chunk_len) {
        const int nq_c = (nqt - c0) < chunk_len ? (nqt - c0) : chunk_len;
        const int nq_p = (nq_c + 1) / 2 * 2;                               // steps of a sweep: padded with an all-dead tile to a multiple of two
        auto tile_q0 = [&](int pos) __attribute__((always_inline)) {       // first query of the tile at position pos of the (padded) chunk
            const int t = (int)qlist[c0 + (pos < nq_c ? pos : 0)];
            return (pos < nq_c ? t : nqt_all) * 32;
        };
        // zero the dQ image of the chunk
        {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            for (int i = tid; i < nq_p * (DQ_TILE_BYTES / 16); i += FT) *(f32x4*)(dq_acc + i * 16) = z;
        }
        // The tiles of the (padded) chunk are walked round and round (once per key block); g counts the steps of the chunk, tile(g) sits in
        // image buffer g & 1.  A global round trip (4 000 - 6 000 cycles here) takes longer than a step, so two tiles are on their way
        // at any time: tile T travels in staging slot T & 1, requested at the top of step T - 3 and stored to LDS at the top of step
        // T - 1.  The step loop is unrolled twice (a sweep is a multiple of two steps), so that a slot is a fixed set of registers and
        // the compiler's own s_waitcnt vmcnt(N) before a slot's store lets the younger slot's load stay out.
        u32x4 sl0, sl1;
        auto ld1 = [&](u32x4& d_, int q0n) __attribute__((always_inline)) {
            const int q = q0n + st_row;
            d_ = *(const u32x4*)(st_base + (int64_t)(q < Lq ? q : Lq - 1) * st_ld);
        };
        int it_load = 0;                                                   // position in the chunk of the next tile to request
        auto next_q0 = [&]() __attribute__((always_inline)) {
            const int q0n = tile_q0(it_load);
            it_load = it_load + 1 < nq_p ? it_load + 1 : 0;
            return q0n;
        };
        {
            const int q00 = next_q0();
            ld1(sl0, q00);                                                 // tile 0
            ld1(sl1, next_q0());                                           // tiles 1, 2 -> slots 1, 0
            *(u32x4*)st_dst = keep_or_zero(sl0, live[q00 + st_row] != 0);   // tile 0 -> buffer 0 (visible behind the key block's first barrier)
            ld1(sl0, next_q0());
        }
        int it_n1 = 1;                                                     // position in the chunk of tile g + 1
        int q0_n1 = __builtin_amdgcn_readfirstlane(tile_q0(it_n1));
        bool live_n1 = live[q0_n1 + st_row] != 0;                          // this thread's row of tile g + 1 is computed
        int g = 0;
        // K / V fragments of the wave's key tile: requested one key block ahead (behind the sweep's last S / dP products, in front of the
        // block's dK / dV stores), so that their round trip runs under the epilogue instead of behind it
        bool wave_live, key_valid, has_masked;
        int kt, key, keyc;
        float bias_key;
        bf16x8 kf[4], vf[4];
        auto load_kv = [&](int kb_) __attribute__((always_inline)) {
            const int ki = kb_ * 8 + w;
            wave_live = ki < nkt;                                          // (wave-uniform)
            kt = __builtin_amdgcn_readfirstlane((int)klist[wave_live ? ki : nkt - 1]);
            key = kt * 32 + r;
            keyc = key < Lk ? key : Lk - 1;
            key_valid = wave_live && key < Lk && (maskg == nullptr || maskg[keyc] != 0.f);
            bias_key = key_valid ? 0.f : -INFINITY;
            has_masked = !__all(key_valid);
            const bf16_t* kp = Kg + (int64_t)keyc * a.ldk;
            const bf16_t* vp = Vg + (int64_t)keyc * a.ldv;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kf[ks] = keep_or_zero(*(const bf16x8*)(kp + ks * 16 + hh * 8), key_valid);
                vf[ks] = keep_or_zero(*(const bf16x8*)(vp + ks * 16 + hh * 8), key_valid);
            }
        };
        load_kv(0);
        for (int kb = 0; kb < nkb; ++kb) {
            // (the previous key block ended with a barrier: the K image and the exchange tiles are free)
            {
                const int row = w * 32 + r;
                const int sk = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) *(bf16x8*)(img_k + row * KP + ((ks ^ sk) * 32) + hh * 16) = kf[ks];
            }
            if (!wave_live) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int xb = 0; xb < 2; ++xb) {
                    *(f32x4*)(ds_x + xb * DSX_BUF + w * (32 * DSP) + lane * 16) = z;
                    *(f32x4*)(ds_x + xb * DSX_BUF + w * (32 * DSP) + 1024 + lane * 16) = z;
                }
            }
            f32x16 dk[2], dv[2];
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) { dk[d][e] = 0.f; dv[d][e] = 0.f; }
            int q0 = __builtin_amdgcn_readfirstlane((int)qlist[c0] * 32);
            // keep masks of the wave's (key tile, query tile): sixteen 64-lane masks, requested a step ahead (right behind their last use)
            uint64_t km[16];
            if constexpr (DROP && BITS) {
                const uint64_t* mp = (const uint64_t*)(kbits + ((bh * nkt_all + kt) * a.ld_bits + q0));
#pragma unroll
                for (int e = 0; e < 16; ++e) km[e] = mp[e];
            }
            __syncthreads();
            STAMP(3);
            int s = 0;

            auto step = [&](u32x4& slot_regs) __attribute__((always_inline)) {
                const int buf = g & 1;
                // tile g + 1 -> the other image buffer (read last during step g - 1); tile g + 4 -> the slot it leaves
                *(u32x4*)(st_dst + (buf ^ 1) * IMG_BUF) = keep_or_zero(slot_regs, live_n1);
                ld1(slot_regs, next_q0());
                const int q0_next = q0_n1;                                 // first query of tile g + 1
                it_n1 = it_n1 + 1 < nq_p ? it_n1 + 1 : 0;
                q0_n1 = __builtin_amdgcn_readfirstlane(tile_q0(it_n1));
                live_n1 = live[q0_n1 + st_row] != 0;
                const unsigned char* iq = img_q + buf * IMG_BUF;
                const unsigned char* ig = img_do + buf * IMG_BUF;
                if (wave_live) {
                    // ---- S = Q K^T, dP = dO V^T - delta / c
                    f32x16 sc, dp;
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const f32x4 n4 = *(const f32x4*)(nd + q0 + 8 * e4 + 4 * hh);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { sc[4 * e4 + j] = 0.f; dp[4 * e4 + j] = n4[j]; }
                    }
                    {
                        bf16x8 qa[4], ga[4];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            qa[ks] = *(const bf16x8*)(iq + r * FP + ks * 32 + hh * 16);
                            ga[ks] = *(const bf16x8*)(ig + r * FP + ks * 32 + hh * 16);
                        }
#pragma unroll
                        for (int ks = 0; ks < ((FUSED_SKIP & 1) ? 0 : 4); ++ks) {
                            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], kf[ks], sc, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[ks], vf[ks], dp, 0, 0, 0);
                        }
                    }
                    // the dQ^T piece of the previous step, under this step's products and arithmetic
                    if (s > 0) dq_phase(s - 1, buf ^ 1);
                    // ---- sc <- Pd, dp <- dS / scale  (the row constants are read again here rather than held over the products)
                    f32x4 ndq[4], nlq[4];
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        nlq[e4] = *(const f32x4*)(nl + q0 + 8 * e4 + 4 * hh);
                        ndq[e4] = *(const f32x4*)(nd + q0 + 8 * e4 + 4 * hh);
                    }
                    if (has_masked) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) sc[e] += bias_key;
                    }
                    if constexpr ((FUSED_SKIP & 2) != 0) {
                    } else if constexpr (DROP && BITS) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], c2, nlq[e >> 2][e & 3]));
                            const float pc = p * dsc;
                            sc[e] = sel0_m(pc, km[e]);
                            dp[e] = pc * sel_m(ndq[e >> 2][e & 3], dp[e], km[e]);
                        }
                        // the masks of the next step
                        const uint64_t* mp = (const uint64_t*)(kbits + ((bh * nkt_all + kt) * a.ld_bits + (q0_next < lqp ? q0_next : lqp - 32)));
#pragma unroll
                        for (int e = 0; e < 16; ++e) km[e] = mp[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], c2, nlq[e >> 2][e & 3]));
                            sc[e] = p;
                            dp[e] = p * dp[e];
                        }
                    }
                    // ---- dV^T += dO^T Pd,  dK^T += Q^T dS;  dS^T -> the exchange image
                    unsigned char* row = ds_x + buf * DSX_BUF + w * (32 * DSP) + r * DSP;
                    const int sw = ((r >> 3) & 1) << 4;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        bf16x8 pf, sf;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { pf[j] = (bf16_t)sc[8 * s2 + j]; sf[j] = (bf16_t)dp[8 * s2 + j]; }
#pragma unroll
                        for (int d = 0; d < ((FUSED_SKIP & 4) ? 0 : 2); ++d) {
                            dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag32(ig, FP, 16 * s2, d * 32, lane), pf, dv[d], 0, 0, 0);
                            dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag32(iq, FP, 16 * s2, d * 32, lane), sf, dk[d], 0, 0, 0);
                        }
                        // register quads 2 s2, 2 s2 + 1 of this lane's key row: queries 8 g + 4 hh .. + 3
                        *(bf16x4*)(row + (((16 * s2 + 4 * hh) ^ sw) * 2)) = __builtin_shufflevector(sf, sf, 0, 1, 2, 3);
                        *(bf16x4*)(row + (((16 * s2 + 8 + 4 * hh) ^ sw) * 2)) = __builtin_shufflevector(sf, sf, 4, 5, 6, 7);
                    }
                } else if (s > 0) {
                    dq_phase(s - 1, buf ^ 1);
                }
                q0 = q0_next;
                ++s; ++g;
                __syncthreads();           // dS^T (step g) and the images of tile g + 1 are in place; buffers of step g - 1 are free
                __builtin_amdgcn_sched_barrier(0);                         // (nothing of the next step is scheduled into this one: register pressure)
            };
            for (int s2 = 0; s2 < nq_p; s2 += 2) { step(sl1); step(sl0); }
            STAMP(4);
            const bool e_live = wave_live, e_valid = key_valid;
            const int e_key = key, e_keyc = keyc;
            if (kb + 1 < nkb) load_kv(kb + 1);
            dq_phase(nq_p - 1, (g - 1) & 1);

            // dK / dV rows of the wave's key tile: the two lane halves exchange register quads (v_permlane32_swap) so that a lane stores
            // 16 contiguous bytes: lane (r, 0) columns 16 gp .. + 7, lane (r, 1) columns 16 gp + 8 .. + 15 (per 32-column half)
            if (e_live) {
                bf16_t* kp = (bf16_t*)a.dK + b * a.dk_bs + (int64_t)e_keyc * a.lddk + h * FHD;
                bf16_t* vp = (bf16_t*)a.dV + b * a.dv_bs + (int64_t)e_keyc * a.lddv + h * FHD;
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    bf16_t* op = t2 ? vp : kp;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {                   // quads 2 gp (kept by the lower half) and 2 gp + 1 (kept by the upper half)
                            const f32x16& acc = t2 ? dv[d] : dk[d];
                            const float m = t2 ? 1.f : a.scale;
                            const int col = d * 32 + 16 * gp + 8 * hh;
                            float x0[4], x1[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) { x0[j] = acc[8 * gp + j] * m; x1[j] = acc[8 * gp + 4 + j] * m; }
                            if (c0 > 0) {
                                // a later chunk of query tiles: add to what the first one stored (this lane's own columns, before the exchange)
                                const bf16x4 o0 = *(const bf16x4*)(op + d * 32 + 16 * gp + 4 * hh), o1 = *(const bf16x4*)(op + d * 32 + 16 * gp + 8 + 4 * hh);
#pragma unroll
                                for (int j = 0; j < 4; ++j) { x0[j] += (float)o0[j]; x1[j] += (float)o1[j]; }
                            }
                            bf16x4 qa, qb;
#pragma unroll
                            for (int j = 0; j < 4; ++j) { qa[j] = (bf16_t)x0[j]; qb[j] = (bf16_t)x1[j]; }
                            const uint2 ua = __builtin_bit_cast(uint2, qa), ub = __builtin_bit_cast(uint2, qb);
                            const auto s0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
                            const u32x4 pk = {s0[0], s1[0], s0[1], s1[1]};
                            if (e_key < Lk) *(u32x4*)(op + col) = e_valid ? pk : u32x4{0u, 0u, 0u, 0u};
                        }
                }
            }
            __syncthreads();               // the last dQ^T pieces are in; the K image and the exchange tiles may be rewritten
            STAMP(5);
        }
        // dQ rows of the chunk: a tile's image is eight pieces (head-dim slice hs, query half qh) of 1 KB, wave w flushes piece w:
        // lane (g16, i16) holds head-dim columns 16 hs + 4 g16 .. + 3 of query 16 qh + i16
#pragma unroll 3
        for (int s = 0; s < nq_c; ++s) {
            const int qt = qlist[c0 + s];
            const int q = qt * 32 + qh * 16 + i16;
            const f32x4 c = *(const f32x4*)(dq_acc + (s * 8 + w) * 1024 + lane * 16);
            if (q < Lq) {
                const bool ok = live[q] != 0;
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(ok ? c[j] * a.scale : 0.f);
                *(bf16x4*)((bf16_t*)a.dQ + b * a.dq_bs + (int64_t)q * a.lddq + h * FHD + hs * 16 + 4 * g16) = o;
            }
        }
        __syncthreads();
        STAMP(6);
    }
#if FUSED_STAMPS
    if (tid == 0) {
        uint32_t* dst = (uint32_t*)(a.delta + a.B * a.H * a.Lq) + pair * 16;
        for (int i = 0; i < 8; ++i) dst[i] = (uint32_t)st_acc[i];
        dst[8] = (uint32_t)(__builtin_amdgcn_s_memtime() - st_first);
        dst[9] = nqt; dst[10] = nkt; dst[11] = nchunks;
        dst[12] = (uint32_t)(st_first >> 8);                // start time (256-cycle units)
        uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); dst[13] = xcc & 15;
    }
#endif
}

constexpr int FUSED_LDS_MAX = 160 * 1024;

}  // namespace

// MADE_OK after launching, a HIP error code, or -1000 when the single-pass form does not apply (the caller then uses the split kernels).
int made_attention_bwd_fused_try(const MadeAttnBwdArgs& a, hipStream_t st) {
    if (a.dtype != MADE_BF16 || a.hd != FHD) return -1000;
    if (a.Lq > 2048 || a.Lk > 2048) return -1000;
    const bool drop = a.drop.p > 0.f, bits = drop && a.keep_bits != nullptr;
    if (drop && !bits) return -1000;                       // (no bit cache: the split kernels re-draw the mask)
    const int lqp = (int)((a.Lq + 31) / 32) * 32, lkp = (int)((a.Lk + 31) / 32) * 32;
    const int fixed = fused_layout(0, lqp, lkp).total;
    int qc = (FUSED_LDS_MAX - fixed) / DQ_TILE_BYTES;
    if (qc > lqp / 32 + 1) qc = lqp / 32 + 1;              // (+ 1: a sweep is padded to an even number of tiles)
    if (qc < 4) return -1000;
    const size_t lds = (size_t)fused_layout(qc, lqp, lkp).total;
    void (*fn)(const MadeAttnBwdArgs, const uint32_t*, const int) = !drop ? attn_bwd_fused_kernel<false, false> : attn_bwd_fused_kernel<true, true>;
    static bool attr_done[2] = {false, false};
    const int vi = !drop ? 0 : 1;
    if (!attr_done[vi]) {
        if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_MAX) != hipSuccess) {
            (void)hipGetLastError();
            return -1000;
        }
        attr_done[vi] = true;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)(a.B * a.H)), dim3(FT), lds, st, a, a.keep_bits, qc);
    return made_check_launch("made_attention_bwd (single pass)");
}
