// made_xpool_inbatch: the in-batch X-Pool contraction -- softmax_s(scale q_n . k_{m,s}) U_m for every (video n, track m) of a batch of at most 64
// videos (reference modules/transformer.py:110-119 with the out projection hoisted onto the values).  gfx950, bf16, D = 256 / 512, S <= 512.
//
// The problem is 64 tracks x (K + U = 1 MB) of operands and 4 GFLOP: memory-bound, and small.  One workgroup per track (56.8 us in round 3)
// streams a track at ONE CU's rate; splitting the keys over four workgroups with f32 partial outputs and a merge launch (33.3 us) writes and re-reads
// 33 MB of partials.  Here the split is along the two operands instead, in two launches of one workgroup per CU each, and nothing larger than
// the probabilities (bf16, 4 MB) passes between them:
//   scores : workgroup (track m, slice of 128 segments).  The K slice (128 x D bf16 = 128 KB at D = 512) goes global -> LDS in ONE batch of
//            global_load_lds (every byte in flight at once: one memory round trip), Q fragments stream from L2; wave w owns the 32-segment tile
//            pair: S^T tile [32 x 32 videos], the tile's own reference per video (its maximum rounded up to an integer), p~ = exp2(s - reference)
//            in bf16 in ACCUMULATOR order, and the tile's (reference, sum) per video.  Q reaches the registers through LDS (coalesced).
//   pv     : workgroup (track m, 128 value columns).  The U column slice (S x 128 bf16 <= 128 KB) in one batch of global_load_lds; per video the
//            track's reference and denominator from the tiles' pairs; the tiles' references are INTEGERS (log2 units), so the factor between a
//            tile's and the track's is an exact power of two: it is applied to the bf16 p~ fragment itself (a saturating subtraction from the
//            exponent fields, four packed instructions per fragment) and O^T += U_tile^T p~_tile accumulates straight in the MFMA; p~ fragments
//            are 16-byte loads (accumulator order is the B-operand order of this product), U^T through ds_read_b64_tr_b16.
// Masked segments: their p~ is exactly 0 and their K / U rows are fetched from the track's first valid segment instead (the buffers behind masked
// rows are never written by the projection that makes K / U).  A track without a valid segment gives NaN rows, like the reference's softmax.
#include "common.h"

namespace {

constexpr int IB_T = 256;                 // threads: 4 waves
constexpr int IB_SEG = 128;               // segments per scores workgroup
constexpr int IB_COL = 128;               // value columns per pv workgroup
constexpr int IB_SMAX = 512;              // segments per track (the pv kernel's LDS image)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct InbatchArgs {
    const bf16_t* Q; int64_t ldq;
    const bf16_t* K; const bf16_t* U; int64_t k_bs, ldk, u_bs, ldu;
    const float* key_mask;
    void* out; int out_dtype; int64_t o_bs, ldo;
    int Nv, Nm, S, Tpad;                  // Tpad: 32-segment tiles per track in the workspace (a multiple of 4)
    float scale;
    bf16_t* wp;                           // [Nm][Tpad][2 video blocks][64 lanes][16] p~ in accumulator order
    float* wml;                           // [Nm][Tpad][64 videos][2] (max, sum) of a tile
};

__device__ __forceinline__ float ib_other_half(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned lo = sw[0], hi = sw[1];
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? lo : hi);
}

// 8 bf16 probabilities (>= 0) times 2^-k, exactly: k << 7 taken off every 16-bit word with saturation (a word whose exponent field is smaller
// than k becomes +0)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 ib_scale_pow2(bf16x8 p, uint32_t kk) {
    u32x4 w = __builtin_bit_cast(u32x4, p);
    const u16x2 d = __builtin_bit_cast(u16x2, kk);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t wi = w[i];
        w[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, wi), d));
    }
    return __builtin_bit_cast(bf16x8, w);
}

// the track's mask -> LDS flags (1 = valid), its first valid segment, one past its last, and whether the valid segments are one run (then a
// row's validity is arithmetic and the flags are not needed): every thread returns the same values.  The barrier is the raw one: a
// __syncthreads() would also wait for the operand loads the caller has in flight.
__device__ __forceinline__ void ib_scan_mask(const float* maskg, int S, float* flags, int* red, int& first, int& s_eff, bool& one_run) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int lo = 0x7fffffff, hi = -1, cnt = 0;
    if (maskg == nullptr) { first = 0; s_eff = S; one_run = true; return; }
    for (int j = tid; j < IB_SMAX; j += IB_T) {
        const bool v = j < S && maskg[j < S ? j : 0] != 0.f;
        flags[j] = v ? 1.f : 0.f;
        if (v) { lo = min(lo, j); hi = max(hi, j); ++cnt; }
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) { lo = min(lo, __shfl_xor(lo, o2)); hi = max(hi, __shfl_xor(hi, o2)); cnt += __shfl_xor(cnt, o2); }
    if (lane == 0) { red[3 * wave] = lo; red[3 * wave + 1] = hi; red[3 * wave + 2] = cnt; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    lo = min(min(red[0], red[3]), min(red[6], red[9]));
    hi = max(max(red[1], red[4]), max(red[7], red[10]));
    cnt = red[2] + red[5] + red[8] + red[11];
    first = hi < 0 ? 0 : lo;
    s_eff = hi + 1;
    one_run = hi < 0 || cnt == hi - lo + 1;
}

// ---------------------------------------------------------------------------------------------------------------- scores
template <int D>
__global__ __launch_bounds__(IB_T, 1) void xpool_inbatch_scores_kernel(const InbatchArgs a) {
    constexpr int ROWB = D * 2;                       // bytes of a K row
    constexpr int RPP = 1024 / ROWB;                  // rows per 1 KB piece (1 at D = 512, 2 at 256)
    constexpr int NKS = D / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* flags = (float*)(lds + IB_SEG * ROWB);     // [IB_SMAX]
    int* red = (int*)(flags + IB_SMAX);               // [12]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int vb = wave & 1, sh = wave >> 1;          // this wave: videos 32 vb .., segments 64 sh .. of the slice (two 32-segment tiles)
    const int nsl = a.Tpad / 4;
    const int m = (int)blockIdx.x / nsl, sl = (int)blockIdx.x % nsl;
    int first, s_eff; bool one_run;
    ib_scan_mask(a.key_mask ? a.key_mask + (int64_t)m * a.S : nullptr, a.S, flags, red, first, s_eff, one_run);
    if (sl * IB_SEG >= s_eff && !(s_eff == 0 && sl == 0)) return;          // nothing valid in this slice: the pv kernel never reads its tiles
    // ---- LDS image of the K slice: row rr at rr * ROWB, its 16-byte chunk c at position c ^ (rr & 31).  A row per lane straight from global
    // memory (the fragment layout) costs the texture path a separate 32-byte access per row and instruction -- 2.7 us of issue for Q alone -- so
    // Q takes the coalesced way too: its 64 rows pass through the LOWER half of the image (same swizzle) into registers while the upper half
    // of the K slice is already in flight; then the lower half of K follows, under the upper half's products.
    const unsigned char* Kb = (const unsigned char*)(a.K + (int64_t)m * a.k_bs);
    const uint32_t ldk_b = (uint32_t)a.ldk * 2u, ldq_b = (uint32_t)a.ldq * 2u;
    constexpr int HP = 64 * ROWB / 1024 / 4;          // 1 KB pieces of half an image (64 rows) per wave: 16 (D = 512) / 8 (D = 256)
    auto k_half_ = [&](int half, const bool arith) __attribute__((always_inline)) {   // (arith: a literal at every call -- no branch per piece)
#pragma unroll
        for (int i = 0; i < HP; ++i) {
            const int piece = half * 4 * HP + wave * HP + i;
            const int rr = piece * RPP + (RPP == 1 ? 0 : (lane >> 5));
            const int cl = RPP == 1 ? lane : (lane & 31);
            const int seg = sl * IB_SEG + rr;
            const bool v = arith ? (seg >= first && seg < s_eff) : (flags[seg] != 0.f);
            const uint32_t off = (uint32_t)(v ? seg : first) * ldk_b + (uint32_t)((cl ^ (rr & 31)) << 4);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(Kb + off), (lds_ptr_t)(lds + piece * 1024), 16, 0, 0);
        }
    };
    auto k_half = [&](int half) __attribute__((always_inline)) { if (one_run) k_half_(half, true); else k_half_(half, false); };
    k_half(1);
#pragma unroll
    for (int i = 0; i < HP; ++i) {                     // Q rows 0 .. 63 (rows past Nv: the last one) into the lower half
        const int piece = wave * HP + i;
        const int rr = piece * RPP + (RPP == 1 ? 0 : (lane >> 5));
        const int cl = RPP == 1 ? lane : (lane & 31);
        const uint32_t off = (uint32_t)(rr < a.Nv ? rr : a.Nv - 1) * ldq_b + (uint32_t)((cl ^ (rr & 31)) << 4);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)((const unsigned char*)a.Q + off), (lds_ptr_t)(lds + piece * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // Q and the upper half of K have landed
    // this wave's 32 videos as B-operand fragments (lane (n, hh) holds Q[n][16 ks + 8 hh ..]): the whole block in registers
    bf16x8 qf[NKS];
    {
        const uint32_t qbase = (uint32_t)((vb * 32 + r) * ROWB);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8*)(lds + qbase + ((uint32_t)(((2 * ks + hh) ^ r) & (ROWB / 16 - 1)) << 4));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(qf[0]), "+v"(qf[NKS - 1]) :: "memory");   // every wave holds its Q block: the lower half is free
    k_half(0);
    if (sh == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");    // (waves of the lower half: wait for it; the others multiply first)
    // ---- S^T of the wave's two tiles [32 segments x 32 videos] side by side (two independent accumulator chains)
    f32x16 sa, sb;
#pragma unroll
    for (int e = 0; e < 16; ++e) { sa[e] = 0.f; sb[e] = 0.f; }
    const uint32_t kbase = (uint32_t)((sh * 64 + r) * ROWB);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const uint32_t co = (uint32_t)(((2 * ks + hh) ^ r) & (ROWB / 16 - 1)) << 4;
        const bf16x8 ka = *(const bf16x8*)(lds + kbase + co), kb = *(const bf16x8*)(lds + kbase + 32 * ROWB + co);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[ks], sa, 0, 0, 0);
        sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb, qf[ks], sb, 0, 0, 0);
    }
    if (sh != 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" : "+v"(sa[0]), "+v"(sb[0]) :: "memory");   // (the lower half's barrier, from the waves that did not need it)
    // ---- per tile: its own maximum per video (= per lane column; the lane halves hold different segments), p~ = exp2(s - max) in bf16 in
    // accumulator order, and the (max, sum) pair
    const float c = a.scale * 1.4426950408889634f;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        f32x16 s = tt == 0 ? sa : sb;
        const int tile = sl * 4 + sh * 2 + tt;
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int seg = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            const bool v = one_run ? (seg >= first && seg < s_eff) : (flags[seg] != 0.f);
            s[e] = v ? s[e] * c : -INFINITY;
            mx = fmaxf(mx, s[e]);
        }
        mx = fmaxf(mx, ib_other_half(mx));
        // the tile's reference is the maximum rounded UP to an integer (the scores are in log2 units): the pv kernel's factor between a tile's
        // reference and the track's is then an exact power of two, which it applies to the bf16 probabilities by an exponent subtraction
        mx = mx == -INFINITY ? mx : __builtin_ceilf(mx);
        const float us = mx == -INFINITY ? 0.f : mx;
        float l = 0.f;
        bf16x8 p[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bf16_t pb = (bf16_t)__builtin_amdgcn_exp2f(s[e] - us);
            l += (float)pb;                                                   // the denominator of what the pv kernel will multiply with
            p[e >> 3][e & 7] = pb;
        }
        l += ib_other_half(l);
        bf16_t* wp = a.wp + ((((int64_t)m * a.Tpad + tile) * 2 + vb) * 64 + lane) * 16;
        *(bf16x8*)(wp) = p[0]; *(bf16x8*)(wp + 8) = p[1];
        if (hh == 0) *(f32x2*)(a.wml + (((int64_t)m * a.Tpad + tile) * 64 + vb * 32 + r) * 2) = (f32x2){mx, l};
    }
}

// ---------------------------------------------------------------------------------------------------------------- pv
__global__ __launch_bounds__(IB_T, 1) void xpool_inbatch_pv_kernel(const InbatchArgs a, int D) {
    constexpr int ROWB = IB_COL * 2;                  // 256-byte rows of the U column slice
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* flags = (float*)(lds + IB_SMAX * ROWB);    // [IB_SMAX]
    int* red = (int*)(flags + IB_SMAX);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int vb = wave & 1, ch = wave >> 1;          // this wave: videos 32 vb .., column blocks 2 ch and 2 ch + 1 (of the slice's four)
    const int ncs = D / IB_COL;
    const int m = (int)blockIdx.x / ncs, cs = (int)blockIdx.x % ncs;
    // ---- requested first (they fly under the mask scan and the U slice): the tiles' (max, sum) pairs of this lane's video and the first eight
    // tiles' p~ fragments (accumulator order = B-operand order); tiles the track does not have are read from its last workspace tile and ignored
    constexpr int TMAX = IB_SMAX / 32, PAH = 8;
    const float* wml = a.wml + ((int64_t)m * a.Tpad * 64 + vb * 32 + r) * 2;
    const bf16_t* wp = a.wp + (((int64_t)m * a.Tpad * 2 + vb) * 64 + lane) * 16;
    f32x2 ts[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) ts[t] = *(const f32x2*)(wml + (int64_t)(t < a.Tpad ? t : a.Tpad - 1) * 64 * 2);
    bf16x8 pq[PAH][2];
    auto load_p = [&](int t, bf16x8 (&dst)[2]) __attribute__((always_inline)) {
        const bf16_t* w = wp + (int64_t)(t < a.Tpad ? t : a.Tpad - 1) * 2 * 64 * 16;
        dst[0] = *(const bf16x8*)(w); dst[1] = *(const bf16x8*)(w + 8);
    };
#pragma unroll
    for (int t = 0; t < PAH; ++t) load_p(t, pq[t]);
    int first, s_eff; bool one_run;
    ib_scan_mask(a.key_mask ? a.key_mask + (int64_t)m * a.S : nullptr, a.S, flags, red, first, s_eff, one_run);
    const int nt = s_eff > 0 ? (s_eff + 31) / 32 : 1;                        // tiles that hold a valid segment (one all-masked tile: NaN rows)
    // ---- the U column slice: segment rr's 64-byte group j (32 columns) at rr * 256 + ((j ^ (rr & 3)) << 6); every piece (4 rows) in flight
    {
        const unsigned char* Ub = (const unsigned char*)(a.U + (int64_t)m * a.u_bs + cs * IB_COL);
        const uint32_t ldu_b = (uint32_t)a.ldu * 2u;
        const int pieces = nt * 8;                                          // 32 rows per tile, 4 rows per piece
        const int cl = lane & 15;
        const uint32_t coff = (uint32_t)cl;                                 // (the chunk's swizzle depends on rr & 3 = (lane >> 4) & 3 only: pieces are 4 rows)
        const uint32_t sw = (uint32_t)(((((coff >> 2) ^ ((lane >> 4) & 3)) << 2) | (coff & 3)) << 4);
        if (one_run) {
            for (int piece = wave; piece < pieces; piece += 4) {
                const int rr = piece * 4 + (lane >> 4);
                const uint32_t off = (uint32_t)((rr >= first && rr < s_eff) ? rr : first) * ldu_b + sw;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(Ub + off), (lds_ptr_t)(lds + piece * 1024), 16, 0, 0);
            }
        } else {
            for (int piece = wave; piece < pieces; piece += 4) {
                const int rr = piece * 4 + (lane >> 4);
                const uint32_t off = (uint32_t)(flags[rr] != 0.f ? rr : first) * ldu_b + sw;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(Ub + off), (lds_ptr_t)(lds + piece * 1024), 16, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- this lane's video: the track's reference (the largest tile reference: integers), per tile the exponent shift k = reference - tile
    // reference as a packed 16-bit decrement of a bf16 word's exponent field (k << 7; 255 flushes every value), and the denominator
    float mg = -INFINITY;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        if (t < nt) mg = fmaxf(mg, ts[t][0]);
    const float ug = mg == -INFINITY ? 0.f : mg;
    float l = 0.f;
    uint32_t kk[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const float d = t < nt ? fminf(ug - ts[t][0], 255.f) : 255.f;       // (a tile without a valid segment: -inf reference -> 255)
        const uint32_t k = (uint32_t)d;
        kk[t] = (k << 7) | (k << 23);
        l = t < nt ? __builtin_fmaf(__builtin_amdgcn_exp2f(-d), ts[t][1], l) : l;
    }
    // ---- O^T (this wave's 2 x 32 columns x 32 videos) = sum over tiles f_tile * (U_tile^T p~_tile)
    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    const int g = lane >> 4, i16 = lane & 15;
    const int trow = 4 * (g >> 1) + (i16 >> 2);
    const uint32_t u_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + trow * ROWB + (g & 1) * 32 + (i16 & 3) * 8;
    const uint32_t u_l0 = u_b + (((2 * ch) ^ (trow & 3)) << 6), u_l1 = u_b + (((2 * ch + 1) ^ (trow & 3)) << 6);
    // Two tiles per step: their sixteen transposing reads are issued together and awaited ONCE, inside one basic block.  (Issuing tile t + 1's
    // reads under tile t's MFMAs was tried: the registers an inline-assembly read has been given are ordinary values to the compiler, and at the
    // `t < nt` block boundaries it copied them BEFORE the data had arrived -- right alone on the chip, wrong in 10-30 % of the launches beside
    // another kernel's workgroups on the same CU.  An asynchronous result must not cross a block boundary.)  The odd last tile is multiplied
    // twice with its second copy's probabilities flushed to zero.
#pragma unroll
    for (int tp = 0; tp < TMAX / 2; ++tp) {
        if (2 * tp < nt) {
            const int ta = 2 * tp, tb = 2 * tp + 1;
            const bool has_b = tb < nt;
            const bf16x8 pa0 = pq[ta % PAH][0], pa1 = pq[ta % PAH][1], pb0 = pq[tb % PAH][0], pb1 = pq[tb % PAH][1];
            if (ta + PAH < TMAX) { load_p(ta + PAH, pq[ta % PAH]); load_p(tb + PAH, pq[tb % PAH]); }
            bf16x4 u[16];
            const uint32_t va = u_l0 + ta * (32 * ROWB), vb_ = u_l1 + ta * (32 * ROWB);
            const uint32_t vc = u_l0 + (has_b ? tb : ta) * (32 * ROWB), vd = u_l1 + (has_b ? tb : ta) * (32 * ROWB);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u[0]) : "v"(va));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(u[1]) : "v"(va));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(u[2]) : "v"(va));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(u[3]) : "v"(va));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u[4]) : "v"(vb_));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(u[5]) : "v"(vb_));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(u[6]) : "v"(vb_));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(u[7]) : "v"(vb_));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u[8]) : "v"(vc));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(u[9]) : "v"(vc));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(u[10]) : "v"(vc));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(u[11]) : "v"(vc));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u[12]) : "v"(vd));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(u[13]) : "v"(vd));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(u[14]) : "v"(vd));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(u[15]) : "v"(vd));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]),
                                                  "+v"(u[8]), "+v"(u[9]), "+v"(u[10]), "+v"(u[11]), "+v"(u[12]), "+v"(u[13]), "+v"(u[14]), "+v"(u[15]));
            const uint32_t kb = has_b ? kk[tb] : 0xffffffffu;               // (no second tile: every word of the repeated one saturates to +0)
            const bf16x8 qa0 = ib_scale_pow2(pa0, kk[ta]), qa1 = ib_scale_pow2(pa1, kk[ta]);
            const bf16x8 qb0 = ib_scale_pow2(pb0, kb), qb1 = ib_scale_pow2(pb1, kb);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(u[0], u[1], 0, 1, 2, 3, 4, 5, 6, 7), qa0, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(u[4], u[5], 0, 1, 2, 3, 4, 5, 6, 7), qa0, o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(u[2], u[3], 0, 1, 2, 3, 4, 5, 6, 7), qa1, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(u[6], u[7], 0, 1, 2, 3, 4, 5, 6, 7), qa1, o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(u[8], u[9], 0, 1, 2, 3, 4, 5, 6, 7), qb0, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(u[12], u[13], 0, 1, 2, 3, 4, 5, 6, 7), qb0, o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(u[10], u[11], 0, 1, 2, 3, 4, 5, 6, 7), qb1, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(u[14], u[15], 0, 1, 2, 3, 4, 5, 6, 7), qb1, o1, 0, 0, 0);
        }
    }
    // ---- O / l -> out[m][n][cs * 128 + 32 (2 ch + cb) + ..]: the lane holds column groups {0-3, 8-11, 16-19, 24-27} + 4 hh of its video's row
    const float inv = 1.f / l;
    const int n = vb * 32 + r;
    if (n < a.Nv) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int64_t off = (int64_t)m * a.o_bs + (int64_t)n * a.ldo + cs * IB_COL + (2 * ch + cb) * 32 + 4 * hh;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (cb == 0 ? o0[4 * q + j] : o1[4 * q + j]) * inv;
                if (a.out_dtype == MADE_BF16) {
                    bf16x4 b;
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[j] = (bf16_t)v[j];
                    *(bf16x4*)((bf16_t*)a.out + off + 8 * q) = b;
                } else {
                    *(f32x4*)((float*)a.out + off + 8 * q) = v;
                }
            }
        }
    }
}

}  // namespace

extern "C" int64_t made_xpool_inbatch_ws_bytes(int64_t Nm, int64_t S) {
    const int64_t tpad = ((S + IB_SEG - 1) / IB_SEG) * 4;
    return Nm * tpad * (2 * 64 * 16 * 2 + 64 * 2 * 4);
}

extern "C" int made_xpool_inbatch(const MadeXpoolInbatchArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_xpool_inbatch: null args");
    const MadeXpoolInbatchArgs& x = *args;
    MADE_REQUIRE(x.Q && x.K && x.U && x.out && x.ws, "made_xpool_inbatch: null pointer");
    MADE_REQUIRE(x.Nv >= 0 && x.Nm >= 0 && x.S > 0, "made_xpool_inbatch: bad dims");
    MADE_REQUIRE(x.out_dtype == MADE_BF16 || x.out_dtype == MADE_F32, "made_xpool_inbatch: bad out_dtype %d", x.out_dtype);
    MADE_UNSUPPORTED(x.D == 256 || x.D == 512, "made_xpool_inbatch: D=%lld not in {256, 512}", (long long)x.D);
    MADE_UNSUPPORTED(x.Nv <= 64, "made_xpool_inbatch: Nv=%lld videos (at most 64; larger batches: made_attention_wide / made_xpool_attention)", (long long)x.Nv);
    MADE_UNSUPPORTED(x.S <= IB_SMAX, "made_xpool_inbatch: S=%lld segments per track (at most %d)", (long long)x.S, IB_SMAX);
    MADE_UNSUPPORTED(x.ldq % 8 == 0 && x.ldk % 8 == 0 && x.ldu % 8 == 0 && x.k_bs % 8 == 0 && x.u_bs % 8 == 0 && x.ldo % 4 == 0 && x.o_bs % 4 == 0 &&
                     ((uintptr_t)x.Q % 16) == 0 && ((uintptr_t)x.K % 16) == 0 && ((uintptr_t)x.U % 16) == 0 && ((uintptr_t)x.out % 16) == 0 &&
                     ((uintptr_t)x.ws % 16) == 0, "made_xpool_inbatch: pointers / strides must keep 16-byte alignment");
    MADE_UNSUPPORTED((uint64_t)x.S * (uint64_t)x.ldk * 2 < (1ull << 32) && (uint64_t)x.S * (uint64_t)x.ldu * 2 < (1ull << 32), "made_xpool_inbatch: a track's rows must span less than 4 GB");
    if (x.Nv == 0 || x.Nm == 0) return MADE_OK;
    InbatchArgs a;
    a.Q = (const bf16_t*)x.Q; a.ldq = x.ldq; a.K = (const bf16_t*)x.K; a.U = (const bf16_t*)x.U;
    a.k_bs = x.k_bs; a.ldk = x.ldk; a.u_bs = x.u_bs; a.ldu = x.ldu; a.key_mask = x.key_mask;
    a.out = x.out; a.out_dtype = x.out_dtype; a.o_bs = x.o_bs; a.ldo = x.ldo;
    a.Nv = (int)x.Nv; a.Nm = (int)x.Nm; a.S = (int)x.S; a.Tpad = (int)(((x.S + IB_SEG - 1) / IB_SEG) * 4);
    a.scale = x.scale;
    a.wp = (bf16_t*)x.ws;
    a.wml = (float*)((unsigned char*)x.ws + (int64_t)a.Nm * a.Tpad * 2 * 64 * 16 * 2);
    const int lds_s = IB_SEG * (int)x.D * 2 + IB_SMAX * 4 + 64, lds_p = IB_SMAX * IB_COL * 2 + IB_SMAX * 4 + 64;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)xpool_inbatch_scores_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, IB_SEG * 512 * 2 + IB_SMAX * 4 + 64);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)xpool_inbatch_scores_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, IB_SEG * 256 * 2 + IB_SMAX * 4 + 64);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)xpool_inbatch_pv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_p);
        if (e != hipSuccess) { made_set_error("made_xpool_inbatch: cannot reserve LDS: %s", hipGetErrorString(e)); return MADE_ERR_HIP; }
        attr_done = true;
    }
    const dim3 g1((unsigned)(a.Nm * (a.Tpad / 4))), g2((unsigned)(a.Nm * (x.D / IB_COL))), blk(IB_T);
    if (x.D == 512) hipLaunchKernelGGL(xpool_inbatch_scores_kernel<512>, g1, blk, lds_s, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(xpool_inbatch_scores_kernel<256>, g1, blk, lds_s, (hipStream_t)stream, a);
    hipLaunchKernelGGL(xpool_inbatch_pv_kernel, g2, blk, lds_p, (hipStream_t)stream, a, (int)x.D);
    return made_check_launch("made_xpool_inbatch");
}
