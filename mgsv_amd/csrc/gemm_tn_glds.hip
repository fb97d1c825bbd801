// Fast path of made_gemm_tn (bf16, N and K multiples of 128, rows all valid or gathered): the weight-gradient product with the
// operands going global -> LDS directly (global_load_lds, no VGPR staging) into a two-stage ring (slab i + 1 in flight while slab i
// is multiplied), 80 KB of LDS so that TWO workgroups share a CU.  Round 1 ran one workgroup per CU with a four-stage ring: three
// slabs in flight did not buy bandwidth (a CU's four waves took in 21 GB/s; the forward GEMMs reach 40-50 GB/s with two to five
// workgroups per CU), a second workgroup does (profiles/r02_c_tn_bench.txt).
//
// LDS image of a stage: A rows [64][256 B] then B rows [64][256 B], unpadded (the LDS-DMA writes 1 KB contiguous per wave
// instruction = 4 rows), with the 64-byte blocks of a row XOR-swizzled by (row & 3) on the SOURCE side (rounds 1-5: the 32-byte pairs, which
// left rows r and r ^ 1 on the same banks -- a 32-lane half of the transposing read takes BOTH pairs of a block from each of four rows: 48 % of the
// LDS cycles were conflicts, SQ_LDS_BANK_CONFLICT; profiles/r06_sq_counters_tn256.txt), so the four rows a
// transposing read (ds_read_b64_tr_b16) touches fall on four different bank groups.
#include "common.h"

#include <type_traits>

namespace {

constexpr int GT_BN = 128, GT_BK = 128, GT_BM = 64, GT_NT = 256, GT_NST = 2;
constexpr int GT_ROW = 256;                              // bytes per LDS row (128 bf16)
constexpr int GT_HALF = GT_BM * GT_ROW;                  // 16 KB: one operand of one stage
constexpr int GT_STAGE = 2 * GT_HALF;                    // 32 KB
constexpr int GT_MAX_ROWS = 4096;                        // row indices a block may hold (64 slabs)
constexpr int GT_LDS = GT_NST * GT_STAGE + GT_MAX_ROWS * 4;  // 80 KB: two workgroups per CU

// The transposing LDS read as inline asm: through the builtin, hipcc's waitcnt pass cannot tell the read from the in-flight
// LDS-DMA writes of OTHER stages and drains vmcnt(0) before every fragment read, which serialises the ring.  The price: the
// lgkmcnt waits for these reads are ours to place (gt_wait below ties them to the registers the MFMAs consume).
template <int OFF>
__device__ __forceinline__ bf16x4 gt_tr(uint32_t addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void gt_wait(bf16x4 (&f)[8]) {           // wait until at most N newer LDS reads are outstanding
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "n"(N));
}

// One problem of a launch: C[N, K] (+)= alpha * A[M, N]^T B[M, K] over the rows of this workgroup's reduction split.
struct GtProblem {
    const bf16_t* A; const bf16_t* B; void* C; float* colsum;
    int64_t lda, ldb, ldc;
    int tiles_k, tile;                                       // tiles along K of this problem, this workgroup's tile inside it
    int c_dtype, accumulate;
    float alpha;
};
struct GtShared {                                            // what the problems of a launch have in common
    int64_t M, split_m, split_y;
    const int32_t* row_index; const int32_t* n_rows;
};

__device__ __forceinline__ void gemm_tn_glds_body(const GtProblem a, const GtShared sh, unsigned char* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const int tiles_k = a.tiles_k;
    const int64_t split_y = sh.split_y;
    const int tile_n = a.tile / tiles_k, tile_k = a.tile % tiles_k;
    const int64_t n0 = (int64_t)tile_n * GT_BN, k0 = (int64_t)tile_k * GT_BK;
    int64_t Mv = sh.M;
    if (sh.n_rows) { const int64_t nv = *sh.n_rows; Mv = nv < sh.M ? nv : sh.M; }
    const int64_t nslab = (Mv + GT_BM - 1) / GT_BM;
    const int64_t sstep = sh.split_m;
    const int64_t nloc = nslab > split_y ? (nslab - split_y + sstep - 1) / sstep : 0;   // slabs y, y + split, ...
    if (nloc == 0) return;
    const bf16_t* Ag = a.A;
    const bf16_t* Bg = a.B;

    // physical rows of every slab this block reduces, resolved ONCE into LDS: a per-slab index load would be a vector-memory
    // operation in front of the slab's LDS-DMA loads, and waiting for it (vmcnt is in-order) would drain the pipeline
    int* lds_rows = (int*)(lds + GT_NST * GT_STAGE);
    for (int64_t t = tid; t < nloc * GT_BM; t += GT_NT) {
        const int64_t g = split_y + (t / GT_BM) * sstep;
        const int64_t m = g * GT_BM + (t % GT_BM);
        const int64_t ml = m < Mv ? m : Mv - 1;
        lds_rows[t] = sh.row_index ? sh.row_index[ml] : (int)ml;
    }
    __syncthreads();

    // wave w, instruction j (0..3) moves rows 16 w + 4 j + (lane >> 4) of the slab: 16 lanes x 16 B per row
    const int pos = lane & 15;                               // 16-byte slot inside the LDS row
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    auto issue = [&](int64_t i) __attribute__((always_inline)) {
        unsigned char* st = lds + (i % GT_NST) * GT_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rowl = 16 * wave + 4 * j + (lane >> 4);
            const int64_t pr = lds_rows[i * GT_BM + rowl];
            const int chunk = (((pos >> 1) ^ ((rowl & 3) << 1)) << 1) | (pos & 1);   // source-side swizzle: the row's 64-byte blocks XORed with row & 3
            const int piece = (16 * wave + 4 * j) * GT_ROW;                          // 1 KB destination of this instruction
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(Ag + pr * a.lda + n0 + chunk * 8), (lds_ptr_t)(st + piece), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(Bg + pr * a.ldb + k0 + chunk * 8), (lds_ptr_t)(st + GT_HALF + piece), 16, 0, 0);
        }
    };

    // per-lane byte offsets of the transposing reads inside a stage (k-step 0; k-step ks adds ks * 16 rows as an immediate):
    // row = 4 * (g >> 1) + (i >> 2), 32-byte pair (c0 / 16 + (g & 1)) ^ 2 (row & 3), 8 * (i & 3) inside the pair
    uint32_t offA[2], offB[2];
    {
        const int g = lane >> 4, i16 = lane & 15;
        const int row = 4 * (g >> 1) + (i16 >> 2);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int pa = ((wn * 64 + t * 32) >> 4) + (g & 1), pb = ((wk * 64 + t * 32) >> 4) + (g & 1);
            offA[t] = (uint32_t)(row * GT_ROW + ((pa ^ ((row & 3) << 1)) << 5) + 8 * (i16 & 3));
            offB[t] = (uint32_t)(GT_HALF + row * GT_ROW + ((pb ^ ((row & 3) << 1)) << 5) + 8 * (i16 & 3));
        }
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // bias gradient = column sums of A: taken from the A fragments the MFMAs consume anyway (lane (r, hh) of an A fragment holds 8
    // consecutive reduction rows of output row n = r), by the K-tile-0 workgroup of every row of tiles, in its wk = 0 waves (the two
    // wk waves read the same A fragments).  Round 1 summed 2-byte LDS reads here: 20-25 % of the kernel.
    float csum[2] = {0.f, 0.f};
    const bool do_colsum = a.colsum != nullptr && tile_k == 0 && wk == 0;

    issue(0);
    for (int64_t i = 0; i < nloc; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // slab i (the only one in flight) has landed
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");             // every wave's part landed; everyone is done with the other stage
        if (i + 1 < nloc) issue(i + 1);
        unsigned char* st = lds + (i % GT_NST) * GT_STAGE;
        const int64_t g = split_y + i * sstep;
        const int64_t live = Mv - g * GT_BM;                 // rows of this slab that exist
        if (live < GT_BM) {
            // last slab of the reduction: the clamped duplicate rows must not be summed -> zero them in LDS (both operands)
            for (int idx = tid; idx < (GT_BM - (int)live) * (GT_ROW / 16) * 2; idx += GT_NT) {
                const int half = idx / ((GT_BM - (int)live) * (GT_ROW / 16));
                const int rem = idx % ((GT_BM - (int)live) * (GT_ROW / 16));
                const int row = (int)live + rem / (GT_ROW / 16), c16 = rem % (GT_ROW / 16);
                f32x4 z; z[0] = z[1] = z[2] = z[3] = 0.f;
                *(f32x4*)(st + half * GT_HALF + row * GT_ROW + c16 * 16) = z;
            }
            __syncthreads();
        }
        // fragment reads of k-step ks + 1 are issued before the MFMAs of k-step ks (two register sets, counted lgkmcnt)
        const uint32_t sbase = lds_base + (uint32_t)((i % GT_NST) * GT_STAGE);
        bf16x4 fr[2][8];
        auto read_step = [&](int buf, auto KS) __attribute__((always_inline)) {
            constexpr int o = decltype(KS)::value * 16 * GT_ROW;
            fr[buf][0] = gt_tr<o>(sbase + offA[0]); fr[buf][1] = gt_tr<o + 8 * GT_ROW>(sbase + offA[0]);
            fr[buf][2] = gt_tr<o>(sbase + offA[1]); fr[buf][3] = gt_tr<o + 8 * GT_ROW>(sbase + offA[1]);
            fr[buf][4] = gt_tr<o>(sbase + offB[0]); fr[buf][5] = gt_tr<o + 8 * GT_ROW>(sbase + offB[0]);
            fr[buf][6] = gt_tr<o>(sbase + offB[1]); fr[buf][7] = gt_tr<o + 8 * GT_ROW>(sbase + offB[1]);
        };
        auto mul_step = [&](int buf) __attribute__((always_inline)) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = __builtin_shufflevector(fr[buf][2 * t], fr[buf][2 * t + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                bfr[t] = __builtin_shufflevector(fr[buf][4 + 2 * t], fr[buf][4 + 2 * t + 1], 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2)
                    acc[i2][j2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i2], bfr[j2], acc[i2][j2], 0, 0, 0);
            if (do_colsum) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) csum[t] += (float)af[t][j];
            }
        };
        read_step(0, std::integral_constant<int, 0>{});
        read_step(1, std::integral_constant<int, 1>{});
        gt_wait<8>(fr[0]); mul_step(0);
        read_step(0, std::integral_constant<int, 2>{});
        gt_wait<8>(fr[1]); mul_step(1);
        read_step(1, std::integral_constant<int, 3>{});
        gt_wait<8>(fr[0]); mul_step(0);
        gt_wait<0>(fr[1]); mul_step(1);
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t k = k0 + wk * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t n = n0 + wn * 64 + i * 32 + acc_row(e, hh);
                const float v = acc[i][j][e] * a.alpha;
                if (a.c_dtype == MADE_F32) {
                    float* p = (float*)a.C + n * a.ldc + k;
                    if (a.accumulate) unsafeAtomicAdd(p, v); else *p = v;
                } else {
                    ((bf16_t*)a.C)[n * a.ldc + k] = (bf16_t)v;
                }
            }
        }
    if (do_colsum) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float tot = csum[t] + __shfl_xor(csum[t], 32);                     // the two lane halves hold different reduction rows
            if (hh == 0) unsafeAtomicAdd(a.colsum + n0 + wn * 64 + t * 32 + r, tot * a.alpha);
        }
    }
}

__global__ __launch_bounds__(GT_NT, 2) void gemm_tn_glds_kernel(const MadeGemmTNArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // XCD-aware placement (1-D grid; workgroup b runs on XCD b % 8): all output tiles of one reduction split -- which read the
    // same rows of A and B -- go to ONE XCD, so each 16 KB panel slab is fetched from HBM once per split and served to the other
    // tiles from that XCD's L2 (with the tiles of a split spread over the XCDs every tile re-fetched its two panels)
    const int tiles_k = (int)(a.K / GT_BK);
    const int tiles = (int)(a.N / GT_BN) * tiles_k;
    // (fewer than 8 splits -- short reductions, where the splits' atomic adds would outweigh the products: round 6 -- are dealt block by block,
    //  tile-major: consecutive blocks = consecutive XCDs, every XCD busy; the operands of so short a reduction sit in every L2 anyway)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int64_t split_y = a.split_m >= 8 ? xcd + 8 * (int64_t)(jx / tiles) : (int64_t)(blockIdx.x % (unsigned)a.split_m);
    if (split_y >= a.split_m) return;
    GtProblem p;
    p.A = (const bf16_t*)a.A; p.B = (const bf16_t*)a.B; p.C = a.C; p.colsum = a.colsum;
    p.lda = a.lda; p.ldb = a.ldb; p.ldc = a.ldc; p.tiles_k = tiles_k; p.tile = a.split_m >= 8 ? jx % tiles : (int)(blockIdx.x / (unsigned)a.split_m);
    p.c_dtype = a.c_dtype; p.accumulate = a.accumulate; p.alpha = a.alpha;
    GtShared sh;
    sh.M = a.M; sh.split_m = a.split_m; sh.split_y = split_y; sh.row_index = a.row_index; sh.n_rows = a.n_rows;
    gemm_tn_glds_body(p, sh, lds);
}

// Several weight gradients that reduce over the SAME rows (the Linears of one transformer layer: dW_i = dY_i^T X_i) in one launch.
// Separate launches need 256 / tiles_i reduction splits each to fill the chip, and every split adds its whole 128 x 128 f32 tile
// to the gradient with atomics (16.8 MB per launch at 1.3 TB/s = 13 us, whatever the problem size); together the problems of a
// layer have 128 tiles, so 3-4 splits are enough: a quarter of the atomic traffic and one ramp-up / drain instead of five.
__global__ __launch_bounds__(GT_NT, 2) void gemm_tn_glds_grouped_kernel(const MadeGemmTNGroup g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tiles = g.tile_end[g.n_problems - 1];
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int64_t split_y = g.split_m >= 8 ? xcd + 8 * (int64_t)(jx / tiles) : (int64_t)(blockIdx.x % (unsigned)g.split_m);   // (see gemm_tn_glds_kernel)
    if (split_y >= g.split_m) return;
    int tile = g.split_m >= 8 ? jx % tiles : (int)(blockIdx.x / (unsigned)g.split_m), pi = 0;
#pragma unroll
    for (int i = 0; i < MADE_GEMM_TN_MAX_GROUP - 1; ++i)
        if (i + 1 < g.n_problems && tile >= g.tile_end[i]) pi = i + 1;
    if (pi > 0) tile -= g.tile_end[pi - 1];
    GtProblem p;
    p.A = (const bf16_t*)g.p[pi].A; p.B = (const bf16_t*)g.p[pi].B; p.C = g.p[pi].C; p.colsum = g.p[pi].colsum;
    p.lda = g.p[pi].lda; p.ldb = g.p[pi].ldb; p.ldc = g.p[pi].ldc; p.tiles_k = (int)(g.p[pi].K / GT_BK); p.tile = tile;
    p.c_dtype = MADE_F32; p.accumulate = 1; p.alpha = g.alpha;
    GtShared sh;
    sh.M = g.M; sh.split_m = g.split_m; sh.split_y = split_y; sh.row_index = g.row_index; sh.n_rows = g.n_rows;
    gemm_tn_glds_body(p, sh, lds);
}

// =================================================================================================
// 256 x 256 output tiles (round 3).  The 128 x 128 kernel above moves 32 KB of operands per 2.1 MFLOP (65 flop / byte): at the
// ~50 GB/s a CU takes in through LDS-DMA that is a third of the MFMA rate, and a layer's five products need 16 reduction splits to
// fill the chip, each adding its whole f32 tile to the gradient with atomics (134 MB per layer).  Here a workgroup of EIGHT waves
// owns a 256 x 256 tile (a wave: 128 x 64 = 4 x 2 MFMA tiles, 128 accumulator registers): 64 KB per 8.4 MFLOP (131 flop / byte), one
// workgroup per CU, a quarter of the atomic traffic, and per k-step 12 transposing LDS reads feed 8 MFMAs (the LDS read rate stays
// below the MFMA rate).  Same slab ring, same source-side swizzle, same bias-gradient trick.
#ifndef TN256_SKIP
#define TN256_SKIP 0          // elimination builds (tools/variant_build.sh ... "-DTN256_SKIP=<mask>"): 1 no flush, 2 no MFMAs, 4 no operand loads
#endif
constexpr int T2_BN = 256, T2_BK = 256, T2_BM = 64, T2_NT = 512, T2_NST = 2;
constexpr int T2_ROW = 512;                               // bytes per LDS row (256 bf16)
constexpr int T2_HALF = T2_BM * T2_ROW;                   // 32 KB: one operand of one stage
constexpr int T2_STAGE = 2 * T2_HALF;                     // 64 KB
constexpr int T2_MAX_ROWS = 4608;                         // row indices a block may hold (72 slabs)
constexpr int T2_LDS = T2_NST * T2_STAGE + T2_MAX_ROWS * 4;   // 146 KB: one workgroup per CU

__device__ __forceinline__ void t2_wait12(bf16x4 (&f)[12], int n_outstanding_is_12) {
    if (n_outstanding_is_12)
        asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                     "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                     "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]));
}

// Work of a launch = (tile, 64-row slab) units.  The slabs are dealt over the 8 XCDs (slab s to the workgroups with blockIdx % 8 == s % 8:
// under the observed round-robin placement the rows of a slab are then read through ONE L2 by all tiles), and the units of an XCD,
// tile-major, are cut into equal consecutive ranges for its workgroups -- any number of tiles keeps all 256 CUs busy to within one
// slab (36 tiles x 8 splits as whole tiles per workgroup were 288 workgroups = two rounds, the second one 12 % full).  A workgroup adds its
// accumulators to C whenever its range leaves a tile (at most ceil(32 / tiles) + 1 times).
struct T2Tile { const bf16_t* A; const bf16_t* B; float* C; float* colsum; int64_t lda, ldb, ldc, n0, k0; int first_k; };

// (The group is read where the launch put it -- the kernel-argument segment, through scalar loads at a computed offset -- not from the by-value
//  parameter: selecting a problem's members out of that costs ~150 scalar registers spilled to vector lanes, and indexing it with a run-time problem
//  number makes the compiler copy all 672 bytes of it to scratch memory.)
typedef const __attribute__((address_space(4))) MadeGemmTNGroup* T2GroupPtr;
__device__ __forceinline__ T2GroupPtr t2_group() { return (T2GroupPtr)__builtin_amdgcn_kernarg_segment_ptr(); }      // the group is the kernels' FIRST parameter

__device__ __forceinline__ T2Tile t2_tile(T2GroupPtr g, int tile) {
    int pi = 0, base = 0;
#pragma unroll
    for (int i = 0; i < MADE_GEMM_TN_MAX_GROUP - 1; ++i)
        if (i + 1 < g->n_problems && tile >= g->tile_end[i]) { pi = i + 1; base = g->tile_end[i]; }
    tile -= base;
    T2Tile t;
    MadeGemmTNProblem p;                                     // (member by member: the struct has no constructor from the constant address space)
    p.A = g->p[pi].A; p.B = g->p[pi].B; p.C = g->p[pi].C; p.colsum = g->p[pi].colsum;
    p.N = g->p[pi].N; p.K = g->p[pi].K; p.lda = g->p[pi].lda; p.ldb = g->p[pi].ldb; p.ldc = g->p[pi].ldc;
    const int tiles_k = (int)(p.K / T2_BK);
    t.A = (const bf16_t*)p.A; t.B = (const bf16_t*)p.B; t.C = (float*)p.C; t.colsum = p.colsum;
    t.lda = p.lda; t.ldb = p.ldb; t.ldc = p.ldc;
    t.n0 = (int64_t)(tile / tiles_k) * T2_BN; t.k0 = (int64_t)(tile % tiles_k) * T2_BK; t.first_k = (tile % tiles_k) == 0;
    return t;
}

// The flush without atomics (round 6).  A tile's reduction is split over ~8 workgroups and each used to ADD its 256 x 256 f32 partial to the
// gradient with atomics: 67 MB per encoder layer at the 1.2 TB/s the L2s' atomic units sustain (one dword per clock and channel; the same whether
// the eight workgroups of a tile sit on one XCD or on eight: tools/probes/xcc_atomics_probe.hip) = 83 of the launch's 146 us.  With a workspace
// (MadeGemmTNGroup.workspace) a workgroup instead STORES its partial (plain 16-byte stores in accumulator order: 4 us) and a second launch,
// gemm_tn_256_reduce_kernel, sums the partials of every 32-row strip of every tile in a fixed order (block number, then flush number: the
// gradient is bitwise reproducible -- the atomics' was not) and adds the sum to the gradient with plain accesses.  (Summing inside the launch by
// each strip's last contributor -- a counter per strip, nobody waiting for anybody -- was built first and measured slower than the atomics: the
// workgroups of a tile finish up to 20 us apart, the last one then sums the whole tile alone, 2 MB through one CU: 9 us per strip;
// profiles/r06_tn256_flush.txt.)  C is updated with plain read-modify-write: launches that update the same C must be ordered (one stream), as the step's are.
struct T2Ws { float* part; int F; };                        // partial slots of 256 x 256 f32 [gridDim * F], flushes a workgroup can make

__global__ __launch_bounds__(T2_NT, 1) void gemm_tn_256_grouped_kernel(const MadeGemmTNGroup g, const T2Ws w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int wn = wave >> 2, wk = wave & 3;                 // 2 x 4 waves: 128 output rows (n) x 64 output columns (k) each
    const int tiles = g.tile_end[g.n_problems - 1];
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    int64_t Mv = g.M;
    if (g.n_rows) { const int64_t nv = *g.n_rows; Mv = nv < g.M ? nv : g.M; }
    const int64_t nslab = (Mv + T2_BM - 1) / T2_BM;
    const int64_t S = nslab > xcd ? (nslab - xcd + 7) / 8 : 0;          // slabs xcd, xcd + 8, ... : this XCD's
    const int64_t U = S * tiles;
    const int64_t u0 = (U * wg) / nwg, u1 = (U * (wg + 1)) / nwg;       // this workgroup's units
    if (u0 >= u1) return;

    int* lds_rows = (int*)(lds + T2_NST * T2_STAGE);         // physical rows of every slab of this XCD, resolved once
    for (int64_t t = tid; t < S * T2_BM; t += T2_NT) {
        const int64_t sl = xcd + (t / T2_BM) * 8;
        const int64_t m = sl * T2_BM + (t % T2_BM);
        const int64_t ml = m < Mv ? m : Mv - 1;
        lds_rows[t] = g.row_index ? g.row_index[ml] : (int)ml;
    }
    __syncthreads();

    // wave w, instruction j (0..3) moves rows 8 w + 2 j + (lane >> 5) of the slab: 32 lanes x 16 B per 512-byte row
    const int pos = lane & 31;                               // 16-byte slot inside the LDS row
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    auto issue = [&](int64_t n, int64_t si, const T2Tile& t) __attribute__((always_inline)) {    // n: how many units this workgroup has issued
        unsigned char* st = lds + (n % T2_NST) * T2_STAGE;
        if (TN256_SKIP & 4) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rowl = 8 * wave + 2 * j + (lane >> 5);
            const int64_t pr = lds_rows[si * T2_BM + rowl];
            const int chunk = (((pos >> 1) ^ ((rowl & 3) << 1)) << 1) | (pos & 1);   // source-side swizzle: the row's 64-byte blocks XORed with row & 3
            const int piece = (8 * wave + 2 * j) * T2_ROW;                           // 1 KB destination of this instruction (two rows)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(t.A + pr * t.lda + t.n0 + chunk * 8), (lds_ptr_t)(st + piece), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(t.B + pr * t.ldb + t.k0 + chunk * 8), (lds_ptr_t)(st + T2_HALF + piece), 16, 0, 0);
        }
    };

    // per-lane byte offsets of the transposing reads inside a stage (k-step 0; k-step ks adds ks * 16 rows as an immediate)
    uint32_t offA[4], offB[2];
    {
        const int gq = lane >> 4, i16 = lane & 15;
        const int row = 4 * (gq >> 1) + (i16 >> 2);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int pa = ((wn * 128 + t * 32) >> 4) + (gq & 1);
            offA[t] = (uint32_t)(row * T2_ROW + ((pa ^ ((row & 3) << 1)) << 5) + 8 * (i16 & 3));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int pb = ((wk * 64 + t * 32) >> 4) + (gq & 1);
            offB[t] = (uint32_t)(T2_HALF + row * T2_ROW + ((pb ^ ((row & 3) << 1)) << 5) + 8 * (i16 & 3));
        }
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

    f32x16 acc[4][2];
    float csum[4];                                           // bias gradient = column sums of A, from the A fragments (see the 128 x 128 body)
    auto clear = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            csum[i] = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
    };
    auto flush = [&](const T2Tile& t) __attribute__((always_inline)) {
        if (TN256_SKIP & 1) {                                // (every accumulator stays live: with one of them tested the compiler drops the other MFMAs)
            float z = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) z += acc[i][j][e];
            if (z == 123.456f) t.C[0] = 1.f;
            return;
        }
        int ftid = threadIdx.x;                              // (see flush_ws: nothing of the flush parked in registers across the unit loop)
        asm volatile("" : "+v"(ftid));
        const int r = ftid & 31, hh = (ftid >> 5) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t k = t.k0 + wk * 64 + j * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t n = t.n0 + wn * 128 + i * 32 + acc_row(e, hh);
                    unsafeAtomicAdd(t.C + n * t.ldc + k, acc[i][j][e] * g.alpha);
                }
            }
        if (t.colsum != nullptr && t.first_k && wk == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float tot = csum[i] + __shfl_xor(csum[i], 32);                 // the two lane halves hold different reduction rows
                if (hh == 0) unsafeAtomicAdd(t.colsum + t.n0 + wn * 128 + i * 32 + r, tot * g.alpha);
            }
        }
    };
    // ---- the flush through the workspace (see above the kernel)
    int nflush = 0;
    const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc((void*)w.part, 0, w.part ? (int)((unsigned)gridDim.x * (unsigned)w.F * 262144u) : 0, 0x00020000);
    auto flush_ws = [&](const T2Tile& t) __attribute__((always_inline)) {
        // (what the flush derives from the thread's number is made here, not in front of the unit loop where the optimiser would park it in registers
        //  the loop has none to spare: the loop spilled to scratch memory without this fence)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, r = lane & 31, hh = lane >> 5;
        const uint32_t slot_b = ((uint32_t)blockIdx.x * (uint32_t)w.F + (uint32_t)nflush) * 262144u;
        ++nflush;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 v = {acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
                    const uint32_t chunk = (uint32_t)(((wn * 4 + i) * 8 + wk * 2 + j) * 4 + g4);       // strip wn * 4 + i: 32 chunks of 1 KB
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_part, slot_b + chunk * 1024u + (uint32_t)lane * 16u, 0, 0);
                }
        if (t.colsum != nullptr && t.first_k && wk == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float tot = csum[i] + __shfl_xor(csum[i], 32);
                if (hh == 0) unsafeAtomicAdd(t.colsum + t.n0 + wn * 128 + i * 32 + r, tot * g.alpha);
            }
        }
    };

    int tile = (int)(u0 / S);
    int64_t si = u0 % S;                                     // slab (index inside this XCD's list) of the current unit
    T2Tile cur = t2_tile(t2_group(), tile), nxt = cur;
    clear();
    issue(0, si, cur);
    for (int64_t u = u0, n = 0; u < u1; ++u, ++n) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this unit's slab (the only one in flight) has landed
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");             // every wave's part landed; everyone is done with the other stage
        const bool last_of_tile = si + 1 == S;
        if (u + 1 < u1) {
            if (last_of_tile) { nxt = t2_tile(t2_group(), tile + 1); issue(n + 1, 0, nxt); }
            else issue(n + 1, si + 1, cur);
        }
        unsigned char* st = lds + (n % T2_NST) * T2_STAGE;
        const int64_t live = Mv - (xcd + si * 8) * T2_BM;    // rows of this slab that exist
        if (live < T2_BM) {
            // last slab of the reduction: the clamped duplicate rows must not be summed -> zero them in LDS (both operands)
            const int per_half = (T2_BM - (int)live) * (T2_ROW / 16);
            for (int idx = tid; idx < per_half * 2; idx += T2_NT) {
                const int half = idx / per_half, rem = idx % per_half;
                const int row = (int)live + rem / (T2_ROW / 16), c16 = rem % (T2_ROW / 16);
                f32x4 z; z[0] = z[1] = z[2] = z[3] = 0.f;
                *(f32x4*)(st + half * T2_HALF + row * T2_ROW + c16 * 16) = z;
            }
            __syncthreads();
        }
        const bool do_colsum = cur.colsum != nullptr && cur.first_k && wk == 0;
        const uint32_t sbase = lds_base + (uint32_t)((n % T2_NST) * T2_STAGE);
        bf16x4 fr[2][12];
        auto read_step = [&](int buf, auto KS) __attribute__((always_inline)) {
            constexpr int o = decltype(KS)::value * 16 * T2_ROW;
#pragma unroll
            for (int t = 0; t < 4; ++t) { fr[buf][2 * t] = gt_tr<o>(sbase + offA[t]); fr[buf][2 * t + 1] = gt_tr<o + 8 * T2_ROW>(sbase + offA[t]); }
#pragma unroll
            for (int t = 0; t < 2; ++t) { fr[buf][8 + 2 * t] = gt_tr<o>(sbase + offB[t]); fr[buf][8 + 2 * t + 1] = gt_tr<o + 8 * T2_ROW>(sbase + offB[t]); }
        };
        auto mul_step = [&](int buf) __attribute__((always_inline)) {
            bf16x8 af[4], bfr[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) af[t] = __builtin_shufflevector(fr[buf][2 * t], fr[buf][2 * t + 1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int t = 0; t < 2; ++t) bfr[t] = __builtin_shufflevector(fr[buf][8 + 2 * t], fr[buf][8 + 2 * t + 1], 0, 1, 2, 3, 4, 5, 6, 7);
            if (!(TN256_SKIP & 2)) {
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2)
                    acc[i2][j2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i2], bfr[j2], acc[i2][j2], 0, 0, 0);
            } else { acc[0][0][0] += (float)af[0][0] + (float)bfr[0][0]; }
            if (do_colsum) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) csum[t] += (float)af[t][j];
            }
        };
        read_step(0, std::integral_constant<int, 0>{});
        read_step(1, std::integral_constant<int, 1>{});
        t2_wait12(fr[0], 1); mul_step(0);
        read_step(0, std::integral_constant<int, 2>{});
        t2_wait12(fr[1], 1); mul_step(1);
        read_step(1, std::integral_constant<int, 3>{});
        t2_wait12(fr[0], 1); mul_step(0);
        t2_wait12(fr[1], 0); mul_step(1);
        if (last_of_tile || u + 1 == u1) {                   // the range leaves this tile: add what was gathered to the gradient
            if (w.part != nullptr && !(TN256_SKIP & 1)) flush_ws(cur);
            else flush(cur);
            clear();
        }
        if (last_of_tile) { cur = nxt; ++tile; si = 0; } else ++si;
    }
}

// Second launch of the workspace form: workgroup (tile, strip) sums the strip's partials -- 32 rows x 256 columns, one 32 KB piece per contributor
// in accumulator order -- over the tile's contributors in block order, and adds alpha * sum to C.  The contributors follow from the launch's
// geometry alone: block b = (wg = b / 8, xcd = b % 8) of the first launch covered units [U wg / nwg, U (wg + 1) / nwg) of its XCD's S_x * tiles
// (tile-major), and made one flush per tile it touched, in order.  Eight waves, wave v chunks 4 v .. 4 v + 3 of the strip's 32; all of a
// lane's loads (4 per contributor, up to 32 in flight) are requested before the first use.
__global__ __launch_bounds__(T2_NT, 1) void gemm_tn_256_reduce_kernel(const MadeGemmTNGroup g, const T2Ws w, int grid1) {
    __shared__ int f_list[256];                              // slots of the tile's partials, in summing order
    __shared__ int f_cnt[5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int tile_idx = blockIdx.x >> 3, s8 = blockIdx.x & 7;
    const int tiles = g.tile_end[g.n_problems - 1];
    const int nwg = grid1 >> 3;
    int64_t Mv = g.M;
    if (g.n_rows) { const int64_t nv = *g.n_rows; Mv = nv < g.M ? nv : g.M; }
    const int64_t nslab = (Mv + T2_BM - 1) / T2_BM;
    const T2Tile t = t2_tile(t2_group(), tile_idx);
    // this lane's C addresses first (their loads fly under the partials')
    float cv[4][4];
    float* cp[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = wave * 4 + c;                         // (wk, j, g4) of the chunk
        const int g4 = ch & 3, j = (ch >> 2) & 1, wkc = ch >> 3;
        cp[c] = t.C + (t.n0 + s8 * 32 + 8 * g4 + 4 * hh) * t.ldc + t.k0 + wkc * 64 + j * 32 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) cv[c][e] = cp[c][e * t.ldc];
    }
    int hit = 0, slot = 0;
    if (tid < grid1) {
        const int bx = tid & 7, bw = tid >> 3;
        const int64_t Sx = nslab > bx ? (nslab - bx + 7) / 8 : 0, Ux = Sx * tiles;
        const int64_t b0 = (Ux * bw) / nwg, b1 = (Ux * (bw + 1)) / nwg;
        if (b0 < b1 && b0 < (int64_t)(tile_idx + 1) * Sx && b1 > (int64_t)tile_idx * Sx) {
            hit = 1;
            slot = tid * w.F + (tile_idx - (int)(b0 / Sx));
        }
    }
    const unsigned long long bal = __ballot(hit);
    if (wave < 4 && lane == 0) f_cnt[wave] = __popcll(bal);
    __syncthreads();
    if (wave < 4 && hit) {
        int base = 0;
        for (int k = 0; k < wave; ++k) base += f_cnt[k];
        f_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = slot;
    }
    if (tid == 0) f_cnt[4] = f_cnt[0] + f_cnt[1] + f_cnt[2] + f_cnt[3];
    __syncthreads();
    const int nc = f_cnt[4];
    f32x4 sum[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) sum[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < nc; k0 += 8) {                     // eight partials (32 loads of 16 bytes per lane) in flight
        f32x4 v[8][4];
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8) {
            const int kk = k0 + u8 < nc ? k0 + u8 : nc - 1;
            const float* sp = w.part + (int64_t)f_list[kk] * 65536 + (s8 * 32 + wave * 4) * 256 + lane * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[u8][c] = *(const f32x4*)(sp + c * 256);
        }
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8)
            if (k0 + u8 < nc) {
#pragma unroll
                for (int c = 0; c < 4; ++c) sum[c] += v[u8][c];
            }
    }
    if (nc == 0) return;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) cp[c][e * t.ldc] = cv[c][e] + sum[c][e] * g.alpha;
}

}  // namespace

// workspace of the 256 x 256-tile form: gridDim * F partial slots of 256 KB
static int t2_flushes(int tiles, int grid) { const int nwg = grid / 8 > 0 ? grid / 8 : 1; return (tiles + nwg - 1) / nwg + 1; }
static int64_t t2_workspace_bytes(int tiles, int grid) { return (int64_t)grid * t2_flushes(tiles, grid) * 262144; }
static bool t2_shape(const MadeGemmTNGroup& g, int* tiles_out, int* grid_out) {
    int tiles2 = 0;
    for (int i = 0; i < g.n_problems; ++i) {
        const auto& p = g.p[i];
        if (p.N <= 0 || p.K <= 0 || p.N % T2_BN != 0 || p.K % T2_BK != 0) return false;
        tiles2 += (int)((p.N / T2_BN) * (p.K / T2_BK));
    }
    const int64_t per_xcd = (((g.M + T2_BM - 1) / T2_BM + 7) / 8) * tiles2;
    const int64_t cap = 32;                                  // (fewer workgroups per XCD -- CUs left to the other stream's kernels -- gained nothing in the step: profiles/r06_ab_tn256.txt)
    *tiles_out = tiles2;
    *grid_out = (int)(8 * (per_xcd < cap ? (per_xcd > 0 ? per_xcd : 1) : cap));
    return true;
}

extern "C" int64_t made_gemm_tn_grouped_workspace(const MadeGemmTNGroup* group) {
    if (group == nullptr || group->tile_size != 256 || group->n_problems < 1 || group->n_problems > MADE_GEMM_TN_MAX_GROUP || group->M <= 0) return 0;
    int tiles = 0, grid = 0;
    if (!t2_shape(*group, &tiles, &grid)) return 0;
    return t2_workspace_bytes(tiles, grid);
}

extern "C" int made_gemm_tn_grouped(const MadeGemmTNGroup* group, void* stream) {
    MADE_REQUIRE(group != nullptr, "made_gemm_tn_grouped: null args");
    MadeGemmTNGroup g = *group;
    MADE_REQUIRE(g.n_problems >= 1 && g.n_problems <= MADE_GEMM_TN_MAX_GROUP, "made_gemm_tn_grouped: n_problems=%d out of [1, %d]", g.n_problems, MADE_GEMM_TN_MAX_GROUP);
    MADE_REQUIRE(g.M >= 0 && g.split_m >= 1, "made_gemm_tn_grouped: bad M / split_m");
    MADE_REQUIRE((g.row_index == nullptr) == (g.n_rows == nullptr), "made_gemm_tn_grouped: row_index and n_rows come together");
    if (g.M == 0) return MADE_OK;
    if (g.tile_size == 256) {                                 // 256 x 256 output tiles, eight waves, one workgroup per CU
        int tiles2 = 0;
        for (int i = 0; i < g.n_problems; ++i) {
            const auto& p = g.p[i];
            MADE_REQUIRE(p.A && p.B && p.C, "made_gemm_tn_grouped: problem %d has a null tensor", i);
            MADE_UNSUPPORTED(p.N > 0 && p.K > 0 && p.N % T2_BN == 0 && p.K % T2_BK == 0, "made_gemm_tn_grouped(256): problem %d: N=%lld, K=%lld must be multiples of 256",
                             i, (long long)p.N, (long long)p.K);
            MADE_UNSUPPORTED(p.lda % 8 == 0 && p.ldb % 8 == 0 && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0,
                             "made_gemm_tn_grouped: problem %d: operand rows must be 16-byte aligned", i);
            tiles2 += (int)((p.N / T2_BN) * (p.K / T2_BK));
            g.tile_end[i] = tiles2;
        }
        MADE_UNSUPPORTED(((g.M + T2_BM - 1) / T2_BM + 7) / 8 * T2_BM <= T2_MAX_ROWS,
                         "made_gemm_tn_grouped(256): more than %d rows (a workgroup keeps an eighth of the row list in LDS)", 8 * T2_MAX_ROWS);
        // one workgroup per CU; fewer when there is less than a slab for each (small M): 8 (the XCDs) x ceil(units of an XCD / 1) capped at 32
        int tiles_chk = 0, grid_n = 0;
        t2_shape(g, &tiles_chk, &grid_n);                       // one workgroup per CU; fewer when there is less than a slab for each (small M)
        dim3 grid2((unsigned)grid_n, 1, 1);
        static const bool once2 = hipFuncSetAttribute((const void*)gemm_tn_256_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS) == hipSuccess;
        (void)once2;
        T2Ws w; w.part = nullptr; w.F = 0;
        if (g.workspace != nullptr && made_variant_env("MADE_TN256_ATOMIC_FLUSH") == nullptr) {
            const int64_t need = t2_workspace_bytes(tiles2, (int)grid2.x);
            MADE_REQUIRE(g.workspace_bytes >= need && ((uintptr_t)g.workspace % 16) == 0,
                         "made_gemm_tn_grouped(256): workspace of %lld bytes, %lld needed (made_gemm_tn_grouped_workspace), 16-byte aligned",
                         (long long)g.workspace_bytes, (long long)need);
            w.part = (float*)g.workspace;
            w.F = t2_flushes(tiles2, (int)grid2.x);
        }
        hipLaunchKernelGGL(gemm_tn_256_grouped_kernel, grid2, dim3(T2_NT), T2_LDS, (hipStream_t)stream, g, w);
        if (w.part != nullptr && made_variant_env("MADE_TN256_NO_REDUCE") == nullptr) {      // (measurement knob: the first launch alone -- wrong results)
            const int rc1 = made_check_launch("made_gemm_tn_grouped(256)");
            if (rc1 != MADE_OK) return rc1;
            hipLaunchKernelGGL(gemm_tn_256_reduce_kernel, dim3((unsigned)tiles2 * 8), dim3(T2_NT), 0, (hipStream_t)stream, g, w, (int)grid2.x);
            return made_check_launch("made_gemm_tn_grouped(256, reduce)");
        }
        return made_check_launch("made_gemm_tn_grouped(256)");
    }
    MADE_REQUIRE(g.tile_size == 0 || g.tile_size == 128, "made_gemm_tn_grouped: tile_size=%d (0 / 128 or 256)", g.tile_size);
    int tiles = 0;
    for (int i = 0; i < g.n_problems; ++i) {
        const auto& p = g.p[i];
        MADE_REQUIRE(p.A && p.B && p.C, "made_gemm_tn_grouped: problem %d has a null tensor", i);
        MADE_UNSUPPORTED(p.N > 0 && p.K > 0 && p.N % GT_BN == 0 && p.K % GT_BK == 0, "made_gemm_tn_grouped: problem %d: N=%lld, K=%lld must be multiples of 128",
                         i, (long long)p.N, (long long)p.K);
        MADE_UNSUPPORTED(p.lda % 8 == 0 && p.ldb % 8 == 0 && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0,
                         "made_gemm_tn_grouped: problem %d: operand rows must be 16-byte aligned", i);
        tiles += (int)((p.N / GT_BN) * (p.K / GT_BK));
        g.tile_end[i] = tiles;
    }
    MADE_UNSUPPORTED(((g.M + GT_BM - 1) / GT_BM + g.split_m - 1) / g.split_m * GT_BM <= GT_MAX_ROWS,
                     "made_gemm_tn_grouped: split_m=%lld leaves more than %d rows per workgroup", (long long)g.split_m, GT_MAX_ROWS);
    dim3 grid((unsigned)(g.split_m >= 8 ? 8 * (int64_t)tiles * ((g.split_m + 7) / 8) : (int64_t)tiles * g.split_m), 1, 1);
    static const bool once = hipFuncSetAttribute((const void*)gemm_tn_glds_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS) == hipSuccess;
    (void)once;
    hipLaunchKernelGGL(gemm_tn_glds_grouped_kernel, grid, dim3(GT_NT), GT_LDS, (hipStream_t)stream, g);
    return made_check_launch("made_gemm_tn_grouped");
}


// called by made_gemm_tn after validation; returns MADE_ERR_UNSUPPORTED-like sentinel 1 when the fast path does not apply
int made_gemm_tn_fast(const MadeGemmTNArgs& a, hipStream_t st) {
    const bool ok = a.ab_dtype == MADE_BF16 && a.N % GT_BN == 0 && a.K % GT_BK == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 &&
                    a.batch1 * a.batch2 == 1 && (a.row_mask == nullptr || a.row_index != nullptr) &&
                    ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.B % 16) == 0 && a.M >= 4 * GT_BM &&
                    ((a.M + GT_BM - 1) / GT_BM + a.split_m - 1) / a.split_m * GT_BM <= GT_MAX_ROWS;
    if (!ok) return 1;
    const int64_t tiles = (a.N / GT_BN) * (a.K / GT_BK);
    dim3 grid((unsigned)(a.split_m >= 8 ? 8 * tiles * ((a.split_m + 7) / 8) : tiles * a.split_m), 1, 1);
    static const bool once = hipFuncSetAttribute((const void*)gemm_tn_glds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS) == hipSuccess;
    (void)once;
    hipLaunchKernelGGL(gemm_tn_glds_kernel, grid, dim3(GT_NT), GT_LDS, st, a);
    return made_check_launch("made_gemm_tn(glds)");
}
