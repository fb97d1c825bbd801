// Fast path of made_gemm_tn (bf16, N and K multiples of 128, rows all valid or gathered): the weight-gradient product with the
// operands going global -> LDS directly (global_load_lds, no VGPR staging) into a four-stage ring, so three slabs of the
// reduction are in flight while a fourth is multiplied -- the register-staged kernel has one, and at one or two workgroups per
// CU its loop is a chain of exposed memory round trips (measured: 3.2 us per 64-row slab vs 0.2 us of MFMA work).
//
// LDS image of a stage: A rows [64][256 B] then B rows [64][256 B], unpadded (the LDS-DMA writes 1 KB contiguous per wave
// instruction = 4 rows), with the 32-byte column pairs of a row XOR-swizzled by (row & 3) on the SOURCE side, so the four rows a
// transposing read (ds_read_b64_tr_b16) touches fall on four different bank groups.
#include "common.h"

#include <type_traits>

namespace {

constexpr int GT_BN = 128, GT_BK = 128, GT_BM = 64, GT_NT = 256, GT_NST = 4;
constexpr int GT_ROW = 256;                              // bytes per LDS row (128 bf16)
constexpr int GT_HALF = GT_BM * GT_ROW;                  // 16 KB: one operand of one stage
constexpr int GT_STAGE = 2 * GT_HALF;                    // 32 KB
constexpr int GT_MAX_ROWS = 6144;                        // row indices a block may hold (96 slabs)

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

__global__ __launch_bounds__(GT_NT, 1) void gemm_tn_glds_kernel(const MadeGemmTNArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[GT_NST * GT_STAGE + GT_MAX_ROWS * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    // XCD-aware placement (1-D grid; workgroup b runs on XCD b % 8): all output tiles of one reduction split -- which read the
    // same rows of A and B -- go to ONE XCD, so each 16 KB panel slab is fetched from HBM once per split and served to the other
    // tiles from that XCD's L2 (with the tiles of a split spread over the XCDs every tile re-fetched its two panels)
    const int tiles_k = (int)(a.K / GT_BK);
    const int tiles = (int)(a.N / GT_BN) * tiles_k;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int64_t split_y = xcd + 8 * (int64_t)(jx / tiles);
    if (split_y >= a.split_m) return;
    const int tile = jx % tiles;
    const int tile_n = tile / tiles_k, tile_k = tile % tiles_k;
    const int64_t n0 = (int64_t)tile_n * GT_BN, k0 = (int64_t)tile_k * GT_BK;
    int64_t Mv = a.M;
    if (a.n_rows) { const int64_t nv = *a.n_rows; Mv = nv < a.M ? nv : a.M; }
    const int64_t nslab = (Mv + GT_BM - 1) / GT_BM;
    const int64_t sstep = a.split_m;
    const int64_t nloc = nslab > split_y ? (nslab - split_y + sstep - 1) / sstep : 0;   // slabs y, y + split, ...
    if (nloc == 0) return;
    const bf16_t* Ag = (const bf16_t*)a.A;
    const bf16_t* Bg = (const bf16_t*)a.B;

    // physical rows of every slab this block reduces, resolved ONCE into LDS: a per-slab index load would be a vector-memory
    // operation in front of the slab's LDS-DMA loads, and waiting for it (vmcnt is in-order) would drain the pipeline
    int* lds_rows = (int*)(lds + GT_NST * GT_STAGE);
    for (int64_t t = tid; t < nloc * GT_BM; t += GT_NT) {
        const int64_t g = split_y + (t / GT_BM) * sstep;
        const int64_t m = g * GT_BM + (t % GT_BM);
        const int64_t ml = m < Mv ? m : Mv - 1;
        lds_rows[t] = a.row_index ? a.row_index[ml] : (int)ml;
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
            const int chunk = (((pos >> 1) ^ (rowl & 3)) << 1) | (pos & 1);          // source-side swizzle of the 32-byte pairs
            const int piece = (16 * wave + 4 * j) * GT_ROW;                          // 1 KB destination of this instruction
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(Ag + pr * a.lda + n0 + chunk * 8), (lds_ptr_t)(st + piece), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(Bg + pr * a.ldb + k0 + chunk * 8), (lds_ptr_t)(st + GT_HALF + piece), 16, 0, 0);
        }
    };

    // per-lane byte offsets of the transposing reads inside a stage (k-step 0; k-step ks adds ks * 16 rows as an immediate):
    // row = 4 * (g >> 1) + (i >> 2), 32-byte pair (c0 / 16 + (g & 1)) ^ (row & 3), 8 * (i & 3) inside the pair
    uint32_t offA[2], offB[2];
    {
        const int g = lane >> 4, i16 = lane & 15;
        const int row = 4 * (g >> 1) + (i16 >> 2);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int pa = ((wn * 64 + t * 32) >> 4) + (g & 1), pb = ((wk * 64 + t * 32) >> 4) + (g & 1);
            offA[t] = (uint32_t)(row * GT_ROW + ((pa ^ (row & 3)) << 5) + 8 * (i16 & 3));
            offB[t] = (uint32_t)(GT_HALF + row * GT_ROW + ((pb ^ (row & 3)) << 5) + 8 * (i16 & 3));
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
    float csum = 0.f;
    const bool do_colsum = a.colsum != nullptr && tid < GT_BN;
    const int cs_step = tiles_k < GT_BM ? tiles_k : GT_BM;

    issue(0);
    if (nloc > 1) issue(1);
    if (nloc > 2) issue(2);
    for (int64_t i = 0; i < nloc; ++i) {
        // slab i landed <=> at most the 8 LDS-DMA loads of each of the (up to two) newer slabs are still outstanding
        if (i + 2 < nloc) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (i + 1 < nloc) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");             // every wave's part landed; everyone is done with stage (i - 1) % 4
        if (i + 3 < nloc) issue(i + 3);
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
        const unsigned char* sa = st;
        if (do_colsum && tile_k < cs_step) {
#pragma unroll 4
            for (int m = tile_k; m < GT_BM; m += cs_step)
                csum += (float)*(const bf16_t*)(sa + m * GT_ROW + ((((tid >> 4) ^ (m & 3))) << 5) + (tid & 15) * 2);
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
    if (do_colsum && tile_k < cs_step) unsafeAtomicAdd(a.colsum + n0 + tid, csum * a.alpha);
}

}  // namespace

// called by made_gemm_tn after validation; returns MADE_ERR_UNSUPPORTED-like sentinel 1 when the fast path does not apply
int made_gemm_tn_fast(const MadeGemmTNArgs& a, hipStream_t st) {
    const bool ok = a.ab_dtype == MADE_BF16 && a.N % GT_BN == 0 && a.K % GT_BK == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 &&
                    a.batch1 * a.batch2 == 1 && (a.row_mask == nullptr || a.row_index != nullptr) &&
                    ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.B % 16) == 0 && a.M >= 4 * GT_BM &&
                    ((a.M + GT_BM - 1) / GT_BM + a.split_m - 1) / a.split_m * GT_BM <= GT_MAX_ROWS;
    if (!ok) return 1;
    const int64_t tiles = (a.N / GT_BN) * (a.K / GT_BK);
    dim3 grid((unsigned)(8 * tiles * ((a.split_m + 7) / 8)), 1, 1);
    hipLaunchKernelGGL(gemm_tn_glds_kernel, grid, dim3(GT_NT), 0, st, a);
    return made_check_launch("made_gemm_tn(glds)");
}
