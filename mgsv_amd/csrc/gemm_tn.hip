// made_gemm_tn: C[N,K] (+)= alpha * sum_m A[m,n] * B[m,k]  -- the weight-gradient product of the training path
// (dW = dY^T X: both operands are stored with the REDUCTION index m as the slow dimension, i.e. as the forward pass
// left them in HBM), gfx950.
//
// One workgroup = 4 waves = one 128 x 128 tile of C; wave (wn, wk) owns a 64 x 64 quadrant = 2 x 2 MFMA 32x32 tiles.
// A and B slabs of BM reduction rows are staged global -> registers -> LDS row-major exactly as they lie in HBM
// (coalesced 16-byte chunks; the next slab's loads are in flight during the current slab's MFMAs); the MFMA operand
// fragments (row index on the lane, reduction index along k) come out of LDS through ds_read_b64_tr_b16, the hardware
// transposing read (bf16), or plain 4-byte reads with the lane on the row index (f32).  No transposed copy of an
// activation is ever written.  The reduction can be split over workgroups (split_m) and over a batch whose C stride is
// 0; partial tiles are then combined with f32 atomic adds (accumulate = 1), which is also how gradients of a weight used
// at several places add up.  Optionally the column sums of A (the bias gradient) are accumulated by the k-tile-0 blocks.
//   bf16 : v_mfma_f32_32x32x16_bf16,  f32 : v_mfma_f32_32x32x2_f32 (exact f32; parity mode)
#include "common.h"

namespace {

__device__ const float kOne = 1.f;     // the "row mask" read when there is none (stride 0)

constexpr int TBN = 128, TBK = 128, TNT = 256;

template <typename TC> struct TnCfg;
// row pitch: 4 consecutive rows must land on disjoint quarters of the 64 LDS banks for the transposing reads
template <> struct TnCfg<bf16_t> { static constexpr int BM = 64, ROW = 128 * 2 + 64, CPR = 16; typedef bf16x8 frag_t; };
// f32: the two reduction rows a wave reads per MFMA must sit 32 banks apart
template <> struct TnCfg<float>  { static constexpr int BM = 32, ROW = 128 * 4 + 128, CPR = 32; typedef f32x4 frag_t; };

template <typename TC, bool X3 = false>        // X3 (f32 only): split-bf16 products (common.h, made_set_f32_products)
__global__ __launch_bounds__(TNT) void gemm_tn_kernel(const MadeGemmTNArgs a) {
    typedef TnCfg<TC> Cfg;
    typedef typename Cfg::frag_t frag_t;
    constexpr int BM = Cfg::BM, ROW = Cfg::ROW, CPR = Cfg::CPR;
    constexpr int PER16 = 16 / (int)sizeof(TC);
    constexpr int NCH = BM * CPR / TNT;             // 16-byte chunks per thread per operand (= 4)
    constexpr bool IS_BF16 = sizeof(TC) == 2;

    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BM * ROW];
    unsigned char* lds_a = lds;
    unsigned char* lds_b = lds + BM * ROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const int tiles_k = (int)((a.K + TBK - 1) / TBK);
    const int tile_n = blockIdx.x / tiles_k, tile_k = blockIdx.x % tiles_k;
    const int64_t n0 = (int64_t)tile_n * TBN, k0 = (int64_t)tile_k * TBK;
    const int64_t z = blockIdx.z, z1 = z / a.batch2, z2 = z % a.batch2;

    const TC* Ag = (const TC*)a.A + z1 * a.a_zs1 + z2 * a.a_zs2;
    const TC* Bg = (const TC*)a.B + z1 * a.b_zs1 + z2 * a.b_zs2;
    const float* maskg = a.row_mask ? a.row_mask + z1 * a.mask_zs1 + z2 * a.mask_zs2 : nullptr;

    // this block's share of the reduction: slabs blockIdx.y, blockIdx.y + split_m, ... (interleaved, so that runs of padded
    // rows -- whose slabs are skipped -- spread evenly over the blocks); local slab i <-> global slab blockIdx.y + i * split_m
    int64_t Mv = a.M;                                       // row gather: the reduction runs over the valid rows only
    if (a.n_rows) { const int64_t nv = *a.n_rows; Mv = nv < a.M ? nv : a.M; }
    const int64_t nslab = (Mv + BM - 1) / BM;
    const int64_t sstep = a.split_m;
    const int64_t s_begin = 0;
    const int64_t s_end = nslab > (int64_t)blockIdx.y ? (nslab - blockIdx.y + sstep - 1) / sstep : 0;
    if (s_begin >= s_end) return;
    auto gslab = [&](int64_t i) __attribute__((always_inline)) { return (int64_t)blockIdx.y + i * sstep; };

    // slabs made only of masked rows contribute nothing: with row_group_valid (one flag per 32 rows, made_row_groups) the
    // block learns which of its slabs to skip from ONE load + ballot (bit i = slab s_begin + i has a valid row)
    uint64_t slab_bits = ~0ull;
    if (a.row_group_valid && (s_end - s_begin) <= 64 && a.batch1 * a.batch2 == 1) {
        const int64_t sl = lane < s_end ? gslab(lane) : nslab;
        bool any = false;
        if (lane < s_end) {
            constexpr int GP = BM / 32;                     // 32-row groups per slab
            const int64_t ngroups = (a.M + 31) / 32;
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                const int64_t gi = sl * GP + g;
                any = any || (gi < ngroups && a.row_group_valid[gi] != 0.f);
            }
        }
        slab_bits = __ballot(any);
    }
    auto slab_live = [&](int64_t i) __attribute__((always_inline)) { return i >= 64 || ((slab_bits >> i) & 1ull) != 0; };

    frag_t ra[NCH], rb[NCH];
    int64_t prow[NCH];                                      // physical rows of the slab that is loaded next (fetched one slab ahead)
    auto load_rows = [&](int64_t slab) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * TNT;
            const int64_t m = slab * BM + c / CPR;
            const int64_t ml = m < Mv ? m : Mv - 1;
            prow[i] = a.row_index ? (int64_t)a.row_index[ml] : ml;
        }
    };
    // load_slab only ISSUES the loads of the next slab (clamped addresses, the row mask read unconditionally: stride 0 over a
    // constant 1.0 when there is none); rows past M, columns past N / K and masked rows are zeroed in store_slab, one slab later,
    // after the current slab's MFMAs -- a use right behind the loads would make the wave wait for them before multiplying
    float rmk[NCH];
    int64_t rslab = 0;
    const float* mrow = maskg ? maskg : &kOne;
    const int64_t mstride = maskg ? 1 : 0;
    auto load_slab = [&](int64_t slab) __attribute__((always_inline)) {
        rslab = slab;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * TNT;
            const int cc = c % CPR;
            const int64_t mc = prow[i];
            int64_t na = n0 + cc * PER16, kb = k0 + cc * PER16;
            const int64_t nac = na < a.N ? na : 0, kbc = kb < a.K ? kb : 0;     // a chunk may run past N / K inside the row pitch
            ra[i] = *(const frag_t*)(Ag + mc * a.lda + nac);
            rb[i] = *(const frag_t*)(Bg + mc * a.ldb + kbc);
            rmk[i] = mrow[mc * mstride];
        }
    };
    auto store_slab = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * TNT;
            const int row = c / CPR, cc = c % CPR;
            const bool rowok = (rslab * BM + row) < Mv && rmk[i] != 0.f;
            // (columns >= N only feed C rows that are not stored)
            *(frag_t*)(lds_a + row * ROW + cc * 16) = keep_or_zero(ra[i], rowok && (n0 + cc * PER16 < a.N));
            *(frag_t*)(lds_b + row * ROW + cc * 16) = keep_or_zero(rb[i], rowok && (k0 + cc * PER16 < a.K));
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float csum = 0.f;
    // bias gradient: the k-tiles of one n-tile share the work (rows tile_k, tile_k + tiles_k, ... of every slab), so no block
    // carries the column sums alone and finishes late
    const bool do_colsum = a.colsum != nullptr && tid < TBN;
    const int cs_step = tiles_k < BM ? tiles_k : BM;

    load_rows(gslab(s_begin));
    if (slab_live(s_begin)) load_slab(gslab(s_begin));
    if (s_begin + 1 < s_end) load_rows(gslab(s_begin + 1));
    for (int64_t s = s_begin; s < s_end; ++s) {
        const bool live = slab_live(s);                     // block-uniform
        if (live) {
            __syncthreads();
            store_slab();
            __syncthreads();
        }
        if (s + 1 < s_end) {
            if (slab_live(s + 1)) load_slab(gslab(s + 1));
            if (s + 2 < s_end) load_rows(gslab(s + 2));
        }
        if (!live) continue;

        if (do_colsum && tile_k < cs_step) {
#pragma unroll 4
            for (int m = tile_k; m < BM; m += cs_step) csum += to_f32(*(const TC*)(lds_a + m * ROW + tid * (int)sizeof(TC)));
        }
        if constexpr (IS_BF16) {
            const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
            for (int ks = 0; ks < BM / 16; ++ks) {
                // reduction rows of this k-step: slots j = 0..7 of lane half hh <-> row 16 ks + 8 (j >> 2) + 4 hh + (j & 3)
                const int kb = 16 * ks + 4 * (g >> 1) + (i16 >> 2);
                const int cofs = ((g & 1) * 16 + 4 * (i16 & 3)) * 2;
                bf16x8 af[2], bfr[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const unsigned char* pa = lds_a + kb * ROW + (wn * 64 + t * 32) * 2 + cofs;
                    const unsigned char* pb = lds_b + kb * ROW + (wk * 64 + t * 32) * 2 + cofs;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)pa);
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(pa + 8 * ROW));
                    af[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)pb);
                    hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(pb + 8 * ROW));
                    bfr[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (X3) {
            // split-bf16 products: sixteen reduction rows per product -- lane half hh takes rows 4 hh .. + 3 and 8 + 4 hh .. + 3 of the step
#pragma unroll 2
            for (int ks = 0; ks < BM / 16; ++ks) {
                SplitF32x4 a0[2], a1[2], b0[2], b1[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 x0, x1, y0, y1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m0 = 16 * ks + 4 * hh + j, m1 = m0 + 8;
                        x0[j] = *(const float*)(lds_a + m0 * ROW + (wn * 64 + t * 32 + r) * 4);
                        x1[j] = *(const float*)(lds_a + m1 * ROW + (wn * 64 + t * 32 + r) * 4);
                        y0[j] = *(const float*)(lds_b + m0 * ROW + (wk * 64 + t * 32 + r) * 4);
                        y1[j] = *(const float*)(lds_b + m1 * ROW + (wk * 64 + t * 32 + r) * 4);
                    }
                    a0[t] = made_split4(x0); a1[t] = made_split4(x1); b0[t] = made_split4(y0); b1[t] = made_split4(y1);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = made_mfma_x3_16(a0[i], a1[i], b0[j], b1[j], acc[i][j]);
            }
        } else {
#pragma unroll 4
            for (int ks = 0; ks < BM / 2; ++ks) {
                const int m = 2 * ks + hh;
                float af[2], bfr[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    af[t] = *(const float*)(lds_a + m * ROW + (wn * 64 + t * 32 + r) * 4);
                    bfr[t] = *(const float*)(lds_b + m * ROW + (wk * 64 + t * 32 + r) * 4);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: C[n][k], n on the accumulator rows, k on the lanes (128-byte runs per half-wave)
    const int64_t cz = z1 * a.c_zs1 + z2 * a.c_zs2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t k = k0 + wk * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t n = n0 + wn * 64 + i * 32 + acc_row(e, hh);
                if (n < a.N && k < a.K) {
                    const float v = acc[i][j][e] * a.alpha;
                    if (a.c_dtype == MADE_F32) {
                        float* p = (float*)a.C + cz + n * a.ldc + k;
                        if (a.accumulate) unsafeAtomicAdd(p, v); else *p = v;
                    } else {
                        ((bf16_t*)a.C)[cz + n * a.ldc + k] = (bf16_t)v;
                    }
                }
            }
        }
    if (do_colsum && n0 + tid < a.N)
        unsafeAtomicAdd(a.colsum + z1 * a.colsum_zs1 + z2 * a.colsum_zs2 + n0 + tid, csum * a.alpha);
}

// shapes the tiled kernel cannot address with 16-byte chunks (tiny heads: N = 2 classes / span coordinates):
// one thread per C element, the reduction split over blockIdx.y
__global__ __launch_bounds__(256) void gemm_tn_small_kernel(const MadeGemmTNArgs a) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t z = blockIdx.z, z1 = z / a.batch2, z2 = z % a.batch2;
    const bool active = e < a.N * a.K;
    const int64_t n = active ? e / a.K : 0, k = active ? e % a.K : 0;
    int64_t Mv = a.M;
    if (a.n_rows) { const int64_t nv = *a.n_rows; Mv = nv < a.M ? nv : a.M; }
    const int64_t per = (Mv + gridDim.y - 1) / gridDim.y;
    const int64_t m0 = (int64_t)blockIdx.y * per, m1 = m0 + per < Mv ? m0 + per : Mv;
    const float* maskg = a.row_mask ? a.row_mask + z1 * a.mask_zs1 + z2 * a.mask_zs2 : nullptr;
    const int64_t ao = z1 * a.a_zs1 + z2 * a.a_zs2, bo = z1 * a.b_zs1 + z2 * a.b_zs2;
    float acc = 0.f, cs = 0.f;
    for (int64_t ml = m0; ml < m1; ++ml) {
        const int64_t m = a.row_index ? (int64_t)a.row_index[ml] : ml;
        if (maskg && maskg[m] == 0.f) continue;
        const float av = load_as_f32(a.A, a.ab_dtype, ao + m * a.lda + n);
        acc += av * load_as_f32(a.B, a.ab_dtype, bo + m * a.ldb + k);
        cs += av;
    }
    if (!active) return;
    const int64_t co = z1 * a.c_zs1 + z2 * a.c_zs2 + n * a.ldc + k;
    if (a.c_dtype == MADE_F32) {
        if (a.accumulate) unsafeAtomicAdd((float*)a.C + co, acc * a.alpha); else ((float*)a.C)[co] = acc * a.alpha;
    } else {
        ((bf16_t*)a.C)[co] = (bf16_t)(acc * a.alpha);
    }
    if (a.colsum && k == 0) unsafeAtomicAdd(a.colsum + z1 * a.colsum_zs1 + z2 * a.colsum_zs2 + n, cs * a.alpha);
}

}  // namespace

int made_gemm_tn_fast(const MadeGemmTNArgs& a, hipStream_t st);      // gemm_tn_glds.hip

extern "C" int made_gemm_tn(const MadeGemmTNArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_gemm_tn: null args");
    MadeGemmTNArgs a = *args;
    MADE_REQUIRE(a.A && a.B && a.C, "made_gemm_tn: null tensor");
    MADE_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0, "made_gemm_tn: bad dims");
    MADE_REQUIRE(a.ab_dtype == MADE_F32 || a.ab_dtype == MADE_BF16, "made_gemm_tn: bad operand dtype %d", a.ab_dtype);
    MADE_REQUIRE(a.c_dtype == MADE_F32 || a.c_dtype == a.ab_dtype, "made_gemm_tn: C must be f32 or the operand dtype");
    if (a.batch1 <= 0) a.batch1 = 1;
    if (a.batch2 <= 0) a.batch2 = 1;
    if (a.split_m <= 0) a.split_m = 1;
    const int64_t nz = a.batch1 * a.batch2;
    MADE_REQUIRE(a.split_m == 1 || a.accumulate, "made_gemm_tn: split_m > 1 needs accumulate = 1 (partials are added atomically)");
    MADE_REQUIRE(!a.accumulate || a.c_dtype == MADE_F32, "made_gemm_tn: accumulation needs an f32 C");
    MADE_UNSUPPORTED(nz < 65536 && a.split_m < 65536, "made_gemm_tn: batch / split too large for the grid");
    MADE_REQUIRE(a.row_group_valid == nullptr || a.row_mask != nullptr, "made_gemm_tn: row_group_valid without row_mask");
    MADE_REQUIRE((a.row_index == nullptr) == (a.n_rows == nullptr), "made_gemm_tn: row_index and n_rows come together");
    MADE_UNSUPPORTED(a.row_index == nullptr || nz == 1, "made_gemm_tn: row gather is for unbatched calls");
    if (a.row_index) a.row_group_valid = nullptr;
    if (a.M == 0) return MADE_OK;                                   /* nothing to add (C is not cleared: callers zero gradients) */
    hipStream_t st = (hipStream_t)stream;
    const int per16 = a.ab_dtype == MADE_F32 ? 4 : 8;
    const int64_t n_up = (a.N + per16 - 1) / per16 * per16, k_up = (a.K + per16 - 1) / per16 * per16;
    const bool aligned = a.lda % per16 == 0 && a.ldb % per16 == 0 && n_up <= a.lda && k_up <= a.ldb &&
                         a.a_zs1 % per16 == 0 && a.a_zs2 % per16 == 0 && a.b_zs1 % per16 == 0 && a.b_zs2 % per16 == 0 &&
                         ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.B % 16) == 0;
    if (!aligned) {
        MADE_UNSUPPORTED(a.N * a.K <= (1 << 20), "made_gemm_tn: unaligned operands are only supported for small outputs");
        int sy = (int)a.split_m;
        if (!a.accumulate) sy = 1;
        dim3 grid((unsigned)((a.N * a.K + 255) / 256), (unsigned)sy, (unsigned)nz);
        hipLaunchKernelGGL(gemm_tn_small_kernel, grid, dim3(256), 0, st, a);
        return made_check_launch("made_gemm_tn(small)");
    }
    const int64_t tiles = ((a.N + TBN - 1) / TBN) * ((a.K + TBK - 1) / TBK);
    MADE_UNSUPPORTED(tiles < (1LL << 31), "made_gemm_tn: too many tiles");
    {
        const int rc = made_gemm_tn_fast(a, st);                     // direct-to-LDS 3-stage kernel when the shapes allow
        if (rc != 1) return rc;
    }
    dim3 grid((unsigned)tiles, (unsigned)a.split_m, (unsigned)nz);
    if (a.ab_dtype == MADE_BF16) hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, grid, dim3(TNT), 0, st, a);
    else if (g_made_f32_products) hipLaunchKernelGGL((gemm_tn_kernel<float, true>), grid, dim3(TNT), 0, st, a);
    else hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, dim3(TNT), 0, st, a);
    return made_check_launch("made_gemm_tn");
}

namespace {
__global__ __launch_bounds__(256) void row_groups_kernel(const float* mask, int64_t M, float* out) {
    const int lane = threadIdx.x & 63;
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;      // one half-wave per 32-row group
    const int64_t ngroups = (M + 31) / 32;
    const int64_t row = g * 32 + (lane & 31);
    const bool v = g < ngroups && row < M && mask[row] != 0.f;
    const uint64_t bal = __ballot(v);
    const uint32_t half = (lane >> 5) ? (uint32_t)(bal >> 32) : (uint32_t)bal;
    if ((lane & 31) == 0 && g < ngroups) out[g] = half != 0 ? 1.f : 0.f;
}
}  // namespace

extern "C" int made_row_groups(const float* mask, int64_t M, float* out, void* stream) {
    MADE_REQUIRE(mask && out && M > 0, "made_row_groups: bad arguments");
    const int64_t ngroups = (M + 31) / 32;
    hipLaunchKernelGGL(row_groups_kernel, dim3((unsigned)((ngroups * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask, M, out);
    return made_check_launch("made_row_groups");
}

namespace {
// stream compaction of a token mask in one workgroup of 16 waves: each wave owns a contiguous range, counts it with
// ballots, the 16 totals are scanned, then the indices are scattered.  The range is walked in batches of 16 x 64 elements whose
// 16 (coalesced) loads are all issued before the first ballot: a dependent load per 64 elements made this a 17 us kernel at the
// head of every step's critical path.
__global__ __launch_bounds__(1024) void row_index_kernel(const float* mask, int64_t M, int32_t* row_index, int32_t* n_rows) {
    constexpr int NB = 16;
    __shared__ int wsum[16];
    __shared__ int wlast[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t per = ((M + 15) / 16 + 63) / 64 * 64;          // elements per wave, multiple of 64
    const int64_t b = (int64_t)wave * per, e = b + per < M ? b + per : M;
    uint64_t bal[NB];
    auto ballots = [&](int64_t i0) __attribute__((always_inline)) {
        float v[NB];
        const float* p = mask + i0 + lane;
        if (i0 + NB * 64 <= e) {
#pragma unroll
            for (int j = 0; j < NB; ++j) v[j] = p[j * 64];
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) v[j] = i0 + j * 64 + lane < e ? p[j * 64] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) bal[j] = __ballot(v[j] != 0.f);
    };
    int cnt = 0, last = -1;
    for (int64_t i0 = b; i0 < e; i0 += NB * 64) {
        ballots(i0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            cnt += __popcll(bal[j]);
            if (bal[j]) last = (int)(i0 + j * 64) + 63 - __clzll(bal[j]);
        }
    }
    if (lane == 0) { wsum[wave] = cnt; wlast[wave] = last; }
    __syncthreads();
    int base = 0, total = 0, lastv = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) base += wsum[w];
        total += wsum[w];
        if (wlast[w] >= 0) lastv = wlast[w];
    }
    for (int64_t i0 = b; i0 < e; i0 += NB * 64) {
        ballots(i0);                                   // the second look at the range comes out of L2
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if ((bal[j] >> lane) & 1ull) row_index[base + __popcll(bal[j] & ((1ull << lane) - 1ull))] = (int32_t)(i0 + j * 64 + lane);
            base += __popcll(bal[j]);
        }
    }
    if (threadIdx.x == 0) n_rows[0] = total;
    for (int64_t i = total + threadIdx.x; i < M; i += 1024) row_index[i] = lastv;
}

// valid length of every sample (one wave per sample, ballots over batches of 8 x 64 entries loaded together), then rank by
// (length descending, index ascending)
__global__ __launch_bounds__(1024) void batch_order_kernel(const float* mask, int B, int64_t T, int32_t* order) {
    __shared__ int len[8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b0 = wave; b0 < B; b0 += 64) {            // four samples of this wave at a time: their loads travel together
        int cnt[4] = {0, 0, 0, 0};
        for (int64_t i0 = 0; i0 < T; i0 += 512) {
            float v[4][8];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int b = b0 + 16 * s4;
                const float* m = mask + (int64_t)(b < B ? b : B - 1) * T;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int64_t i = i0 + j * 64 + lane;
                    v[s4][j] = i < T ? m[i] : 0.f;
                }
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int j = 0; j < 8; ++j) cnt[s4] += __popcll(__ballot(v[s4][j] != 0.f));
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
            if (lane == 0 && b0 + 16 * s4 < B) len[b0 + 16 * s4] = cnt[s4];
    }
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += 1024) {
        const int mine = len[b];
        int rank = 0;
        for (int j = 0; j < B; ++j) {
            const int l = len[j];
            rank += (l > mine || (l == mine && j < b)) ? 1 : 0;
        }
        order[rank] = b;
    }
}
}  // namespace

extern "C" int made_batch_order(const float* mask, int64_t B, int64_t T, int32_t* order, void* stream) {
    MADE_REQUIRE(mask && order && B > 0 && T > 0, "made_batch_order: bad arguments");
    MADE_UNSUPPORTED(B <= 8192, "made_batch_order: B=%lld exceeds the single-workgroup table (8192)", (long long)B);
    hipLaunchKernelGGL(batch_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, (int)B, T, order);
    return made_check_launch("made_batch_order");
}

extern "C" int made_row_index(const float* mask, int64_t M, int32_t* row_index, int32_t* n_rows, void* stream) {
    MADE_REQUIRE(mask && row_index && n_rows && M > 0, "made_row_index: bad arguments");
    MADE_UNSUPPORTED(M <= (1 << 22), "made_row_index: M=%lld too large for the single-workgroup scan", (long long)M);
    hipLaunchKernelGGL(row_index_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, M, row_index, n_rows);
    return made_check_launch("made_row_index");
}
