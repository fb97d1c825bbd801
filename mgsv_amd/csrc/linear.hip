// made_linear: fused Linear (GEMM + prologue/epilogue) on MFMA, gfx950.
//
// Tile: 128 x 128 outputs per 256-thread workgroup (4 waves, 2 x 2, each 64 x 64 = 2 x 2 MFMA
// 32x32 tiles), K consumed in 128-byte slabs (64 bf16 / 32 f32 per row).  Both operands are
// K-contiguous ([M,K] activations, [N,K] nn.Linear weights), staged global -> registers -> LDS
// (the activation prologue: row mask, +A2, f32->bf16 conversion happens in registers), one slab
// prefetched in registers while the previous one is multiplied.  LDS rows are padded to 144 B so
// the 16-byte fragment reads of a wave are bank-conflict free.
//   bf16 : v_mfma_f32_32x32x16_bf16, one per 16-byte fragment pair
//   f32  : v_mfma_f32_32x32x2_f32, four per fragment pair (exact f32 FMA chain)
// Transposed output segments swap the MFMA operands so that lanes run along the row (time) axis
// and the per-batch transposed store stays coalesced.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int KB = 128;               // bytes of K per row per stage, in the compute type
constexpr int LDS_ROW = KB + 16;      // padded row stride in bytes
constexpr int NTHREADS = 256;

template <typename TC> struct Frag;
template <> struct Frag<float>  { typedef f32x4  type; };
template <> struct Frag<bf16_t> { typedef bf16x8 type; };

template <typename TC>
__device__ __forceinline__ typename Frag<TC>::type zero_frag() {
    typename Frag<TC>::type z;
#pragma unroll
    for (int i = 0; i < elem_traits<TC>::per16; ++i) z[i] = (TC)0.f;
    return z;
}

// Load one 16-byte compute-type fragment of A' = (A [+ A2]) at (row gm, element k).
template <typename TA, typename TC>
__device__ __forceinline__ typename Frag<TC>::type load_a_frag(const TA* __restrict__ A, int64_t lda,
                                                               const TA* __restrict__ A2, int64_t lda2,
                                                               int64_t a2_row_mod, int64_t gm, int64_t k) {
    constexpr int n = elem_traits<TC>::per16;
    float v[n];
    const TA* p = A + gm * lda + k;
    if constexpr (sizeof(TA) == 4) {
#pragma unroll
        for (int i = 0; i < n; i += 4) {
            f32x4 t = *(const f32x4*)(p + i);
            v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
        }
    } else {
        static_assert(n == 8, "bf16 activations need bf16 compute");
        bf16x8 t = *(const bf16x8*)p;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
    }
    if (A2) {
        int64_t r2 = a2_row_mod > 0 ? gm % a2_row_mod : gm;
        const TA* p2 = A2 + r2 * lda2 + k;
        if constexpr (sizeof(TA) == 4) {
#pragma unroll
            for (int i = 0; i < n; i += 4) {
                f32x4 t = *(const f32x4*)(p2 + i);
                v[i] += t[0]; v[i + 1] += t[1]; v[i + 2] += t[2]; v[i + 3] += t[3];
            }
        } else {
            bf16x8 t = *(const bf16x8*)p2;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += (float)t[i];
        }
    }
    typename Frag<TC>::type f;
#pragma unroll
    for (int i = 0; i < n; ++i) f[i] = from_f32<TC>(v[i]);
    return f;
}

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case MADE_ACT_RELU: return fmaxf(x, 0.f);
        case MADE_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
        case MADE_ACT_QUICKGELU: return x / (1.f + expf(-1.702f * x));
        case MADE_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        default: return x;
    }
}

template <typename TA, typename TC>
__global__ __launch_bounds__(NTHREADS, 2) void linear_kernel(const MadeLinearArgs a) {
    typedef typename Frag<TC>::type frag_t;
    constexpr int PER16 = elem_traits<TC>::per16;
    constexpr int KE = KB / (int)sizeof(TC);          // elements of K per stage

    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BM * LDS_ROW];
    unsigned char* lds_a = lds;
    unsigned char* lds_w = lds + BM * LDS_ROW;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int64_t n_tiles = (a.N + BN - 1) / BN;
    const int64_t tile_m = blockIdx.x / n_tiles, tile_n = blockIdx.x % n_tiles;
    const int64_t m0 = tile_m * BM, n0 = tile_n * BN;
    const int64_t z = blockIdx.z;

    // segment of this column tile
    int si = 0;
#pragma unroll
    for (int s = 1; s < 4; ++s)
        if (s < a.nseg && n0 >= a.seg[s].col_begin) si = s;
    const MadeLinearSeg seg = a.seg[si];
    const bool transposed = seg.transposed != 0;

    const TA* A = (const TA*)a.A + z * a.a_z_stride;
    const TA* A2 = (seg.use_a2 && a.A2) ? (const TA*)a.A2 : nullptr;
    const TC* W = (const TC*)a.W + z * a.w_z_stride;

    // staging assignment: 4 chunks of A and 4 of W per thread
    frag_t ra[4], rw[4];
    int srow[4], skc[4];
    bool a_ok[4], w_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + i * NTHREADS;
        srow[i] = c >> 3;
        skc[i] = c & 7;
        int64_t gm = m0 + srow[i];
        a_ok[i] = gm < a.M && (a.a_row_mask == nullptr || a.a_row_mask[gm] != 0.f);
        w_ok[i] = (n0 + srow[i]) < a.N;
    }

    auto load_stage = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t k = k0 + skc[i] * PER16;
            bool kin = k < a.K;
            ra[i] = (a_ok[i] && kin) ? load_a_frag<TA, TC>(A, a.lda, A2, a.lda2, a.a2_row_mod, m0 + srow[i], k)
                                     : zero_frag<TC>();
            rw[i] = (w_ok[i] && kin) ? *(const frag_t*)(W + (n0 + srow[i]) * a.ldw + k) : zero_frag<TC>();
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(frag_t*)(lds_a + srow[i] * LDS_ROW + skc[i] * 16) = ra[i];
            *(frag_t*)(lds_w + srow[i] * LDS_ROW + skc[i] * 16) = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int64_t nk = (a.K + KE - 1) / KE;
    load_stage(0);
    store_stage();
    __syncthreads();
    for (int64_t kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_stage((kt + 1) * KE);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_t fa[2], fw[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *(const frag_t*)(lds_a + (wm * 64 + t * 32 + r) * LDS_ROW + ks * 32 + hh * 16);
                fw[t] = *(const frag_t*)(lds_w + (wn * 64 + t * 32 + r) * LDS_ROW + ks * 32 + hh * 16);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if constexpr (sizeof(TC) == 2) {
                        if (transposed)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nt], fa[mt], acc[mt][nt], 0, 0, 0);
                        else
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt], fw[nt], acc[mt][nt], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (transposed)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[nt][e], fa[mt][e], acc[mt][nt], 0, 0, 0);
                            else
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][e], fw[nt][e], acc[mt][nt], 0, 0, 0);
                        }
                    }
                }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_stage();
            __syncthreads();
        }
    }

    // ---- epilogue -------------------------------------------------------------------------
    // acc[mt][nt][e]: normal     -> row m = (mt, e, hh), col n = (nt, lane)
    //                 transposed -> row m = (mt, lane),  col n = (nt, e, hh)
    unsigned char* outp = (unsigned char*)seg.out;
    const int64_t out_z = z * seg.out_z_stride;
    const int M = (int)a.M, N = (int)a.N;
    const int mbase = (int)m0 + wm * 64, nbase = (int)n0 + wn * 64;
    const int rpb = (int)seg.rows_per_batch, rmod = (int)a.r_row_mod;
    const int colb = (int)seg.col_begin;

    // per-row quantities: validity, output row offset, residual row offset
    auto row_info = [&](int m, bool& ok, bool& zero, int64_t& orow, int64_t& rrow) {
        ok = m < M;
        zero = false; orow = 0; rrow = 0;
        if (!ok) return;
        zero = a.out_row_mask != nullptr && a.out_row_mask[m] == 0.f;
        if (rpb > 0) {
            int b = m / rpb, t = m - b * rpb;
            orow = transposed ? (int64_t)b * seg.out_batch_stride + t
                              : (int64_t)b * seg.out_batch_stride + (int64_t)t * seg.ldo;
        } else {
            orow = transposed ? (int64_t)m : (int64_t)m * seg.ldo;
        }
        int rr = rmod > 0 ? m % rmod : m;
        rrow = (int64_t)rr * a.ldr;
    };
    auto finish = [&](float v, int n, bool zero, int64_t orow, int64_t rrow) {
        if (a.bias) v += a.bias[n];
        v = apply_act(v, a.act);
        if (a.R) v += load_as_f32(a.R, a.r_dtype, rrow + n);
        if (zero) v = 0.f;
        int col = n - colb;
        int64_t off = transposed ? orow + (int64_t)col * seg.ldo : orow + col;
        store_from_f32(outp, seg.out_dtype, out_z + off, v);
    };

    if (!transposed) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                bool ok, zero; int64_t orow, rrow;
                row_info(mbase + mt * 32 + acc_row(e, hh), ok, zero, orow, rrow);
                if (ok) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        int n = nbase + nt * 32 + r;
                        if (n < N) finish(acc[mt][nt][e], n, zero, orow, rrow);
                    }
                }
            }
    } else {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            bool ok, zero; int64_t orow, rrow;
            row_info(mbase + mt * 32 + r, ok, zero, orow, rrow);
            if (ok) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        int n = nbase + nt * 32 + acc_row(e, hh);
                        if (n < N) finish(acc[mt][nt][e], n, zero, orow, rrow);
                    }
            }
        }
    }
}

}  // namespace

extern "C" int made_linear(const MadeLinearArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_linear: null args");
    const MadeLinearArgs& a = *args;
    MADE_REQUIRE(a.A && a.W, "made_linear: null A or W");
    MADE_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0, "made_linear: bad dims M=%lld N=%lld K=%lld",
                 (long long)a.M, (long long)a.N, (long long)a.K);
    MADE_REQUIRE(a.nseg >= 1 && a.nseg <= 4, "made_linear: nseg=%d out of range", a.nseg);
    MADE_REQUIRE(a.batch >= 1 && a.batch <= 65535, "made_linear: batch=%lld out of range", (long long)a.batch);
    MADE_REQUIRE(a.w_dtype == MADE_F32 || a.w_dtype == MADE_BF16, "made_linear: bad w_dtype %d", a.w_dtype);
    MADE_UNSUPPORTED(!(a.a_dtype == MADE_BF16 && a.w_dtype == MADE_F32),
                     "made_linear: bf16 activations with f32 weights are not supported");
    const int per16 = a.w_dtype == MADE_F32 ? 4 : 8;
    const int a_align = a.a_dtype == MADE_F32 ? 4 : 8;
    MADE_UNSUPPORTED(a.K % per16 == 0, "made_linear: K=%lld must be a multiple of %d", (long long)a.K, per16);
    MADE_UNSUPPORTED(a.lda % a_align == 0 && a.ldw % per16 == 0 && ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0,
                     "made_linear: A/W rows must be 16-byte aligned (lda=%lld ldw=%lld)", (long long)a.lda, (long long)a.ldw);
    if (a.A2)
        MADE_UNSUPPORTED(a.lda2 % a_align == 0 && ((uintptr_t)a.A2 % 16) == 0, "made_linear: A2 rows must be 16-byte aligned");
    if (a.batch > 1)
        MADE_UNSUPPORTED(a.a_z_stride % a_align == 0 && a.w_z_stride % per16 == 0, "made_linear: batch strides must keep 16-byte alignment");
    for (int s = 0; s < a.nseg; ++s) {
        MADE_REQUIRE(a.seg[s].out != nullptr, "made_linear: segment %d has null out", s);
        MADE_REQUIRE(a.seg[s].col_begin >= 0 && a.seg[s].col_begin < a.N, "made_linear: segment %d col_begin out of range", s);
        if (s > 0) {
            MADE_REQUIRE(a.seg[s].col_begin > a.seg[s - 1].col_begin, "made_linear: segments must be ascending");
            MADE_UNSUPPORTED(a.seg[s].col_begin % BN == 0, "made_linear: segment boundaries must be multiples of %d", BN);
        } else {
            MADE_REQUIRE(a.seg[0].col_begin == 0, "made_linear: first segment must start at column 0");
        }
    }
    if (a.M == 0) return MADE_OK;
    const int64_t tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    MADE_UNSUPPORTED(tiles < (1LL << 31), "made_linear: too many tiles");
    dim3 grid((unsigned)tiles, 1, (unsigned)a.batch), block(NTHREADS);
    hipStream_t st = (hipStream_t)stream;
    if (a.w_dtype == MADE_BF16) {
        if (a.a_dtype == MADE_F32) hipLaunchKernelGGL((linear_kernel<float, bf16_t>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((linear_kernel<bf16_t, bf16_t>), grid, block, 0, st, a);
    } else {
        hipLaunchKernelGGL((linear_kernel<float, float>), grid, block, 0, st, a);
    }
    return made_check_launch("made_linear");
}
