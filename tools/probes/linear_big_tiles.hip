// EXPERIMENT, not built into libmade_hip.so (profiles/r01_g_linear_big_tile_experiment.txt).
// made_linear with 256 x 256 tiles, one workgroup per CU.  This is a fragment of mgsv_amd/csrc/linear.hip (it uses that file's
// helpers: swz, epilogue8, store8, KB, NTHREADS) kept for the next attempt.  Parity-green when it was wired in
// (tests/test_ops_gpu.py::test_linear_big_tiles), but 50 us against the 64 x 128 / 128 x 128 kernels' 34-36 us on
// 32768 x 512 x 512: its K loop alone takes 20 us (the tiled kernels' whole launch: 34), and the epilogue -- 17-22 us of
// instruction latency with one wave per SIMD and nothing to overlap it with -- eats the gain.

// =================================================================================================
// Encoder-sized problems (tens of thousands of rows): 256 x 256 tiles.
// What bounds the 64 x 128 / 128 x 128 kernels above on these shapes is neither MFMA nor HBM but what a CU can take in through
// its L1: every tile re-fetches its A and W panels (M N K 2 (1/TN + 1/TM) bytes in all: 400 MB for 32768 x 512 x 512 at
// 64 x 128) and a CU sustains ~45 GB/s of such reads (1.5 MB per CU in 36 us measured).  A 256 x 256 tile moves a third of
// those bytes.  One workgroup per CU (all 160 KB of LDS): 4 waves of 128 x 128 (16 accumulator tiles each = 256 AGPRs),
// 64-deep K slabs in the sub-tiled image of linear_glds_kernel (full 128-byte lines: 64-byte rows fetched every line twice),
// A in a three-stage ring (it comes from HBM / Infinity Cache: two slabs ahead), W in a two-stage ring (it is L2-resident:
// one slab ahead), one barrier per slab.  The epilogue is ONE copy of the code run four times (64 rows each): unrolled four
// times it no longer fits the instruction cache, and with one wave per SIMD nothing hides the misses (32 us measured).
constexpr int Q_BM = 256, Q_BN = 256;
constexpr int Q_PART = Q_BM * KB;                                       // 32 KB: 256 rows x 128 bytes of one slab
constexpr int Q_NSA = 3, Q_NSW = 2;
constexpr int Q_CT_LD = Q_BN + 4;
constexpr int Q_LDS = (Q_NSA + Q_NSW) * Q_PART;                         // 160 KB

template <bool TRAIN>
__global__ __launch_bounds__(NTHREADS, 1) void linear_glds256_kernel(const MadeLinearArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qlds[];
    unsigned char* lds_a = qlds;
    unsigned char* lds_w = qlds + Q_NSA * Q_PART;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int M = (int)a.M, N = (int)a.N, K = (int)a.K;
    const int n_tiles = (N + Q_BN - 1) / Q_BN;
    int Mv = M;
    if (a.n_rows) { const int nv = *a.n_rows; Mv = nv < M ? nv : M; }
    const int nwg = ((Mv + Q_BM - 1) / Q_BM) * n_tiles;
    if ((int)blockIdx.x >= nwg) return;
    int tile_id;                                           // XCD-aware order, see linear_glds_kernel
    {
        const int xcd = blockIdx.x & 7, q = nwg >> 3, rem = nwg & 7;
        tile_id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
    }
    const int tile_m = tile_id / n_tiles, tile_n = tile_id % n_tiles;
    const int m0 = tile_m * Q_BM, n0 = tile_n * Q_BN;
    int si = 0;
#pragma unroll
    for (int s = 1; s < 4; ++s)
        if (s < a.nseg && n0 >= a.seg[s].col_begin) si = s;
    const MadeLinearSeg seg = a.seg[si];

    // ---- LDS-DMA sources: wave w issues the 1 KB pieces 8w..8w+7 (rows 64w..64w+63) of the A part and of the W part of
    // every slab; lane l -> row 8j + l/8, LDS slot l%8 holding global chunk (l%8) ^ swz(row)
    const bool use2 = seg.use_a2 && a.A2 && a.a2_replace;
    const bf16_t* Abase = use2 ? (const bf16_t*)a.A2 : (const bf16_t*)a.A;
    const int64_t lda = use2 ? a.lda2 : a.lda;
    const bf16_t* pa[8];
    const bf16_t* pw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 8 * (8 * wave + i) + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        int gm = m0 + row; gm = gm < Mv ? gm : Mv - 1;            // rows past the edge are fetched from a valid row, never stored
        if (a.row_index) gm = a.row_index[gm];
        pa[i] = Abase + (int64_t)gm * lda + chunk * 8;
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
        pw[i] = (const bf16_t*)a.W + (int64_t)gn * a.ldw + chunk * 8;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    auto issue_a = [&](int kt) __attribute__((always_inline)) {
        unsigned char* st = lds_a + (kt % Q_NSA) * Q_PART + wave * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pa[i] + kt * 64), (lds_ptr_t)(st + i * 1024), 16, 0, 0);
    };
    auto issue_w = [&](int kt) __attribute__((always_inline)) {
        unsigned char* st = lds_w + (kt % Q_NSW) * Q_PART + wave * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pw[i] + kt * 64), (lds_ptr_t)(st + i * 1024), 16, 0, 0);
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int offa[4], offw[4], sa[4], sw[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ra = wm * 128 + t * 32 + r, rw = wn * 128 + t * 32 + r;
        offa[t] = ra * KB; sa[t] = swz(ra);
        offw[t] = rw * KB; sw[t] = swz(rw);
    }

    // Issue order W(kt+1), A(kt+2) per trip (W(0), A(0), A(1) up front): when slab kt is needed the only requests that may
    // still be outstanding are the 8 of A(kt+1).  The barrier also proves every wave has finished with slab kt-1, whose
    // stages W(kt+1) and A(kt+2) overwrite.
    const int nk = a.split_k == -2 ? 0 : K / 64;   // DBG
    if (nk > 0) { issue_w(0);
    issue_a(0); }
    if (nk > 1) issue_a(1);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (kt + 1 < nk) issue_w(kt + 1);
        if (kt + 2 < nk) issue_a(kt + 2);
        const unsigned char* sta = lds_a + (kt % Q_NSA) * Q_PART;
        const unsigned char* stw = lds_w + (kt % Q_NSW) * Q_PART;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[4], fw[4];
            const int c = 2 * ks + hh;
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = *(const bf16x8*)(sta + offa[t] + ((c ^ sa[t]) << 4));
#pragma unroll
            for (int t = 0; t < 4; ++t) fw[t] = *(const bf16x8*)(stw + offw[t] + ((c ^ sw[t]) << 4));
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt], fw[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    __syncthreads();                                       // the epilogue reuses the staging LDS
    if (a.split_k == -1) return;   // DBG
    if (a.split_k == -3 && blockIdx.x >= 64) return;   // DBG

    // ---- epilogue: four passes of 64 rows through LDS (one copy of the code); 32 threads per row, 8 consecutive columns each
    float* Ct = (float*)qlds;
    unsigned char* outp = (unsigned char*)seg.out;
    const int rpb = (int)seg.rows_per_batch, rmod = (int)a.r_row_mod;
    const int colb = (int)seg.col_begin;
    const int odt = seg.out_dtype;
    const int cc = tid & 31;
    const int n = n0 + cc * 8;
    int nvalid = N - n; nvalid = nvalid > 8 ? 8 : nvalid;
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = (a.bias && j < nvalid) ? a.bias[n + j] : 0.f;
    const bool out_vec = (seg.ldo % 8 == 0) && (seg.out_batch_stride % 8 == 0) && (((uintptr_t)outp & 15) == 0) && (colb % 8 == 0);
    const bool r_vec = a.R && (a.ldr % 8 == 0) && (((uintptr_t)a.R & 15) == 0);
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        if (m0 + pass * 64 >= Mv) break;                   // block-uniform
        if (wm == (pass >> 1)) {
            const bool hi = pass & 1;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        Ct[(mh * 32 + acc_row(e, hh)) * Q_CT_LD + wn * 128 + nt * 32 + r] = hi ? acc[2 + mh][nt][e] : acc[mh][nt][e];
        }
        __syncthreads();
        if (nvalid > 0) {
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
                const int row = (tid >> 5) + 8 * i;
                const int ml = m0 + pass * 64 + row;
                if (ml >= Mv) break;
                const int m = a.row_index ? a.row_index[ml] : ml;
                const float* cp = Ct + row * Q_CT_LD + cc * 8;
                f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
                float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                epilogue8<TRAIN>(a, m, n, nvalid, v, bv, rmod, r_vec);
                int64_t orow;
                if (rpb > 0) {
                    const int b = m / rpb, t = m - b * rpb;
                    orow = (int64_t)b * seg.out_batch_stride + (int64_t)t * seg.ldo;
                } else {
                    orow = (int64_t)m * seg.ldo;
                }
                store8(outp, odt, orow + (n - colb), v, nvalid, out_vec);
            }
        }
        __syncthreads();
    }
}
