// Stand-alone probe (not part of libmade_hip.so): bf16 GEMM C[M,N] = A[M,K] W[N,K]^T + bias on gfx950 with a
// multi-stage LDS-DMA ring (global_load_lds, counted vmcnt, one raw s_barrier per K slab), templated on the tile shape,
// the wave grid, the ring depth and the slab depth -- to choose the shape of made_linear's fast path by measurement on the
// path's own problem sizes.  Build: hipcc -O3 --offload-arch=gfx950 -o gemm_ring_probe gemm_ring_probe.hip
// Run:   ./gemm_ring_probe            (sweeps the configurations over the encoder-sized shapes, checks each against a naive kernel)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 t; t[0] = (bf16_t)a; t[1] = (bf16_t)b;
    return __builtin_bit_cast(unsigned, t);
}

// BM x BN tile, WM x WN waves, NST ring stages, BKB bytes of K per row per slab (64 / 128 / 256 -> 32 / 64 / 128 bf16), MINW = min waves per SIMD
// EPI: 0 = no output (K loop only), 1 = LDS-staged coalesced bf16 stores with bias
template <int BM, int BN, int WM, int WN, int NST, int BKB, int MINW, int EPI>
__global__ __launch_bounds__(WM * WN * 64, MINW) void gemm_ring(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                               const float* __restrict__ bias, bf16_t* __restrict__ C, int M, int N, int K) {
    constexpr int NW = WM * WN, NT_ = NW * 64;
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NTL = TN / 32;
    constexpr int STAGE = (BM + BN) * BKB;
    constexpr int RPP = 1024 / BKB;                     // rows per 1 KB piece
    constexpr int CPR = BKB / 16;                       // 16-byte chunks per row
    constexpr int RPB = 256 / BKB > 0 ? 256 / BKB : 1;  // rows per 256-byte bank row
    constexpr int PA = BM / RPP / NW, PWN = BN / RPP / NW;   // pieces per wave per slab
    static_assert(BM % (RPP * NW) == 0 && BN % (RPP * NW) == 0, "pieces must divide over the waves");
    constexpr int PPW = PA + PWN;
    constexpr int KSTEPS = BKB / 32;                    // 16-deep MFMA k-steps per slab
    constexpr int OUT_LD = BN * 2 + 16;                 // bytes per row of the staged output tile (padded)
    constexpr int LDS_BYTES = NST * STAGE > BM * OUT_LD ? NST * STAGE : BM * OUT_LD;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    (void)LDS_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int n_tiles = (N + BN - 1) / BN, m_tiles = (M + BM - 1) / BM;
    const int nwg = m_tiles * n_tiles;
    int tile_id;
    {
        const int xcd = blockIdx.x & 7, q = nwg >> 3, rem = nwg & 7;
        tile_id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
    }
    const int tile_m = tile_id / n_tiles, tile_n = tile_id % n_tiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    auto swz = [](int row) { return (row / RPB) % CPR; };
    const bf16_t* pa[PA];
    const bf16_t* pw[PWN];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = RPP * (PA * wave + i) + lane / CPR;
        const int chunk = (lane % CPR) ^ swz(row);
        int gm = m0 + row; gm = gm < M ? gm : M - 1;
        pa[i] = A + (int64_t)gm * K + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < PWN; ++i) {
        const int row = RPP * (PWN * wave + i) + lane / CPR;
        const int chunk = (lane % CPR) ^ swz(row);
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
        pw[i] = W + (int64_t)gn * K + chunk * 8;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int offa[MT], offw[NTL], sa[MT], sw[NTL];
#pragma unroll
    for (int t = 0; t < MT; ++t) { const int ra = wm * TM + t * 32 + r; offa[t] = ra * BKB; sa[t] = swz(ra); }
#pragma unroll
    for (int t = 0; t < NTL; ++t) { const int rw = wn * TN + t * 32 + r; offw[t] = BM * BKB + rw * BKB; sw[t] = swz(rw); }

    constexpr int KE = BKB / 2;
    const int nk = K / KE;
    auto issue = [&](int kt) __attribute__((always_inline)) {
        unsigned char* st = lds + (kt % NST) * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pa[i] + kt * KE), (lds_ptr_t)(st + (PA * wave + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PWN; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pw[i] + kt * KE), (lds_ptr_t)(st + BM * BKB + (PWN * wave + i) * 1024), 16, 0, 0);
    };
    // swapped operands: acc = W-fragment (rows = n) x A-fragment (cols = m): each lane then holds 4 consecutive n of one row m
    auto multiply = [&](int kt) __attribute__((always_inline)) {
        const unsigned char* st = lds + (kt % NST) * STAGE;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            bf16x8 fa[MT], fw[NTL];
            const int c = 2 * ks + hh;
#pragma unroll
            for (int t = 0; t < MT; ++t) fa[t] = *(const bf16x8*)(st + offa[t] + ((c ^ sa[t]) << 4));
#pragma unroll
            for (int t = 0; t < NTL; ++t) fw[t] = *(const bf16x8*)(st + offw[t] + ((c ^ sw[t]) << 4));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nt], fa[mt], acc[mt][nt], 0, 0, 0);
        }
    };

    // ---- ring: slab kt lives in stage kt % NST; NST - 1 slabs are in flight ahead of the one being multiplied
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue(s);
    const int n_main = nk - (NST - 1);                   // iterations that still issue a new slab
    int kt = 0;
    for (; kt < n_main; ++kt) {
        wait_vmcnt<(NST - 2) * PPW>();                   // slab kt (this wave's pieces) has landed
        asm volatile("s_barrier" ::: "memory");          // ... everyone's; and everyone is done reading stage (kt - 1) % NST
        issue(kt + NST - 1);
        multiply(kt);
    }
    // tail: nothing left to issue; `ahead` slabs are still in flight behind the current one
    for (; kt < nk; ++kt) {
        const int ahead = nk - kt - 1;
        if constexpr (NST >= 4) { if (ahead >= 2) wait_vmcnt<2 * PPW>(); }
        if constexpr (NST >= 3) { if (ahead == 1) wait_vmcnt<1 * PPW>(); }
        if (ahead <= 0) wait_vmcnt<0>();
        asm volatile("s_barrier" ::: "memory");
        multiply(kt);
    }

    if constexpr (EPI == 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        if (s == 123.456f) C[tid] = (bf16_t)s;
        return;
    } else if constexpr (EPI == 3) {
        // ---- direct epilogue: no LDS.  A lane holds 4 consecutive columns of one row per register group; permlane32_swap pairs two
        // groups so that each lane owns 8 consecutive columns (f32), then + bias, -> bf16, one 16-byte store per pair.
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + wm * TM + mt * 32 + r;
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    float lo[4], hi[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned x = __builtin_bit_cast(unsigned, acc[mt][nt][4 * g + j]);
                        const unsigned y = __builtin_bit_cast(unsigned, acc[mt][nt][4 * (g + 1) + j]);
                        auto sw = __builtin_amdgcn_permlane32_swap(x, y, false, false);
                        lo[j] = __builtin_bit_cast(float, (unsigned)sw[0]);
                        hi[j] = __builtin_bit_cast(float, (unsigned)sw[1]);
                    }
                    const int n = n0 + wn * TN + nt * 32 + 8 * (g + hh);
                    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
                    if (bias) { b0 = *(const f32x4*)(bias + n); b1 = *(const f32x4*)(bias + n + 4); }
                    u32x4 pk;
                    pk[0] = pack2(lo[0] + b0[0], lo[1] + b0[1]); pk[1] = pack2(lo[2] + b0[2], lo[3] + b0[3]);
                    pk[2] = pack2(hi[0] + b1[0], hi[1] + b1[1]); pk[3] = pack2(hi[2] + b1[2], hi[3] + b1[3]);
                    if (m < M) *(u32x4*)(C + (int64_t)m * N + n) = pk;
                }
        }
    } else {
        // ---- epilogue: + bias, -> bf16, 8-byte pieces into a padded row-major LDS tile, then whole rows out with 16-byte stores
        asm volatile("s_barrier" ::: "memory");          // every wave is done reading the last stage
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            const int nb = n0 + wn * TN + nt * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb + 8 * g;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (bias && n + 3 < N) bv = *(const f32x4*)(bias + n);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int ml = wm * TM + mt * 32 + r;
                    u32x2 pk;
                    pk[0] = pack2(acc[mt][nt][4 * g + 0] + bv[0], acc[mt][nt][4 * g + 1] + bv[1]);
                    pk[1] = pack2(acc[mt][nt][4 * g + 2] + bv[2], acc[mt][nt][4 * g + 3] + bv[3]);
                    *(u32x2*)(lds + ml * OUT_LD + (wn * TN + nt * 32 + 8 * g + 4 * hh) * 2) = pk;
                }
            }
        }
        __syncthreads();
        constexpr int CPRO = BN / 8;                     // 16-byte chunks per output row
        constexpr int ROWS_PER_PASS = NT_ / CPRO;
        const int cc = tid % CPRO, rr = tid / CPRO;
        const int n = n0 + cc * 8;
#pragma unroll 4
        for (int p = 0; p < BM / ROWS_PER_PASS; ++p) {
            const int row = rr + p * ROWS_PER_PASS;
            const int m = m0 + row;
            if (m < M && n < N) {
                const u32x4 v = *(const u32x4*)(lds + row * OUT_LD + cc * 16);
                *(u32x4*)(C + (int64_t)m * N + n) = v;
            }
        }
    }
}

__global__ void naive_gemm(const bf16_t* A, const bf16_t* W, const float* bias, float* C, int M, int N, int K, int mstep) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y * mstep;
    if (n >= N || m >= M) return;
    float s = bias ? bias[n] : 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(int64_t)m * K + k] * (float)W[(int64_t)n * K + k];
    C[(int64_t)blockIdx.y * N + n] = s;
}

struct Shape { int M, N, K; };

template <int BM, int BN, int WM, int WN, int NST, int BKB, int MINW, int EPI>
static void run_cfg(const char* name, const std::vector<Shape>& shapes, int nbuf, bf16_t** dA, bf16_t** dW, float* dbias, bf16_t** dC, float* dref, int iters) {
    constexpr int STAGE = (BM + BN) * BKB;
    constexpr int OUT_LD = BN * 2 + 16;
    constexpr int LDS_BYTES = NST * STAGE > BM * OUT_LD ? NST * STAGE : BM * OUT_LD;
    auto kern = gemm_ring<BM, BN, WM, WN, NST, BKB, MINW, EPI>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, WM * WN * 64, LDS_BYTES));
    printf("%-34s lds=%6d occ=%d |", name, LDS_BYTES, occ);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        if (s.K % (BKB / 2) != 0 || s.K / (BKB / 2) < 1) { printf("   n/a    "); continue; }
        const int grid = ((s.M + BM - 1) / BM) * ((s.N + BN - 1) / BN);
        for (int i = 0; i < 3; ++i)
            hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), LDS_BYTES, 0, dA[i % nbuf], dW[0], dbias, dC[i % nbuf], s.M, s.N, s.K);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), LDS_BYTES, 0, dA[i % nbuf], dW[0], dbias, dC[i % nbuf], s.M, s.N, s.K);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        const double tf = 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12;
        // correctness on sampled rows (every mstep-th row), buffer 0
        double maxerr = 0.0;
        if (EPI) {
            CHECK(hipMemset(dC[0], 0, (size_t)s.M * s.N * 2));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), LDS_BYTES, 0, dA[0], dW[0], dbias, dC[0], s.M, s.N, s.K);
            const int mstep = 97, nrows = (s.M + mstep - 1) / mstep;
            hipLaunchKernelGGL(naive_gemm, dim3((s.N + 255) / 256, nrows), dim3(256), 0, 0, dA[0], dW[0], dbias, dref, s.M, s.N, s.K, mstep);
            CHECK(hipDeviceSynchronize());
            std::vector<float> ref((size_t)nrows * s.N);
            std::vector<bf16_t> got((size_t)s.M * s.N);
            CHECK(hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(got.data(), dC[0], got.size() * 2, hipMemcpyDeviceToHost));
            for (int i = 0; i < nrows; ++i)
                for (int n = 0; n < s.N; ++n) {
                    const double d = fabs((double)(float)got[(size_t)(i * mstep) * s.N + n] - (double)ref[(size_t)i * s.N + n]);
                    const double tol = 0.02 + 0.01 * fabs(ref[(size_t)i * s.N + n]);
                    if (d / tol > maxerr) maxerr = d / tol;
                }
        }
        printf(" %6.1fus %5.0fTF%s |", us, tf, maxerr > 1.0 ? " BAD" : "");
    }
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    std::vector<Shape> shapes = {{18432, 512, 512}, {18432, 1536, 512}, {18432, 1024, 512}, {18432, 512, 1024}, {32768, 512, 512}, {17000, 512, 768}, {8192, 8192, 8192}};
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const int nbuf = 3;
    size_t maxA = 0, maxW = 0, maxC = 0;
    for (auto& s : shapes) {
        maxA = std::max(maxA, (size_t)s.M * s.K); maxW = std::max(maxW, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N);
    }
    bf16_t* dA[nbuf]; bf16_t* dW[1]; bf16_t* dC[nbuf]; float* dbias; float* dref;
    std::vector<bf16_t> h(std::max(maxA, maxW));
    srand(1);
    for (int b = 0; b < nbuf; ++b) {
        CHECK(hipMalloc(&dA[b], maxA * 2)); CHECK(hipMalloc(&dC[b], maxC * 2));
        for (size_t i = 0; i < maxA; ++i) h[i] = (bf16_t)((rand() % 2001 - 1000) / 1000.0f);
        CHECK(hipMemcpy(dA[b], h.data(), maxA * 2, hipMemcpyHostToDevice));
    }
    CHECK(hipMalloc(&dW[0], maxW * 2));
    for (size_t i = 0; i < maxW; ++i) h[i] = (bf16_t)((rand() % 2001 - 1000) / 1000.0f * 0.05f);
    CHECK(hipMemcpy(dW[0], h.data(), maxW * 2, hipMemcpyHostToDevice));
    std::vector<float> hb(8192);
    for (auto& x : hb) x = (rand() % 2001 - 1000) / 1000.0f;
    CHECK(hipMalloc(&dbias, 8192 * 4)); CHECK(hipMemcpy(dbias, hb.data(), 8192 * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dref, (size_t)(32768 / 97 + 2) * 8192 * 4));

    printf("%-34s %-18s |", "config (BMxBN waves NST BK)", "");
    for (auto& s : shapes) printf(" %6dx%4dx%4d  |", s.M, s.N, s.K);
    printf("\n");
#define RUN(BM, BN, WM, WN, NST, BKB, MINW) \
    run_cfg<BM, BN, WM, WN, NST, BKB, MINW, 1>(#BM "x" #BN " w" #WM "x" #WN " st" #NST " kb" #BKB " mw" #MINW, shapes, nbuf, dA, dW, dbias, dC, dref, iters); \
    run_cfg<BM, BN, WM, WN, NST, BKB, MINW, 3>("   (direct epilogue)", shapes, nbuf, dA, dW, dbias, dC, dref, iters); \
    run_cfg<BM, BN, WM, WN, NST, BKB, MINW, 0>("   (K loop only)", shapes, nbuf, dA, dW, dbias, dC, dref, iters);
    RUN(128, 128, 2, 2, 2, 128, 2)
    RUN(128, 128, 2, 4, 2, 128, 4)
    RUN(128, 128, 4, 2, 2, 128, 4)
    RUN(128, 128, 4, 4, 2, 128, 8)
    RUN(64, 128, 1, 4, 2, 128, 4)
    RUN(64, 128, 2, 4, 2, 128, 6)
    RUN(128, 256, 2, 4, 2, 128, 2)
    RUN(256, 128, 4, 2, 2, 128, 2)
    RUN(256, 128, 4, 4, 2, 128, 4)
    return 0;
}
