// made_xpool_fused: the whole per-pair chain of the X-Pool block in ONE kernel, for all-pairs retrieval scoring
// (reference test-MaDe.py:392-403 = modules/transformer.py:156-180 + modules/metrics.py:10-24).  gfx950, bf16, D = 256.
//
// For every (video n, track m):   scores over the track's segments -> softmax -> pooled U rows (out_proj hoisted onto the
// values: rows of the softmax sum to 1) -> LayerNorm2 -> + Linear (residual) -> LayerNorm3 -> cosine with the video.
// Done as separate launches this chain writes and re-reads three [Nm*Nv, D] tensors -- 0.65 TB of HBM traffic at 53 k x 4 k
// pairs, two thirds of the retrieval time.  Here nothing per-pair ever leaves the chip:
//
//   * one workgroup = 128 videos x 1 track; each of the four waves owns 32 videos with the FULL width D, so its flash
//     attention state O^T [256 x 32] (8 accumulator tiles) and everything after it stay in that wave's registers;
//   * K / U tiles of 32 segments are staged through LDS once per workgroup (128 videos share them);
//   * LayerNorm statistics are per video = per accumulator COLUMN = per lane (+ one shuffle with the other lane half);
//   * the Linear is a second MFMA whose B operand is the normalised O^T straight from the accumulator registers: a lane
//     holds rows {0-3, 8-11} (+4 for the upper lane half) of every 16-row group, so the K index of that product is
//     permuted accordingly and the weight tile is stored in LDS with the same permutation (4-element groups reordered
//     [g0, g2, g1, g3] within every 16) -- no cross-lane traffic between the two products;
//   * the weight is staged in two halves of 128 output rows through the LDS the K / U tiles used.
#include "common.h"

namespace {

constexpr int XD = 256;                 // model width
constexpr int XQ = 128;                 // videos per workgroup
constexpr int XKEY = 32;                // segments per tile
constexpr int XT = 256;                 // threads
constexpr int K_ROW = XD * 2 + 16;      // padded: conflict-free 16-byte row reads
constexpr int V_ROW = XD * 2 + 64;      // 4 consecutive rows on disjoint bank quarters (ds_read_b64_tr_b16)
constexpr int W_ROW = XD * 2 + 16;
constexpr int KV_BYTES = XKEY * K_ROW + XKEY * V_ROW;
constexpr int W_BYTES = 128 * W_ROW;
constexpr int STAGE_BYTES = W_BYTES > KV_BYTES ? W_BYTES : KV_BYTES;
constexpr int XLDS = STAGE_BYTES + 5 * XD * 4 + XKEY * 4 + 16;

__global__ __launch_bounds__(XT, 1) void xpool_fused_kernel(const MadeXpoolFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* vec = (float*)(lds + STAGE_BYTES);          // [5][256]: ln2 gamma, ln2 beta, linear bias, ln3 gamma, ln3 beta
    float* lds_bias = vec + 5 * XD;                    // [32]
    int* red = (int*)(lds_bias + XKEY);                // [4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int64_t m = blockIdx.y;
    const int64_t my_n = (int64_t)blockIdx.x * XQ + wave * 32 + r;
    const int64_t nc = my_n < a.Nv ? my_n : a.Nv - 1;

    for (int i = tid; i < XD; i += XT) {
        vec[i] = a.ln2_g[i]; vec[XD + i] = a.ln2_b[i]; vec[2 * XD + i] = a.bl[i];
        vec[3 * XD + i] = a.ln3_g[i]; vec[4 * XD + i] = a.ln3_b[i];
    }

    // ---- Q fragments (B operand of S^T = K Q^T): lane (r, hh) holds Q[n][ks*16 + hh*8 ..]
    bf16x8 qf[XD / 16];
    {
        const bf16_t* qp = (const bf16_t*)a.Q + nc * a.ldq;
#pragma unroll
        for (int ks = 0; ks < XD / 16; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16 + hh * 8);
    }
    const bf16_t* Kg = (const bf16_t*)a.K + m * a.k_bs;
    const bf16_t* Ug = (const bf16_t*)a.U + m * a.u_bs;
    const float* maskg = a.key_mask ? a.key_mask + m * a.S : nullptr;

    // segments after the last valid one contribute exactly 0: stop there
    int64_t s_eff = a.S;
    if (maskg) {
        int last = -1;
        for (int j = tid; j < (int)a.S; j += XT)
            if (maskg[j] != 0.f) last = j;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) last = max(last, __shfl_xor(last, o2));
        if (lane == 0) red[wave] = last;
        __syncthreads();
        s_eff = max(max(red[0], red[1]), max(red[2], red[3])) + 1;
    }
    const int64_t ntiles = (s_eff + XKEY - 1) / XKEY;

    f32x16 o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c = a.scale * 1.4426950408889634f;     // scores in the log2 domain: one FMA + exp2 per element
    unsigned char* lds_k = lds;
    unsigned char* lds_v = lds + XKEY * K_ROW;
    const int g = lane >> 4, i16 = lane & 15;

    for (int64_t t = 0; t < ntiles; ++t) {
        __syncthreads();                               // previous tile consumed (and the vectors / s_eff published)
        {
            bf16x8 rk[4], rv[4];
            float mk[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {              // branch-free: always load (row clamped), mask on the registers
                const int ch = tid + i * XT, row = ch >> 5, cc = ch & 31;
                const int64_t key = t * XKEY + row;
                const int64_t kcl = key < a.S ? key : a.S - 1;
                rk[i] = *(const bf16x8*)(Kg + kcl * a.ldk + cc * 8);
                rv[i] = *(const bf16x8*)(Ug + kcl * a.ldu + cc * 8);
                mk[i] = maskg ? maskg[kcl] : 1.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = tid + i * XT, row = ch >> 5, cc = ch & 31;
                const bool keep = (t * XKEY + row) < a.S && mk[i] != 0.f;
                *(bf16x8*)(lds_k + row * K_ROW + cc * 16) = keep_or_zero(rk[i], keep);
                *(bf16x8*)(lds_v + row * V_ROW + cc * 16) = keep_or_zero(rv[i], keep);
            }
            if (tid < XKEY) {
                const int64_t key = t * XKEY + tid;
                const int64_t kcl = key < a.S ? key : a.S - 1;
                const float mkb = maskg ? maskg[kcl] : 1.f;
                lds_bias[tid] = (key < a.S && mkb != 0.f) ? 0.f : -INFINITY;
            }
        }
        __syncthreads();

        // ---- S^T [32 segments x 32 videos]
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < XD / 16; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(lds_k + r * K_ROW + ks * 32 + hh * 16);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        }
        // ---- online softmax (per video = per lane column; the two lane halves hold different segments)
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = s[e] * c + lds_bias[acc_row(e, hh)];
            mx = fmaxf(mx, s[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = __builtin_amdgcn_exp2f(s[e] - m_use);
            psum += s[e];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (!__all(alpha == 1.f)) {
#pragma unroll
            for (int d = 0; d < 8; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
        }
        // ---- O^T += U^T [256 x 32 segments] P^T [32 segments x 32 videos]; U^T read transposed out of the row-major tile
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[8 * s2 + j];
            const int kb = 16 * s2 + 4 * (g >> 1);
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const int dcol = d * 32 + (g & 1) * 16 + 4 * (i16 & 3);
                const unsigned char* vp = lds_v + (kb + (i16 >> 2)) * V_ROW + dcol * 2;
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)vp);
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + 8 * V_ROW));
                const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
            }
        }
    }

    // ---- LayerNorm2 per video (column): the lane holds 128 of the 256 values, its partner (lane ^ 32) the rest
    const float inv_l = 1.f / (l_run + __shfl_xor(l_run, 32));
    float sum = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { o[d][e] *= inv_l; sum += o[d][e]; }
    sum += __shfl_xor(sum, 32);
    const float mean2 = sum * (1.f / XD);
    float sq = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float x = o[d][e] - mean2; sq += x * x; }
    sq += __shfl_xor(sq, 32);
    const float rstd2 = 1.0f / sqrtf(sq * (1.f / XD) + a.eps);
    bf16x8 a3[8][2];                                   // normalised O^T as B-operand fragments of the Linear (and its residual)
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int dd = d * 32 + acc_row(e, hh);
            a3[d][e >> 3][e & 7] = (bf16_t)((o[d][e] - mean2) * rstd2 * vec[dd] + vec[XD + dd]);
        }

    // ---- Y^T = W_l a3^T, output rows in two halves of 128 staged through the LDS the K / U tiles used
    float s1 = 0.f;                                    // LayerNorm3 statistics are gathered as the tiles complete
    f32x16 y[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                               // K / U tile (or the previous half) consumed by every wave
        {
            const bf16_t* Wg = (const bf16_t*)a.Wl + (int64_t)(128 * h) * a.ldw;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int ch = tid + i * XT, row = ch >> 5, cc = ch & 31;      // 16-byte chunk cc of row: 4-groups 2cc, 2cc+1
                const bf16x8 w = *(const bf16x8*)(Wg + (int64_t)row * a.ldw + cc * 8);
                const bf16x4 w0 = __builtin_shufflevector(w, w, 0, 1, 2, 3), w1 = __builtin_shufflevector(w, w, 4, 5, 6, 7);
                // within its 16-element group the chunk holds 4-groups (0,1) [cc even] or (2,3) [cc odd]; they go to slots
                // 0->0, 1->2, 2->1, 3->3 of the permuted group
                unsigned char* dst = lds + row * W_ROW + (cc >> 1) * 32;
                *(bf16x4*)(dst + ((cc & 1) ? 8 : 0)) = w0;
                *(bf16x4*)(dst + ((cc & 1) ? 24 : 16)) = w1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 wf = *(const bf16x8*)(lds + (t * 32 + r) * W_ROW + (kt * 32 + s2 * 16 + hh * 8) * 2);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a3[kt][s2], acc, 0, 0, 0);
                }
            const int dt = 4 * h + t;
#pragma unroll
            for (int e = 0; e < 16; ++e) {             // + bias + residual (the normalised row itself, reference :177)
                acc[e] += vec[2 * XD + dt * 32 + acc_row(e, hh)] + (float)a3[dt][e >> 3][e & 7];
                s1 += acc[e];
            }
            y[dt] = acc;
        }
    }

    // ---- LayerNorm3 + cosine with the (already L2-normalised) video
    s1 += __shfl_xor(s1, 32);
    const float mean3 = s1 * (1.f / XD);
    float sq3 = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float x = y[d][e] - mean3; sq3 += x * x; }
    sq3 += __shfl_xor(sq3, 32);
    const float rstd3 = 1.0f / sqrtf(sq3 * (1.f / XD) + a.eps);
    float dot = 0.f, zz = 0.f;
    const float* vp = a.vn + nc * a.ldvn;
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int dd = d * 32 + 8 * g4 + 4 * hh;
            const f32x4 v4 = *(const f32x4*)(vp + dd);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = (y[d][4 * g4 + j] - mean3) * rstd3 * vec[3 * XD + dd + j] + vec[4 * XD + dd + j];
                dot += z * v4[j];
                zz += z * z;
            }
        }
    dot += __shfl_xor(dot, 32);
    zz += __shfl_xor(zz, 32);
    if (hh == 0 && my_n < a.Nv) a.sims[my_n * a.ld_sims + m] = dot / sqrtf(zz);
}

}  // namespace

extern "C" int made_xpool_fused(const MadeXpoolFusedArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr, "made_xpool_fused: null args");
    const MadeXpoolFusedArgs& a = *args;
    MADE_REQUIRE(a.Q && a.K && a.U && a.ln2_g && a.ln2_b && a.Wl && a.bl && a.ln3_g && a.ln3_b && a.vn && a.sims,
                 "made_xpool_fused: null pointer");
    MADE_REQUIRE(a.Nv >= 0 && a.Nm >= 0 && a.S > 0, "made_xpool_fused: bad dims");
    MADE_UNSUPPORTED(a.D == XD, "made_xpool_fused: D=%lld (built for %d)", (long long)a.D, XD);
    MADE_UNSUPPORTED(a.Nm <= 65535, "made_xpool_fused: more than 65535 tracks per call (chunk them)");
    MADE_UNSUPPORTED(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldu % 8 == 0 && a.k_bs % 8 == 0 && a.u_bs % 8 == 0 && a.ldw % 8 == 0 &&
                     a.ldvn % 4 == 0 && ((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.K % 16) == 0 && ((uintptr_t)a.U % 16) == 0 &&
                     ((uintptr_t)a.Wl % 16) == 0 && ((uintptr_t)a.vn % 16) == 0,
                     "made_xpool_fused: pointers / strides must keep 16-byte alignment");
    if (a.Nv == 0 || a.Nm == 0) return MADE_OK;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)xpool_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, XLDS);
        if (e != hipSuccess) {
            made_set_error("made_xpool_fused: cannot reserve %d bytes of LDS: %s", XLDS, hipGetErrorString(e));
            return MADE_ERR_HIP;
        }
        attr_done = true;
    }
    dim3 grid((unsigned)((a.Nv + XQ - 1) / XQ), (unsigned)a.Nm), block(XT);
    hipLaunchKernelGGL(xpool_fused_kernel, grid, block, XLDS, (hipStream_t)stream, a);
    return made_check_launch("made_xpool_fused");
}
