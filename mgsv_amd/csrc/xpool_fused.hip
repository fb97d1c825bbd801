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
static_assert(XT == XD, "the prologue stages one vector element per thread");
constexpr int XS_MAX = 1024;            // segments per track this kernel accepts (its mask row lives in LDS)
constexpr int XLDS = STAGE_BYTES + 5 * XD * 4 + XKEY * 4 + XS_MAX * 4;

__global__ __launch_bounds__(XT, 1) void xpool_fused_kernel(const MadeXpoolFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* vec = (float*)(lds + STAGE_BYTES);          // [5][256]: ln2 gamma, ln2 beta, linear bias, ln3 gamma^2, ln3 gamma*beta
    float* lds_bias = vec + 5 * XD;                    // [32]
    float* lds_mask = lds_bias + XKEY;                 // [S]: the track's mask row (all ones without a mask)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int64_t m = blockIdx.y;
    const int64_t my_n = (int64_t)blockIdx.x * XQ + wave * 32 + r;
    const int64_t nc = my_n < a.Nv ? my_n : a.Nv - 1;

    const bf16_t* Kg = (const bf16_t*)a.K + m * a.k_bs;
    const bf16_t* Ug = (const bf16_t*)a.U + m * a.u_bs;
    const float* maskg = a.key_mask ? a.key_mask + m * a.S : nullptr;

    // ---- everything the workgroup needs first is requested in ONE round trip: the small vectors, the track's mask row, the
    // per-video scalars, the Q fragments and the first K / U tile (one workgroup per CU: nobody else hides these latencies)
    const float v0 = a.ln2_g[tid], v1 = a.ln2_b[tid], v2 = a.bl[tid], g3 = a.ln3_g[tid], b3 = a.ln3_b[tid];
    float mrow[XS_MAX / XT];
#pragma unroll
    for (int i = 0; i < XS_MAX / XT; ++i) {
        const int j = tid + i * XT;
        float mv = 1.f;
        if (maskg) mv = maskg[j < (int)a.S ? j : 0];   // (uniform branch; the load itself is unconditional, index clamped)
        mrow[i] = j < (int)a.S ? mv : 0.f;
    }
    const float* pvp = a.ws + a.Nv * XD + nc * 2;      // sum g v, sum b v of this lane's video
    const float* csp = a.ws + a.Nv * (XD + 2);         // sum g^2, sum g b, sum b^2
    const float p0 = pvp[0], pb = pvp[1], c0 = csp[0], e0 = csp[1], f0 = csp[2];
    // Q fragments (B operand of S^T = K Q^T): lane (r, hh) holds Q[n][ks*16 + hh*8 ..]
    bf16x8 qf[XD / 16];
    {
        const bf16_t* qp = (const bf16_t*)a.Q + nc * a.ldq + hh * 8;
#pragma unroll
        for (int ks = 0; ks < XD / 16; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }
    bf16x8 rk[4], rv[4];
    bf16x8 rw[16];
    auto load_kv = [&](int64_t t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                  // branch-free: always load (row clamped), mask on the registers
            const int ch = tid + i * XT, row = ch >> 5, cc = ch & 31;
            const int64_t key = t * XKEY + row;
            const int64_t kcl = key < a.S ? key : a.S - 1;
            rk[i] = *(const bf16x8*)(Kg + kcl * a.ldk + cc * 8);
            rv[i] = *(const bf16x8*)(Ug + kcl * a.ldu + cc * 8);
        }
    };
    load_kv(0);
    vec[tid] = v0; vec[XD + tid] = v1; vec[2 * XD + tid] = v2; vec[3 * XD + tid] = g3 * g3; vec[4 * XD + tid] = g3 * b3;
#pragma unroll
    for (int i = 0; i < XS_MAX / XT; ++i) lds_mask[tid + i * XT] = mrow[i];
    __syncthreads();
    // segments after the last valid one contribute exactly 0: stop there (every wave scans the row itself: no second barrier)
    int64_t s_eff;
    {
        int last = -1;
        for (int j = lane; j < (int)a.S; j += 64)
            if (lds_mask[j] != 0.f) last = j;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) last = max(last, __shfl_xor(last, o2));
        s_eff = last + 1;
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

    // K / U tile t+1 travels global -> registers while tile t is multiplied, the Linear's weight halves likewise
    auto load_w = [&](int h) __attribute__((always_inline)) {
        const bf16_t* Wg = (const bf16_t*)a.Wl + (int64_t)(128 * h) * a.ldw;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ch = tid + i * XT, row = ch >> 5, cc = ch & 31;
            rw[i] = *(const bf16x8*)(Wg + (int64_t)row * a.ldw + cc * 8);
        }
    };
    auto store_w = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ch = tid + i * XT, row = ch >> 5, cc = ch & 31;          // 16-byte chunk cc of row: 4-groups 2cc, 2cc+1
            const bf16x4 w0 = __builtin_shufflevector(rw[i], rw[i], 0, 1, 2, 3), w1 = __builtin_shufflevector(rw[i], rw[i], 4, 5, 6, 7);
            // within its 16-element group the chunk holds 4-groups (0,1) [cc even] or (2,3) [cc odd]; they go to slots
            // 0->0, 1->2, 2->1, 3->3 of the permuted group
            unsigned char* dst = lds + row * W_ROW + (cc >> 1) * 32;
            *(bf16x4*)(dst + ((cc & 1) ? 8 : 0)) = w0;
            *(bf16x4*)(dst + ((cc & 1) ? 24 : 16)) = w1;
        }
    };
    for (int64_t t = 0; t < ntiles; ++t) {
        __syncthreads();                               // previous tile consumed (and the vectors / s_eff published)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = tid + i * XT, row = ch >> 5, cc = ch & 31;
            const int64_t key = t * XKEY + row;
            const bool keep = key < a.S && lds_mask[key < a.S ? key : 0] != 0.f;
            *(bf16x8*)(lds_k + row * K_ROW + cc * 16) = keep_or_zero(rk[i], keep);
            *(bf16x8*)(lds_v + row * V_ROW + cc * 16) = keep_or_zero(rv[i], keep);
        }
        if (tid < XKEY) {
            const int64_t key = t * XKEY + tid;
            lds_bias[tid] = (key < a.S && lds_mask[key < a.S ? key : 0] != 0.f) ? 0.f : -INFINITY;
        }
        __syncthreads();
        if (t + 1 < ntiles) load_kv(t + 1);

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
        // the running maximum moves only when a score beats it by more than 2^8 (probabilities stay <= 256, exact in f32 and
        // harmless in bf16): the 128-value rescale of O^T is then rare instead of per tile
        const bool move = mx > m_run + 8.f || m_run == -INFINITY;
        const float m_new = move ? fmaxf(m_run, mx) : m_run;
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
        if (t > 0 && __any(move)) {                    // (first tile: O^T is still zero)
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

    // ---- LayerNorm2 per video (column): the lane holds 128 of the 256 values, its partner (lane ^ 32) the rest.  One pass
    // (sum, sum of squares) on the un-normalised O^T; the 1/l of the softmax is folded into the scale.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float inv_l = 1.f / (l_run + __shfl_xor(l_run, 32));
    f32x2 su = {0.f, 0.f}, sq = {0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const f32x2 x = {o[d][e], o[d][e + 1]};
            su += x; sq += x * x;
        }
    float sum1 = su[0] + su[1], sum2 = sq[0] + sq[1];
    sum1 += __shfl_xor(sum1, 32); sum2 += __shfl_xor(sum2, 32);
    const float mean_o = sum1 * (1.f / XD);
    const float var2 = fmaxf(sum2 * (1.f / XD) - mean_o * mean_o, 0.f) * inv_l * inv_l;
    const float rstd2 = 1.0f / sqrtf(var2 + a.eps);
    const float k1 = inv_l * rstd2, k2 = -mean_o * inv_l * rstd2;          // a3 = o * (k1 g) + (b + k2 g)
    bf16x8 a3[8][2];                                   // normalised O^T as B-operand fragments of the Linear (and its residual)
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int dd = d * 32 + 8 * g4 + 4 * hh;
            const f32x4 gm = *(const f32x4*)(vec + dd), bt = *(const f32x4*)(vec + XD + dd);
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const int e = 4 * g4 + j;
                const f32x2 gg = {gm[j], gm[j + 1]}, bb = {bt[j], bt[j + 1]}, x = {o[d][e], o[d][e + 1]};
                const f32x2 v = x * (gg * k1) + (bb + gg * k2);
                a3[d][e >> 3][e & 7] = (bf16_t)v[0];
                a3[d][e >> 3][(e & 7) + 1] = (bf16_t)v[1];
            }
        }

    // ---- Y^T = W_l a3^T + b + a3, output rows in two halves of 128 staged through the LDS the K / U tiles used.  Every 32-row
    // tile of Y^T is consumed as it completes: with z = r (y - mu) g + b (LayerNorm3) the cosine needs only running sums,
    //   <z, v>  = r (sum y (g v) - mu sum g v) + sum b v
    //   <z, z>  = r^2 (sum y^2 g^2 - 2 mu sum y g^2 + mu^2 sum g^2) + 2 r (sum y g b - mu sum g b) + sum b^2
    // of which everything without y was summed once by xpool_prep_kernel (per video: sum g v, sum b v, and g v itself; per
    // model: sum g^2, sum g b, sum b^2) -- no tile of Y is kept and a row costs eight packed FMAs per two elements.
    f32x2 S1 = {0.f, 0.f}, S2 = {0.f, 0.f}, P1 = {0.f, 0.f}, C2 = {0.f, 0.f}, C1 = {0.f, 0.f}, E1 = {0.f, 0.f};
    const float* vp = a.ws + nc * XD;                  // g3 * vn of this lane's video
    f32x4 v4[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) v4[g4] = *(const f32x4*)(vp + 8 * g4 + 4 * hh);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                               // K / U tile (or the previous half) consumed by every wave
        if (h == 0) load_w(0);
        store_w();
        __syncthreads();
        if (h == 0) load_w(1);                         // the second half travels while the first is multiplied
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            __builtin_amdgcn_sched_barrier(0);         // keep the tiles apart: hoisting every tile's LDS reads costs > 256 registers
            const int dt = 4 * h + t;
            f32x4 v4n[4];                              // next tile's video components travel under this tile's MFMAs
            if (dt + 1 < 8) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) v4n[g4] = *(const f32x4*)(vp + (dt + 1) * 32 + 8 * g4 + 4 * hh);
            }
            f32x4 g2v[4], gbv[4];                      // the epilogue's LayerNorm3 vectors: read under the MFMAs, not after them
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                g2v[g4] = *(const f32x4*)(vec + 3 * XD + dt * 32 + 8 * g4 + 4 * hh);
                gbv[g4] = *(const f32x4*)(vec + 4 * XD + dt * 32 + 8 * g4 + 4 * hh);
            }
            f32x16 acc;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {           // accumulator starts from the Linear's bias
                const f32x4 bl4 = *(const f32x4*)(vec + 2 * XD + dt * 32 + 8 * g4 + 4 * hh);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * g4 + j] = bl4[j];
            }
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 wf = *(const bf16x8*)(lds + (t * 32 + r) * W_ROW + (kt * 32 + s2 * 16 + hh * 8) * 2);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a3[kt][s2], acc, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 g2 = g2v[g4], gb = gbv[g4];
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int e = 4 * g4 + j;
                    // + residual (the normalised row itself, reference modules/transformer.py:177)
                    const f32x2 res = {(float)a3[dt][e >> 3][e & 7], (float)a3[dt][e >> 3][(e & 7) + 1]};
                    const f32x2 yv = (f32x2){acc[e], acc[e + 1]} + res;
                    const f32x2 yy = yv * yv;
                    const f32x2 gg = {g2[j], g2[j + 1]}, bb = {gb[j], gb[j + 1]}, vv = {v4[g4][j], v4[g4][j + 1]};
                    S1 += yv; S2 += yy;
                    P1 += yv * vv;
                    C2 += yy * gg; C1 += yv * gg;
                    E1 += yv * bb;
                }
            }
            if (dt + 1 < 8) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) v4[g4] = v4n[g4];
            }
        }
    }
    float s1 = S1[0] + S1[1], s2 = S2[0] + S2[1], p1 = P1[0] + P1[1], c2 = C2[0] + C2[1], c1 = C1[0] + C1[1], e1 = E1[0] + E1[1];
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); p1 += __shfl_xor(p1, 32);
    c2 += __shfl_xor(c2, 32); c1 += __shfl_xor(c1, 32); e1 += __shfl_xor(e1, 32);
    const float mu = s1 * (1.f / XD);
    const float var = fmaxf(s2 * (1.f / XD) - mu * mu, 0.f);
    const float rs = 1.0f / sqrtf(var + a.eps);
    const float dot = rs * (p1 - mu * p0) + pb;
    const float zz = rs * rs * (c2 - 2.f * mu * c1 + mu * mu * c0) + 2.f * rs * (e1 - mu * e0) + f0;
    if (hh == 0 && my_n < a.Nv) a.sims[my_n * a.ld_sims + m] = dot / sqrtf(zz);
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
        hipError_t e = hipFuncSetAttribute((const void*)xpool_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, XLDS);
        if (e != hipSuccess) {
            made_set_error("made_xpool_fused: cannot reserve %d bytes of LDS: %s", XLDS, hipGetErrorString(e));
            return MADE_ERR_HIP;
        }
        attr_done = true;
    }
    if (a.prepare_ws)
        hipLaunchKernelGGL(xpool_prep_kernel, dim3((unsigned)((a.Nv + 3) / 4)), dim3(XT), 0, (hipStream_t)stream, a.vn, a.ldvn, a.ln3_g,
                           a.ln3_b, a.ws, a.Nv);
    dim3 grid((unsigned)((a.Nv + XQ - 1) / XQ), (unsigned)a.Nm), block(XT);
    hipLaunchKernelGGL(xpool_fused_kernel, grid, block, XLDS, (hipStream_t)stream, a);
    return made_check_launch("made_xpool_fused");
}
