// Row kernels of the MaDe hot path (HBM-bound): LayerNorm, masked mean, L2 normalise, the
// mask-aware sine position embedding, the masked softmax over segments, the X-Pool tail
// (LayerNorm3 + cosine with the video), and the symmetric cross-entropy.  One wave (64 lanes)
// per row with 16-byte loads wherever a row is contiguous; reductions by wave shuffles.
#include "common.h"

namespace {

constexpr int ROW_THREADS = 256;           // 4 waves = 4 rows per workgroup
constexpr int MAX_VEC = 8;                 // 8 x 4 elements per lane -> D <= 2048

// load 4 consecutive elements of runtime dtype as f32
__device__ __forceinline__ f32x4 load4(const void* p, int dtype, int64_t idx) {
    f32x4 v;
    if (dtype == MADE_F32) {
        v = *(const f32x4*)((const float*)p + idx);
    } else {
        bf16x4 t = *(const bf16x4*)((const bf16_t*)p + idx);
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    return v;
}
__device__ __forceinline__ void store4(void* p, int dtype, int64_t idx, f32x4 v) {
    if (dtype == MADE_F32) {
        *(f32x4*)((float*)p + idx) = v;
    } else {
        bf16x4 t;
        t[0] = (bf16_t)v[0]; t[1] = (bf16_t)v[1]; t[2] = (bf16_t)v[2]; t[3] = (bf16_t)v[3];
        *(bf16x4*)((bf16_t*)p + idx) = t;
    }
}

// ---- LayerNorm -----------------------------------------------------------------------------
// Two-pass (mean, then centred variance) in registers: the row is read from HBM once.
template <int NV, bool FULL>
__global__ __launch_bounds__(ROW_THREADS, NV <= 4 ? 4 : 2) void layernorm_kernel(const void* x, int xdt, int64_t ldx, int64_t rpb, int64_t xbs,
                                                                const float* gamma, const float* beta,
                                                                void* y, int ydt, int64_t ldy,
                                                                int64_t rows, int D, float eps, const float* row_skip) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (ROW_THREADS / WAVE) + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (row_skip && row_skip[row] == 0.f) return;
    const int64_t xoff = rpb > 0 ? (row / rpb) * xbs + (row % rpb) * ldx : row * ldx;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * WAVE + lane) * 4;
        if (FULL || c < D) {
            v[i] = load4(x, xdt, xoff + c);
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * WAVE + lane) * 4;
        if (FULL || c < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = v[i][j] - mean; sq += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * WAVE + lane) * 4;
        if (FULL || c < D) {
            f32x4 g = *(const f32x4*)(gamma + c), bb = *(const f32x4*)(beta + c), o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + bb[j];
            store4(y, ydt, row * ldy + c, o);
        }
    }
}

// ---- masked mean over the sequence axis -------------------------------------------------------
// grid (D/64 column groups, B); 256 threads = 64 columns x 4 row phases; LDS combine.
__global__ __launch_bounds__(ROW_THREADS) void masked_mean_kernel(const void* x, int xdt, int64_t x_bs, int64_t ldx,
                                                                  const float* mask, float* out,
                                                                  int64_t T, int D) {
    __shared__ float part[4][64];
    __shared__ float cnt[4];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int phase = threadIdx.x >> 6;
    const int64_t b = blockIdx.y;
    float acc = 0.f, n = 0.f;
    const int colc = col < D ? col : D - 1;            // every lane always loads; the mask is applied on the value
#pragma unroll 8
    for (int64_t t = phase; t < T; t += 4) {
        float mk = mask ? mask[b * T + t] : 1.f;
        n += mk;
        float xv = load_as_f32(x, xdt, b * x_bs + t * ldx + colc);
        acc += (mk != 0.f) ? xv : 0.f;
    }
    part[phase][threadIdx.x & 63] = acc;
    if ((threadIdx.x & 63) == 0) cnt[phase] = n;
    __syncthreads();
    if (phase == 0 && col < D) {
        float s = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (mask) s = s / ((cnt[0] + cnt[1]) + (cnt[2] + cnt[3]));
        out[b * D + col] = s;
    }
}

// The same for rows that allow 4-column vector loads: grid (D/256 column slabs, B), 16 waves, wave w takes tokens w, w + 16, ...
// (four of its tokens in flight), a lane 4 consecutive columns.  The scalar kernel above moves 2 bytes per load instruction:
// 41 us for the audio branch's 64 x 512 x 512 bf16 tokens, on the critical path of every step.
__global__ __launch_bounds__(1024) void masked_mean_vec_kernel(const void* x, int xdt, int64_t x_bs, int64_t ldx,
                                                               const float* mask, float* out, int64_t T, int D) {
    __shared__ f32x4 part[16][64];
    __shared__ float cnt[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + lane * 4;
    const int64_t b = blockIdx.y;
    const int colc = col < D ? col : D - 4;
    f32x4 acc; acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
    float n = 0.f;
    for (int64_t t0 = wave; t0 < T; t0 += 64) {
        f32x4 v[4];
        float mk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t t = t0 + 16 * j;
            const int64_t tc = t < T ? t : T - 1;
            mk[j] = t < T ? (mask ? mask[b * T + tc] : 1.f) : 0.f;
            v[j] = load4(x, xdt, b * x_bs + tc * ldx + colc);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            n += mk[j];
            if (mk[j] != 0.f) { acc[0] += v[j][0]; acc[1] += v[j][1]; acc[2] += v[j][2]; acc[3] += v[j][3]; }
        }
    }
    part[wave][lane] = acc;
    if (lane == 0) cnt[wave] = n;
    __syncthreads();
    if (wave == 0 && col < D) {
        f32x4 s = part[0][lane];
        float c = cnt[0];
        for (int w = 1; w < 16; ++w) {
            const f32x4 p = part[w][lane];
            s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];
            c += cnt[w];
        }
        if (mask) { s[0] /= c; s[1] /= c; s[2] /= c; s[3] /= c; }
        *(f32x4*)(out + b * D + col) = s;
    }
}

// ---- L2 normalise rows -------------------------------------------------------------------------
template <int NV, bool FULL>
__global__ __launch_bounds__(ROW_THREADS, NV <= 4 ? 4 : 2) void l2norm_kernel(const void* x, int xdt, int64_t ldx, float* y32,
                                                             void* yalt, int yadt, int64_t ldy,
                                                             int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (ROW_THREADS / WAVE) + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 v[NV];
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * WAVE + lane) * 4;
        if (FULL || c < D) {
            v[i] = load4(x, xdt, row * ldx + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) sq += v[i][j] * v[i][j];
        }
    }
    const float nrm = fmaxf(sqrtf(wave_sum(sq)), eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * WAVE + lane) * 4;
        if (FULL || c < D) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = v[i][j] / nrm;
            if (y32) *(f32x4*)(y32 + row * ldy + c) = o;
            if (yalt) store4(yalt, yadt, row * ldy + c, o);
        }
    }
}

// ---- mask-aware sine position embedding --------------------------------------------------------
// grid (ceil(L/PE_ROWS), B).  Every workgroup re-reduces its batch row's mask (L floats) to get the
// running count before its rows and the total, then writes PE_ROWS x D outputs.
constexpr int PE_ROWS = 16;
__global__ __launch_bounds__(ROW_THREADS) void sine_pe_kernel(const float* mask, const float* dim_t, void* out, int odt,
                                                              int L, int D) {
    __shared__ float red_before[4], red_total[4];
    const int64_t b = blockIdx.y;
    const int t0 = blockIdx.x * PE_ROWS;
    const float* mrow = mask + b * L;
    float before = 0.f, total = 0.f;
    for (int t = threadIdx.x; t < L; t += ROW_THREADS) {
        float mk = mrow[t];
        total += mk;
        if (t < t0) before += mk;
    }
    before = wave_sum(before);
    total = wave_sum(total);
    if ((threadIdx.x & 63) == 0) { red_before[threadIdx.x >> 6] = before; red_total[threadIdx.x >> 6] = total; }
    __syncthreads();
    before = (red_before[0] + red_before[1]) + (red_before[2] + red_before[3]);
    total = (red_total[0] + red_total[1]) + (red_total[2] + red_total[3]);
    const float denom = total + 1e-6f;
    const float two_pi = 6.283185307179586f;
    float c = before;
    for (int rr = 0; rr < PE_ROWS; ++rr) {
        int t = t0 + rr;
        if (t >= L) break;
        c += mrow[t];                                   // inclusive cumsum (counts are exact in f32)
        const float xe = __fmul_rn(__fdiv_rn(c, denom), two_pi);
        if ((D & 3) == 0) {
            // four columns per thread = two (sin, cos) pairs; a pair shares its frequency (dim_t[2k] == dim_t[2k+1] in the
            // reference's table), so one range reduction serves both; one 8- or 16-byte store
            for (int i = threadIdx.x * 4; i < D; i += ROW_THREADS * 4) {
                const f32x4 dt = *(const f32x4*)(dim_t + i);
                f32x4 v;
                if (odt == MADE_BF16) {
                    // bf16 output (the bf16 engine / trainer): the hardware sine / cosine on the angle in revolutions (|angle| <= 2 pi here;
                    // their error is ~1e-6, three orders below a bf16 ulp) instead of libm's sincosf -- 56 -> ~10 us per step at the headline
                    // shape.  The f32 path below keeps the exact arithmetic the parity mode is pinned with.
                    const float xr = xe * 0.15915494309189535f;
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const float a0 = xr * __builtin_amdgcn_rcpf(dt[2 * pr]);
                        v[2 * pr] = __builtin_amdgcn_sinf(a0);
                        v[2 * pr + 1] = __builtin_amdgcn_cosf(dt[2 * pr] == dt[2 * pr + 1] ? a0 : xr * __builtin_amdgcn_rcpf(dt[2 * pr + 1]));
                    }
                    store4(out, odt, (b * L + t) * (int64_t)D + i, v);
                    continue;
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const float a0 = __fdiv_rn(xe, dt[2 * pr]);
                    if (dt[2 * pr] == dt[2 * pr + 1]) {
                        float sn, cs;
                        sincosf(a0, &sn, &cs);
                        v[2 * pr] = sn; v[2 * pr + 1] = cs;
                    } else {
                        v[2 * pr] = sinf(a0);
                        v[2 * pr + 1] = cosf(__fdiv_rn(xe, dt[2 * pr + 1]));
                    }
                }
                store4(out, odt, (b * L + t) * (int64_t)D + i, v);
            }
        } else {
            for (int i = threadIdx.x; i < D; i += ROW_THREADS) {
                float ang = __fdiv_rn(xe, dim_t[i]);
                float v = (i & 1) ? cosf(ang) : sinf(ang);
                store_from_f32(out, odt, (b * L + t) * (int64_t)D + i, v);
            }
        }
    }
}

// ---- masked softmax over segments ---------------------------------------------------------------
// logits [M_outer, R, S] f32; one wave per (m, r) row; S <= 64*MAXS.
__global__ __launch_bounds__(ROW_THREADS) void masked_softmax_kernel(const float* logits, int64_t ldl, const float* mask,
                                                                     int64_t ldm, void* probs, int pdt, int64_t ldp,
                                                                     int64_t rows, int64_t R, int S, int S_pad, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (ROW_THREADS / WAVE) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t m = row / R;
    const float* lp = logits + row * ldl;
    const float* mp = mask ? mask + m * ldm : nullptr;
    float mx = -INFINITY;
    for (int s = lane; s < S; s += WAVE) {
        float v = (mp && mp[s] == 0.f) ? -INFINITY : lp[s] * scale;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int s = lane; s < S; s += WAVE) {
        float v = (mp && mp[s] == 0.f) ? -INFINITY : lp[s] * scale;
        sum += expf(v - mx);
    }
    sum = wave_sum(sum);
    for (int s = lane; s < S_pad; s += WAVE) {
        float p = 0.f;
        if (s < S) {
            float v = (mp && mp[s] == 0.f) ? -INFINITY : lp[s] * scale;
            p = expf(v - mx) / sum;                     // all-masked row: exp(nan) -> NaN like the reference
        }
        store_from_f32(probs, pdt, row * ldp + s, p);
    }
}

// ---- X-Pool tail: LayerNorm3 + cosine with the video --------------------------------------------
template <int NV, bool FULL>
__global__ __launch_bounds__(ROW_THREADS, NV <= 4 ? 4 : 2) void xpool_tail_kernel(const void* y, int ydt, int64_t ldy, const float* gamma,
                                                                 const float* beta, const float* video, int64_t ldv,
                                                                 float* pooled, float* sims, int64_t lds_, int64_t rows,
                                                                 int64_t Nv, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (ROW_THREADS / WAVE) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t m = row / Nv, n = row % Nv;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * WAVE + lane) * 4;
        if (FULL || c < D) {
            v[i] = load4(y, ydt, row * ldy + c);
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * WAVE + lane) * 4;
        if (FULL || c < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = v[i][j] - mean; sq += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
    float pp = 0.f, vv = 0.f, pv = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * WAVE + lane) * 4;
        if (FULL || c < D) {
            f32x4 g = *(const f32x4*)(gamma + c), bb = *(const f32x4*)(beta + c), o;
            f32x4 vid = *(const f32x4*)(video + n * ldv + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (v[i][j] - mean) * rstd * g[j] + bb[j];
                pp += o[j] * o[j];
                vv += vid[j] * vid[j];
                pv += o[j] * vid[j];
            }
            if (pooled) *(f32x4*)(pooled + row * (int64_t)D + c) = o;
        }
    }
    pp = wave_sum(pp); vv = wave_sum(vv); pv = wave_sum(pv);
    if (lane == 0) sims[n * lds_ + m] = pv / (sqrtf(pp) * sqrtf(vv));
}

// ---- per-row affine + activation: out[r, :] = act(x[r, :] * scale[r % period] + shift[r % period]) -------------------
// (eval-mode BatchNorm1d over the token axis of a [B, T, F] tensor, reference model/model_Base.py:224-229)
__global__ __launch_bounds__(ROW_THREADS) void row_affine_kernel(const void* x, int xdt, int64_t ldx, const float* scale, const float* shift,
                                                                int period, int act, void* out, int odt, int64_t ldo, int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (ROW_THREADS / WAVE) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float sc = scale[row % period], sh = shift[row % period];
    for (int c = lane; c < cols; c += WAVE) {
        float v = load_as_f32(x, xdt, row * ldx + c) * sc + sh;
        if (act == MADE_ACT_RELU) v = fmaxf(v, 0.f);
        store_from_f32(out, odt, row * ldo + c, v);
    }
}

// ---- symmetric cross entropy ----------------------------------------------------------------------
// One workgroup of 1024 threads; rows then columns, each wave strides over lines.
// row_exclude [n, n] (optional): entries that are 1 are left out of the ROW-direction softmax (video -> music): the negatives that
// share the row's own music track (reference modules/loss.py:90-114); the column direction always uses every entry.
__global__ __launch_bounds__(1024) void clip_loss_kernel(const float* sims, int64_t ld, int n, const float* logit_scale,
                                                         float weight, int accumulate, float* loss_out, const float* row_exclude) {
    __shared__ float partial[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const float gsc = expf(logit_scale[0]);
    float acc = 0.f;                       // sum over lines of (lse - z_ii), both directions
    for (int dir = 0; dir < 2; ++dir) {
        for (int i = wave; i < n; i += nwaves) {
            const float* ex = (dir == 0 && row_exclude) ? row_exclude + (int64_t)i * n : nullptr;
            float mx = -INFINITY;
            for (int j = lane; j < n; j += WAVE) {
                float z = (dir == 0 ? sims[(int64_t)i * ld + j] : sims[(int64_t)j * ld + i]) * gsc;
                if (ex && ex[j] != 0.f) z = -INFINITY;
                mx = fmaxf(mx, z);
            }
            mx = wave_max(mx);
            float se = 0.f;
            for (int j = lane; j < n; j += WAVE) {
                float z = (dir == 0 ? sims[(int64_t)i * ld + j] : sims[(int64_t)j * ld + i]) * gsc;
                if (ex && ex[j] != 0.f) z = -INFINITY;
                se += expf(z - mx);
            }
            se = wave_sum(se);
            if (lane == 0) acc += (mx + logf(se)) - sims[(int64_t)i * ld + i] * gsc;
        }
    }
    if (lane == 0) partial[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < nwaves; ++w) t += partial[w];
        float loss = weight * 0.5f * (t / (float)n);
        loss_out[0] = accumulate ? loss_out[0] + loss : loss;
    }
}


// ---- split-K finish: sum partials + bias/act/residual (+ LayerNorm, + second LayerNorm) -----------
__device__ __forceinline__ float finish_act(float x, int act) {
    switch (act) {
        case MADE_ACT_RELU: return fmaxf(x, 0.f);
        case MADE_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
        case MADE_ACT_QUICKGELU: return x / (1.f + expf(-1.702f * x));
        case MADE_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        default: return x;
    }
}

template <int NV, bool FULL>
__device__ __forceinline__ void wave_layernorm(f32x4 (&v)[NV], int D, int lane, const float* g, const float* b, float eps) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (FULL || (i * WAVE + lane) * 4 < D) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (FULL || (i * WAVE + lane) * 4 < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = v[i][j] - mean; sq += d * d; }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * WAVE + lane) * 4;
        if (FULL || c < D) {
            f32x4 gg = *(const f32x4*)(g + c), bb = *(const f32x4*)(b + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = (v[i][j] - mean) * rstd * gg[j] + bb[j];
        }
    }
}


// ---- LayerNorm with a second output y2 = y + add (gamma == nullptr: y = x) --------------------------
template <int NV, bool FULL>
__global__ __launch_bounds__(ROW_THREADS, NV <= 4 ? 4 : 2) void layernorm_add_kernel(const void* x, int xdt, int64_t ldx,
                                                                const float* gamma, const float* beta, void* y, int ydt, int64_t ldy,
                                                                const void* add, int adt, int64_t lda, void* y2, int64_t ldy2,
                                                                int64_t rows, int D, float eps, const float* row_skip) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (ROW_THREADS / WAVE) + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (row_skip && row_skip[row] == 0.f) return;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * WAVE + lane) * 4;
        if (FULL || c < D) v[i] = load4(x, xdt, row * ldx + c);
    }
    if (gamma) wave_layernorm<NV, FULL>(v, D, lane, gamma, beta, eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * WAVE + lane) * 4;
        if (FULL || c < D) {
            if (y) store4(y, ydt, row * ldy + c, v[i]);
            f32x4 o = v[i];
            if (ydt == MADE_BF16) {                       // y2 is built from the value y actually holds
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (float)(bf16_t)o[j];
            }
            f32x4 ad = load4(add, adt, row * lda + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += ad[j];
            store4(y2, ydt, row * ldy2 + c, o);
        }
    }
}

template <bool WITH_LN, int NV, bool FULL>
__global__ __launch_bounds__(ROW_THREADS, NV <= 4 ? 4 : 2) void splitk_finish_kernel(const MadeFinishArgs a) {
    // WITH_LN: one WORKGROUP per row: the 4 waves each sum a quarter of the K-splits (all their partial loads in flight
    //          at once), the quarters meet in LDS and wave 0 applies bias / act / residual and the LayerNorm(s).
    // !WITH_LN: one wave per (row, 256-column chunk): blockIdx.y = chunk.
    __shared__ f32x4 part[WITH_LN ? 3 * NV * WAVE : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = WITH_LN ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * (ROW_THREADS / WAVE) + wave;
    if (row >= a.M) return;
    const int N = (int)a.N;
    const int64_t rr = a.r_row_mod > 0 ? row % a.r_row_mod : row;
    const int64_t s_begin = WITH_LN ? wave : 0, s_step = WITH_LN ? 4 : 1;
    f32x4 keep[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = ((WITH_LN ? i : (int)blockIdx.y) * WAVE + lane) * 4;
        f32x4 acc; acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
        if (FULL || c < N) {
            int64_t s = s_begin;
            for (; s + 7 * s_step < a.split_k; s += 8 * s_step) {           // eight independent loads in flight
                f32x4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(a.ws + ((s + u * s_step) * a.M + row) * N + c);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] += ((t[0][j] + t[1][j]) + (t[2][j] + t[3][j])) + ((t[4][j] + t[5][j]) + (t[6][j] + t[7][j]));
            }
            for (; s < a.split_k; s += s_step) {
                f32x4 t = *(const f32x4*)(a.ws + (s * a.M + row) * N + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += t[j];
            }
        }
        keep[i] = acc;
    }
    if constexpr (WITH_LN) {
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) part[((wave - 1) * NV + i) * WAVE + lane] = keep[i];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                f32x4 t = part[(w * NV + i) * WAVE + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) keep[i][j] += t[j];
            }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = ((WITH_LN ? i : (int)blockIdx.y) * WAVE + lane) * 4;
        if (!FULL && c >= N) continue;
        f32x4 acc = keep[i];
        if (a.bias) {
            f32x4 bb = *(const f32x4*)(a.bias + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += bb[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = finish_act(acc[j], a.act);
        if (a.R) {
            f32x4 t = load4(a.R, a.r_dtype, rr * a.ldr + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += t[j];
        }
        if (a.out) store4(a.out, a.out_dtype, row * a.ldo + c, acc);
        keep[i] = acc;
    }
    if constexpr (WITH_LN) {
        wave_layernorm<NV, FULL>(keep, N, lane, a.ln1_g, a.ln1_b, a.eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = (i * WAVE + lane) * 4;
            if ((FULL || c < N) && a.ln1_out) store4(a.ln1_out, a.ln1_dtype, row * a.ln1_ld + c, keep[i]);
        }
        if (a.ln2_g) {
            // the second norm sees what the first one STORED (rounded to its dtype), like a separate kernel would
            if (a.ln1_out && a.ln1_dtype == MADE_BF16) {
#pragma unroll
                for (int i = 0; i < NV; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) keep[i][j] = (float)(bf16_t)keep[i][j];
            }
            wave_layernorm<NV, FULL>(keep, N, lane, a.ln2_g, a.ln2_b, a.eps);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                int c = (i * WAVE + lane) * 4;
                if (FULL || c < N) store4(a.ln2_out, a.ln2_dtype, row * a.ln2_ld + c, keep[i]);
            }
        }
    }
}

// ---- f32 rows -> compute dtype with the row mask applied (feeds the direct-to-LDS GEMM) -------------
__global__ __launch_bounds__(ROW_THREADS) void cast_mask_rows_kernel(const float* x, int64_t ldx, const float* mask, void* y, int ydt,
                                                                     int64_t ldy, int64_t rows, int D) {
    const int64_t chunks_per_row = D / 8;
    const int64_t idx = (int64_t)blockIdx.x * ROW_THREADS + threadIdx.x;
    if (idx >= rows * chunks_per_row) return;
    const int64_t row = idx / chunks_per_row;
    const int c = (int)(idx % chunks_per_row) * 8;
    f32x4 a0 = *(const f32x4*)(x + row * ldx + c), a1 = *(const f32x4*)(x + row * ldx + c + 4);
    if (mask && mask[row] == 0.f) { a0[0] = a0[1] = a0[2] = a0[3] = 0.f; a1 = a0; }
    store4(y, ydt, row * ldy + c, a0);
    store4(y, ydt, row * ldy + c + 4, a1);
}

// out[b, :Ca] = a[b, :], out[b, Ca:] = c[b, :] (f32): the DETR token mask = [frame mask ; segment mask] (reference model/model_Uni.py:209)
__global__ void concat_cols_kernel(const float* a, int Ca, const float* c, int Cc, float* out, int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C = Ca + Cc;
    if (idx >= rows * C) return;
    const int64_t b = idx / C;
    const int j = (int)(idx % C);
    out[idx] = j < Ca ? a[b * Ca + j] : c[b * Cc + (j - Ca)];
}

inline unsigned row_blocks(int64_t rows) { return (unsigned)((rows + 3) / 4); }
inline int nv_for(int64_t D) { return D <= 512 ? 2 : (D <= 1024 ? 4 : 8); }
// NV = 16-byte vectors per lane; FULL = the row fills them exactly (D == 256*NV), so the bounds checks fold away and the
// loads are unconditional (conditional loads get serialised by hipcc: one s_waitcnt per load)
#define DISPATCH_NV(D, CALL)                                                      \
    switch (nv_for(D)) {                                                          \
        case 2: { constexpr int NV = 2; if ((D) == 512) { constexpr bool FULL = true; CALL; } else { constexpr bool FULL = false; CALL; } } break; \
        case 4: { constexpr int NV = 4; if ((D) == 1024) { constexpr bool FULL = true; CALL; } else { constexpr bool FULL = false; CALL; } } break; \
        default: { constexpr int NV = 8; constexpr bool FULL = false; CALL; } break; \
    }

// One record per music track for the sharded retrieval's single all-gather (reference test-MaDe.py:386-403 run over 8 GPUs: the music side
// is exchanged once): [S * D segment embeddings in the travel dtype | S mask floats | D floats of the pooled vector | pad to 16 bytes].
// One workgroup per record; rows n .. n_pad - 1 (the shards are padded to the largest one) are zero-filled.
__global__ __launch_bounds__(ROW_THREADS) void pack_music_records_kernel(const void* seg, int seg_dtype, int64_t seg_bs, const float* mask, int64_t ld_mask,
                                                                          const float* music, int64_t ld_music, unsigned char* out, int pack_dtype,
                                                                          int64_t rec, int64_t n, int S, int D) {
    const int64_t m = blockIdx.x;
    unsigned char* o = out + m * rec;
    const int tid = threadIdx.x;
    if (m >= n) {
        for (int64_t i = tid; i < rec / 16; i += ROW_THREADS) ((u32x4*)o)[i] = (u32x4){0u, 0u, 0u, 0u};
        return;
    }
    const int64_t ne = (int64_t)S * D;
    const int64_t a = ne * (pack_dtype == MADE_F32 ? 4 : 2), b = a + (int64_t)S * 4, c = b + (int64_t)D * 4;
    // segment embeddings: 8 elements per thread and step (ne is a multiple of 8: D is)
    for (int64_t e = (int64_t)tid * 8; e < ne; e += (int64_t)ROW_THREADS * 8) {
        float v[8];
        if (seg_dtype == MADE_F32) {
            const f32x4 x0 = *(const f32x4*)((const float*)seg + m * seg_bs + e), x1 = *(const f32x4*)((const float*)seg + m * seg_bs + e + 4);
            v[0] = x0[0]; v[1] = x0[1]; v[2] = x0[2]; v[3] = x0[3]; v[4] = x1[0]; v[5] = x1[1]; v[6] = x1[2]; v[7] = x1[3];
        } else {
            const bf16x8 x = *(const bf16x8*)((const bf16_t*)seg + m * seg_bs + e);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
        }
        if (pack_dtype == MADE_F32) {
            *(f32x4*)(o + e * 4) = (f32x4){v[0], v[1], v[2], v[3]};
            *(f32x4*)(o + e * 4 + 16) = (f32x4){v[4], v[5], v[6], v[7]};
        } else {
            bf16x8 t;
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = (bf16_t)v[j];
            *(bf16x8*)(o + e * 2) = t;
        }
    }
    for (int i = tid; i < S; i += ROW_THREADS) ((float*)(o + a))[i] = mask[m * ld_mask + i];
    for (int i = tid; i < D; i += ROW_THREADS) ((float*)(o + b))[i] = music[m * ld_music + i];
    for (int64_t i = c + (int64_t)tid * 4; i < rec; i += (int64_t)ROW_THREADS * 4) *(uint32_t*)(o + i) = 0u;
}

}  // namespace

extern "C" int made_layernorm(const void* x, int32_t x_dtype, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride,
                              const float* gamma, const float* beta,
                              void* y, int32_t y_dtype, int64_t ldy, int64_t rows, int64_t D, float eps,
                              const float* row_skip, void* stream) {
    MADE_REQUIRE(x && y && gamma && beta, "made_layernorm: null pointer");
    MADE_UNSUPPORTED(D > 0 && D % 4 == 0 && D <= 64 * 4 * MAX_VEC, "made_layernorm: D=%lld must be a multiple of 4 and <= %d",
                     (long long)D, 64 * 4 * MAX_VEC);
    MADE_UNSUPPORTED(ldx % 4 == 0 && ldy % 4 == 0 && x_batch_stride % 4 == 0, "made_layernorm: row strides must be multiples of 4");
    MADE_REQUIRE(x_rows_per_batch >= 0, "made_layernorm: negative x_rows_per_batch");
    if (rows <= 0) return MADE_OK;
    DISPATCH_NV(D, hipLaunchKernelGGL((layernorm_kernel<NV, FULL>), dim3(row_blocks(rows)), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       x, x_dtype, ldx, x_rows_per_batch, x_batch_stride, gamma, beta, y, y_dtype, ldy, rows, (int)D, eps, row_skip));
    return made_check_launch("made_layernorm");
}

extern "C" int made_masked_mean(const void* x, int32_t x_dtype, int64_t x_bs, int64_t ldx, const float* mask,
                                float* out, int64_t B, int64_t T, int64_t D, void* stream) {
    MADE_REQUIRE(x && out, "made_masked_mean: null pointer");
    MADE_REQUIRE(B >= 0 && T > 0 && D > 0 && B <= 65535, "made_masked_mean: bad dims");
    if (B == 0) return MADE_OK;
    const int esz = x_dtype == MADE_F32 ? 4 : 2;
    if (D % 4 == 0 && ldx % 4 == 0 && x_bs % 4 == 0 && ((uintptr_t)x % (4 * esz)) == 0 && ((uintptr_t)out % 16) == 0)
        hipLaunchKernelGGL(masked_mean_vec_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)B), dim3(1024), 0,
                           (hipStream_t)stream, x, x_dtype, x_bs, ldx, mask, out, T, (int)D);
    else
        hipLaunchKernelGGL(masked_mean_kernel, dim3((unsigned)((D + 63) / 64), (unsigned)B), dim3(ROW_THREADS), 0,
                           (hipStream_t)stream, x, x_dtype, x_bs, ldx, mask, out, T, (int)D);
    return made_check_launch("made_masked_mean");
}

extern "C" int made_l2norm_rows(const void* x, int32_t x_dtype, int64_t ldx, float* y_f32, void* y_alt, int32_t y_alt_dtype,
                                int64_t ldy, int64_t rows, int64_t D, float eps, void* stream) {
    MADE_REQUIRE(x && (y_f32 || y_alt), "made_l2norm_rows: null pointer");
    MADE_UNSUPPORTED(D > 0 && D % 4 == 0 && D <= 64 * 4 * MAX_VEC && ldx % 4 == 0 && ldy % 4 == 0, "made_l2norm_rows: bad D/strides");
    if (rows <= 0) return MADE_OK;
    DISPATCH_NV(D, hipLaunchKernelGGL((l2norm_kernel<NV, FULL>), dim3(row_blocks(rows)), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       x, x_dtype, ldx, y_f32, y_alt, y_alt_dtype, ldy, rows, (int)D, eps));
    return made_check_launch("made_l2norm_rows");
}

extern "C" int made_sine_pe(const float* mask, const float* dim_t, void* out, int32_t out_dtype,
                            int64_t B, int64_t L, int64_t D, void* stream) {
    MADE_REQUIRE(mask && dim_t && out, "made_sine_pe: null pointer");
    MADE_REQUIRE(B >= 0 && L > 0 && D > 0 && B <= 65535 && L < (1 << 30), "made_sine_pe: bad dims");
    if (B == 0) return MADE_OK;
    hipLaunchKernelGGL(sine_pe_kernel, dim3((unsigned)((L + PE_ROWS - 1) / PE_ROWS), (unsigned)B), dim3(ROW_THREADS), 0,
                       (hipStream_t)stream, mask, dim_t, out, out_dtype, (int)L, (int)D);
    return made_check_launch("made_sine_pe");
}

extern "C" int made_masked_softmax(const float* logits, int64_t ld_logits, const float* mask, int64_t ld_mask,
                                   void* probs, int32_t out_dtype, int64_t ldp,
                                   int64_t M_outer, int64_t R, int64_t S, int64_t S_pad, float scale, void* stream) {
    MADE_REQUIRE(logits && probs, "made_masked_softmax: null pointer");
    MADE_REQUIRE(S > 0 && S_pad >= S && ldp >= S_pad && ld_logits >= S, "made_masked_softmax: bad S/S_pad/strides");
    const int64_t rows = M_outer * R;
    if (rows <= 0) return MADE_OK;
    hipLaunchKernelGGL(masked_softmax_kernel, dim3(row_blocks(rows)), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       logits, ld_logits, mask, ld_mask, probs, out_dtype, ldp, rows, R, (int)S, (int)S_pad, scale);
    return made_check_launch("made_masked_softmax");
}

extern "C" int made_xpool_tail(const void* y, int32_t y_dtype, int64_t ldy, const float* gamma, const float* beta,
                               const float* video, int64_t ld_video, float* pooled_out, float* sims, int64_t ld_sims,
                               int64_t Nm, int64_t Nv, int64_t D, float eps, void* stream) {
    MADE_REQUIRE(y && gamma && beta && video && sims, "made_xpool_tail: null pointer");
    MADE_UNSUPPORTED(D > 0 && D % 4 == 0 && D <= 64 * 4 * MAX_VEC && ldy % 4 == 0 && ld_video % 4 == 0, "made_xpool_tail: bad D/strides");
    const int64_t rows = Nm * Nv;
    if (rows <= 0) return MADE_OK;
    DISPATCH_NV(D, hipLaunchKernelGGL((xpool_tail_kernel<NV, FULL>), dim3(row_blocks(rows)), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       y, y_dtype, ldy, gamma, beta, video, ld_video, pooled_out, sims, ld_sims, rows, Nv, (int)D, eps));
    return made_check_launch("made_xpool_tail");
}

extern "C" int made_clip_loss(const float* sims, int64_t ld, int64_t n, const float* logit_scale, float weight,
                              int32_t accumulate, float* loss_out, const float* row_exclude, void* stream) {
    MADE_REQUIRE(sims && logit_scale && loss_out, "made_clip_loss: null pointer");
    MADE_REQUIRE(n > 0 && n <= 4096 && ld >= n, "made_clip_loss: n=%lld out of range", (long long)n);
    hipLaunchKernelGGL(clip_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sims, ld, (int)n, logit_scale,
                       weight, accumulate, loss_out, row_exclude);
    return made_check_launch("made_clip_loss");
}

extern "C" int made_row_affine(const void* x, int32_t x_dtype, int64_t ldx, const float* scale, const float* shift, int64_t period,
                               int32_t act, void* out, int32_t out_dtype, int64_t ldo, int64_t rows, int64_t cols, void* stream) {
    MADE_REQUIRE(x && scale && shift && out && period > 0 && cols > 0, "made_row_affine: bad arguments");
    MADE_UNSUPPORTED(act == MADE_ACT_NONE || act == MADE_ACT_RELU, "made_row_affine: act %d (none or ReLU)", act);
    if (rows <= 0) return MADE_OK;
    hipLaunchKernelGGL(row_affine_kernel, dim3(row_blocks(rows)), dim3(ROW_THREADS), 0, (hipStream_t)stream, x, x_dtype, ldx, scale, shift,
                       (int)period, act, out, out_dtype, ldo, rows, (int)cols);
    return made_check_launch("made_row_affine");
}

extern "C" int made_splitk_finish(const MadeFinishArgs* args, void* stream) {
    MADE_REQUIRE(args != nullptr && args->ws != nullptr, "made_splitk_finish: null args/ws");
    const MadeFinishArgs& a = *args;
    MADE_REQUIRE(a.split_k >= 1 && a.M >= 0 && a.N > 0, "made_splitk_finish: bad dims");
    MADE_UNSUPPORTED(a.N % 4 == 0 && a.ldo % 4 == 0 && a.ldr % 4 == 0 && a.ln1_ld % 4 == 0 && a.ln2_ld % 4 == 0,
                     "made_splitk_finish: N and row strides must be multiples of 4");
    const bool ln = a.ln1_g != nullptr;
    MADE_REQUIRE(a.out != nullptr || ln, "made_splitk_finish: nothing to write");
    if (ln) {
        MADE_REQUIRE(a.ln1_b != nullptr && (a.ln1_out != nullptr || a.ln2_g != nullptr), "made_splitk_finish: incomplete ln1 arguments");
        MADE_UNSUPPORTED(a.N <= 64 * 4 * MAX_VEC, "made_splitk_finish: LayerNorm variant needs N <= %d", 64 * 4 * MAX_VEC);
        if (a.ln2_g) MADE_REQUIRE(a.ln2_b != nullptr && a.ln2_out != nullptr, "made_splitk_finish: incomplete ln2 arguments");
    } else {
        MADE_REQUIRE(a.ln2_g == nullptr, "made_splitk_finish: ln2 without ln1");
    }
    if (a.M == 0) return MADE_OK;
    if (ln) { DISPATCH_NV(a.N, hipLaunchKernelGGL((splitk_finish_kernel<true, NV, FULL>), dim3((unsigned)a.M), dim3(ROW_THREADS), 0, (hipStream_t)stream, a)); }
    else if (a.N % 256 == 0) hipLaunchKernelGGL((splitk_finish_kernel<false, 1, true>), dim3(row_blocks(a.M), (unsigned)(a.N / 256)), dim3(ROW_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((splitk_finish_kernel<false, 1, false>), dim3(row_blocks(a.M), (unsigned)((a.N + 255) / 256)), dim3(ROW_THREADS), 0, (hipStream_t)stream, a);
    return made_check_launch("made_splitk_finish");
}

extern "C" int made_layernorm_add(const void* x, int32_t x_dtype, int64_t ldx, const float* gamma, const float* beta,
                                  void* y, int32_t y_dtype, int64_t ldy, const void* add, int32_t add_dtype, int64_t ld_add,
                                  void* y2, int64_t ldy2, int64_t rows, int64_t D, float eps, const float* row_skip, void* stream) {
    MADE_REQUIRE(x && add && y2 && (gamma == nullptr || beta != nullptr), "made_layernorm_add: null pointer");
    MADE_UNSUPPORTED(D > 0 && D % 4 == 0 && D <= 64 * 4 * MAX_VEC && ldx % 4 == 0 && ldy % 4 == 0 && ld_add % 4 == 0 && ldy2 % 4 == 0,
                     "made_layernorm_add: D and row strides must be multiples of 4 (D <= %d)", 64 * 4 * MAX_VEC);
    if (rows <= 0) return MADE_OK;
    DISPATCH_NV(D, hipLaunchKernelGGL((layernorm_add_kernel<NV, FULL>), dim3(row_blocks(rows)), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       x, x_dtype, ldx, gamma, beta, y, y_dtype, ldy, add, add_dtype, ld_add, y2, ldy2, rows, (int)D, eps, row_skip));
    return made_check_launch("made_layernorm_add");
}

extern "C" int made_cast_mask_rows(const float* x, int64_t ldx, const float* mask, void* y, int32_t y_dtype, int64_t ldy,
                                   int64_t rows, int64_t D, void* stream) {
    MADE_REQUIRE(x && y, "made_cast_mask_rows: null pointer");
    MADE_UNSUPPORTED(D > 0 && D % 8 == 0 && ldx % 4 == 0 && ldy % 8 == 0, "made_cast_mask_rows: D must be a multiple of 8, strides aligned");
    if (rows <= 0) return MADE_OK;
    const int64_t n = rows * (D / 8);
    hipLaunchKernelGGL(cast_mask_rows_kernel, dim3((unsigned)((n + ROW_THREADS - 1) / ROW_THREADS)), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       x, ldx, mask, y, y_dtype, ldy, rows, (int)D);
    return made_check_launch("made_cast_mask_rows");
}

extern "C" int made_concat_cols(const float* a, int64_t cols_a, const float* c, int64_t cols_c, float* out, int64_t rows, void* stream) {
    MADE_REQUIRE(out && rows >= 0 && cols_a >= 0 && cols_c >= 0 && (a || cols_a == 0) && (c || cols_c == 0), "made_concat_cols: bad arguments");
    const int64_t n = rows * (cols_a + cols_c);
    if (n == 0) return MADE_OK;
    hipLaunchKernelGGL(concat_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, (int)cols_a, c, (int)cols_c, out, rows);
    return made_check_launch("made_concat_cols");
}

extern "C" int made_pack_music_records(const void* seg, int32_t seg_dtype, int64_t seg_bs, const float* mask, int64_t ld_mask, const float* music, int64_t ld_music,
                                       void* out, int32_t pack_dtype, int64_t rec_bytes, int64_t n, int64_t n_pad, int64_t S, int64_t D, void* stream) {
    MADE_REQUIRE(out != nullptr && n >= 0 && n_pad >= n && S > 0 && D > 0, "made_pack_music_records: bad arguments");
    MADE_REQUIRE(n == 0 || (seg && mask && music), "made_pack_music_records: null input");
    MADE_REQUIRE((seg_dtype == MADE_F32 || seg_dtype == MADE_BF16) && (pack_dtype == MADE_F32 || pack_dtype == MADE_BF16), "made_pack_music_records: bad dtype");
    const int64_t c = S * D * (pack_dtype == MADE_F32 ? 4 : 2) + S * 4 + D * 4;
    MADE_UNSUPPORTED(D % 8 == 0 && rec_bytes % 16 == 0 && rec_bytes >= c && ((uintptr_t)out % 16) == 0 && ((uintptr_t)seg % 16) == 0 &&
                     seg_bs % 8 == 0, "made_pack_music_records: D must be a multiple of 8, records of whole 16-byte groups, aligned buffers");
    MADE_UNSUPPORTED(n_pad <= 0x7fffffff, "made_pack_music_records: too many records");
    if (n_pad == 0) return MADE_OK;
    hipLaunchKernelGGL(pack_music_records_kernel, dim3((unsigned)n_pad), dim3(ROW_THREADS), 0, (hipStream_t)stream, seg, (int)seg_dtype, seg_bs, mask, ld_mask,
                       music, ld_music, (unsigned char*)out, (int)pack_dtype, rec_bytes, n, (int)S, (int)D);
    return made_check_launch("made_pack_music_records");
}
