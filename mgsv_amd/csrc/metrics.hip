// Evaluation metrics that follow the similarity matrix (reference utils/util_test.py:44-96, music_detr/span_utils.py:119-170),
// on the device: the [N_v, N_m] matrix never leaves HBM, only one rank per video does.
#include "common.h"

namespace {

constexpr int MT = 256;

__device__ __forceinline__ int float_order(float f) {          // monotone map float -> int (for atomicMax on floats)
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}

// One workgroup per video row.  Music tracks that share an id form a group (gid); the de-duplicated rank of the ground-truth
// track is the number of OTHER groups whose best similarity beats the ground-truth group's best one -- exactly what walking the
// descending sort while skipping already-seen ids counts (reference utils/util_test.py:44-70).
__global__ __launch_bounds__(MT) void recall_rank_kernel(const float* sims, int64_t ld, const int32_t* gid, const int32_t* gt_gid,
                                                         int Nm, int G, int32_t* rank_out, int32_t* top1_out) {
    extern __shared__ int gmax[];                              // [G] ordered-int maxima
    __shared__ int red_cnt[MT / 64];
    __shared__ float red_best[MT / 64];
    __shared__ int red_arg[MT / 64];
    const int64_t row = blockIdx.x;
    const float* s = sims + row * ld;
    const int NEG = float_order(-INFINITY);
    for (int g = threadIdx.x; g < G; g += MT) gmax[g] = NEG;
    __syncthreads();
    float best = -INFINITY;
    int arg = 0x7FFFFFFF;
    for (int j = threadIdx.x; j < Nm; j += MT) {
        const float v = s[j];
        atomicMax(&gmax[gid[j]], float_order(v));
        if (v > best || (v == best && j < arg)) { best = v; arg = j; }
    }
    __syncthreads();
    const int mine = gt_gid[row];
    const int ref = gmax[mine];
    int cnt = 0;
    for (int g = threadIdx.x; g < G; g += MT) cnt += (g != mine && gmax[g] > ref) ? 1 : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o);
        const float ob = __shfl_xor(best, o);
        const int oa = __shfl_xor(arg, o);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) { red_cnt[wave] = cnt; red_best[wave] = best; red_arg[wave] = arg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        float b = -INFINITY;
        int a = 0x7FFFFFFF;
        for (int w = 0; w < MT / 64; ++w) {
            c += red_cnt[w];
            if (red_best[w] > b || (red_best[w] == b && red_arg[w] < a)) { b = red_best[w]; a = red_arg[w]; }
        }
        rank_out[row] = c;
        if (top1_out) top1_out[row] = a;
    }
}

// top-scoring query's span -> seconds, clamped, IoU with the ground-truth moment (reference test-MaDe.py:304-313,
// music_detr/span_utils.py:119-170 with discounted = False)
__global__ void span_iou_kernel(const float* logits, const float* spans, const float* gt_moment, const float* m_duration,
                                int64_t N, int Q, int fg, float max_m_duration, float* iou_out, float* pred_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int bq = 0;
    float bs = -INFINITY;
    for (int q = 0; q < Q; ++q) {
        const float l0 = logits[(i * Q + q) * 2], l1 = logits[(i * Q + q) * 2 + 1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const float p = (fg == 0 ? e0 : e1) / (e0 + e1);
        if (p > bs) { bs = p; bq = q; }                          // sorted(..., reverse=True) is stable: the first maximum wins
    }
    const float c = spans[(i * Q + bq) * 2], w = spans[(i * Q + bq) * 2 + 1];
    float ps = (c - 0.5f * w) * max_m_duration, pe = (c + 0.5f * w) * max_m_duration;
    if (pred_out) { pred_out[i * 3] = ps; pred_out[i * 3 + 1] = pe; pred_out[i * 3 + 2] = bs; }
    ps = fmaxf(ps, 0.f);
    pe = fminf(pe, max_m_duration);                               // detr_iou clamps to args.max_m_duration ...
    const float gs = gt_moment[i * 2], ge = gt_moment[i * 2 + 1], dur = m_duration[i];
    float iou = 0.f;
    if (gs < ge) {
        ps = fmaxf(ps, 0.f);
        pe = fminf(pe, dur);                                      // ... and individual_IoU_tensor to the track's duration
        const float inter = fmaxf(fminf(ge, pe) - fmaxf(gs, ps), 0.f);
        const float uni = (pe - ps) + (ge - gs) - inter;
        iou = uni > 0.f ? inter / uni : 0.f;
    }
    iou_out[i] = iou;
}

// ---- the free functions the reference's drivers import ------------------------------------------------
// out[a * sa + p * sp] = cos(anchor[a], pooled[p, a]) -- reference modules/metrics.py:10-24 (music pooling: anchor = videos,
// pooled [bs_m, bs_v, D], out [bs_v, bs_m]) and :26-41 (video pooling: anchor = tracks, pooled [bs_v, bs_m, D]).  One wave per pair.
template <typename TP>
__global__ void pooled_cosine_kernel(const float* anchor, int64_t lda, const TP* pooled, float* out, int64_t sa, int64_t sp,
                                     int64_t A, int64_t P, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pair >= A * P) return;
    const int64_t p = pair / A, ai = pair % A;
    const float* ar = anchor + ai * lda;
    const TP* pr = pooled + pair * D;
    float dot = 0.f, na = 0.f, np_ = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float x = ar[c], y = (float)pr[c];
        dot += x * y; na += x * x; np_ += y * y;
    }
    dot = wave_sum(dot); na = wave_sum(na); np_ = wave_sum(np_);
    if (lane == 0) out[ai * sa + p * sp] = (dot / sqrtf(na)) / sqrtf(np_);       // x / |x| . y / |y| as the reference forms it
}

// out = x * exp(*logit_scale): the logits CLIPLoss / InfoNCELoss return beside the loss (reference modules/loss.py:12-13,86-88)
__global__ void scale_exp_kernel(const float* x, const float* logit_scale, float* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] * expf(*logit_scale);
}

// mode 0: (centre, width) -> (start, end); mode 1: (start, end) -> (centre, width)   (reference music_detr/span_utils.py:4-24)
__global__ void span_convert_kernel(const float* in, float* out, int64_t N, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float a = in[2 * i], b = in[2 * i + 1];
    if (mode == 0) { out[2 * i] = a - 0.5f * b; out[2 * i + 1] = a + 0.5f * b; }
    else { out[2 * i] = (a + b) * 0.5f; out[2 * i + 1] = b - a; }
}

// all pairs of two span lists: IoU, union, generalised IoU, intersection over the second span (reference span_utils.py:39-115)
__global__ void span_pairwise_kernel(const float* s1, const float* s2, float* iou, float* uni, float* giou, float* iop, int64_t N, int64_t M) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * M) return;
    const int64_t i = idx / M, j = idx % M;
    const float a0 = s1[2 * i], a1 = s1[2 * i + 1], b0 = s2[2 * j], b1 = s2[2 * j + 1];
    const float inter = fmaxf(fminf(a1, b1) - fmaxf(a0, b0), 0.f);
    const float u = (a1 - a0) + (b1 - b0) - inter;
    const float io = inter / u;
    if (iou) iou[idx] = io;
    if (uni) uni[idx] = u;
    if (giou) { const float enc = fmaxf(fmaxf(a1, b1) - fminf(a0, b0), 0.f); giou[idx] = io - (enc - u) / enc; }
    if (iop) iop[idx] = inter / (b1 - b0);
}

// IoU of one predicted (start, end) [seconds] per sample with its ground truth (reference span_utils.py:119-170: detr_iou clamps the
// prediction to [0, max_m_duration], individual_IoU_tensor to [0, m_duration]; discounted: x (1 - |d start| / dur)(1 - |d end| / dur))
__global__ void span_iou_se_kernel(const float* pred, const float* gt, const float* dur, int64_t N, float max_dur, int clamp_max, int discounted, float* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float ps = pred[2 * i], pe = pred[2 * i + 1];
    if (clamp_max) { ps = fmaxf(ps, 0.f); pe = fminf(pe, max_dur); }
    const float gs = gt[2 * i], ge = gt[2 * i + 1], d = dur[i];
    float v = 0.f;
    if (gs < ge) {
        ps = fmaxf(ps, 0.f); pe = fminf(pe, d);
        const float inter = fmaxf(fminf(ge, pe) - fmaxf(gs, ps), 0.f);
        const float u = (pe - ps) + (ge - gs) - inter;
        if (u > 0.f) {
            v = inter / u;
            if (discounted) v = v * (1.f - fabsf(gs - ps) / d) * (1.f - fabsf(ge - pe) / d);
        }
    }
    out[i] = v;
}

}  // namespace

extern "C" int made_recall_ranks(const float* sims, int64_t ld, const int32_t* group_id, const int32_t* gt_group, int64_t Nv, int64_t Nm,
                                 int64_t n_groups, int32_t* rank_out, int32_t* top1_out, void* stream) {
    MADE_REQUIRE(sims && group_id && gt_group && rank_out, "made_recall_ranks: null pointer");
    MADE_REQUIRE(Nv >= 0 && Nm > 0 && n_groups > 0 && ld >= Nm, "made_recall_ranks: bad dims");
    MADE_UNSUPPORTED(n_groups <= 32768 && Nv < (1LL << 31), "made_recall_ranks: at most 32768 distinct ids (LDS table)");
    if (Nv == 0) return MADE_OK;
    const size_t lds = (size_t)n_groups * sizeof(int);
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)recall_rank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(recall_rank_kernel, dim3((unsigned)Nv), dim3(MT), lds, (hipStream_t)stream, sims, ld, group_id, gt_group, (int)Nm,
                       (int)n_groups, rank_out, top1_out);
    return made_check_launch("made_recall_ranks");
}

extern "C" int made_span_iou(const float* pred_logits, const float* pred_spans, const float* gt_moment, const float* m_duration,
                             int64_t N, int64_t Q, int32_t fg_label, float max_m_duration, float* iou_out, float* pred_out, void* stream) {
    MADE_REQUIRE(pred_logits && pred_spans && gt_moment && m_duration && iou_out, "made_span_iou: null pointer");
    MADE_REQUIRE(N >= 0 && Q >= 1, "made_span_iou: bad dims");
    if (N == 0) return MADE_OK;
    hipLaunchKernelGGL(span_iou_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred_logits, pred_spans, gt_moment,
                       m_duration, N, (int)Q, (int)fg_label, max_m_duration, iou_out, pred_out);
    return made_check_launch("made_span_iou");
}

extern "C" int made_pooled_cosine(const float* anchor, int64_t lda, const void* pooled, int32_t pooled_dtype, float* out, int64_t sa,
                                  int64_t sp, int64_t A, int64_t P, int64_t D, void* stream) {
    MADE_REQUIRE(anchor && pooled && out, "made_pooled_cosine: null pointer");
    MADE_REQUIRE(A >= 0 && P >= 0 && D > 0 && lda >= D, "made_pooled_cosine: bad dims");
    if (A * P == 0) return MADE_OK;
    const dim3 grid((unsigned)((A * P + 3) / 4)), block(256);
    if (pooled_dtype == MADE_F32) hipLaunchKernelGGL(pooled_cosine_kernel<float>, grid, block, 0, (hipStream_t)stream, anchor, lda, (const float*)pooled, out, sa, sp, A, P, (int)D);
    else hipLaunchKernelGGL(pooled_cosine_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, anchor, lda, (const bf16_t*)pooled, out, sa, sp, A, P, (int)D);
    return made_check_launch("made_pooled_cosine");
}

extern "C" int made_scale_exp(const float* x, const float* logit_scale, float* out, int64_t n, void* stream) {
    MADE_REQUIRE(x && logit_scale && out && n >= 0, "made_scale_exp: bad arguments");
    if (n == 0) return MADE_OK;
    hipLaunchKernelGGL(scale_exp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, logit_scale, out, n);
    return made_check_launch("made_scale_exp");
}

extern "C" int made_span_convert(const float* in, float* out, int64_t N, int32_t mode, void* stream) {
    MADE_REQUIRE(in && out && N >= 0 && (mode == 0 || mode == 1), "made_span_convert: bad arguments");
    if (N == 0) return MADE_OK;
    hipLaunchKernelGGL(span_convert_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, N, (int)mode);
    return made_check_launch("made_span_convert");
}

extern "C" int made_span_pairwise(const float* spans1, const float* spans2, float* iou, float* uni, float* giou, float* inter_over_2,
                                  int64_t N, int64_t M, void* stream) {
    MADE_REQUIRE(spans1 && spans2 && N >= 0 && M >= 0, "made_span_pairwise: bad arguments");
    if (N * M == 0) return MADE_OK;
    hipLaunchKernelGGL(span_pairwise_kernel, dim3((unsigned)((N * M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, spans1, spans2, iou, uni, giou,
                       inter_over_2, N, M);
    return made_check_launch("made_span_pairwise");
}

extern "C" int made_span_iou_se(const float* pred_se, const float* gt_moment, const float* m_duration, int64_t N, float max_m_duration,
                                int32_t clamp_to_max, int32_t discounted, float* iou_out, void* stream) {
    MADE_REQUIRE(pred_se && gt_moment && m_duration && iou_out && N >= 0, "made_span_iou_se: bad arguments");
    if (N == 0) return MADE_OK;
    hipLaunchKernelGGL(span_iou_se_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred_se, gt_moment, m_duration, N,
                       max_m_duration, (int)clamp_to_max, (int)discounted, iou_out);
    return made_check_launch("made_span_iou_se");
}
