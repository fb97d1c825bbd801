// Hungarian matcher (cost + rectangular LSAP) and the DETR set criterion, on device.
//
// The reference moves the cost matrix to the host and calls SciPy once per sample and decoder layer
// (music_detr/matcher.py:89-91: 6 D2H syncs per forward).  Here one launch handles all
// n_layers*B samples: a wave computes the sample's cost block in f32 with the reference's operation
// order (explicit *_rn intrinsics AND contraction switched off for this file ON THE COMMAND LINE, csrc/Makefile: -ffp-contract=off.  HIP's
// __fadd_rn / __fmul_rn are plain + and * defined in headers, which the default -ffp-contract=fast fuses into FMAs; the pragma below
// does not reach those header bodies -- rounds 3-4 shipped fused products here without knowing),
// then lane 0 runs the shortest-augmenting-path LSAP (Crouse 2016, as SciPy implements it, same
// tie-break) on the f64-promoted block.  Q, G <= 64, so all solver state lives in LDS.
#include "common.h"
#pragma clang fp contract(off)

namespace {

constexpr int MAXN = 64;

__device__ __forceinline__ float giou_se(float s1, float e1, float s2, float e2) {
    // reference music_detr/span_utils.py:39-66, :86-115 (f32, left to right)
    float a1 = __fsub_rn(e1, s1), a2 = __fsub_rn(e2, s2);
    float inter = fmaxf(__fsub_rn(fminf(e1, e2), fmaxf(s1, s2)), 0.f);
    float uni = __fsub_rn(__fadd_rn(a1, a2), inter);
    float iou = __fdiv_rn(inter, uni);
    float enc = fmaxf(__fsub_rn(fmaxf(e1, e2), fminf(s1, s2)), 0.f);
    return __fsub_rn(iou, __fdiv_rn(__fsub_rn(enc, uni), enc));
}

__device__ __forceinline__ void cw_to_se(float c, float w, float& s, float& e) {
    float hw = __fmul_rn(0.5f, w);                     // reference span_utils.py:22-23
    s = __fsub_rn(c, hw);
    e = __fadd_rn(c, hw);
}

__global__ __launch_bounds__(64) void hungarian_kernel(const float* logits, const float* spans, const float* targets,
                                                       int B, int Q, int G, int fg, float w_span, float w_giou, float w_class,
                                                       float* cost_ws, int cost_is_input, int64_t* out_pred, int64_t* out_tgt,
                                                       int32_t* out_count, int32_t* status) {
    __shared__ int kept[MAXN];
    __shared__ int n_kept;
    __shared__ double u[MAXN], v[MAXN], spc[MAXN];
    __shared__ int path[MAXN], col4row[MAXN], row4col[MAXN], remaining[MAXN];
    __shared__ unsigned char SR[MAXN], SC[MAXN];
    __shared__ int bad;

    const int s = blockIdx.x, b = s % B, lane = threadIdx.x;
    const float* tg = targets + (int64_t)b * G * 2;
    if (lane == 0) {
        int k = 0;
        for (int g = 0; g < G; ++g)
            if (tg[2 * g + 1] != 0.f) kept[k++] = g;      // reference matcher.py:59-61
        n_kept = k;
        bad = 0;
    }
    __syncthreads();
    const int Gk = n_kept;
    float* C = cost_ws + (int64_t)s * Q * G;
    const float* lg = logits + (int64_t)s * Q * 2;
    const float* sp = spans + (int64_t)s * Q * 2;
    for (int idx = lane; idx < Q * Gk; idx += 64) {
        int q = idx / Gk, j = idx % Gk, g = kept[j];
        if (cost_is_input) {
            float c = C[q * G + j];
            if (c != c || c == -INFINITY) bad = 1;
            continue;
        }
        float l0 = lg[2 * q], l1 = lg[2 * q + 1];
        float mx = fmaxf(l0, l1);
        // Foreground probability of the 2-class softmax in the operation order torch's CPU softmax runs for the reference's line
        // (music_detr/matcher.py:58; aten vec_softmax_lastdim: e = exp(x - max) per element, s = sum, r = 1 / s, p = e * r, all f32),
        // with a CORRECTLY ROUNDED f32 exponential (the f64 exp rounded once).  torch's own exp there is a <= 1-ulp (SLEEF, torch 1.13)
        // or <= 2-ulp (exp_u20, torch 2.x) approximation whose bits differ between torch versions and between AVX2 and AVX-512 hosts,
        // so it has no single bit pattern to copy; what can be copied is the order of the roundings around it.  With it all 127
        // samples of the matcher fixture (SciPy on torch-CPU costs) are assigned identically; rounds 1-4 evaluated the quotient in
        // f64 and rounded once, which is closer to the real number but differs from torch's bits in 41 % of random arguments
        // (this form: 4.6 %, all of them torch's exp) and left two tie samples assigned the other way.
        const float e0 = (float)exp((double)__fsub_rn(l0, mx)), e1 = (float)exp((double)__fsub_rn(l1, mx));
        const float rs = __fdiv_rn(1.0f, __fadd_rn(e0, e1));
        float p = __fmul_rn(fg == 0 ? e0 : e1, rs);
        float pc = sp[2 * q], pw = sp[2 * q + 1], tc = tg[2 * g], tw = tg[2 * g + 1];
        float cost_span = __fadd_rn(fabsf(__fsub_rn(pc, tc)), fabsf(__fsub_rn(pw, tw)));
        float ps, pe, ts, te;
        cw_to_se(pc, pw, ps, pe);
        cw_to_se(tc, tw, ts, te);
        float cost_giou = -giou_se(ps, pe, ts, te);
        float cost_class = -p;
        float c = __fadd_rn(__fadd_rn(__fmul_rn(w_span, cost_span), __fmul_rn(w_giou, cost_giou)), __fmul_rn(w_class, cost_class));
        C[q * G + j] = c;
        if (c != c || c == -INFINITY) bad = 1;
    }
    __syncthreads();

    const int cnt = Q < Gk ? Q : Gk;
    const int width = Q < G ? Q : G;
    int64_t* op = out_pred + (int64_t)s * width;
    int64_t* ot = out_tgt + (int64_t)s * width;
    for (int i = lane; i < width; i += 64) { op[i] = -1; ot[i] = -1; }
    __syncthreads();
    if (lane != 0) return;
    out_count[s] = cnt;
    if (cnt == 0) return;
    if (bad) { atomicExch(status, 1); out_count[s] = 0; return; }

    const bool tr = Gk < Q;                               // SciPy transposes when nc < nr
    const int nr = tr ? Gk : Q, nc = tr ? Q : Gk;
    auto cost = [&](int i, int j) -> double { return (double)(tr ? C[j * G + i] : C[i * G + j]); };
    for (int i = 0; i < nr; ++i) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = 0; j < nc; ++j) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    const double INF = 1.0 / 0.0;
    for (int cur = 0; cur < nr; ++cur) {
        double min_val = 0.0;
        int i = cur, num_remaining = nc, sink = -1;
        for (int it = 0; it < nc; ++it) { remaining[it] = nc - it - 1; spc[it] = INF; SC[it] = 0; }
        for (int it = 0; it < nr; ++it) SR[it] = 0;
        while (sink == -1) {
            int index = -1;
            double lowest = INF;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                int j = remaining[it];
                double rr = min_val + cost(i, j) - u[i] - v[j];
                if (rr < spc[j]) { path[j] = i; spc[j] = rr; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            min_val = lowest;
            if (min_val == INF) { atomicExch(status, 1); out_count[s] = 0; return; }   // infeasible
            int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += min_val;
        for (int i2 = 0; i2 < nr; ++i2)
            if (SR[i2] && i2 != cur) u[i2] += min_val - spc[col4row[i2]];
        for (int j2 = 0; j2 < nc; ++j2)
            if (SC[j2]) v[j2] -= min_val - spc[j2];
        int j = sink;
        while (true) {
            int i2 = path[j];
            row4col[j] = i2;
            int tmp = col4row[i2]; col4row[i2] = j; j = tmp;
            if (i2 == cur) break;
        }
    }
    if (!tr) {
        for (int i = 0; i < nr; ++i) { op[i] = i; ot[i] = col4row[i]; }
    } else {
        // rows of the transposed problem are targets: emit pairs sorted by prediction index
        int done = 0, last = -1;
        while (done < nr) {
            int best = -1;
            for (int t = 0; t < nr; ++t)
                if (col4row[t] > last && (best < 0 || col4row[t] < col4row[best])) best = t;
            op[done] = col4row[best]; ot[done] = best; last = col4row[best]; ++done;
        }
    }
}

// ---- set criterion ---------------------------------------------------------------------------------
constexpr int CRIT_THREADS = 256;
constexpr int CRIT_MAX_BQ = 4096;

__device__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = (red[0] + red[1]) + (red[2] + red[3]);
    return t;
}

__device__ __forceinline__ int kth_kept(const float* tg, int G, int k) {
    int seen = 0;
    for (int g = 0; g < G; ++g)
        if (tg[2 * g + 1] != 0.f) { if (seen == k) return g; ++seen; }
    return 0;
}

__global__ __launch_bounds__(CRIT_THREADS) void criterion_kernel(const float* logits, const float* spans, const float* targets,
                                                                 const int64_t* pred_idx, const int64_t* tgt_idx, const int32_t* count,
                                                                 const float* proj_q, const float* vid_sum, const float* empty_w,
                                                                 int n_layers, int B, int Q, int G, int Dc, int fg, float temperature,
                                                                 const float* weights, float* losses, float* total) {
    __shared__ unsigned char matched[CRIT_MAX_BQ];
    __shared__ float lgt[CRIT_MAX_BQ];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int width = Q < G ? Q : G;
    const int bg = 1 - fg;
    {
        const int l = blockIdx.x;                      // one workgroup per decoder layer
        const float* lg = logits + (int64_t)l * B * Q * 2;
        const float* sp = spans + (int64_t)l * B * Q * 2;
        for (int i = tid; i < B * Q; i += CRIT_THREADS) matched[i] = 0;
        // contrastive-align logits first: they depend on nothing the matching produces, so their loads travel together with
        // the first loads of the matched-pair pass instead of adding a round trip at the end.  Four lanes per (b, q) pair,
        // 16-byte loads where the rows allow them.
        if (proj_q && vid_sum) {
            const bool vec = (Dc % 4 == 0) && (((uintptr_t)proj_q | (uintptr_t)vid_sum) % 16 == 0);
            for (int i = tid >> 2; i < B * Q; i += CRIT_THREADS / 4) {
                const int b = i / Q, part = tid & 3;
                const float* pq = proj_q + ((int64_t)l * B * Q + i) * Dc;
                const float* vs = vid_sum + (int64_t)b * Dc;
                float d = 0.f;
                if (vec) {
                    for (int k = part * 4; k < Dc; k += 16) {
                        const f32x4 x = *(const f32x4*)(pq + k), y = *(const f32x4*)(vs + k);
                        d += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
                    }
                } else {
                    for (int k = part; k < Dc; k += 4) d += pq[k] * vs[k];
                }
                d += __shfl_xor(d, 1);
                d += __shfl_xor(d, 2);
                if (part == 0) lgt[i] = d / temperature;
            }
        }
        __syncthreads();
        // matched pairs: one thread per (b, slot)
        float span_sum = 0.f, giou_sum = 0.f, correct = 0.f, npairs = 0.f;
        for (int i = tid; i < B * width; i += CRIT_THREADS) {
            int b = i / width, slot = i % width;
            int s = l * B + b;
            if (slot < count[s]) {
                int q = (int)pred_idx[(int64_t)s * width + slot];
                int g = kth_kept(targets + (int64_t)b * G * 2, G, (int)tgt_idx[(int64_t)s * width + slot]);
                float pc = sp[(b * Q + q) * 2], pw = sp[(b * Q + q) * 2 + 1];
                float tc = targets[((int64_t)b * G + g) * 2], tw = targets[((int64_t)b * G + g) * 2 + 1];
                span_sum += fabsf(pc - tc) + fabsf(pw - tw);
                float ps, pe, ts, te;
                cw_to_se(pc, pw, ps, pe);
                cw_to_se(tc, tw, ts, te);
                giou_sum += 1.f - giou_se(ps, pe, ts, te);
                float l0 = lg[(b * Q + q) * 2], l1 = lg[(b * Q + q) * 2 + 1];
                int top = l1 > l0 ? 1 : 0;
                correct += (top == fg) ? 1.f : 0.f;
                npairs += 1.f;
                matched[b * Q + q] = 1;
            }
        }
        span_sum = block_sum(span_sum, red);
        giou_sum = block_sum(giou_sum, red);
        correct = block_sum(correct, red);
        npairs = block_sum(npairs, red);          // also orders the matched[] writes
        // classification: weighted NLL, plain mean over B*Q (reference loss_detr.py:101-105)
        float label_sum = 0.f;
        for (int i = tid; i < B * Q; i += CRIT_THREADS) {
            int cls = matched[i] ? fg : bg;
            float l0 = lg[i * 2], l1 = lg[i * 2 + 1];
            float mx = fmaxf(l0, l1);
            float lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
            label_sum += empty_w[cls] * (lse - (cls == 0 ? l0 : l1));
        }
        label_sum = block_sum(label_sum, red);
        // contrastive align (reference loss_detr.py:112-128)
        float contr = 0.f;
        if (proj_q && vid_sum) {
            float csum = 0.f;
            for (int b = tid; b < B; b += CRIT_THREADS) {
                float mx = -INFINITY, pos = 0.f, npos = 0.f;
                for (int q = 0; q < Q; ++q) {
                    float z = lgt[b * Q + q];
                    mx = fmaxf(mx, z);
                    if (matched[b * Q + q]) { pos += z; npos += 1.f; }
                }
                float se = 0.f;
                for (int q = 0; q < Q; ++q) se += expf(lgt[b * Q + q] - mx);
                csum += -pos / npos + (mx + logf(se));
            }
            contr = block_sum(csum, red) / (float)B;
        }
        if (tid == 0) {
            float ls = span_sum / (2.f * npairs);
            float lgi = giou_sum / npairs;
            float ll = label_sum / (float)(B * Q);
            float ce = 100.f - correct * (100.f / npairs);
            float* o = losses + l * 5;
            o[0] = ls; o[1] = lgi; o[2] = ll; o[3] = ce; o[4] = contr;
        }
    }
}

// total = sum_l sum_k weights[k] * losses[l][k], fixed order (deterministic)
__global__ void criterion_total_kernel(const float* losses, const float* weights, int n_layers, float* total) {
    if (threadIdx.x != 0) return;
    float t = 0.f;
    for (int l = 0; l < n_layers; ++l) {
        const float* o = losses + l * 5;
        t += weights[0] * o[0] + weights[1] * o[1] + weights[2] * o[2] + weights[4] * o[4];
    }
    total[0] = t;
}

// ---- set criterion backward --------------------------------------------------------------------------
// Gradients of total = sum_l sum_k weights[k] * losses[l][k] w.r.t. logits, spans, proj_queries and vid_sum, scaled by
// upstream[0] (the gradient arriving at localization_loss).  One workgroup per decoder layer; vid_sum gradients of all
// layers are added atomically.  Ties of min/max inside the GIoU have measure zero and take the first operand's branch.
__global__ __launch_bounds__(CRIT_THREADS) void criterion_bwd_kernel(const float* logits, const float* spans, const float* targets,
                                                                     const int64_t* pred_idx, const int64_t* tgt_idx, const int32_t* count,
                                                                     const float* proj_q, const float* vid_sum, const float* empty_w,
                                                                     int B, int Q, int G, int Dc, int fg, float temperature,
                                                                     const float* weights, const float* upstream,
                                                                     float* dlogits, float* dspans, int ldo, int through_sigmoid,
                                                                     float* dproj_q, float* dvid_sum) {
    __shared__ unsigned char matched[CRIT_MAX_BQ];
    __shared__ float lgt[CRIT_MAX_BQ];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int width = Q < G ? Q : G;
    const int bg = 1 - fg;
    const int l = blockIdx.x;
    // grid.y slices the samples of the contrastive term (its rows cost a pass over Dc each); slice 0 also writes the span / label
    // gradients.  Every slice rebuilds the (cheap) matched flags of the whole layer.
    const int slice = blockIdx.y, nslice = gridDim.y;
    const float up = upstream ? upstream[0] : 1.f;
    const float* lg = logits + (int64_t)l * B * Q * 2;
    const float* sp = spans + (int64_t)l * B * Q * 2;
    float* dlg = dlogits + (int64_t)l * B * Q * ldo;
    float* dsp = dspans + (int64_t)l * B * Q * ldo;
    for (int i = tid; i < B * Q; i += CRIT_THREADS) {
        matched[i] = 0;
        if (slice == 0) { dsp[ldo * i] = 0.f; dsp[ldo * i + 1] = 0.f; }
    }
    __syncthreads();
    float npairs = 0.f;
    for (int i = tid; i < B * width; i += CRIT_THREADS) {
        int b = i / width, slot = i % width;
        if (slot < count[l * B + b]) npairs += 1.f;
    }
    npairs = block_sum(npairs, red);
    for (int i = tid; i < B * width; i += CRIT_THREADS) {
        int b = i / width, slot = i % width;
        int s = l * B + b;
        if (slot < count[s]) {
            int q = (int)pred_idx[(int64_t)s * width + slot];
            matched[b * Q + q] = 1;
            if (slice != 0) continue;
            int g = kth_kept(targets + (int64_t)b * G * 2, G, (int)tgt_idx[(int64_t)s * width + slot]);
            float pc = sp[(b * Q + q) * 2], pw = sp[(b * Q + q) * 2 + 1];
            float tc = targets[((int64_t)b * G + g) * 2], tw = targets[((int64_t)b * G + g) * 2 + 1];
            // L1: mean over 2 * npairs entries
            const float wl1 = up * weights[0] / (2.f * npairs);
            float dc = wl1 * (pc > tc ? 1.f : (pc < tc ? -1.f : 0.f));
            float dw = wl1 * (pw > tw ? 1.f : (pw < tw ? -1.f : 0.f));
            // GIoU: loss = mean(1 - giou)
            float ps, pe, ts, te;
            cw_to_se(pc, pw, ps, pe);
            cw_to_se(tc, tw, ts, te);
            const float mn_e = fminf(pe, te), mx_s = fmaxf(ps, ts);
            const float inter = fmaxf(mn_e - mx_s, 0.f);
            const float uni = (pe - ps) + (te - ts) - inter;
            const float enc = fmaxf(fmaxf(pe, te) - fminf(ps, ts), 0.f);
            const float pos_i = (mn_e - mx_s) > 0.f ? 1.f : 0.f;
            const float di_pe = pos_i * (pe <= te ? 1.f : 0.f), di_ps = -pos_i * (ps >= ts ? 1.f : 0.f);
            const float du_pe = 1.f - di_pe, du_ps = -1.f - di_ps;
            const float pos_e = enc > 0.f ? 1.f : 0.f;
            const float de_pe = pos_e * (pe >= te ? 1.f : 0.f), de_ps = -pos_e * (ps <= ts ? 1.f : 0.f);
            // giou = inter/uni - (enc - uni)/enc = inter/uni - 1 + uni/enc
            const float dg_pe = (di_pe * uni - inter * du_pe) / (uni * uni) + (du_pe * enc - uni * de_pe) / (enc * enc);
            const float dg_ps = (di_ps * uni - inter * du_ps) / (uni * uni) + (du_ps * enc - uni * de_ps) / (enc * enc);
            const float wg = -up * weights[1] / npairs;
            dc += wg * (dg_pe + dg_ps);
            dw += wg * 0.5f * (dg_pe - dg_ps);
            if (through_sigmoid) { dc *= pc * (1.f - pc); dw *= pw * (1.f - pw); }   // pred_spans = sigmoid(z): gradient w.r.t. z
            dsp[(b * Q + q) * ldo] = dc;
            dsp[(b * Q + q) * ldo + 1] = dw;
        }
    }
    __syncthreads();
    // weighted NLL, plain mean over B*Q
    for (int i = tid; i < B * Q && slice == 0; i += CRIT_THREADS) {
        int cls = matched[i] ? fg : bg;
        float l0 = lg[i * 2], l1 = lg[i * 2 + 1];
        float mx = fmaxf(l0, l1);
        float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        const float w = up * weights[2] * empty_w[cls] / (float)(B * Q);
        dlg[i * ldo] = w * (p0 - (cls == 0 ? 1.f : 0.f));
        dlg[i * ldo + 1] = w * (p1 - (cls == 1 ? 1.f : 0.f));
    }
    if (proj_q && vid_sum && dproj_q && dvid_sum) {
        const int nrow = ((B - slice + nslice - 1) / nslice) * Q;      // rows of the samples b = slice, slice + nslice, ...
        for (int j = wave; j < nrow; j += CRIT_THREADS / 64) {
            const int b = slice + (j / Q) * nslice, i = b * Q + j % Q;
            const float* pq = proj_q + ((int64_t)l * B * Q + i) * Dc;
            const float* vs = vid_sum + (int64_t)b * Dc;
            float d = 0.f;
            for (int k = lane; k < Dc; k += 64) d += pq[k] * vs[k];
            d = wave_sum(d);
            if (lane == 0) lgt[i] = d / temperature;
        }
        __syncthreads();
        const float wc = up * weights[4] / ((float)B * temperature);
        for (int j = wave; j < nrow; j += CRIT_THREADS / 64) {
            const int b = slice + (j / Q) * nslice, i = b * Q + j % Q;
            float mx = -INFINITY, npos = 0.f;
            for (int q = 0; q < Q; ++q) { mx = fmaxf(mx, lgt[b * Q + q]); npos += matched[b * Q + q] ? 1.f : 0.f; }
            float se = 0.f;
            for (int q = 0; q < Q; ++q) se += expf(lgt[b * Q + q] - mx);
            const float dl = wc * (expf(lgt[i] - mx) / se - (matched[i] ? 1.f / npos : 0.f));
            const float* pq = proj_q + ((int64_t)l * B * Q + i) * Dc;
            const float* vs = vid_sum + (int64_t)b * Dc;
            float* dpq = dproj_q + ((int64_t)l * B * Q + i) * Dc;
            for (int k = lane; k < Dc; k += 64) {
                dpq[k] = dl * vs[k];
                unsafeAtomicAdd(dvid_sum + (int64_t)b * Dc + k, dl * pq[k]);
            }
        }
    }
}

}  // namespace

extern "C" int made_hungarian_match(const float* pred_logits, const float* pred_spans, const float* targets,
                                    int64_t NS, int64_t B, int64_t Q, int64_t G, int32_t fg_label,
                                    float w_span, float w_giou, float w_class,
                                    float* cost_ws, int32_t cost_is_input, int64_t* out_pred_idx, int64_t* out_tgt_idx,
                                    int32_t* out_count, int32_t* status, void* stream) {
    MADE_REQUIRE(pred_logits && pred_spans && targets && cost_ws && out_pred_idx && out_tgt_idx && out_count && status,
                 "made_hungarian_match: null pointer");
    MADE_REQUIRE(NS >= 0 && B > 0 && NS % B == 0, "made_hungarian_match: NS=%lld must be a multiple of B=%lld", (long long)NS, (long long)B);
    MADE_REQUIRE(fg_label == 0 || fg_label == 1, "made_hungarian_match: fg_label must be 0 or 1");
    MADE_UNSUPPORTED(Q >= 1 && G >= 1 && Q <= MAXN && G <= MAXN, "made_hungarian_match: Q=%lld, G=%lld must be in [1,%d]",
                     (long long)Q, (long long)G, MAXN);
    if (NS == 0) return MADE_OK;
    hipLaunchKernelGGL(hungarian_kernel, dim3((unsigned)NS), dim3(64), 0, (hipStream_t)stream, pred_logits, pred_spans, targets,
                       (int)B, (int)Q, (int)G, (int)fg_label, w_span, w_giou, w_class, cost_ws, (int)cost_is_input, out_pred_idx, out_tgt_idx,
                       out_count, status);
    return made_check_launch("made_hungarian_match");
}

extern "C" int made_set_criterion(const float* pred_logits, const float* pred_spans, const float* targets,
                                  const int64_t* pred_idx, const int64_t* tgt_idx, const int32_t* count,
                                  const float* proj_queries, const float* vid_sum, const float* empty_weight,
                                  int64_t n_layers, int64_t B, int64_t Q, int64_t G, int64_t Dc, int32_t fg_label,
                                  float temperature, const float* weights, float* losses, float* total, void* stream) {
    MADE_REQUIRE(pred_logits && pred_spans && targets && pred_idx && tgt_idx && count && empty_weight && weights && losses,
                 "made_set_criterion: null pointer");
    MADE_REQUIRE(n_layers >= 1 && B >= 1 && Q >= 1 && G >= 1, "made_set_criterion: bad dims");
    MADE_UNSUPPORTED(n_layers <= 65535 && B * Q <= CRIT_MAX_BQ, "made_set_criterion: B*Q=%lld exceeds %d", (long long)(B * Q), CRIT_MAX_BQ);
    hipLaunchKernelGGL(criterion_kernel, dim3((unsigned)n_layers), dim3(CRIT_THREADS), 0, (hipStream_t)stream, pred_logits, pred_spans, targets,
                       pred_idx, tgt_idx, count, proj_queries, vid_sum, empty_weight, (int)n_layers, (int)B, (int)Q, (int)G,
                       (int)Dc, (int)fg_label, temperature, weights, losses, total);
    int rc = made_check_launch("made_set_criterion");
    if (rc != MADE_OK || total == nullptr) return rc;
    hipLaunchKernelGGL(criterion_total_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, losses, weights, (int)n_layers, total);
    return made_check_launch("made_set_criterion(total)");
}

extern "C" int made_set_criterion_bwd(const float* pred_logits, const float* pred_spans, const float* targets,
                                      const int64_t* pred_idx, const int64_t* tgt_idx, const int32_t* count,
                                      const float* proj_queries, const float* vid_sum, const float* empty_weight,
                                      int64_t n_layers, int64_t B, int64_t Q, int64_t G, int64_t Dc, int32_t fg_label,
                                      float temperature, const float* weights, const float* upstream,
                                      float* d_logits, float* d_spans, int64_t ld_out, int32_t through_sigmoid,
                                      float* d_proj_queries, float* d_vid_sum, void* stream) {
    MADE_REQUIRE(pred_logits && pred_spans && targets && pred_idx && tgt_idx && count && empty_weight && weights && d_logits && d_spans,
                 "made_set_criterion_bwd: null pointer");
    MADE_REQUIRE(n_layers >= 1 && B >= 1 && Q >= 1 && G >= 1, "made_set_criterion_bwd: bad dims");
    MADE_UNSUPPORTED(n_layers <= 65535 && B * Q <= CRIT_MAX_BQ, "made_set_criterion_bwd: B*Q=%lld exceeds %d", (long long)(B * Q), CRIT_MAX_BQ);
    const unsigned nslice = (proj_queries && vid_sum && d_proj_queries && d_vid_sum) ? (unsigned)(B < 16 ? B : 16) : 1u;
    hipLaunchKernelGGL(criterion_bwd_kernel, dim3((unsigned)n_layers, nslice), dim3(CRIT_THREADS), 0, (hipStream_t)stream, pred_logits, pred_spans,
                       targets, pred_idx, tgt_idx, count, proj_queries, vid_sum, empty_weight, (int)B, (int)Q, (int)G, (int)Dc,
                       (int)fg_label, temperature, weights, upstream, d_logits, d_spans, (int)(ld_out > 0 ? ld_out : 2), (int)through_sigmoid,
                       d_proj_queries, d_vid_sum);
    return made_check_launch("made_set_criterion_bwd");
}
