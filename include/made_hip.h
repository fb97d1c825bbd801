/*
 * made_hip.h -- C ABI of libmade_hip.so: the MI355X (gfx950) kernels behind the MaDe hot path.
 *
 * The reference (xxayt/MGSV) has NO native/FFI layer: every op on its hot path is a stock ATen
 * call made from Python (SURVEY.md section 2 "Native components: none").  The drop-in boundary is
 * therefore the Python class `Uni_model` (reference model/model_Uni.py:15,177) and this C ABI is
 * the contract the build defines underneath it ("C-ABI level").  Each entry
 * point cites the reference call sites whose arithmetic it replaces.
 *
 * Conventions
 *   - every argument is POD: device pointers, int64 sizes/strides (in ELEMENTS), dtype enums;
 *   - the caller allocates all outputs and workspaces; the library owns no memory;
 *   - every call enqueues on `stream` (a hipStream_t passed as void*) and returns without
 *     synchronising; return value 0 = OK, < 0 = error (text via made_last_error(), thread-local);
 *   - nothing throws across the boundary; calls are re-entrant on distinct streams;
 *   - masks are float32 with 1 = valid, 0 = padding, exactly as the reference's dataset emits them
 *     (reference dataloaders/dataloader_MGSV_EC_feature.py:61,67).
 */
#ifndef MADE_HIP_H
#define MADE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MADE_ABI_VERSION 8

enum MadeDtype { MADE_F32 = 0, MADE_BF16 = 1 };

enum MadeAct {
    MADE_ACT_NONE = 0,
    MADE_ACT_RELU = 1,       /* DETR FFN / MLP heads: reference music_detr/transformer.py:205,302,358 */
    MADE_ACT_GELU = 2,       /* erf GELU, temporal block FFN: reference model/model_Base.py:72 */
    MADE_ACT_QUICKGELU = 3,  /* x*sigmoid(1.702x): reference model/model_Base.py:17-20 */
    MADE_ACT_SIGMOID = 4     /* span head: reference model/model_Uni.py:135 */
};

/* act'(.) factors of the backward Linears (made_linear `gate`): out = (A W^T) * act'(G) * gate_scale */
enum MadeGate {
    MADE_GATE_NONE = 0,
    MADE_GATE_RELU_OUT = 1,      /* G = saved output of ReLU (after dropout): 1 where G != 0 */
    MADE_GATE_GELU_Z = 2,        /* G = saved pre-activation of the erf GELU */
    MADE_GATE_QUICKGELU_Z = 3,   /* G = saved pre-activation of x*sigmoid(1.702x) */
    MADE_GATE_SIGMOID_OUT = 4    /* G = saved sigmoid output: G (1 - G) */
};

/* Stateless dropout: keep(seed, site, idx) <=> (made_rng_mix(seed, site, idx) >> 8) >= floor(p * 2^24); kept values are
 * scaled by 1/(1-p).  Forward and backward kernels regenerate the same mask from (seed, site, logical element index), so
 * no mask is ever stored (mgsv_amd/dropout.py documents the index conventions and restates the mix in numpy).
 * Replaces torch's Philox draws at reference model/model_Base.py:69-75, modules/transformer.py:177,
 * music_detr/transformer.py:153-162,229-241 (value semantics identical, stream different by construction). */
typedef struct MadeDropout {
    uint64_t seed;
    uint32_t site;
    float    p;              /* 0 = no dropout */
    const uint64_t* seed_device;   /* non-NULL: the kernels read the seed from this device word instead of `seed`, so a captured
                                      hipGraph of the training step draws fresh masks on every replay (the host updates one word) */
} MadeDropout;

#if defined(__HIPCC__)
#define MADE_HOST_DEVICE __host__ __device__      /* the kernels call the same definition */
#else
#define MADE_HOST_DEVICE
#endif
MADE_HOST_DEVICE static inline uint32_t made_rng_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
/* keep threshold of probability p (a float32): floor(p * 2^24), evaluated exactly */
MADE_HOST_DEVICE static inline uint32_t made_drop_threshold(float p) { return (uint32_t)((double)p * 16777216.0); }
/* mix(seed, site, idx) = fmix32(lo32(idx) ^ key(seed, site, hi32(idx))): kernels hoist the key out of their inner loops */
MADE_HOST_DEVICE static inline uint32_t made_rng_key(uint64_t seed, uint32_t site, uint32_t idx_hi) {
    return made_rng_fmix32((uint32_t)seed ^ (site * 0x9E3779B9u) ^ ((uint32_t)(seed >> 32) * 0x85EBCA6Bu) ^ (idx_hi * 0xC2B2AE35u));
}
MADE_HOST_DEVICE static inline uint32_t made_rng_mix(uint64_t seed, uint32_t site, uint64_t idx) {
    return made_rng_fmix32((uint32_t)idx ^ made_rng_key(seed, site, (uint32_t)(idx >> 32)));
}
#if defined(__HIPCC__)
/* the seed a kernel uses: one scalar load per kernel when it lives in device memory */
__device__ static inline uint64_t made_drop_seed(const MadeDropout& d) { return d.seed_device ? *d.seed_device : d.seed; }
#endif


enum MadeStatus {
    MADE_OK = 0,
    MADE_ERR_INVALID_ARG = -1,
    MADE_ERR_UNSUPPORTED = -2,
    MADE_ERR_HIP = -3
};

int         made_abi_version(void);
/* f32 products of every kernel that multiplies f32 operands (MadeDtype MADE_F32 compute): mode 0 = exact (v_mfma_f32_32x32x2_f32, 157 TFLOP/s
 * peak), mode 1 = split-bf16: each operand as hi + lo bf16 halves and a . b ~ hi.hi + hi.lo + lo.hi on the bf16 matrix pipe (three products at
 * 16x the f32 rate; the operands' last 7 mantissa bits are dropped: relative 8e-6 per operand, f32 accumulation, storage and elementwise
 * steps unchanged).  Process-wide, read when a launch is issued; the reference's arithmetic is fp32 throughout (train-MaDe.py:237), its
 * tolerance <= 1e-4 on logits / spans -- the engine's "f32x3" mode is held to that gate by the same tests as "f32". */
int         made_set_f32_products(int mode);
int         made_get_f32_products(void);
const char* made_last_error(void);
/* fills name (<= name_len chars), compute-unit count and 1 if the device is gfx950 */
int         made_device_info(char* name, int name_len, int* cu_count, int* is_gfx950);

/* ------------------------------------------------------------------------------------------
 * made_linear: out = act(A' W^T + bias) (+ R), the fused Linear of the path.
 * Replaces: nn.Linear at reference model/model_Base.py:559,598 (vit_proj/ast_proj, with the
 * masked_fill of :556,:595 as `a_row_mask` and the PE add of :533 as `R` with r_row_mod=T),
 * :75-80,:81 (temporal FFN / final_linear, masked_fill :541 as `out_row_mask`), the packed in-proj
 * and out-proj of every nn.MultiheadAttention (reference music_detr/transformer.py:153,229,230;
 * model/model_Base.py:69) with the `with_pos_embed` adds of :193,:284,:293-294 as `A2`, FFN
 * linears :205,:302, heads at reference model/model_Uni.py:131-146 and the X-Pool projections
 * reference modules/transformer.py:96-103,122,176.
 *
 *   A' = (A + A2[row % a2_row_mod]) for column segments flagged use_a2 (or A2 itself when a2_replace is set),
 *   else A; rows of A whose a_row_mask is 0 are read as zero.  W is [N,K] row-major (nn.Linear layout), K contiguous.
 *   Compute type = w_dtype: MADE_BF16 -> v_mfma_f32_32x32x16_bf16, MADE_F32 -> v_mfma_f32_32x32x2_f32
 *   (exact f32 FMA chain); accumulation is always f32.
 *   The N columns are split into up to 4 segments, each with its own output tensor; a segment may
 *   be written transposed per batch (out[b][n][t], t = row % rows_per_batch) which is the layout
 *   made_attention wants for V.  `batch` > 1 runs independent problems (grid.z) with the given
 *   element strides (used for the per-track QK^T / PV products of the X-Pool block).
 *   split_k > 1 (skinny problems: the decoder's M = B*Q rows): grid.z splits K, every block writes its raw
 *   f32 partial tile to split_ws[split][M][N] and NO epilogue runs; made_splitk_finish sums the splits and
 *   applies bias / act / residual (and optionally LayerNorm), so a 64-row GEMM fills the chip.
 */
typedef struct MadeLinearSeg {
    int64_t col_begin;         /* first column of this segment (multiple of 128 unless nseg == 1) */
    void*   out;
    int32_t out_dtype;         /* MadeDtype */
    int32_t transposed;        /* 0: out[row*ldo + col]; 1: out[(row/rpb)*obs + col*ldo + row%rpb] */
    int64_t ldo;
    int64_t rows_per_batch;    /* rpb; 0 = plain rows */
    int64_t out_batch_stride;  /* obs (elements), used when rows_per_batch > 0 */
    int64_t out_z_stride;      /* per-problem stride when batch > 1 */
    int32_t use_a2;
    int32_t _pad;
} MadeLinearSeg;

typedef struct MadeLinearArgs {
    const void*  A;  int32_t a_dtype; int32_t w_dtype;
    int64_t      lda;
    const void*  A2; int64_t lda2; int64_t a2_row_mod;      /* A2 has a_dtype; may be NULL */
    int32_t      a2_replace;                                /* 1: flagged segments read A2 INSTEAD of A (e.g. src+pos) */
    int32_t      drop_col_div;                              /* > 1: dropout draws once per `drop_col_div` columns -- element index
                                                               row * drop_ld + col / drop_col_div (one attention-weight draw per head
                                                               on the value path of a one-query self-attention); 0 / 1: per element */
    const float* a_row_mask;                                /* [M] or NULL */
    const void*  W;  int64_t ldw;
    const float* bias;                                      /* [N] f32 or NULL */
    int64_t      M, N, K;
    int64_t      batch, a_z_stride, w_z_stride;             /* batch >= 1 */
    int32_t      act; int32_t r_dtype;
    const void*  R;  int64_t ldr; int64_t r_row_mod;        /* added after act; may be NULL */
    const float* out_row_mask;                              /* [M] or NULL: masked rows -> 0 */
    const float* tile_skip_mask;                            /* [M] or NULL: a tile whose rows are ALL 0 here is not computed
                                                               (its rows are left untouched, or zeroed when out_row_mask is
                                                               set): padded tokens cost nothing.  Rows are independent, so
                                                               valid rows are unaffected. */
    int32_t      nseg; int32_t split_k;                     /* split_k > 1: see below */
    float*       split_ws;                                  /* [split_k, M, N] f32 workspace or NULL */
    MadeLinearSeg seg[4];
    /* training-path epilogue (plain segments only; element order: z = A'W^T + bias [-> Zout], act, gate, dropout, +R, row mask) */
    const void*  G;  int32_t g_dtype; int32_t gate;         /* MadeGate: multiply by act'(G[row*ldg + col]) * gate_scale */
    int64_t      ldg; float gate_scale; int32_t z_dtype;
    void*        Zout; int64_t ldz;                         /* optional: the pre-activation z is also stored (saved for backward) */
    MadeDropout  drop; int64_t drop_ld;                     /* dropout after act/gate, element index row*drop_ld + col */
    /* row gather (padded tokens cost nothing): the kernel works on logical rows r < *n_rows (device scalar) that live at
       physical rows row_index[r] of A / A2 / R / G / out / masks (made_row_index of the token mask); every per-row quantity
       (residual, dropout index, rows_per_batch addressing) keeps using the PHYSICAL row, so valid rows are bit-identical to
       the ungathered call and padded rows are simply never read or written.  M stays the physical row count. */
    const int32_t* row_index; const int32_t* n_rows;
    /* per-(row, problem) scaled bias (tiny-M kernel only): the bias term is bias[z * bias_z_stride + col] * bias_row_scale[row * batch + z]
       for problem z of a batched call -- the value-projection bias of the memory-space cross-attention, whose dropped attention
       weights no longer sum to 1 (reference music_detr/transformer.py:293-296 under dropout; replaces a made_head_bias launch) */
    const float* bias_row_scale; int64_t bias_z_stride;
} MadeLinearArgs;

int made_linear(const MadeLinearArgs* args, void* stream);

/* Which kernel made_linear dispatches these arguments to (no launch; for profilers and bench.py's per-kernel roofline, so a
 * HIP-event average can be set beside rocprofv3's per-symbol average). */
enum MadeLinearVariant {
    MADE_LINEAR_GENERAL_F32 = 0,    /* linear_kernel<float, float>: exact-f32 MFMA, every option */
    MADE_LINEAR_GENERAL_F32IN = 1,  /* linear_kernel<float, bf16>: f32 activations converted in the prologue */
    MADE_LINEAR_GENERAL_BF16 = 2,   /* linear_kernel<bf16, bf16>: split-K, additive A2, transposed segments */
    MADE_LINEAR_TINY = 3,           /* linear_tiny_kernel: 64 x 32 tiles, fragments straight from global memory (K <= 1024) */
    MADE_LINEAR_SKINNY = 4,         /* linear_skinny_kernel: 64 x 64 tiles, 8-stage LDS-DMA ring */
    MADE_LINEAR_GLDS3 = 5,          /* linear_glds_kernel<3, ., 128>: at most one workgroup per CU, three LDS stages */
    MADE_LINEAR_GLDS64 = 6,         /* linear_glds_kernel<1, ., 64>: 64 x 128 tiles */
    MADE_LINEAR_GLDS128 = 7,        /* linear_glds_kernel<1, ., 128>: 128 x 128 tiles */
    MADE_LINEAR_BIG256 = 8,         /* linear_big_kernel<256>: persistent, 256 x 256 tiles, two LDS stages, register epilogue (round 4; the
                                       slot was round 2's ring kernel, removed) */
    MADE_LINEAR_TINY16 = 9,         /* linear_t16_kernel: at most 64 rows, 16 x 16 tiles (a third of the bytes per workgroup), 128 <= K <= 1024 */
    MADE_LINEAR_BIG128 = 10,        /* linear_big_kernel<128>: the same with 128 x 256 tiles (MADE_LINEAR_TILE=256; the slot was round 3's
                                       W-stationary kernel, removed) */
    MADE_LINEAR_GLDS64_F32 = 11,    /* linear_glds_kernel<1, ., 64, float>: the f32 parity mode's large Linears on the LDS-DMA loop (exact-f32 MFMA) */
    MADE_LINEAR_GLDS128_F32 = 12    /* linear_glds_kernel<1, ., 128, float> */
};
int made_linear_variant(const MadeLinearArgs* args);

/* One stage of the moment-DETR decoder's chain of B*Q-row Linears with the PREVIOUS stage's LayerNorm in its prologue
 * (reference music_detr/transformer.py:273-307, forward_post; :136 for the shared output norm):
 *     x   = LayerNorm(Zin; ln_g, ln_b)          Zin raw rows [M, K] (f32, or bf16 in the training chain); ln_g == NULL: x = Zin
 *     x2  = LayerNorm(x; ln2_g, ln2_b)          optional -> x2_out (bf16): the decoder output of the previous layer
 *     A   = bf16(x) (+ add[row % add_row_mod])  add: bf16 rows of K elements (query_pos) or NULL; x itself -> x_out (bf16) or NULL
 *     out = act(A W^T + bias) + R               R: bf16 [M, N] or NULL; res_from_x: + bf16(x) instead (needs N == K, no add)
 * bf16 MFMA, f32 accumulate; K (the LayerNorm width) is 256 or 512; out is f32 (raw rows for the next stage's norm) or bf16.
 * One launch of ceil(N / 32) x ceil(M / 64) workgroups, no split-K workspace, no finish launch. */
typedef struct MadeDecStageArgs {
    const void*  Zin; int64_t ldz;                 /* f32 rows (zin_dtype = MADE_F32, the eval chain) or bf16 rows (MADE_BF16, the training chain) */
    const float* ln_g; const float* ln_b;
    const float* ln2_g; const float* ln2_b; void* x2_out; int64_t ldx2;
    const void*  add; int64_t add_row_mod;
    void*        x_out; int64_t ldx;
    const void*  W; int64_t ldw; const float* bias;
    const void*  R; int64_t ldr;
    void*        out; int64_t ldo;
    int32_t      out_dtype; int32_t act; int32_t res_from_x; float eps;
    int64_t      M, N, K;
    /* training chain (bf16 Zin only): A = bf16(x) + add is also stored (a_out: the weight gradient's operand), and the stateless
       dropout of this header follows the activation -- element index row * drop_ld + col / max(drop_col_div, 1) */
    int32_t      zin_dtype; int32_t drop_col_div;
    void*        a_out; int64_t lda_out;
    MadeDropout  drop; int64_t drop_ld;
} MadeDecStageArgs;
int made_dec_stage(const MadeDecStageArgs* args, void* stream);

/* made_dec_stage_bwd: a stage of the decoder's BACKWARD chain with a LayerNorm backward in the prologue of the dX product that
 * consumes it (reference music_detr/transformer.py:273-307 read backwards; bf16 rows, K = the norm's width = 256 or 512):
 *     g   = dy (+ add),  or with a second norm stacked on the first (xb != NULL: norm 3 + the shared output norm, :306 and :136)
 *     g   = LN_b'(dy; xb, gamma_b) + add
 *     dx  = LN_a'(g; xa, gamma_a)       -> dx_out [M, K] (may be NULL);  dgamma / dbeta of the norm(s) accumulated (may be NULL)
 *     A   = dropout_a(dx)               -> a_out [M, K] (may be NULL): element index row * drop_a_ld + col
 *     out = dropout_o((A W^T) * [G != 0] * gate_scale) + R        W [N, K] bf16; G, R, out [M, N] bf16; drop_o index row * drop_o_ld + col / drop_o_col_div
 * One launch of ceil(N / 32) x ceil(M / 64) workgroups replaces made_layernorm_bwd (or made_layernorm_bwd2) + made_linear. */
typedef struct MadeDecStageBwdArgs {
    const void* xa; const float* gamma_a; int64_t ldxa;
    const void* xb; const float* gamma_b; int64_t ldxb;      /* second (outer) norm or NULL */
    const void* dy; int64_t lddy;
    const void* add; int64_t ldadd;                          /* may be NULL */
    void* dx_out; int64_t lddx;
    void* a_out; int64_t lda_out;
    float *dgamma_a, *dbeta_a, *dgamma_b, *dbeta_b;
    MadeDropout drop_a; int64_t drop_a_ld;
    const void* W; int64_t ldw;
    const void* G; int64_t ldg; float gate_scale; float eps;
    MadeDropout drop_o; int64_t drop_o_ld; int32_t drop_o_col_div; int32_t _pad;
    const void* R; int64_t ldr;
    void* out; int64_t ldo;
    int64_t M, N, K;
} MadeDecStageBwdArgs;
int made_dec_stage_bwd(const MadeDecStageBwdArgs* args, void* stream);

/* y = act(sum_s ws[s] + bias) + R[row % r_row_mod]  -> out (any dtype, may be NULL);
 * then optionally z1 = LayerNorm(y; ln1) -> ln1_out, and z2 = LayerNorm(z1; ln2) -> ln2_out (the decoder's
 * per-layer norm followed by the shared output norm, reference music_detr/transformer.py:306,136).
 * One wave per row; the LayerNorm variants need N <= 2048.  ws is [split_k, M, N] f32. */
typedef struct MadeFinishArgs {
    const float* ws; int64_t split_k, M, N;
    const float* bias; int32_t act; int32_t r_dtype;
    const void*  R; int64_t ldr; int64_t r_row_mod;
    void* out; int32_t out_dtype; int32_t _pad0; int64_t ldo;
    const float* ln1_g; const float* ln1_b; void* ln1_out; int32_t ln1_dtype; int32_t _pad1; int64_t ln1_ld;
    const float* ln2_g; const float* ln2_b; void* ln2_out; int32_t ln2_dtype; int32_t _pad2; int64_t ln2_ld;
    float eps; int32_t _pad3;
} MadeFinishArgs;

int made_splitk_finish(const MadeFinishArgs* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * made_attention: fused multi-head attention core, softmax(Q K^T * scale + key mask) V, flash
 * style (scores never touch HBM).  Replaces the inner part of nn.MultiheadAttention at reference
 * model/model_Base.py:87 and music_detr/transformer.py:199,287,293-296
 * and the CrossAttention core at reference model/model_Base.py:144-163.
 *   Q  [B, Lq, H*hd]  element (b,i,h,d) at Q + b*q_bs + i*ldq + h*hd + d
 *   K  [B, Lk, H*hd]  likewise with k_bs/ldk
 *   V  [B, Lk, H*hd]  likewise with v_bs/ldv (row-major; transposed on the fly by ds_read_b64_tr_b16)
 *   key_mask [B, Lk] f32, 0 = padded key (-inf before the softmax), may be NULL
 *   q_mask   [B, Lq] f32, 0 = output row forced to 0 AFTER the softmax (reference
 *            model/model_Base.py:163), may be NULL
 *   O  [B, Lq, H*hd] (dtype = compute dtype), element stride ldo / o_bs
 * hd in {32, 64, 128}.  A query whose keys are all masked yields NaN like the reference.
 */
typedef struct MadeAttnArgs {
    const void* Q; const void* K; const void* V; void* O;
    int32_t dtype; int32_t hd;
    int64_t B, H, Lq, Lk;
    int64_t q_bs, ldq, k_bs, ldk, v_bs, ldv, o_bs, ldo;
    const float* key_mask; const float* q_mask;
    float scale; int32_t _pad;
    const float* q_skip_mask;  /* [B, Lq] or NULL: queries that are 0 here are padding whose output nobody reads; 32-query
                                  groups (and whole workgroups) made only of them are skipped and their O rows left untouched */
    /* training path */
    float*      lse;           /* [B, H, Lq] f32 or NULL: log-sum-exp of the scaled, masked scores (+inf for a row with no valid
                                  key), saved for made_attention_bwd */
    MadeDropout drop;          /* dropout on the attention weights (after the softmax, as nn.MultiheadAttention does),
                                  element index ((b*H + h)*Lq + i)*Lk + j */
    const int32_t* batch_order; /* [B] or NULL: a permutation of the batch (made_batch_order: longest sequence first).  Only the
                                  ORDER in which workgroups are issued changes -- a padded batch then finishes with its short
                                  samples instead of waiting on a long one that started last; results are unchanged */
    uint32_t* keep_bits;       /* NULL, or (with drop.p > 0) [B*H*ceil(Lk/32), ld_bits] words: the dropout decisions of this call, for
                                  made_attention_bwd (which then tests a bit per score instead of re-drawing it: the draw is ~11 VALU
                                  instructions).  Row (b*H + h)*ceil(Lk/32) + kt holds, for key tile kt (keys 32 kt .. 32 kt + 31), one word
                                  per query: bit j of the word at slot 32 (i / 32) + s(i % 32) = element (i, 32 kt + j) is KEPT, with
                                  s(q) = 2 ((q & 3) + 4 (q >> 3)) + ((q >> 2) & 1) -- the 32 words of a (key tile, query group) pair are
                                  then the sixteen 64-lane masks of a 32 x 32 MFMA accumulator tile (register e: rows (e & 3) + 8 (e >> 2)
                                  and + 4), which the single-pass backward loads with scalar loads.  ld_bits = a multiple of 32 >= Lq;
                                  words of keys behind the sample's last valid key are not written (and not read) */
    int64_t ld_bits;
} MadeAttnArgs;

int made_attention(const MadeAttnArgs* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * made_attention_wide: single-head attention with head dimension = model width D (256 or 512),
 * keys and values ROW-major, flash style.  O[b, nq, :] = softmax_t(<Q[b,nq], K[b,t]+Kadd[b,t]>*scale
 * + key mask) . V[b, t, :].
 * Replaces (a) the X-Pool attention core, reference modules/transformer.py:110-119 (queries = all
 * videos, shared by every track: q_bs = 0), and (b) the DETR decoder cross-attention of reference
 * music_detr/transformer.py:293-296 evaluated in memory space: with W_k moved onto the query and
 * W_v onto the pooled rows, keys = memory + pos (Kadd), values = memory, and the H*Q rows
 * q'_{h,q} = W_k,h^T q_h are the "queries" -- this removes the [B*L, D] x [D, 2*dec*D] projection of
 * the memory (25 % of the forward flops) for the small Q the model uses.
 *   query index nq = i1*NQ2 + i2 addresses Q at b*q_bs + i1*q_s1 + i2*q_s2 and O at
 *   b*o_bs + i1*o_s1 + i2*o_s2; K/Kadd/V rows at b*{k,kadd,v}_bs + t*ld{k,kadd,v}.
 *   Kadd may be NULL; V may alias K (then loaded once).  key_mask [B, L] f32 or NULL.
 *   dtype = compute dtype of Q/K/Kadd/V; o_dtype = dtype of O.  All-masked rows give NaN.
 */
typedef struct MadeWideAttnArgs {
    const void* Q; const void* K; const void* Kadd; const void* V; void* O;
    const float* key_mask;
    int32_t dtype; int32_t o_dtype;
    int64_t B, NQ1, NQ2, L, D;
    int64_t q_bs, q_s1, q_s2, k_bs, ldk, kadd_bs, ldkadd, v_bs, ldv, o_bs, o_s1, o_s2;
    float scale; int32_t _pad;
    int64_t n_split;           /* > 1: keys are split over grid.z (few queries, long memory: fills the chip) and */
    float*  part_o;            /*      the slices are merged by a second launch; [B, n_split, NQ, D] f32 */
    float*  part_ml;           /*      [B, n_split, NQ, 4] f32 (running max, sum, sum of the dropped weights, unused) */
    /* training path */
    MadeDropout drop;          /* dropout on the attention weights, element index ((b*NQ1 + i1)*NQ2 + i2)*L + key */
    float*  sum_out;           /* [B, NQ1*NQ2] f32 or NULL: sum of the dropped weights of each row (1 without dropout) */
    float*  lse_out;           /* [B, NQ1*NQ2] f32 or NULL: log-sum-exp of the scaled scores of each row (saved for made_attention_wide_bwd) */
} MadeWideAttnArgs;

int made_attention_wide(const MadeWideAttnArgs* args, void* stream);

/* made_attention_wide_bwd: backward of made_attention_wide for the decoder's memory-space cross-attention (few query rows per sample,
 * bf16), one launch per decoder layer (reference music_detr/transformer.py:293-296 under model.train(); replaces a batched Linear,
 * made_softmax_bwd, a batched made_gemm_tn and made_head_bias_bwd on the backward's critical path):
 *   P_j = exp(scale <Q_q, K_j> - lse_q) on valid keys, Pd = dropout(P) (the forward's mask: index (b*NQ + q)*L + j),
 *   dPd_j = <dO_q, V_j> + extra_q,  dS_j = scale P_j (dropout'(dPd_j) - delta_q),  delta_q = <dO_q, O_q> + extra_q ssum_q,
 *   dQ_q = sum_j dS_j K_j.   Pd and dS ([B, NQ, ld_p] bf16, zero on masked keys and on the pad columns L..ld_p-1) are written for
 *   the memory-gradient products dV = Pd^T dO, dK = dS^T Q that follow as one batched made_gemm_tn over all layers.
 * extra_q = gradient of the sum of row q's dropped weights (the value bias enters the forward as ssum_q b_v): `extra` [B, NQ] f32, or,
 * when `extra` is NULL and `dattc` is given, reduced here as <dattc[b, q*hd .. q*hd+hd), vbias[q*hd ..)> (one head per query row).
 * O = the forward's output rows, lse / ssum = its lse_out / sum_out.  Keys may be split over workgroups (n_split, part_dq
 * [B, n_split, NQ, D] f32); the slices' dQ are summed in slice order by a small second launch.  NQ <= 8, D in {256, 512}. */
typedef struct MadeWideAttnBwdArgs {
    const void *Q, *dO, *O, *K, *V;
    const float* key_mask;                 /* [B, L] or NULL */
    const float *lse, *ssum;               /* [B, NQ]; ssum may be NULL (= 1) */
    const float* extra;                    /* [B, NQ] or NULL */
    const void* dattc; int64_t ld_dattc; const float* vbias; int64_t hd;
    void *Pd, *dS; int64_t p_bs, ld_p;
    void* dQ; int64_t dq_bs, ld_dq;
    int64_t B, NQ, L, D;
    int64_t q_bs, ld_q, do_bs, ld_do, o_bs, ld_o, k_bs, ldk, v_bs, ldv;
    float scale; int32_t _pad;
    int64_t n_split; float* part_dq;
    MadeDropout drop;
} MadeWideAttnBwdArgs;
int made_attention_wide_bwd(const MadeWideAttnBwdArgs* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Launch tape: record the library's launches of one training (or eval) step once, replay them from one C loop.
 * Between made_tape_begin() and made_tape_end() every kernel launch of the calling thread is executed as usual AND appended to the
 * tape (function, grid, block, LDS bytes, stream, argument bytes); made_stream_wait / made_memset_async / made_copy_async are the
 * stream-to-stream dependency, fill and device-to-device copy that are recorded the same way (use them instead of the framework's
 * own calls inside a recorded region).  made_tape_replay re-issues the sequence to the same streams: everything that changes
 * between replays must therefore live in device memory at fixed addresses (batch buffers, MadeDropout.seed_device,
 * MadeAdamDeviceState).  Replaces what the reference gets from the framework's eager dispatch (train-MaDe.py:337-381: ~5 300 ATen
 * dispatches per step); a replayed launch costs the host the HIP runtime's 2-3 us instead of ~10 us of interpreter + ctypes or
 * ~9 us of hipGraphLaunch's node walk, and the two streams keep their overlap. */
int made_tape_begin(void);
int made_tape_end(uint64_t* handle);
int made_tape_replay(uint64_t handle);
int made_tape_free(uint64_t handle);
/* re-orders the ISSUE order of a finished tape so that all its streams are fed at the same time (per-stream order, event order and
 * hence every result unchanged): `main_weight` operations of the busiest stream for one of each other stream, round-robin */
int made_tape_interleave(uint64_t handle, int32_t main_weight);
int made_tape_count(uint64_t handle, int64_t* kernels, int64_t* waits, int64_t* others);
int made_tape_replay_range(uint64_t handle, int64_t first, int64_t count);   /* operations [first, first + count) only, in the tape's issue order:
                                                                               one phase of a step on its own (measurements) */
int made_tape_op(uint64_t handle, int64_t index, int32_t* kind, uint64_t* function, uint64_t* stream, uint32_t* grid3);
                                                              /* what operation `index` is: kind 0 = kernel launch (function = its host-side
                                                                 address, grid3 = workgroups per axis), other kinds: events, fills, copies */
int made_stream_wait(void* src_stream, void* dst_stream);            /* dst waits for all work queued on src so far */
/* recording only, executes nothing: a HOST callback at this point of the issue order; a replay calls fn(user) there (non-zero return:
 * the replay stops with an error).  For work the library does not launch itself but that must sit between the step's launches: the
 * data-parallel gradient all-reduces (RCCL calls of the framework, reference train-MaDe.py:238-241,371).  made_tape_interleave moves
 * nothing across a callback. */
typedef int (*MadeTapeCallback)(void* user);
int made_tape_callback(MadeTapeCallback fn, void* user);
int made_tape_event(int32_t op_kind, int32_t slot, void* stream);    /* recording only, executes nothing: 0 = "record event `slot` on
                                                                         stream", 1 = "stream waits for event `slot`" -- mirrors the
                                                                         framework's own Event.record / Stream.wait_event calls */
int made_memset_async(void* dst, int32_t value, int64_t nbytes, void* stream);
int made_copy_async(void* dst, const void* src, int64_t nbytes, void* stream);
/* dst[0 .. n_words) = words[0 .. n_words) (n_words <= 4, 32-bit words, dst 4-byte aligned): the words travel as kernel arguments of a
 * one-wave launch, so the call is stream-ordered and the host buffer may be reused at once -- the per-step scalars of a replayed
 * training step (dropout seed, learning rates, step count; replaces the framework's fill kernels in TrainStepGraph.step). */
int made_store_words(void* dst, const uint32_t* words, int32_t n_words, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row kernels (HBM-bound).                                                                   */

/* y = LayerNorm(x) * gamma + beta, eps inside the sqrt; one wave per row, D <= 2048, D % 4 == 0.
 * Input row r sits at x + (r / rpb) * x_batch_stride + (r % rpb) * ldx when rpb = x_rows_per_batch > 0
 * (a [B,T,D] view of a larger buffer), else at x + r * ldx; output rows are y + r * ldy.
 * (rows are independent, so skipping padded rows leaves every valid row bit-identical.)
 * Replaces nn.LayerNorm at reference model/model_Base.py:83,85; music_detr/transformer.py:202,
 * 209,290,300,306,136; modules/transformer.py:164-165,174,178. */
int made_layernorm(const void* x, int32_t x_dtype, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride,
                   const float* gamma, const float* beta,
                   void* y, int32_t y_dtype, int64_t ldy, int64_t rows, int64_t D, float eps,
                   const float* row_skip /* [rows] f32 or NULL: rows that are 0 here (padding) are not computed */, void* stream);

/* Same, plus a second output y2 = y + add (row r of `add` at add + r*ld_add; y2 has y's dtype and stride ldy2):
 * the DETR layers need both src and src + pos (reference music_detr/transformer.py:193), so the norm that produces
 * src also emits src + pos and the next projection reads it directly. gamma == NULL skips the normalisation
 * (y = x), which turns the kernel into a fused copy/add for the first encoder layer. */
int made_layernorm_add(const void* x, int32_t x_dtype, int64_t ldx, const float* gamma, const float* beta,
                       void* y, int32_t y_dtype, int64_t ldy, const void* add, int32_t add_dtype, int64_t ld_add,
                       void* y2, int64_t ldy2, int64_t rows, int64_t D, float eps, const float* row_skip, void* stream);

/* The music side of the sharded retrieval as ONE buffer for ONE all-gather (reference test-MaDe.py:386-403 over the 8 GPUs of a node): a
 * record per track, [S * D segment embeddings in pack_dtype | S mask floats | D floats of the pooled music vector | zero pad to rec_bytes
 * (a multiple of 16)].  seg [n, S, D] (f32 or bf16, track stride seg_bs elements, S * D contiguous), mask [n, S] f32, music [n, D] f32;
 * records n .. n_pad - 1 (the shards are padded to the largest) are zero-filled. */
int made_pack_music_records(const void* seg, int32_t seg_dtype, int64_t seg_bs, const float* mask, int64_t ld_mask, const float* music, int64_t ld_music,
                            void* out, int32_t pack_dtype, int64_t rec_bytes, int64_t n, int64_t n_pad, int64_t S, int64_t D, void* stream);

/* y[r, :] = x[r, :] * (mask[r] != 0) converted to y_dtype: the masked_fill of reference model/model_Base.py:556,595
 * fused with the f32 -> bf16 conversion of the pre-extracted features, so the input projection can use the
 * direct-to-LDS GEMM path.  mask may be NULL. */
int made_cast_mask_rows(const float* x, int64_t ldx, const float* mask, void* y, int32_t y_dtype, int64_t ldy,
                        int64_t rows, int64_t D, void* stream);

/* out[b, :] = sum_t x[b,t,:] * (mask[b,t] != 0) / sum_t mask[b,t]   (mask NULL: plain column sum,
 * no division).  Replaces reference model/model_Base.py:579,615 and the sum over frames inside
 * reference music_detr/loss_detr.py:116-117. */
int made_masked_mean(const void* x, int32_t x_dtype, int64_t x_bs, int64_t ldx, const float* mask,
                     float* out, int64_t B, int64_t T, int64_t D, void* stream);

/* y = x / max(||x||_2, eps) per row (F.normalize): reference model/model_Base.py:580,616,
 * model/model_Uni.py:142,146.  y_f32 and/or y_alt (dtype y_alt_dtype) may be NULL. */
int made_l2norm_rows(const void* x, int32_t x_dtype, int64_t ldx, float* y_f32, void* y_alt, int32_t y_alt_dtype,
                     int64_t ldy, int64_t rows, int64_t D, float eps, void* stream);

/* Mask-aware normalised sine position embedding: reference music_detr/position_encoding.py:51-71.
 * mask [B,L] f32; dim_t [D] f32 (= 10000^(2*floor(i/2)/D), computed once by the host exactly as the
 * reference does); out [B,L,D] in out_dtype. */
int made_sine_pe(const float* mask, const float* dim_t, void* out, int32_t out_dtype,
                 int64_t B, int64_t L, int64_t D, void* stream);

/* In-place-capable masked softmax over the last axis of logits [M_outer, R, S] (f32, row stride
 * lds_): p = softmax(logits*scale + (mask[m, s]==0 ? -inf : 0)); columns in [S, S_pad) are written
 * as 0.  Output dtype out_dtype, row stride ldp.  The softmax over segments ("clips") of the
 * X-Pool block: reference modules/transformer.py:110-117. */
int made_masked_softmax(const float* logits, int64_t ld_logits, const float* mask, int64_t ld_mask,
                        void* probs, int32_t out_dtype, int64_t ldp,
                        int64_t M_outer, int64_t R, int64_t S, int64_t S_pad, float scale, void* stream);

/* made_xpool_fused: all-pairs X-Pool scoring in one kernel (bf16, D = 256): for every (video n, track m)
 *   softmax_s(<Q[n], K[m,s]> * scale + mask) . U[m,s,:] -> LayerNorm2 -> x + (Wl x + bl) -> LayerNorm3 -> cosine with vn[n]
 *   -> sims[n*ld_sims + m].
 * Replaces, for retrieval (reference test-MaDe.py:392-403), the chain modules/transformer.py:110-123 (attention; out_proj is
 * hoisted onto the values by the caller: U = out_proj(v_proj(LN1(seg))), valid because the softmax rows sum to 1),
 * :172-178 (LayerNorm2, linear_proj with residual, LayerNorm3) and modules/metrics.py:19-24 (cosine) -- no [Nm*Nv, D]
 * tensor is ever written.  Q = q_proj(LN1(video)) [Nv, D]; K = k_proj(LN1(seg)), U [Nm, S, D] (rows at m*{k,u}_bs + s*ld);
 * key_mask [Nm, S] f32 or NULL; vn = video / |video| [Nv, D] f32.  A track with no valid segment gives NaN, like the
 * reference's softmax over -inf.  LayerNorm variances are taken in one pass (E[x^2] - E[x]^2, f32): a bf16-path kernel. */
typedef struct MadeXpoolFusedArgs {
    const void* Q; int64_t ldq;
    const void* K; const void* U; int64_t k_bs, ldk, u_bs, ldu;
    const float* key_mask;
    const float* ln2_g; const float* ln2_b;
    const void* Wl; int64_t ldw; const float* bl;
    const float* ln3_g; const float* ln3_b;
    const float* vn; int64_t ldvn;
    float* sims; int64_t ld_sims;
    int64_t Nv, Nm, S, D;
    float scale, eps;
    float* ws;                 /* workspace, Nv*(D+2) + 4 + 2*D + D*D/2 + 4*Nm floats, 16-byte aligned: per-video terms of LayerNorm3 +
                                  cosine that do not depend on the track, the Linear folded with LayerNorm2 (bf16 weight, two
                                  vectors), four ints per track (valid range of its segments) */
    int32_t prepare_ws; int32_t _pad;   /* 1: fill the per-video and per-model parts of ws first (small launches); 0: they are still
                                           valid from a previous call with the same vn, ln2, Wl, bl and ln3 (the caller loops over
                                           chunks of tracks); the per-track part is rebuilt by every call */
} MadeXpoolFusedArgs;

int made_xpool_fused(const MadeXpoolFusedArgs* args, void* stream);

/* made_xpool_attention: the attention of the X-Pool block at retrieval scale (every video against the segments of every track, ONE head
 * of width D = 256 or 512; reference modules/transformer.py:87-123 as called by Transformer_XA.forward :156-180 and test-MaDe.py:392-395)
 * plus the normalisation of LayerNorm2 (:172) without its affine part:
 *   o[m, n, :]   = softmax_s(<Q[n], K[m, s]> * scale + (key_mask[m, s] == 0 ? -inf : 0)) . U[m, s, :]
 *   out[m, n, :] = normalize ? (o - mean(o)) * rsqrt(var(o) + eps) : o                       (bf16, row (m * Nv + n) * ldo)
 * The caller folds LayerNorm2's gamma / beta into the Linear that follows (W'' = (W + I) diag(gamma), b'' = (W + I) beta + b), so the
 * residual of modules/transformer.py:176 costs nothing.  Two passes per track with the probabilities parked in LDS (bf16), K / U rows
 * through LDS-DMA; S <= 512 segments.  Q [Nv, D], K / U [Nm, S, D] (rows at m * {k,u}_bs + s * ld{k,u}), all bf16; key_mask [Nm, S] f32
 * or NULL; ws: Nm * 32 ints (valid range and one bit per segment of every track), 16-byte aligned.  A track without a valid segment
 * gives NaN rows, like the reference's softmax over -inf. */
typedef struct MadeXpoolAttnArgs {
    const void* Q; int64_t ldq;
    const void* K; const void* U; int64_t k_bs, ldk, u_bs, ldu;
    const float* key_mask;
    void* out; int64_t ldo;
    int64_t Nv, Nm, S, D;
    float scale, eps;
    int32_t normalize; int32_t _pad;
    void* ws;
} MadeXpoolAttnArgs;

int made_xpool_attention(const MadeXpoolAttnArgs* args, void* stream);

/* made_xpool_sims: all-pairs X-Pool scoring with the per-pair Linear moved onto the values (round 4; D = 256, tracks of at most 96 segments).  The chain of
 * reference modules/transformer.py:156-180 + modules/metrics.py:10-24 after the attention is
 *   y = W'' xhat + b'',  xhat = (o - mean(o)) rstd(o),  W'' = (W + I) diag(g2), b'' = (W + I) b2 + b          (LayerNorm2 + Linear + residual)
 * and o = sum_s p_s u_s is a convex combination of the track's value rows, so W'' o = sum_s p_s (W'' u_s): with u''_s = W'' u_s made ONCE per
 * segment by a GEMM over the tracks (the caller: columns [D, 2D) of the value rows), the per-pair 2 D^2 flops of the Linear become a second
 * P.V product of 2 S D flops, and   y = k1 (P.U'') + k2 Bv + Av,   k1 = rstd(o), k2 = -mean(o) rstd(o), Av = b'', Bv = W'' 1.
 * LayerNorm3 and the cosine with the video are the six sums of made_xpool_fused; nothing per pair is written but sims[n * ld_sims + m].
 * Q [Nv, D] bf16; K [Nm, S, D] bf16 (rows at m * k_bs + s * ldk); UU [Nm, S, 2 D] bf16 (rows at m * u_bs + s * ldu: u_s | u''_s);
 * key_mask [Nm, S] f32 or NULL; av, bv, ln3_g, ln3_b [D] f32; vn = video / |video| [Nv, D] f32; ws: made_xpool_sims_ws_bytes(Nv, Nm, D)
 * bytes, 16-byte aligned (per-video terms of LayerNorm3 + cosine -- since round 5 with a second, bf16 copy of g3 * vn in the order the 64-video kernel's lanes
 * read it; Nv * ldq * 2 and Nv * (1.5 D + 4) * 4 must stay below 4 GB -- filled when prepare_ws != 0; 32 ints per track, rebuilt by every call).
 * Two kernels serve the call: 32 videos per workgroup (the default) and 64 (MADE_XPOOL_SIMS_PQ=64: the same sums in the same order, results equal to 1e-7).
 * A track without a valid segment gives NaN, like the reference's softmax over -inf. */
typedef struct MadeXpoolSimsArgs {
    const void* Q; int64_t ldq;
    const void* K; const void* UU; int64_t k_bs, ldk, u_bs, ldu;
    const float* key_mask;
    const float* av; const float* bv;
    const float* ln3_g; const float* ln3_b;
    const float* vn; int64_t ldvn;
    float* sims; int64_t ld_sims;
    int64_t Nv, Nm, S, D;
    float scale, eps;
    void* ws;
    int32_t prepare_ws; int32_t _pad;
} MadeXpoolSimsArgs;

int     made_xpool_sims(const MadeXpoolSimsArgs* args, void* stream);
int64_t made_xpool_sims_ws_bytes(int64_t Nv, int64_t Nm, int64_t D);

/* made_xpool_inbatch: the in-batch X-Pool contraction for a batch of at most 64 videos -- out[m, n, :] = softmax_s(scale q_n . k_{m,s} + mask) U_m
 * (reference modules/transformer.py:110-119, out projection hoisted onto the values; the north-star contraction).  Two launches of one
 * workgroup per CU -- scores per (track, 128 segments) with the tile-local softmax pieces, then P.V per (track, 128 value columns) -- with only
 * the bf16 probabilities between them; replaces made_attention_wide's key split + merge on this shape.  Q [Nv <= 64, D], K / U [Nm, S <= 512, D]
 * (rows at m * {k,u}_bs + s * ld{k,u}), bf16, D = 256 / 512; key_mask [Nm, S] f32 or NULL; out[m * o_bs + n * ldo + d] bf16 or f32;
 * ws: made_xpool_inbatch_ws_bytes(Nm, S) bytes, 16-byte aligned (a larger workspace may be reused for a smaller call).  A track without a valid
 * segment gives NaN rows (the reference's softmax).  (Round 5's opt-in one-launch form -- both roles in one workgroup, the probabilities handed
 * over inside the launch -- was 4 % faster alone and is gone: docs/EXPERIMENTS.md.) */
typedef struct MadeXpoolInbatchArgs {
    const void* Q; int64_t ldq;
    const void* K; const void* U; int64_t k_bs, ldk, u_bs, ldu;
    const float* key_mask;
    void* out; int32_t out_dtype; int32_t _pad; int64_t o_bs, ldo;
    int64_t Nv, Nm, S, D;
    float scale; int32_t _pad2;
    void* ws;
} MadeXpoolInbatchArgs;

int     made_xpool_inbatch(const MadeXpoolInbatchArgs* args, void* stream);
int64_t made_xpool_inbatch_ws_bytes(int64_t Nm, int64_t S);

/* made_row_affine: out[r, :] = act(x[r, :] * scale[r % period] + shift[r % period]), act = none | ReLU.  Eval-mode
 * BatchNorm1d of the EmbeddingNet aggregator (reference model/model_Base.py:224-229): its input is [B, T, F], so the
 * "channels" are the T token positions and the running statistics reduce to one (scale, shift) per position. */
int made_row_affine(const void* x, int32_t x_dtype, int64_t ldx, const float* scale, const float* shift, int64_t period,
                    int32_t act, void* out, int32_t out_dtype, int64_t ldo, int64_t rows, int64_t cols, void* stream);

/* X-Pool tail: y [Nm*Nv, D] (the pre-LayerNorm3 sum, reference modules/transformer.py:177) ->
 * LayerNorm3 -> (optional) pooled[m,n,:] -> cosine with video n -> sims[n*ld_sims + m].
 * Fuses reference modules/transformer.py:178 with modules/metrics.py:19-24 so the pooled tensor
 * need not be materialised.  video [Nv,D] f32. */
int made_xpool_tail(const void* y, int32_t y_dtype, int64_t ldy, const float* gamma, const float* beta,
                    const float* video, int64_t ld_video, float* pooled_out /* [Nm*Nv,D] or NULL */,
                    float* sims, int64_t ld_sims, int64_t Nm, int64_t Nv, int64_t D, float eps, void* stream);

/* Symmetric cross-entropy of reference modules/loss.py:5-24 (== InfoNCELoss with audio_id=None,
 * :116-122): loss_out[0] (+)= weight * 0.5*(CE_rows + CE_cols) of sims*exp(*logit_scale).
 * accumulate != 0 adds to loss_out[0] instead of overwriting.  n <= 4096. */
/* row_exclude [n, n] f32 or NULL (made_clip_loss and made_clip_loss_bwd): entries equal to 1 are left out of the row-direction
 * (video -> music) softmax -- the negatives that share the row's own music track when the drivers run with
 * --ignore_same_music 0 (reference modules/loss.py:90-114); the column direction always uses every entry. */
int made_clip_loss(const float* sims, int64_t ld, int64_t n, const float* logit_scale, float weight,
                   int32_t accumulate, float* loss_out, const float* row_exclude, void* stream);

/* ------------------------------------------------------------------------------------------
 * Hungarian matcher and set criterion (reference music_detr/matcher.py:36-92, loss_detr.py).   */

/* Cost + assignment for NS = n_layers*B independent samples (sample s uses targets[s % B]).
 *   pred_logits [NS,Q,2] f32, pred_spans [NS,Q,2] f32 (centre,width), targets [B,G,2] f32;
 *   targets with width == 0 are dropped (reference matcher.py:59-61) and tgt indices refer to the
 *   kept targets in order.  cost = w_span*L1 + w_giou*(-GIoU) + w_class*(-softmax(logits)[fg]) in
 *   f32, evaluated left to right without FMA contraction (reference matcher.py:88); the assignment
 *   is SciPy's rectangular LSAP on the f64-promoted block (reference matcher.py:91), same tie-break.
 *   cost_ws [NS,Q,G] f32 workspace (holds the per-sample cost blocks on return).  With
 *   cost_is_input != 0 the cost step is skipped and cost_ws is taken as given (block of sample s =
 *   cost_ws[s, :Q, :kept targets of s]); used to check the assignment bit-for-bit on identical costs:
 *   the class term goes through exp(), whose last bit differs between libms, and a 1-ulp change can
 *   flip an assignment whose two best solutions tie to ~1e-7 (both are then optimal).
 *   out_pred_idx/out_tgt_idx [NS, min(Q,G)] int64, rows ascending in pred index, unused = -1;
 *   out_count [NS] int32; status [1] int32: 0, or 1 if any cost was NaN/-inf or infeasible
 *   (SciPy raises ValueError there).  Q, G <= 64. */
int made_hungarian_match(const float* pred_logits, const float* pred_spans, const float* targets,
                         int64_t NS, int64_t B, int64_t Q, int64_t G, int32_t fg_label,
                         float w_span, float w_giou, float w_class,
                         float* cost_ws, int32_t cost_is_input, int64_t* out_pred_idx, int64_t* out_tgt_idx,
                         int32_t* out_count, int32_t* status, void* stream);

/* Set criterion for `n_layers` decoder layers at once (reference music_detr/loss_detr.py:74-169):
 *   losses [n_layers, 5] f32 = {loss_span, loss_giou, loss_label, class_error, loss_contrastive_align}
 *   per layer, and *total = sum_l sum_k weights[k]*losses[l,k] (class_error has weight 0).
 *   proj_queries [n_layers,B,Q,Dc] f32 and vid_sum [B,Dc] f32 (= sum over frames of proj_vid_mem)
 *   may be NULL (no contrastive term).  empty_weight [2] f32. */
int made_set_criterion(const float* pred_logits, const float* pred_spans, const float* targets,
                       const int64_t* pred_idx, const int64_t* tgt_idx, const int32_t* count,
                       const float* proj_queries, const float* vid_sum, const float* empty_weight,
                       int64_t n_layers, int64_t B, int64_t Q, int64_t G, int64_t Dc, int32_t fg_label,
                       float temperature, const float* weights /* [5] */,
                       float* losses, float* total, void* stream);


/* ==========================================================================================
 * Training path (backward kernels).  The reference trains through torch autograd
 * (reference train-MaDe.py:337-381: forward, loss.backward(), three clip_grad_norm_, Adam); the
 * entries below are the hand-written counterparts of the autograd nodes on the hot path.
 * ========================================================================================== */

/* made_attention_bwd: gradients of made_attention (flash style: the probabilities are recomputed from Q, K and the saved
 * log-sum-exp; nothing of size Lq x Lk touches HBM).  With Pd = dropout(P):
 *   delta_i = dO_i . O_i      dV = Pd^T dO      dP = dropout'(dO V^T)      dS = P * (dP - delta)
 *   dQ = scale * dS K         dK = scale * dS^T Q
 * Replaces autograd through nn.MultiheadAttention (reference model/model_Base.py:87, music_detr/transformer.py:199,287,
 * 293-296).  All tensors use the addressing of made_attention (element (b,i,h,d) at base + b*bs + i*ld + h*hd + d); dQ, dK,
 * dV may be column blocks of one [rows, 3*H*hd] buffer.  `delta` is a [B,H,Lq] f32 workspace the call fills (bf16: by the dQ kernel, for the
 * dK / dV kernel behind it; f32: by a launch of its own).  Rows of dQ
 * whose q_skip_mask is 0 and rows of dK/dV whose key_mask is 0 are written as zeros.  The per-sample mask / log-sum-exp / delta
 * rows live in LDS for the duration of a workgroup: MADE_ERR_UNSUPPORTED beyond ~8 000 queries or ~24 000 keys (fewer at
 * f32 head dim 128). */
typedef struct MadeAttnBwdArgs {
    const void* Q; const void* K; const void* V; const void* O; const void* dO;
    void* dQ; void* dK; void* dV;
    const float* lse; float* delta;
    int32_t dtype; int32_t hd;
    int64_t B, H, Lq, Lk;
    int64_t q_bs, ldq, k_bs, ldk, v_bs, ldv, o_bs, ldo, do_bs, lddo, dq_bs, lddq, dk_bs, lddk, dv_bs, lddv;
    const float* key_mask; const float* q_skip_mask;
    float scale; int32_t _pad;
    MadeDropout drop;
    const int32_t* batch_order; /* [B] or NULL: issue order of the batch, as in MadeAttnArgs */
    const uint32_t* keep_bits; int64_t ld_bits;   /* NULL, or the forward's dropout decisions (MadeAttnArgs.keep_bits; bf16 path): the
                                                     same mask as re-drawing it, bit for bit */
} MadeAttnBwdArgs;

int made_attention_bwd(const MadeAttnBwdArgs* args, void* stream);

/* made_gemm_tn: C[N,K] (+)= alpha * sum_m A[m,n] * B[m,k]   (reduction index m is the slow dimension of both operands).
 * The weight gradient of every nn.Linear on the path: dW = dY^T X with A = dY [M,N], B = X [M,K], and
 * colsum[n] += alpha * sum_m A[m,n] is the bias gradient.  Also the P^T dO / dS^T Q products of the wide-head attention
 * backward (batched).  Batch index z = z1 * batch2 + z2 with independent element strides per level (a C stride of 0 sums
 * the batch into one C).  split_m > 1 splits the reduction over workgroups; partial tiles are combined with f32 atomic
 * adds, so it requires accumulate = 1 and an f32 C that the caller has initialised (gradients are zeroed once per step).
 * Rows whose row_mask is 0 are read as zero in BOTH operands (padded tokens may hold stale data). */
typedef struct MadeGemmTNArgs {
    const void* A; const void* B; void* C;
    int32_t ab_dtype; int32_t c_dtype;             /* MadeDtype; C is f32 or the operand dtype */
    int64_t M, N, K;
    int64_t lda, ldb, ldc;
    int64_t batch1, batch2;
    int64_t a_zs1, a_zs2, b_zs1, b_zs2, c_zs1, c_zs2;
    const float* row_mask; int64_t mask_zs1, mask_zs2;
    float   alpha; int32_t accumulate;
    int64_t split_m;
    float*  colsum; int64_t colsum_zs1, colsum_zs2; /* optional, always accumulated */
    const float* row_group_valid;                   /* optional [ceil(M/32)] (made_row_groups of row_mask; unbatched calls): slabs
                                                       whose rows are all masked are skipped without being loaded */
    const int32_t* row_index; const int32_t* n_rows; /* optional row gather (unbatched calls): the reduction runs over the
                                                       *n_rows rows row_index[0..] only (made_row_index); row_mask is not needed */
} MadeGemmTNArgs;

int made_gemm_tn(const MadeGemmTNArgs* args, void* stream);

/* made_gemm_tn_grouped: up to 8 weight gradients that reduce over the same rows -- the Linears of one transformer layer
 * (reference music_detr/transformer.py:191-210, model/model_Base.py:64-91: dW_i += alpha * dY_i^T X_i, db_i += alpha * colsum(dY_i)) --
 * in ONE launch: bf16 operands, f32 C accumulated (with atomics, or through the workspace below; the caller zeroes gradients once per step), N_i and K_i multiples
 * of 128, one row list (row_index / n_rows as in made_gemm_tn, or all M rows) and one split of the reduction for all problems.
 * tile_end is filled in by the library. */
#define MADE_GEMM_TN_MAX_GROUP 8
typedef struct MadeGemmTNProblem {
    const void* A; const void* B; float* C; float* colsum;      /* A [M, N] = dY, B [M, K] = X, C [N, K], colsum [N] or NULL */
    int64_t N, K, lda, ldb, ldc;
} MadeGemmTNProblem;
typedef struct MadeGemmTNGroup {
    int32_t n_problems; float alpha;
    int64_t M, split_m;
    const int32_t* row_index; const int32_t* n_rows;
    MadeGemmTNProblem p[MADE_GEMM_TN_MAX_GROUP];
    int32_t tile_end[MADE_GEMM_TN_MAX_GROUP];
    int32_t tile_size; int32_t _pad;     /* 0 / 128: 128 x 128 output tiles (four waves, two workgroups per CU); 256: 256 x 256 tiles (eight waves, one
                                            workgroup per CU, N and K multiples of 256, M <= 36864): twice the flops per operand byte, a quarter of
                                            the atomics; the (tile, 64-row slab) units are cut into equal ranges for the workgroups, split_m is not used */
    void* workspace; int64_t workspace_bytes;   /* optional, 256 x 256 tiles only (ABI 8): with it the workgroups that share a tile's reduction STORE their f32
                                            partials there and a second launch of the same call sums them in a fixed order and updates C with plain
                                            accesses -- instead of adding every partial to C with atomics (67 MB of atomic adds per encoder layer at the
                                            1.2 TB/s the atomic units sustain: 83 of the launch's 146 us).  At least
                                            made_gemm_tn_grouped_workspace(group) bytes, 16-byte aligned, contents arbitrary; calls that share a
                                            workspace, or update the same C, must be ordered on one stream.  The result is then bitwise
                                            reproducible.  NULL: the atomics */
} MadeGemmTNGroup;
int made_gemm_tn_grouped(const MadeGemmTNGroup* group, void* stream);
/* bytes of workspace the group's launch can use (0: none -- not the 256 x 256-tile form); reads n_problems, tile_size, M and the problems' N, K */
int64_t made_gemm_tn_grouped_workspace(const MadeGemmTNGroup* group);
/* out[g] = 1 if any of mask[32g .. 32g+31] is nonzero else 0 (computed once per batch, shared by every weight-gradient product) */
int made_row_groups(const float* mask, int64_t M, float* out, void* stream);
/* made_row_index: compaction of a [M] token mask: row_index[r] = index of the r-th nonzero entry (r < n), entries r >= n repeat
 * the last valid row (0 when there is none), n_rows[0] = n.  One workgroup, M <= 2^22.  Feeds the row gather of made_linear /
 * made_gemm_tn: the GEMMs of a padded batch then cost what its valid tokens cost. */
int made_row_index(const float* mask, int64_t M, int32_t* row_index, int32_t* n_rows, void* stream);
/* made_batch_order: order[r] = index of the sample with the r-th largest number of nonzero mask entries (mask [B, T]; ties in
 * ascending sample index, so the result is a deterministic permutation).  The `batch_order` of made_attention / _bwd.
 * One workgroup, B <= 8192. */
int made_batch_order(const float* mask, int64_t B, int64_t T, int32_t* order, void* stream);

/* Row kernels of the backward pass (all parameter gradients are ACCUMULATED into f32 buffers the caller zeroes once per step).
 *
 * made_layernorm_bwd: dx = LN'(x; gamma)(dy) [+ add]; optionally also dx_drop = dropout(dx) (the gradient entering a
 *   residual branch whose output was dropped: `x + dropout(branch)`); dgamma += sum dy*xhat, dbeta += sum dy.  Rows whose
 *   row_skip is 0 are neither read nor written and contribute nothing (their consumers gather / mask the valid rows).  Autograd of nn.LayerNorm at reference model/model_Base.py:77-78,
 *   music_detr/transformer.py:157-158,235-237, modules/transformer.py:141-143. */
int made_layernorm_bwd(const void* x, int32_t x_dtype, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride,
                       const float* gamma, const void* dy, int32_t dy_dtype, int64_t lddy,
                       const void* add, int32_t add_dtype, int64_t ld_add,
                       void* dx, int32_t dx_dtype, int64_t lddx,
                       void* dx_drop, int64_t lddxd, const MadeDropout* drop, int64_t drop_ld,
                       float* dgamma, float* dbeta, int64_t rows, int64_t D, float eps, const float* row_skip, void* stream);

/* made_pool_bwd: backward of vec = normalize(masked_mean(local)) (reference model/model_Base.py:579-580,615-616) merged with
 *   the other gradients of `local`: out[b,t,:] = mask[b,t] * (in1[b,t,:] + in2[b,t,:] + dmean[b,:] / count_b),
 *   dmean = (dvec - vhat (vhat.dvec)) / max(|mean|, eps).  in1 / in2 may be NULL. */
int made_pool_bwd(const float* mean, const float* dvec, const float* mask,
                  const void* in1, int32_t in1_dtype, int64_t in1_bs, int64_t in1_ld,
                  const void* in2, int32_t in2_dtype, int64_t in2_bs, int64_t in2_ld,
                  void* out, int32_t out_dtype, int64_t out_bs, int64_t out_ld,
                  int64_t B, int64_t T, int64_t D, float eps, void* stream);

/* made_l2norm_bwd: y = x / max(|x|, eps)  ->  dx = (dy - yhat (yhat.dy)) / max(|x|, eps); dx f32 (stored or accumulated)
 *   and/or dx_alt in another dtype (always the plain gradient).  dy row of x row r is r / dy_rows_per (0 or 1: one each).  F.normalize at reference model/model_Uni.py:142-146, cosine at modules/loss.py:52-56. */
int made_l2norm_bwd(const void* x, int32_t x_dtype, int64_t ldx, const float* dy, int64_t lddy, int64_t dy_rows_per,
                    float* dx, int64_t lddx, int32_t accumulate, void* dx_alt, int32_t alt_dtype, int64_t lddxa,
                    int64_t rows, int64_t D, float eps, void* stream);

/* made_clip_loss_bwd: gradient of weight * CLIPLoss(sims, logit_scale) (reference modules/loss.py:5-24) times upstream[0]
 *   (NULL = 1): dsims [n,n] and its transpose dsims_t (may be NULL), d_logit_scale[0] += .  lse_ws: [2n] f32 workspace. */
int made_clip_loss_bwd(const float* sims, int64_t ld, int64_t n, const float* logit_scale, float weight,
                       const float* upstream, float* lse_ws, float* dsims, float* dsims_t, int32_t accumulate,
                       float* d_logit_scale, const float* row_exclude, void* stream);

/* made_xpool_tail_bwd: backward of made_xpool_tail (LayerNorm3 + cosine with the video, reference modules/transformer.py:178,
 *   modules/metrics.py:10-24): dy [Nm*Nv, D], optionally dy_drop = dropout(dy) (element index row*D + col), dgamma/dbeta
 *   accumulated, dvideo[n,:] += (atomic).  dpool [Nm, D] f32 or NULL: an additional gradient of the pooled rows themselves,
 *   dpool[m, :] * dpool_scale for every n (moment_query_type = "xpool": the decoder's content query is the mean over the
 *   videos of a track's pooled vectors, reference model/model_Uni.py:222-223). */
int made_xpool_tail_bwd(const void* y, int32_t y_dtype, int64_t ldy, const float* gamma, const float* beta,
                        const float* video, int64_t ld_video, const float* dsims, int64_t ld_dsims,
                        void* dy, int32_t dy_dtype, int64_t lddy, void* dy_drop, const MadeDropout* drop,
                        float* dgamma, float* dbeta, float* dvideo, int64_t ld_dvideo,
                        const float* dpool, int64_t ld_dpool, float dpool_scale,
                        int64_t Nm, int64_t Nv, int64_t D, float eps, void* stream);

/* made_softmax_bwd: softmax backward of the wide-head attention with materialised scores (few query rows per batch):
 *   P = softmax(scale*S + mask), Pd = dropout(P), dP = dropout'(dPd + extra[row]), dS = scale * P * (dP - sum_k P_k dP_k).
 *   Writes Pd and dS [rows, ldo] (columns L..ldo-1 zero) and dS^T [rows/rows_per_batch, L, ldt] in out_dtype; the products
 *   dV = Pd^T dO, dK = dS^T Q and dQ = dS K are then made_gemm_tn calls.  mask row = row / rows_per_mask. */
int made_softmax_bwd(const float* S, int64_t ld_s, const float* dP, int64_t ld_dp, const float* mask, int64_t rows_per_mask,
                     const float* extra, float scale, const MadeDropout* drop,
                     void* Pd, void* dS, void* dSt, int32_t out_dtype, int64_t ldo, int64_t ldt,
                     int64_t out_batch_stride, int64_t t_batch_stride,   /* 0: dense (rows_per_batch*ldo, L*ldt) */
                     int64_t rows, int64_t rows_per_batch, int64_t L, void* stream);

/* x[row, h*hd + j] += s[row, h] * bias[h*hd + j]: the value-projection bias of the memory-space cross-attention when the
 * attention weights of a row no longer sum to 1 (dropout); s = that sum.  And its backward. */
int made_head_bias(void* x, int32_t x_dtype, int64_t ldx, const float* s, const float* bias, int64_t rows, int64_t H, int64_t hd,
                   void* stream);
int made_head_bias_bwd(const void* dy, int32_t dtype, int64_t ld, const float* s, const float* bias, float* dbias, float* ds,
                       int64_t rows, int64_t H, int64_t hd, void* stream);

/* made_layernorm_bwd2: two chained LayerNorms backward in one pass -- the decoder layer's norm 3 followed by the shared output norm
 * (reference music_detr/transformer.py:306 and :136): with t3 = LN_a(xa) and hs = LN_b(xb = the saved t3),
 *   g = LN_b'(dy) + add,  dx = LN_a'(g),  dx_drop = dropout(dx)  (element index row * drop_ld + col);
 * dgamma / dbeta of both norms are accumulated.  All row tensors share `dtype`; D <= 1024. */
int made_layernorm_bwd2(const void* xa, const float* gamma_a, int64_t ldxa, const void* xb, const float* gamma_b, int64_t ldxb,
                        const void* dy, int64_t lddy, const void* add, int64_t ld_add, void* dx, int64_t lddx,
                        void* dx_drop, int64_t lddxd, const MadeDropout* drop, int64_t drop_ld, int32_t dtype,
                        float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b, int64_t rows, int64_t D, float eps,
                        void* stream);

/* made_gate_rows: out[r, c] = dropout(x[r, c] * act'(G[r, c]) * scale)  (element index of the dropout r*drop_ld + c/drop_col_div,
 * drop_col_div <= 0 meaning 1: with drop_col_div = head width and drop_ld = H the mask is one draw per (row, head) -- the
 * attention-weight dropout of a self-attention over a single key, whose only weight is 1; rows whose row_skip is 0 are left
 * untouched).  The element-wise options of a made_linear epilogue for the places of the backward chain
 * where no GEMM can carry them: the activation after the input projection (reference model/model_Base.py:559-561, whose
 * input needs no gradient) and a residual stream that enters a dropped branch (temporal blocks deeper than one layer). */
int made_gate_rows(const void* x, int32_t x_dtype, int64_t ldx, const void* G, int32_t g_dtype, int64_t ldg, int32_t gate,
                   float scale, const MadeDropout* drop, int64_t drop_ld, int64_t drop_col_div, void* out, int32_t out_dtype, int64_t ldo,
                   const float* row_skip, int64_t rows, int64_t cols, void* stream);
/* out = a + b + c over n contiguous elements (b, c may be NULL; any mix of f32 / bf16): merges gradient streams.
 * b_mod > 0: b is broadcast, b[i % b_mod] (the decoder's query embedding added to every sample). */
int made_add3(void* out, int32_t out_dtype, const void* a, int32_t a_dtype, const void* b, int32_t b_dtype,
              const void* c, int32_t c_dtype, int64_t n, int64_t b_mod, void* stream);
/* out[c] += sum over rows of x[row, c]  (f32, accumulated): gradient of a row vector that was broadcast over the rows. */
int made_colsum(const void* x, int32_t dtype, int64_t ld, int64_t rows, int64_t cols, float* out, void* stream);

/* made_posbn_relu_fwd / _bwd: y = relu(BatchNorm1d_over_positions(x)) of the EmbeddingNet aggregator (agg_module = "mlp", reference
 *   model/model_Base.py:216-249: nn.BatchNorm1d(num_features = T) applied to [B, T, F], so position t is the channel and its
 *   statistics run over the B * F values there).  x / y: rows b * T + t of pitch ldx / ldy, F columns.  batch_stats != 0
 *   (model.train()): biased batch statistics normalise, running_mean / running_var (may both be NULL) move by `momentum` towards
 *   the batch mean / unbiased variance; batch_stats == 0 (model.eval()): the running statistics normalise.  save_mean /
 *   save_rstd [T] keep what was used, for the backward.  _bwd: dx (stored), dweight[t] += , dbias[t] += (may be NULL). */
int made_posbn_relu_fwd(const void* x, int32_t x_dtype, int64_t ldx, const float* weight, const float* bias,
                        float* running_mean, float* running_var, float momentum, float eps, int32_t batch_stats,
                        float* save_mean, float* save_rstd, void* y, int32_t y_dtype, int64_t ldy,
                        int64_t B, int64_t T, int64_t F, void* stream);
int made_posbn_relu_bwd(const void* x, int32_t x_dtype, int64_t ldx, const void* y, int32_t y_dtype, int64_t ldy,
                        const void* dy, int32_t dy_dtype, int64_t lddy, const float* weight, const float* save_mean,
                        const float* save_rstd, int32_t batch_stats, void* dx, int32_t dx_dtype, int64_t lddx,
                        float* dweight, float* dbias, int64_t B, int64_t T, int64_t F, void* stream);

/* made_set_criterion_bwd: gradients of made_set_criterion's total (times upstream[0]) w.r.t. pred_logits, pred_spans
 *   [n_layers,B,Q,2], proj_queries [n_layers,B,Q,Dc] (stored) and vid_sum [B,Dc] (accumulated).  d_logits / d_spans rows have
 *   pitch ld_out >= 2 (only columns 0,1 are written: a zero-padded pitch of 8 lets the row feed made_linear as an A operand);
 *   through_sigmoid != 0: d_spans is the gradient w.r.t. the pre-sigmoid value (pred_spans = sigmoid(z), model_Uni.py:135). */
int made_set_criterion_bwd(const float* pred_logits, const float* pred_spans, const float* targets,
                           const int64_t* pred_idx, const int64_t* tgt_idx, const int32_t* count,
                           const float* proj_queries, const float* vid_sum, const float* empty_weight,
                           int64_t n_layers, int64_t B, int64_t Q, int64_t G, int64_t Dc, int32_t fg_label,
                           float temperature, const float* weights, const float* upstream,
                           float* d_logits, float* d_spans, int64_t ld_out, int32_t through_sigmoid,
                           float* d_proj_queries, float* d_vid_sum, void* stream);

/* made_adam_step: the optimizer tail of one training iteration on the flat f32 master buffer (reference train-MaDe.py:
 * 262-266,375-381): for each group the L2 norm of its (grad_scale-scaled) gradient is clipped to max_norm exactly as
 * nn.utils.clip_grad_norm_ does (coef = max_norm / (norm + 1e-6), capped at 1; max_norm <= 0: no clipping), then
 * torch.optim.Adam's update (amsgrad off, weight_decay 0) with the group's lr (host-side schedule) and the given step
 * count (>= 1, for the bias corrections).  Elements outside every group are left untouched (the reference keeps
 * decoder_query_embed out of the optimizer).  grad_scale folds the 1/world_size of a summed data-parallel all-reduce.
 * A group with begin == end is skipped (nothing of it is read or written): a step may be applied in parts -- the groups whose gradients
 * are final early, under the rest of the backward pass; the others at the end -- with the same step count in both calls.
 * norm_ws: [MADE_ADAM_MAX_GROUPS * (1 + MADE_ADAM_NORM_BLOCKS)] f32 device workspace; its first MADE_ADAM_MAX_GROUPS entries
 * return the squared group norms.  The norms are reduced in a fixed order (no atomics), so data-parallel ranks that hold the
 * same all-reduced gradients apply bit-identical updates. */
#define MADE_ADAM_MAX_GROUPS 4
#define MADE_ADAM_NORM_BLOCKS 1024
typedef struct MadeAdamGroup { int64_t begin, end; float lr; float max_norm; } MadeAdamGroup;
int made_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   const MadeAdamGroup* groups, int32_t n_groups, float beta1, float beta2, float eps, int64_t step,
                   float grad_scale, float* norm_ws, void* stream);

/* made_adam_step_device: the same tail with its per-step scalars in DEVICE memory, so that the whole training iteration can sit in
 * one captured hipGraph: `state->step` is incremented by the launch (advance_state != 0: the first call of a step
 * applied in parts; 0: a later part of the same step) and then used for the bias corrections;
 * `state->lr[g]` replaces groups[g].lr (the host-side LambdaLR schedule writes the three floats before it replays the graph). */
typedef struct MadeAdamDeviceState { int64_t step; float lr[MADE_ADAM_MAX_GROUPS]; float bc1, bc2_sqrt; } MadeAdamDeviceState;
int made_adam_step_device(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                          const MadeAdamGroup* groups, int32_t n_groups, float beta1, float beta2, float eps,
                          MadeAdamDeviceState* state_device, int32_t advance_state, float grad_scale, float* norm_ws, void* stream);

/* made_repack: rebuild the kernel-facing copies of every matrix parameter from the f32 masters in one launch:
 * w (rows x cols, `dtype`; NULL = the kernels read the master itself) and wt = W^T (cols x wt_ld, wt_ld >= rows; NULL = not
 * needed).  `descs_device` is a device array sorted by tile_begin (prefix sum of ceil(rows/64)*ceil(cols/64): one workgroup per
 * 64 x 64 tile). */
typedef struct MadeRepackDesc {
    const float* src; void* w; void* wt;
    int64_t rows, cols, wt_ld, tile_begin;
    int32_t dtype; int32_t _pad;
} MadeRepackDesc;
int made_repack(const MadeRepackDesc* descs_device, int32_t n_desc, int64_t total_tiles, void* stream);

/* ==========================================================================================
 * Evaluation metrics that follow the similarity matrix: computed where the matrix lives.
 * ========================================================================================== */

/* made_recall_ranks: de-duplicated rank of every video's ground-truth music (reference utils/util_test.py:44-70, Recall_metrics
 * with dedup=True).  group_id [Nm] int32 maps every music column to the index of its music id (columns with the same id form a
 * group; built once on the host from the id strings), gt_group [Nv] int32 is the group of each video's ground truth.
 * rank_out[i] = number of other groups whose best similarity in row i exceeds the ground-truth group's best one (0 = retrieved
 * first) -- what walking the descending sort while skipping already-seen ids counts.  Exact ties between different ids are
 * counted as ranked after the ground truth (the reference's order among equal values is that of an unstable sort).
 * top1_out [Nv] (may be NULL): column of the row maximum (lowest index among equal maxima).  n_groups <= 32768. */
int made_recall_ranks(const float* sims, int64_t ld, const int32_t* group_id, const int32_t* gt_group, int64_t Nv, int64_t Nm,
                      int64_t n_groups, int32_t* rank_out, int32_t* top1_out, void* stream);

/* made_span_iou: per sample, the highest-scoring query's span (centre, width) -> (start, end) seconds and its IoU with the
 * ground-truth moment (reference test-MaDe.py:304-313 ranked_preds[0]; music_detr/span_utils.py:119-170 detr_iou /
 * individual_IoU_tensor, discounted = False).  pred_out [N,3] (may be NULL) = (start, end, foreground probability) unclamped. */
int made_span_iou(const float* pred_logits, const float* pred_spans, const float* gt_moment, const float* m_duration,
                  int64_t N, int64_t Q, int32_t fg_label, float max_m_duration, float* iou_out, float* pred_out, void* stream);

/* out[b, :cols_a] = a[b, :], out[b, cols_a:] = c[b, :]; contiguous f32 rows.  The DETR token mask [frame mask ; segment mask]
 * (reference model/model_Uni.py:209 torch.cat); either part may be empty. */
int made_concat_cols(const float* a, int64_t cols_a, const float* c, int64_t cols_c, float* out, int64_t rows, void* stream);

/* ---- the free functions the reference's drivers import (train-MaDe.py:16,20,22); Python mirrors with the
 * reference's names: mgsv_amd/modules/metrics.py, mgsv_amd/modules/loss.py, mgsv_amd/music_detr/span_utils.py ---- */

/* out[a * sa + p * sp] = cos(anchor[a, :], pooled[p, a, :]); anchor f32 [A, D] (row stride lda), pooled [P, A, D] contiguous (f32 or
 * bf16).  reference modules/metrics.py:10-24 sim_matrix_music_pooling (anchor = videos, sa = ld, sp = 1) and :26-41
 * sim_matrix_video_pooling (anchor = tracks, sa = 1, sp = ld). */
int made_pooled_cosine(const float* anchor, int64_t lda, const void* pooled, int32_t pooled_dtype, float* out, int64_t sa, int64_t sp,
                       int64_t A, int64_t P, int64_t D, void* stream);
/* out[i] = x[i] * exp(*logit_scale): the scaled logits of reference modules/loss.py:12-13 (CLIPLoss), :86-88 (InfoNCELoss). */
int made_scale_exp(const float* x, const float* logit_scale, float* out, int64_t n, void* stream);
/* mode 0: (centre, width) -> (start, end) (reference music_detr/span_utils.py:15-24 span_cw_to_se); 1: the inverse (:4-13). [N, 2] f32. */
int made_span_convert(const float* in, float* out, int64_t N, int32_t mode, void* stream);
/* all pairs of spans1 [N, 2] x spans2 [M, 2] (start, end): IoU and union (reference span_utils.py:39-66 temporal_iou), generalised
 * IoU (:86-115), intersection over the second span (:69-83); any output may be NULL.  [N, M] f32 each. */
int made_span_pairwise(const float* spans1, const float* spans2, float* iou, float* uni, float* giou, float* inter_over_2,
                       int64_t N, int64_t M, void* stream);
/* IoU of one predicted (start, end) in seconds per sample with its ground-truth moment (reference span_utils.py:119-170:
 * individual_IoU_tensor; clamp_to_max = 1 adds detr_iou's clamp of the prediction to [0, max_m_duration]). */
int made_span_iou_se(const float* pred_se, const float* gt_moment, const float* m_duration, int64_t N, float max_m_duration,
                     int32_t clamp_to_max, int32_t discounted, float* iou_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MADE_HIP_H */
