"""Seeded synthetic weights and inputs for the MaDe hot path.

There are no datasets or checkpoints in this environment (reference:
.MISSING_LARGE_BLOBS), so parity tests, golden fixtures and the benchmark all
draw weights and inputs from this generator (NumPy PCG64; seed 0 = weights,
seed 1 = data).  The parameter names and shapes are the
reference's `state_dict` layout, so the same dict loads
strictly into the reference's `Uni_model` (done in tests/golden/make_golden.py)
and into this package's `Uni_model`.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

from .config import MadeConfig


def param_shapes(cfg: MadeConfig) -> "OrderedDict[str, tuple]":
    """name -> shape for every learnable tensor and persistent buffer on the path."""
    D = cfg.D
    F = cfg.temporal_ffn_dim
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def lin(name, out_f, in_f, bias=True):
        s[name + ".weight"] = (out_f, in_f)
        if bias:
            s[name + ".bias"] = (out_f,)

    def ln(name, d=D):
        s[name + ".weight"] = (d,)
        s[name + ".bias"] = (d,)

    def mha(name, d=D):
        s[name + ".in_proj_weight"] = (3 * d, d)
        s[name + ".in_proj_bias"] = (3 * d,)
        lin(name + ".out_proj", d, d)

    # feature projections (model_Base.py:282,289)
    lin("ast_proj", D, cfg.ast_dim)
    lin("vit_proj", D, cfg.vit_dim)
    if cfg.agg_module == "mlp":
        # EmbeddingNet (model_Base.py:216-249,357-377): Linear(D,1024) - BatchNorm1d(T) - ReLU - Linear(1024,D) - BatchNorm1d(T) -
        # ReLU - Linear(D,D); BatchNorm1d sees [B, T, F] so its channels are the T token positions
        for mod, T in (("Video_encoder_projection", cfg.max_v_frames), ("Music_encoder_projection", cfg.max_snippet_num)):
            lin(mod + ".net.0", 1024, D)
            for bn in (".net.1", ".net.4"):
                s[mod + bn + ".weight"] = (T,); s[mod + bn + ".bias"] = (T,)
                s[mod + bn + ".running_mean"] = (T,); s[mod + bn + ".running_var"] = (T,)
                s[mod + bn + ".num_batches_tracked"] = ()
            lin(mod + ".net.3", D, 1024)
            lin(mod + ".net.6", D, D)
    else:
        if cfg.with_cls_token:                              # model_Base.py:314-321
            s["video_cls_token"] = (1, 1, D)
            s["audio_cls_token"] = (1, 1, D)
        # fixed sin/cos tables are persistent buffers (model_Base.py:58)
        s["video_position_embedding.pe"] = (1, cfg.video_attention_seqlen, D)
        s["audio_position_embedding.pe"] = (1, cfg.audio_attention_seqlen, D)
        # temporal blocks (model_Base.py:64-80); one shared block when transformer_is_share (model_Base.py:300-302,322-331)
        share = bool(cfg.transformer_is_share) and cfg.video_transformer_depth == cfg.audio_transformer_depth and cfg.video_transformer_depth > 0
        mods = ((("share_transformer", cfg.video_transformer_depth),) if share else
                (("video_transformer", cfg.video_transformer_depth), ("audio_transformer", cfg.audio_transformer_depth)))
        for mod, depth in mods:
            for l in range(depth):
                p = f"{mod}.layers.{l}"
                ln(p + ".0")
                mha(p + ".1")
                ln(p + ".2")
                lin(p + ".3.0", F, D)
                lin(p + ".3.3", D, F)
            lin(mod + ".final_linear", D, D)
    # DETR transformer (music_detr/transformer.py)
    Fd = cfg.detr_dim_feedforward
    for l in range(cfg.detr_enc_layers):
        p = f"detr_transformer.encoder.layers.{l}"
        mha(p + ".self_attn")
        lin(p + ".linear1", Fd, D)
        lin(p + ".linear2", D, Fd)
        ln(p + ".norm1")
        ln(p + ".norm2")
    if cfg.detr_enc_layers > 0 and getattr(cfg, "detr_pre_norm", False):
        ln("detr_transformer.encoder.norm")                       # (music_detr/transformer.py:34: only with normalize_before)
    for l in range(cfg.detr_dec_layers):
        p = f"detr_transformer.decoder.layers.{l}"
        mha(p + ".self_attn")
        mha(p + ".multihead_attn")
        lin(p + ".linear1", Fd, D)
        lin(p + ".linear2", D, Fd)
        ln(p + ".norm1")
        ln(p + ".norm2")
        ln(p + ".norm3")
    if cfg.detr_dec_layers > 0:
        ln("detr_transformer.decoder.norm")
    # X-Pool block (modules/transformer.py:128-146)
    if "XA" in cfg.vmr_fusion:
        towers = []
        if "music" in cfg.vmr_fusion:
            towers.append("video_guided_to_music_pooling_cross_transformer")
        if "video" in cfg.vmr_fusion:
            towers.append("music_guided_to_video_pooling_cross_transformer")
        for xa in towers:
            for pj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                lin(f"{xa}.cross_attn.{pj}", D, D)
            lin(xa + ".linear_proj", D, D)
            ln(xa + ".layer_norm1")
            ln(xa + ".layer_norm2")
            ln(xa + ".layer_norm3")
    s["logit_scale"] = ()
    # CA fusion (model_Uni.py:31-43, model_Base.py:99-193)
    if "CA" in cfg.mml_fusion:
        ca = "video_music_fusion_cross_transformer"
        inner = cfg.ca_heads * cfg.ca_dim_head
        lin(ca + ".layers.0.0.to_q", inner, D, bias=False)
        lin(ca + ".layers.0.0.to_kv", 2 * inner, D, bias=False)
        lin(ca + ".layers.0.0.to_out.0", D, inner)
        lin(ca + ".layers.0.1.net.0", cfg.ca_ffn_dim, D)
        lin(ca + ".layers.0.1.net.3", D, cfg.ca_ffn_dim)
        ln(ca + ".attention_query_layer_norms.0")
        ln(ca + ".attention_context_layer_norms.0")
        ln(ca + ".ff_layer_norms.0")
        lin(ca + ".final_linear", D, D)
    # heads (model_Uni.py:46-64)
    s["decoder_query_embed.weight"] = (cfg.num_moment_queries, D)
    span_dim = 1 if cfg.predict_center == 1 else 2
    if "regression" in cfg.mml_localization:               # model_Uni.py:66-69: MLP(D, 256, span_dim, 3), no criterion
        lin("reg_mlp.layers.0", 256, D)
        lin("reg_mlp.layers.1", 256, 256)
        lin("reg_mlp.layers.2", span_dim, 256)
        return s
    lin("span_embed.layers.0", D, D)
    lin("span_embed.layers.1", D, D)
    lin("span_embed.layers.2", span_dim, D)
    lin("class_embed", 2, D)
    if cfg.moment_loss:                                   # model_Uni.py:55-56: MLP(D, D, D, 3)
        for i in range(3):
            lin(f"moment_embed.layers.{i}", D, D)
    if cfg.contrastive_align_loss:
        lin("contrastive_align_projection_query", cfg.contrastive_hdim, D)
        lin("contrastive_align_projection_vid", cfg.contrastive_hdim, D)
    s["criterion.empty_weight"] = (2,)
    return s


def sincos_table(seq_len: int, D: int) -> np.ndarray:
    """Fixed sin/cos table, float32 arithmetic (reference: model_Base.py:48-60).

    pe[p, 2j] = sin(p * exp(-2j ln(1e4)/D)), pe[p, 2j+1] = cos(same)."""
    pos = np.arange(seq_len, dtype=np.float32)[:, None]
    two_j = np.arange(0, D, 2, dtype=np.float32)
    div = np.exp(two_j * np.float32(-(math.log(10000.0) / D))).astype(np.float32)
    ang = (pos * div[None, :]).astype(np.float32)
    pe = np.zeros((seq_len, D), dtype=np.float32)
    pe[:, 0::2] = np.sin(ang)
    pe[:, 1::2] = np.cos(ang)
    return pe[None]


def make_state_dict(cfg: MadeConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Seeded float32 weights in the reference's key layout.

    Linear weights ~ N(0, 1/fan_in) so activations stay O(1) through the stack
    (the reference's own initialisers, e.g. eye_ for the X-Pool projections,
    modules/transformer.py:148-154, would make several kernels trivially testable);
    biases ~ N(0, 0.05^2); LayerNorm gamma = 1 + 0.1 N, beta = 0.1 N.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        if name.endswith(".pe"):
            sd[name] = sincos_table(shape[1], shape[2])
        elif name == "logit_scale":
            sd[name] = np.array(math.log(1.0 / cfg.temperature_init_value), dtype=np.float32)
        elif name == "criterion.empty_weight":
            w = np.ones(2, dtype=np.float32)
            w[cfg.background_label] = 0.1      # eos_coef, model_Uni.py:64 / loss_detr.py:55-57
            sd[name] = w
        elif name == "decoder_query_embed.weight":
            sd[name] = rng.standard_normal(shape, dtype=np.float32)
        elif name.endswith(".num_batches_tracked"):
            sd[name] = np.array(0, dtype=np.int64)
        elif name.endswith(".running_var"):
            sd[name] = (0.5 + np.abs(rng.standard_normal(shape, dtype=np.float32))).astype(np.float32)
        elif name.endswith(".running_mean"):
            sd[name] = (0.3 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif name.endswith("_cls_token"):
            sd[name] = (0.02 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif len(shape) == 2:                  # Linear weight / in_proj_weight
            sd[name] = (rng.standard_normal(shape, dtype=np.float32) / np.float32(math.sqrt(shape[1])))
        else:                                  # 1-D: bias or LayerNorm
            is_ln_weight = name.endswith(".weight")
            v = rng.standard_normal(shape, dtype=np.float32)
            sd[name] = (1.0 + 0.1 * v).astype(np.float32) if is_ln_weight else (0.05 * v).astype(np.float32)
    # LayerNorm biases get 0.1 N (they were drawn as 0.05 N above; scale in place, same stream)
    for name in sd:
        if name.endswith(".bias") and name[:-5] + ".weight" in sd and sd[name[:-5] + ".weight"].ndim == 1:
            sd[name] = (sd[name] * 2.0).astype(np.float32)
    return sd


def make_inputs(cfg: MadeConfig, B: int, T_v: int | None = None, T_a: int | None = None,
                seed: int = 1, min_len_v: int = 5, min_len_a: int = 12) -> dict:
    """Seeded synthetic batch with the dataset's schema (reference:
    dataloaders/dataloader_MGSV_EC_feature.py:46-75): prefix-ones float masks, padded
    feature rows zeroed, spans_target = (centre, width) / max_m_duration."""
    T_v = cfg.max_v_frames if T_v is None else T_v
    T_a = cfg.max_snippet_num if T_a is None else T_a
    rng = np.random.Generator(np.random.PCG64(seed))
    frame_feats = rng.standard_normal((B, T_v, cfg.vit_dim), dtype=np.float32)
    segment_feats = rng.standard_normal((B, T_a, cfg.ast_dim), dtype=np.float32)
    len_v = rng.integers(min(min_len_v, T_v), T_v + 1, size=B)
    len_a = rng.integers(min(min_len_a, T_a), T_a + 1, size=B)
    frame_masks = (np.arange(T_v)[None, :] < len_v[:, None]).astype(np.float32)
    segment_masks = (np.arange(T_a)[None, :] < len_a[:, None]).astype(np.float32)
    frame_feats *= frame_masks[:, :, None]
    segment_feats *= segment_masks[:, :, None]
    c = rng.uniform(0.25, 0.75, size=(B, 1)).astype(np.float32)
    w = rng.uniform(0.02, 0.22, size=(B, 1)).astype(np.float32)
    spans_target = np.stack([c, w], axis=-1).astype(np.float32)     # [B, 1, 2]
    v_duration = rng.uniform(5.0, 45.0, size=B).astype(np.float32)
    return dict(
        frame_feats=frame_feats, segment_feats=segment_feats,
        frame_masks=frame_masks, segment_masks=segment_masks,
        spans_target=spans_target, v_duration=v_duration,
        video_ids=[str(i) for i in range(B)], music_ids=[str(i) for i in range(B)],
    )


def make_retrieval_inputs(N_v: int, N_m: int, S: int, D: int, seed: int = 2, min_len: int = 12) -> dict:
    """Embeddings of a whole split as test-MaDe.py:386-391 assembles them: L2-normalised
    video/music vectors, per-segment music embeddings with zeroed padded rows."""
    rng = np.random.Generator(np.random.PCG64(seed))
    v = rng.standard_normal((N_v, D), dtype=np.float32)
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    seg = rng.standard_normal((N_m, S, D), dtype=np.float32)
    lens = rng.integers(min(min_len, S), S + 1, size=N_m)
    masks = (np.arange(S)[None, :] < lens[:, None]).astype(np.float32)
    seg *= masks[:, :, None]
    m = seg.sum(1) / masks.sum(1, keepdims=True)
    m /= np.linalg.norm(m, axis=-1, keepdims=True)
    return dict(video_embeds=v, segment_embeds=seg, segment_masks=masks, music_embeds=m.astype(np.float32))


def make_ranked_retrieval_inputs(N: int, S: int, D: int, seed: int = 21, hard: bool = False, sigma: float = 4.0, eps: float = 2e-3) -> dict:
    """A split of N (video i, track i) samples whose similarity rows have a meaningful ranking (the metric test-MaDe.py reads through
    utils/util_test.py:32-96): video i is a noisy copy of its track's pooled vector, so the ground truth ranks near the top but not
    always first.  hard: tracks 2k and 2k + 1 are near-duplicates (the same segments up to eps relative noise), so every row has two
    candidates whose similarities differ by far less than 1e-2 -- the case a reduced-precision scorer can rank the other way."""
    rng = np.random.Generator(np.random.PCG64(seed))
    seg = rng.standard_normal((N, S, D), dtype=np.float32)
    lens = rng.integers(min(12, S), S + 1, size=N)
    if hard:
        seg[1::2] = seg[0::2][: len(seg[1::2])] + eps * rng.standard_normal(seg[1::2].shape, dtype=np.float32)
        lens[1::2] = lens[0::2][: len(lens[1::2])]
    masks = (np.arange(S)[None, :] < lens[:, None]).astype(np.float32)
    seg *= masks[:, :, None]
    m = seg.sum(1) / masks.sum(1, keepdims=True)
    m /= np.linalg.norm(m, axis=-1, keepdims=True)
    v = m + sigma / np.sqrt(D) * rng.standard_normal((N, D), dtype=np.float32)
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    return dict(video_embeds=v.astype(np.float32), segment_embeds=seg, segment_masks=masks, music_embeds=m.astype(np.float32))
