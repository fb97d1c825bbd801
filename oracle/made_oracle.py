"""CPU oracle for the MaDe hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product path (`mgsv_amd/`) never routes through it and
raises when the HIP library is missing.

What it is: a clean-room, functional (no nn.Module) float32 restatement of the
reference's `Uni_model.forward` and of the all-pairs retrieval scoring, written from
the math in SURVEY.md Appendix A.  Every function cites the reference lines it
restates.  Parameters are looked up by the reference's `state_dict` key names.

How it is pinned:
  * against the reference itself, imported in the build container from
    /root/reference by `oracle/validate_against_reference.py` (max-abs error per
    output is recorded in tests/golden/VALIDATION.json);
  * against golden vectors produced by the reference (tests/golden/*.npz, generated
    by tests/golden/make_golden.py) -- these travel to the GPU box, the reference
    does not;
  * against the two known-answer tests the reference holds: the matcher example in
    music_detr/test_matcher.py:15-29 and the IoU/GIoU doctests in
    music_detr/span_utils.py:48-54,99-103.
Third-party arithmetic restated here: SciPy's `linear_sum_assignment` (unpinned in
the reference's requirements.txt; container has 1.15.3) -- the shortest-augmenting-
path algorithm of Crouse (2016) as SciPy implements it, including its tie-break.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LN_EPS = 1e-5


# ----------------------------------------------------------------------------- basics
def _t(x) -> Tensor:
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x))


def to_torch_params(sd: Dict[str, np.ndarray]) -> Dict[str, Tensor]:
    return {k: _t(v).clone() for k, v in sd.items()}


def linear(x: Tensor, P, name: str, bias: bool = True) -> Tensor:
    y = x @ P[name + ".weight"].t()
    if bias and (name + ".bias") in P:
        y = y + P[name + ".bias"]
    return y


def layer_norm(x: Tensor, P, name: str) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * P[name + ".weight"] + P[name + ".bias"]


def gelu_erf(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def quick_gelu(x: Tensor) -> Tensor:
    # reference: model/model_Base.py:17-20
    return x * torch.sigmoid(1.702 * x)


def l2_normalize(x: Tensor, eps: float = 1e-12) -> Tensor:
    # F.normalize semantics (used at model_Base.py:580,616; model_Uni.py:142-146)
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


class Drop:
    """Train-mode dropout with the build's stateless masks (mgsv_amd/dropout.py): keep(seed, site, flat index).
    The reference draws from torch's global generator instead; value semantics (zero or x/(1-p)) are the same
    (torch.nn.Dropout / the attention-weight dropout inside nn.MultiheadAttention)."""

    def __init__(self, seed: int, p_detr: float = 0.1, p_temporal: float = 0.8, p_xpool: float = 0.3):
        self.seed, self.p_detr, self.p_temporal, self.p_xpool = int(seed), p_detr, p_temporal, p_xpool

    def __call__(self, x: Tensor, site: str, p: float) -> Tensor:
        """x in the site's logical layout (contiguous order = flat index)."""
        if p <= 0.0:
            return x
        from mgsv_amd import dropout as dr
        keep = dr.keep_mask(self.seed, dr.site_id(site), p, x.numel()).reshape(tuple(x.shape))
        return x * torch.from_numpy(keep).to(x.dtype) * (1.0 / (1.0 - p))


def _drop(drop: Optional["Drop"], x: Tensor, site: str, p: float) -> Tensor:
    return x if drop is None else drop(x, site, p)


def mha(xq: Tensor, xk: Tensor, xv: Tensor, P, name: str, H: int,
        key_is_pad: Optional[Tensor], drop: Optional[Drop] = None, site: str = "", p: float = 0.0) -> Tensor:
    """torch.nn.MultiheadAttention as the reference uses it, batch-first here.

    xq [B,Lq,D], xk/xv [B,Lk,D], key_is_pad [B,Lk] bool (True = padded key)."""
    B, Lq, D = xq.shape
    Lk = xk.shape[1]
    hd = D // H
    W = P[name + ".in_proj_weight"]
    b = P[name + ".in_proj_bias"]
    q = xq @ W[:D].t() + b[:D]
    k = xk @ W[D:2 * D].t() + b[D:2 * D]
    v = xv @ W[2 * D:].t() + b[2 * D:]
    q = q.view(B, Lq, H, hd).transpose(1, 2) / math.sqrt(hd)
    k = k.view(B, Lk, H, hd).transpose(1, 2)
    v = v.view(B, Lk, H, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)                                   # [B,H,Lq,Lk]
    if key_is_pad is not None:
        s = s.masked_fill(key_is_pad[:, None, None, :], float("-inf"))
    a = _drop(drop, torch.softmax(s, dim=-1), site, p)            # dropout on the attention weights ([B,H,Lq,Lk])
    o = (a @ v).transpose(1, 2).reshape(B, Lq, D)
    return linear(o, P, name + ".out_proj")


# ------------------------------------------------------------ K1-K4: feature encoders
def temporal_block(x: Tensor, mask: Tensor, P, mod: str, depth: int, H: int, drop: Optional[Drop] = None, tag: Optional[str] = None) -> Tensor:
    """reference: model/model_Base.py:82-91 (Transformer_enhancement.forward), SURVEY A4.
    Residuals are taken onto the *normalised* tensors; erf-GELU FFN; final Linear.
    Train mode: dropout (p = 0.8, model_Uni.py:41) on the attention weights, after the GELU and after the second
    Linear (model_Base.py:69-75)."""
    pad = ~(mask.bool())
    pt = drop.p_temporal if drop is not None else 0.0
    tag = tag or ("video" if mod.startswith("video") else "audio")        # dropout site names follow the tower, not the module
    for l in range(depth):
        p = f"{mod}.layers.{l}"
        x = layer_norm(x, P, p + ".0")
        x = mha(x, x, x, P, p + ".1", H, pad, drop, f"{tag}.{l}.attn", pt) + x
        x = layer_norm(x, P, p + ".2")
        h = _drop(drop, gelu_erf(linear(x, P, p + ".3.0")), f"{tag}.{l}.ffn_act", pt)
        x = _drop(drop, linear(h, P, p + ".3.3"), f"{tag}.{l}.ffn_out", pt) + x
    return linear(x, P, mod + ".final_linear")


def embedding_net(x: Tensor, P, mod: str, train: bool = False, updates: Optional[dict] = None) -> Tensor:
    """reference: model/model_Base.py:216-249 (hidden_size = 1024, use_bn): Linear - BatchNorm1d - ReLU - Linear - BatchNorm1d -
    ReLU - Linear.  BatchNorm1d gets [B, T, F], so its channel axis is the token position t.  Eval mode: the per-position affine
    (y - mean[t]) / sqrt(var[t] + 1e-5) * weight[t] + bias[t] of the running statistics.  Train mode: the statistics of position
    t are those of the batch's B * F values there (biased variance; padded samples count like any other), and the running
    buffers move by `momentum` (0.1 for net.1, 0.99 for net.4: model_Base.py:224,228) towards the mean / UNBIASED variance --
    the new buffer values are stored in `updates` (torch.nn.BatchNorm1d semantics)."""
    def bn(y, name, momentum):
        w, b = P[name + ".weight"][None, :, None], P[name + ".bias"][None, :, None]
        if not train:
            sc = 1.0 / torch.sqrt(P[name + ".running_var"] + 1e-5)
            return (y - P[name + ".running_mean"][None, :, None]) * sc[None, :, None] * w + b
        n = y.shape[0] * y.shape[2]
        mean = y.mean(dim=(0, 2))
        var = ((y - mean[None, :, None]) ** 2).mean(dim=(0, 2))
        if updates is not None:
            with torch.no_grad():
                updates[name + ".running_mean"] = (1 - momentum) * P[name + ".running_mean"] + momentum * mean
                updates[name + ".running_var"] = (1 - momentum) * P[name + ".running_var"] + momentum * var * (n / max(n - 1, 1))
                updates[name + ".num_batches_tracked"] = P[name + ".num_batches_tracked"] + 1
        return (y - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5) * w + b
    h = torch.relu(bn(linear(x, P, mod + ".net.0"), mod + ".net.1", 0.1))
    h = torch.relu(bn(linear(h, P, mod + ".net.3"), mod + ".net.4", 0.99))
    return linear(h, P, mod + ".net.6")


def encode_features(feats: Tensor, mask: Tensor, P, cfg, which: str, drop: Optional[Drop] = None, train: Optional[bool] = None,
                    updates: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """reference: model/model_Base.py:544-581 (video) / :583-617 (audio) with
    temporal_transformer :520-542.  Returns (local_feats [B,T,D], global_feats [B,D]).
    Variants: one shared temporal block (transformer_is_share), a learned CLS token whose output is the clip vector
    (with_cls_token), the EmbeddingNet aggregator instead of the temporal block (agg_module = "mlp"; train = batch statistics,
    default: whenever a dropout source is given, i.e. the reference's model.train())."""
    proj, mod, pe, depth = (("vit_proj", "video_transformer", "video_position_embedding.pe",
                             cfg.video_transformer_depth) if which == "video" else
                            ("ast_proj", "audio_transformer", "audio_position_embedding.pe",
                             cfg.audio_transformer_depth))
    if cfg.transformer_is_share and cfg.video_transformer_depth == cfg.audio_transformer_depth and depth > 0:
        mod = "share_transformer"
    valid = (mask != 0).unsqueeze(-1)
    x = feats * valid                                            # masked_fill(mask==0, 0)
    x = linear(x, P, proj)
    if cfg.with_act_after_proj:
        x = quick_gelu(x)
    if cfg.agg_module == "mlp":
        x = embedding_net(x, P, "Video_encoder_projection" if which == "video" else "Music_encoder_projection",
                          train=(drop is not None) if train is None else train, updates=updates)
        x = x * valid
    elif depth > 0:
        if cfg.with_cls_token:                                   # model_Base.py:527-530: token first, its mask entry is 1
            tok = P[("video" if which == "video" else "audio") + "_cls_token"].to(x.dtype)
            x = torch.cat([tok.expand(x.shape[0], -1, -1), x], dim=1)
            mask = torch.cat([torch.ones_like(mask[:, :1]), mask], dim=1)
            valid = (mask != 0).unsqueeze(-1)
        T = x.shape[1]
        table = P[pe]
        if table.shape[1] < T:
            raise ValueError(f"{pe} holds {table.shape[1]} positions < T={T} (model_Base.py:533)")
        x = x + table[:, :T]
        x = temporal_block(x, mask, P, mod, depth, cfg.SA_temporal_heads, drop, tag=which)
        x = x * valid
        if cfg.with_cls_token:                                   # model_Base.py:572-574: the token's output, not the mean
            return x[:, 1:], l2_normalize(x[:, 0])
    g = x.sum(1) / mask.sum(1, keepdim=True)
    return x, l2_normalize(g)


# ------------------------------------------------------- K5/K6: X-Pool + similarities
def xpool(video_embeds: Tensor, seg_embeds: Tensor, seg_masks: Optional[Tensor], P,
          xa: str = "video_guided_to_music_pooling_cross_transformer", drop: Optional[Drop] = None) -> Tensor:
    """reference: modules/transformer.py:156-180 with :87-123 (masked) / :27-70 (unmasked).
    video_embeds [Nv,D], seg_embeds [Nm,S,D], seg_masks [Nm,S] -> pooled [Nm,Nv,D]  (SURVEY A5)."""
    D = video_embeds.shape[-1]
    v = layer_norm(video_embeds, P, xa + ".layer_norm1")
    s = layer_norm(seg_embeds, P, xa + ".layer_norm1")
    q = linear(v, P, xa + ".cross_attn.q_proj")                  # [Nv,D]
    k = linear(s, P, xa + ".cross_attn.k_proj")                  # [Nm,S,D]
    u = linear(s, P, xa + ".cross_attn.v_proj")
    logits = torch.einsum("nd,msd->mns", q, k) / math.sqrt(D)    # [Nm,Nv,S]
    if seg_masks is not None:
        logits = logits.masked_fill((seg_masks == 0)[:, None, :], float("-inf"))
    a = torch.softmax(logits, dim=-1)
    o = torch.einsum("mns,msd->mnd", a, u)
    o = linear(o, P, xa + ".cross_attn.out_proj")
    o = layer_norm(o, P, xa + ".layer_norm2")                    # no residual (:172-174)
    lin = _drop(drop, linear(o, P, xa + ".linear_proj"), "xa.linear_out", drop.p_xpool if drop is not None else 0.0)
    o = layer_norm(o + lin, P, xa + ".layer_norm3")                # dropout 0.3 on linear_out only (:133,:177)
    return o


def sim_music_pooling(video_embeds: Tensor, pooled: Tensor) -> Tensor:
    """reference: modules/metrics.py:10-24.  [Nv,D],[Nm,Nv,D] -> [Nv,Nm]; plain norm division."""
    v = video_embeds / video_embeds.norm(dim=-1, keepdim=True)
    p = pooled / pooled.norm(dim=-1, keepdim=True)
    return torch.einsum("nd,mnd->nm", v, p)


def sim_video_pooling(pooled_v: Tensor, music_embeds: Tensor) -> Tensor:
    """reference: modules/metrics.py:26-41.  [Nv,Nm,D],[Nm,D] -> [Nv,Nm]."""
    p = pooled_v / pooled_v.norm(dim=-1, keepdim=True)
    m = music_embeds / music_embeds.norm(dim=-1, keepdim=True)
    return torch.einsum("nmd,md->nm", p, m)


def cos_sim(x: Tensor, y: Tensor) -> Tensor:
    """reference: modules/loss.py:52-56 (cal_distance, "COS")."""
    return (x / x.norm(dim=1, keepdim=True)) @ (y / y.norm(dim=1, keepdim=True)).t()


def clip_loss(sims: Tensor, logit_scale: Tensor) -> Tensor:
    """reference: modules/loss.py:5-24; InfoNCELoss(audio_id=None) :116-122 is the same value."""
    z = sims * logit_scale.exp()
    row = torch.diagonal(torch.log_softmax(z, dim=1))
    col = torch.diagonal(torch.log_softmax(z, dim=0))
    return (-(row.mean()) - col.mean()) / 2.0


def info_nce_same_music(sims: Tensor, logit_scale: Tensor, music_ids: Sequence) -> Tensor:
    """reference: modules/loss.py:90-114 (training, ignore_same_music == 0): in the
    video->music direction other videos of the same track are dropped from the negatives."""
    z = sims * logit_scale.exp()
    n = z.shape[0]
    ids = list(music_ids)
    loss = z.new_zeros(())
    for i in range(n):
        keep = [j for j in range(n) if ids[j] != ids[i]]
        row = torch.cat([z[i, i:i + 1], z[i, keep]])
        loss = loss - torch.log_softmax(row, dim=0)[0]
    v2a = loss / n
    a2v = -torch.diagonal(torch.log_softmax(z.t(), dim=1)).mean()
    return (v2a + a2v) / 2


# ------------------------------------------------------------------- K8: sine PE (DETR)
def sine_position_embedding(mask: Tensor, D: int, temperature: float = 10000.0) -> Tensor:
    """reference: music_detr/position_encoding.py:51-71 (normalize=True, scale=2pi), SURVEY A7."""
    c = mask.cumsum(1, dtype=torch.float32)
    x = c / (c[:, -1:] + 1e-6) * (2 * math.pi)
    i = torch.arange(D, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / D)
    ang = x[:, :, None] / dim_t
    out = torch.empty_like(ang)
    out[:, :, 0::2] = ang[:, :, 0::2].sin()
    out[:, :, 1::2] = ang[:, :, 1::2].cos()
    return out


def sine_pe_dim_t(D: int, temperature: float = 10000.0) -> Tensor:
    i = torch.arange(D, dtype=torch.float32)
    return temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / D)


# ------------------------------------------------------------------ K9/K10: DETR stack
def detr_encoder_layer(x: Tensor, pos: Tensor, pad: Tensor, P, p: str, H: int, drop: Optional[Drop] = None, tag: str = "") -> Tensor:
    """reference: music_detr/transformer.py:191-210 (forward_post, ReLU FFN), SURVEY A8.
    Train mode: attention-weight dropout, dropout1 on the attention branch, dropout inside the FFN, dropout2 (:153-162)."""
    pd = drop.p_detr if drop is not None else 0.0
    qk = x + pos
    a = _drop(drop, mha(qk, qk, x, P, p + ".self_attn", H, pad, drop, tag + ".attn", pd), tag + ".drop1", pd)
    x = layer_norm(x + a, P, p + ".norm1")
    h = _drop(drop, torch.relu(linear(x, P, p + ".linear1")), tag + ".ffn_act", pd)
    return layer_norm(x + _drop(drop, linear(h, P, p + ".linear2"), tag + ".drop2", pd), P, p + ".norm2")


def detr_encoder_layer_pre(x: Tensor, pos: Tensor, pad: Tensor, P, p: str, H: int, drop: Optional[Drop] = None, tag: str = "") -> Tensor:
    """reference: music_detr/transformer.py:170-189 (forward_pre): the norms sit in front of the attention and of the FFN, the residual stream
    is never normalised inside the layer (TransformerEncoder applies one more norm behind the last layer, :33-35,107-108)."""
    pd = drop.p_detr if drop is not None else 0.0
    n1 = layer_norm(x, P, p + ".norm1")
    qk = n1 + pos
    x = x + _drop(drop, mha(qk, qk, n1, P, p + ".self_attn", H, pad, drop, tag + ".attn", pd), tag + ".drop1", pd)
    n2 = layer_norm(x, P, p + ".norm2")
    h = _drop(drop, torch.relu(linear(n2, P, p + ".linear1")), tag + ".ffn_act", pd)
    return x + _drop(drop, linear(h, P, p + ".linear2"), tag + ".drop2", pd)


def detr_decoder_layer_pre(t: Tensor, qp: Tensor, mem: Tensor, pos: Tensor, pad: Tensor, P, p: str, H: int,
                           drop: Optional[Drop] = None, tag: str = "") -> Tensor:
    """reference: music_detr/transformer.py:246-271 (forward_pre; self-attention always runs here, whatever decoder_SA says)."""
    pd = drop.p_detr if drop is not None else 0.0
    n1 = layer_norm(t, P, p + ".norm1")
    q = n1 + qp
    t = t + _drop(drop, mha(q, q, n1, P, p + ".self_attn", H, None, drop, tag + ".sa_attn", pd), tag + ".drop1", pd)
    n2 = layer_norm(t, P, p + ".norm2")
    c = mha(n2 + qp, mem + pos, mem, P, p + ".multihead_attn", H, pad, drop, tag + ".ca_attn", pd)
    t = t + _drop(drop, c, tag + ".drop2", pd)
    n3 = layer_norm(t, P, p + ".norm3")
    h = _drop(drop, torch.relu(linear(n3, P, p + ".linear1")), tag + ".ffn_act", pd)
    return t + _drop(drop, linear(h, P, p + ".linear2"), tag + ".drop3", pd)


def detr_decoder_layer(t: Tensor, qp: Tensor, mem: Tensor, pos: Tensor, pad: Tensor, P, p: str, H: int,
                       drop: Optional[Drop] = None, tag: str = "") -> Tensor:
    """reference: music_detr/transformer.py:273-307 (forward_post; the self-attention
    branch always runs because build_transformer never forwards args, :325-335), SURVEY A9.
    Train mode: dropout on both attentions' weights, dropout1/2/3 on the three branches, dropout inside the FFN (:229-241)."""
    pd = drop.p_detr if drop is not None else 0.0
    q = t + qp
    t = layer_norm(t + _drop(drop, mha(q, q, t, P, p + ".self_attn", H, None, drop, tag + ".sa_attn", pd), tag + ".drop1", pd),
                   P, p + ".norm1")
    c = mha(t + qp, mem + pos, mem, P, p + ".multihead_attn", H, pad, drop, tag + ".ca_attn", pd)
    t = layer_norm(t + _drop(drop, c, tag + ".drop2", pd), P, p + ".norm2")
    h = _drop(drop, torch.relu(linear(t, P, p + ".linear1")), tag + ".ffn_act", pd)
    return layer_norm(t + _drop(drop, linear(h, P, p + ".linear2"), tag + ".drop3", pd), P, p + ".norm3")


def detr_transformer(src: Tensor, mask: Tensor, pos: Tensor, target: Optional[Tensor], P, cfg,
                     drop: Optional[Drop] = None) -> Tuple[Tensor, Tensor]:
    """reference: music_detr/transformer.py:51-81, :92-107, :119-145.
    src [B,L,D], mask [B,L] (1 valid), target [B,Q,D] -> hs [dec,B,Q,D], memory [B,L,D]."""
    pad = ~(mask.bool())
    H = cfg.detr_nheads
    mem = src
    pre = bool(getattr(cfg, "detr_pre_norm", False))          # normalize_before (music_detr/transformer.py:325-335 <- args.detr_pre_norm)
    enc_layer, dec_layer = (detr_encoder_layer_pre, detr_decoder_layer_pre) if pre else (detr_encoder_layer, detr_decoder_layer)
    for l in range(cfg.detr_enc_layers):
        mem = enc_layer(mem, pos, pad, P, f"detr_transformer.encoder.layers.{l}", H, drop, f"enc.{l}")
    if pre and cfg.detr_enc_layers > 0:
        mem = layer_norm(mem, P, "detr_transformer.encoder.norm")
    B = src.shape[0]
    qp = P["decoder_query_embed.weight"][None].expand(B, -1, -1)
    t = torch.zeros_like(qp) if target is None else target
    hs = []
    for l in range(cfg.detr_dec_layers):
        t = dec_layer(t, qp, mem, pos, pad, P, f"detr_transformer.decoder.layers.{l}", H, drop, f"dec.{l}")
        hs.append(layer_norm(t, P, "detr_transformer.decoder.norm"))
    return torch.stack(hs), mem


# ------------------------------------------------------------------- a17: CA fusion
def ca_fusion(query: Tensor, context: Tensor, q_mask: Tensor, kv_mask: Tensor, P, cfg, drop: Optional[Drop] = None) -> Tensor:
    """reference: model/model_Base.py:194-213 + :130-167 + :22-45 (depth 1), SURVEY a17.
    kv-mask before the softmax, q-mask after it; bias-free q/kv projections.  Train mode: dropout (p = 0.8, model_Uni.py:41)
    after the attention's output Linear (:113), after the GELU and after the second FFN Linear (:28-30)."""
    pc = drop.p_temporal if drop is not None else 0.0
    ca = "video_music_fusion_cross_transformer"
    H, dh = cfg.ca_heads, cfg.ca_dim_head
    B, Lq, D = query.shape
    Lk = context.shape[1]
    nx = layer_norm(query, P, ca + ".attention_query_layer_norms.0")
    nc = layer_norm(context, P, ca + ".attention_context_layer_norms.0")
    q = linear(nx, P, ca + ".layers.0.0.to_q", bias=False).view(B, Lq, H, dh).transpose(1, 2)
    kv = linear(nc, P, ca + ".layers.0.0.to_kv", bias=False)
    k = kv[..., :H * dh].view(B, Lk, H, dh).transpose(1, 2)
    v = kv[..., H * dh:].view(B, Lk, H, dh).transpose(1, 2)
    dots = (q @ k.transpose(-1, -2)) * dh ** -0.5
    dots = dots.masked_fill((kv_mask == 0)[:, None, None, :], float("-inf"))
    attn = torch.softmax(dots, dim=-1).masked_fill((q_mask == 0)[:, None, :, None], 0)
    o = (attn @ v).transpose(1, 2).reshape(B, Lq, H * dh)
    x = _drop(drop, linear(o, P, ca + ".layers.0.0.to_out.0"), "ca.attn_out", pc) + query
    nx = layer_norm(x, P, ca + ".ff_layer_norms.0")
    h = _drop(drop, gelu_erf(linear(nx, P, ca + ".layers.0.1.net.0")), "ca.ffn_act", pc)
    x = _drop(drop, linear(h, P, ca + ".layers.0.1.net.3"), "ca.ffn_out", pc) + x
    return linear(x, P, ca + ".final_linear")


# --------------------------------------------------------------------- span algebra
def span_cw_to_se(cw: Tensor) -> Tensor:
    """reference: music_detr/span_utils.py:15-24."""
    return torch.stack([cw[:, 0] - 0.5 * cw[:, 1], cw[:, 0] + 0.5 * cw[:, 1]], dim=-1)


def temporal_iou(a: Tensor, b: Tensor) -> Tuple[Tensor, Tensor]:
    """reference: music_detr/span_utils.py:39-66.  [N,2],[M,2] (start,end) -> iou [N,M], union [N,M]."""
    inter = (torch.min(a[:, None, 1], b[None, :, 1]) - torch.max(a[:, None, 0], b[None, :, 0])).clamp(min=0)
    union = (a[:, 1] - a[:, 0])[:, None] + (b[:, 1] - b[:, 0])[None, :] - inter
    return inter / union, union


def generalized_temporal_iou(a: Tensor, b: Tensor) -> Tensor:
    """reference: music_detr/span_utils.py:86-115."""
    a, b = a.float(), b.float()
    iou, union = temporal_iou(a, b)
    enc = (torch.max(a[:, None, 1], b[None, :, 1]) - torch.min(a[:, None, 0], b[None, :, 0])).clamp(min=0)
    return iou - (enc - union) / enc


# ------------------------------------------------------------ K12/K13: Hungarian matcher
def lsap(cost: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Rectangular linear sum assignment -- restatement of the algorithm behind
    scipy.optimize.linear_sum_assignment (Crouse 2016, "On implementing 2D rectangular
    assignment algorithms"; SciPy's rectangular_lsap), which the reference calls at
    music_detr/matcher.py:91.  float64 costs; returns row-sorted (row_ind, col_ind);
    among equal shortest-path costs prefers an unassigned column, then the lowest column
    index (the candidate list is filled in reverse so the first minimum met is the lowest
    index); NaN or -inf entries and infeasible matrices raise ValueError like SciPy."""
    c = np.array(cost, dtype=np.float64, copy=True)
    if c.ndim != 2:
        raise ValueError("expected a matrix (2-D array), got a %r array" % (c.shape,))
    nr, nc = c.shape
    if nr == 0 or nc == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    transposed = nc < nr
    if transposed:
        c = np.ascontiguousarray(c.T)
        nr, nc = nc, nr
    if np.isnan(c).any() or np.isneginf(c).any():
        raise ValueError("matrix contains invalid numeric entries")
    u = np.zeros(nr); v = np.zeros(nc)
    col4row = -np.ones(nr, dtype=np.int64)
    row4col = -np.ones(nc, dtype=np.int64)
    path = -np.ones(nc, dtype=np.int64)
    for cur in range(nr):
        spc = np.full(nc, np.inf)
        SR = np.zeros(nr, dtype=bool); SC = np.zeros(nc, dtype=bool)
        remaining = [nc - 1 - it for it in range(nc)]
        min_val = 0.0
        i = cur
        sink = -1
        while sink == -1:
            index = -1
            lowest = np.inf
            SR[i] = True
            for it, j in enumerate(remaining):
                r = min_val + c[i, j] - u[i] - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            min_val = lowest
            if min_val == np.inf:
                raise ValueError("cost matrix is infeasible")
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            remaining[index] = remaining[-1]
            remaining.pop()
        u[cur] += min_val
        for r_ in range(nr):
            if SR[r_] and r_ != cur:
                u[r_] += min_val - spc[col4row[r_]]
        for j_ in range(nc):
            if SC[j_]:
                v[j_] -= min_val - spc[j_]
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    if transposed:
        order = np.argsort(col4row, kind="stable")
        return col4row[order].astype(np.int64), order.astype(np.int64)
    return np.arange(nr, dtype=np.int64), col4row.astype(np.int64)


def matcher_cost(pred_logits: Tensor, pred_spans: Tensor, tgt_spans: Tensor, fg: int,
                 w_span: float = 10.0, w_giou: float = 1.0, w_class: float = 4.0) -> Tensor:
    """reference: music_detr/matcher.py:58-88 for ONE sample: [Q,2],[Q,2],[G,2] -> C [Q,G] float32,
    C = w_span*L1 + w_giou*(-GIoU) + w_class*(-p_fg), evaluated left to right."""
    p = torch.softmax(pred_logits.float(), dim=-1)
    cost_class = -p[:, fg][:, None].expand(-1, tgt_spans.shape[0])
    ps, ts = pred_spans.float(), tgt_spans.float()
    cost_span = (ps[:, None, 0] - ts[None, :, 0]).abs() + (ps[:, None, 1] - ts[None, :, 1]).abs()
    cost_giou = -generalized_temporal_iou(span_cw_to_se(ps), span_cw_to_se(ts))
    return w_span * cost_span + w_giou * cost_giou + w_class * cost_class


def hungarian_match(pred_logits: Tensor, pred_spans: Tensor, targets: Tensor, fg: int) -> List[Tuple[np.ndarray, np.ndarray]]:
    """reference: music_detr/matcher.py:36-92.  [B,Q,2],[B,Q,2],[B,G,2] -> per-sample
    (pred_idx, tgt_idx) int64; zero-width targets are dropped first (:59-61) and tgt_idx
    indexes the *kept* targets of that sample."""
    out = []
    for b in range(pred_spans.shape[0]):
        keep = targets[b, :, 1] != 0
        tg = targets[b][keep]
        C = matcher_cost(pred_logits[b], pred_spans[b], tg, fg)
        out.append(lsap(C.detach().cpu().numpy()))
    return out


# ---------------------------------------------------------------- K14: set criterion
def set_criterion(outputs: dict, targets: Tensor, P, cfg, matcher=hungarian_match) -> Dict[str, Tensor]:
    """reference: music_detr/loss_detr.py:130-169 with :74-128 and misc.py:4-21."""
    fg, bg = cfg.foreground_label, cfg.background_label
    w = P["criterion.empty_weight"]

    def one(out: dict) -> Dict[str, Tensor]:
        idx = matcher(out["pred_logits"].detach(), out["pred_spans"].detach(), targets, fg)
        bi = torch.cat([torch.full((len(i),), b, dtype=torch.int64) for b, (i, _) in enumerate(idx)])
        qi = torch.cat([torch.as_tensor(i, dtype=torch.int64) for i, _ in idx])
        kept = [targets[b][targets[b, :, 1] != 0] for b in range(targets.shape[0])]
        tg = torch.cat([kept[b][torch.as_tensor(j, dtype=torch.int64)] for b, (_, j) in enumerate(idx)], dim=0)
        src = out["pred_spans"][bi, qi]
        res: Dict[str, Tensor] = {}
        if cfg.l1_loss:
            res["loss_span"] = (src - tg).abs().mean()
        res["loss_giou"] = (1 - torch.diagonal(generalized_temporal_iou(span_cw_to_se(src), span_cw_to_se(tg)))).mean()
        logits = out["pred_logits"]
        cls = torch.full(logits.shape[:2], bg, dtype=torch.int64)
        cls[bi, qi] = fg
        logp = torch.log_softmax(logits, dim=-1)
        nll = -logp.gather(-1, cls[..., None]).squeeze(-1) * w[cls]
        res["loss_label"] = nll.mean()                               # plain mean (:104-105)
        top1 = logits[bi, qi].argmax(dim=-1)
        res["class_error"] = 100 - (top1 == fg).float().sum() * (100.0 / max(len(bi), 1))
        if cfg.contrastive_align_loss or cfg.moment_loss:
            lg = torch.einsum("bmd,bnd->bmn", out["proj_queries"], out["proj_vid_mem"]).sum(2) / 0.07
            posmap = torch.zeros_like(lg, dtype=torch.bool)
            posmap[bi, qi] = True
            pos_term = lg.masked_fill(~posmap, 0).sum(1)
            res["loss_contrastive_align"] = (-pos_term / posmap.sum(1) + lg.logsumexp(1)).mean()
        return res

    losses = one({k: v for k, v in outputs.items() if k != "aux_outputs"})
    for i, aux in enumerate(outputs.get("aux_outputs", [])):
        losses.update({f"{k}_{i}": v for k, v in one(aux).items()})
    return losses


def criterion_weight_dict(cfg) -> Dict[str, float]:
    """reference: music_detr/loss_detr.py:36-45."""
    wd = {"loss_span": 4, "loss_giou": 1, "loss_label": 0.8}
    if cfg.contrastive_align_loss:
        wd["loss_contrastive_align"] = 0.2
    if cfg.aux_loss:
        base = dict(wd)
        for i in range(cfg.detr_dec_layers - 1):
            wd.update({f"{k}_{i}": v for k, v in base.items()})
    return wd


# ---------------------------------------------------------------------------- heads
def mlp3(x: Tensor, P, name: str) -> Tensor:
    """reference: music_detr/transformer.py:348-360 (3 layers, ReLU between)."""
    x = torch.relu(linear(x, P, name + ".layers.0"))
    x = torch.relu(linear(x, P, name + ".layers.1"))
    return linear(x, P, name + ".layers.2")


def calc_output(hs: Tensor, frame_feats: Tensor, music_feats: Tensor, P, cfg,
                width_proportion: Optional[Tensor] = None) -> dict:
    """reference: model/model_Uni.py:117-173.  hs [dec,B,Q,D]."""
    out: dict = {}
    cls = linear(hs, P, "class_embed")
    coord = torch.sigmoid(mlp3(hs, P, "span_embed"))
    if cfg.predict_center == 1:
        coord = torch.cat([coord, width_proportion[None].expand(coord.shape[0], -1, -1, -1)], dim=-1)
    out["pred_logits"], out["pred_spans"] = cls[-1], coord[-1]
    pq = pv = None
    if cfg.contrastive_align_loss:
        pq = l2_normalize(linear(hs, P, "contrastive_align_projection_query"))
        if cfg.audio_short_cut:
            pq = l2_normalize(pq + music_feats[:, None])
        pv = l2_normalize(linear(frame_feats, P, "contrastive_align_projection_vid"))
        out["proj_queries"], out["proj_vid_mem"] = pq[-1], pv
    if cfg.moment_loss:                                   # model_Uni.py:152-159: an extra normalised embedding of the last layer's queries
        mf = l2_normalize(mlp3(hs[-1], P, "moment_embed"))
        if cfg.audio_short_cut:
            mf = l2_normalize(mf + music_feats[:, None])
        out["moment_feats"] = mf
    if cfg.aux_loss:
        aux = [{"pred_logits": a, "pred_spans": b} for a, b in zip(cls[:-1], coord[:-1])]
        if cfg.contrastive_align_loss:
            for i, d in enumerate(pq[:-1]):
                if cfg.audio_short_cut:
                    d = l2_normalize(d + music_feats[:, None])
                aux[i].update(proj_queries=d, proj_vid_mem=pv)
        out["aux_outputs"] = aux
    return out


# --------------------------------------------------------------------- full forward
def forward(P, cfg, frame_feats, segment_feats, frame_masks, segment_masks, spans_target,
            v_duration=None, music_ids=None, is_train: bool = False, with_losses: bool = True,
            drop: Optional[Drop] = None) -> dict:
    """reference: model/model_Uni.py:177-322; eval mode (all dropouts off) unless `drop` is given (model.train()).
    All tensors stay on the autograd tape: call .backward() on the losses for reference gradients.

    Returns a flat dict: the reference's output_map entries, feat_map entries,
    `music_feats_pooled`, `sims_single`, `sims_dual`, `detr_pos`, `memory`, `hs`,
    `retrieval_loss`, `localization_loss`, `loss_dict`, `matcher_indices`."""
    dt = P["vit_proj.weight"].dtype                 # float32; float64 only when the validation script asks for it
    ff, sf = _t(frame_feats).to(dt), _t(segment_feats).to(dt)
    fm, sm = _t(frame_masks).to(dt), _t(segment_masks).to(dt)
    tg = _t(spans_target).to(dt)
    D = cfg.D
    r: dict = {}
    upd: dict = {}                                  # train-mode BatchNorm buffer updates (agg_module = "mlp")
    frame, video = encode_features(ff, fm, P, cfg, "video", drop, updates=upd)
    seg, music = encode_features(sf, sm, P, cfg, "audio", drop, updates=upd)
    r.update(frame_feats=frame, video_feats=video, segment_feats=seg, music_feats=music, buffer_updates=upd)

    pooled = None
    if "XA" in cfg.vmr_fusion and "music" in cfg.vmr_fusion:
        pooled = xpool(video, seg, sm if cfg.fusion_mask == 1 else None, P, drop=drop)
        r["music_feats_pooled"] = pooled
    pooled_v = None
    if "XA" in cfg.vmr_fusion and "video" in cfg.vmr_fusion:
        pooled_v = xpool(music, frame, fm if cfg.fusion_mask == 1 else None, P,
                         xa="music_guided_to_video_pooling_cross_transformer", drop=drop)

    if "concat" in cfg.mml_fusion:
        fus = torch.cat([frame, seg], dim=1)
        fus_mask = torch.cat([fm, sm], dim=1)
    else:                                                           # "CA" (model_Uni.py:209-212)
        fus = ca_fusion(seg, frame, sm, fm, P, cfg, drop) * (sm != 0).unsqueeze(-1)
        fus_mask = sm
    pos = sine_position_embedding(fus_mask, D)
    r["detr_pos"] = pos
    if cfg.moment_query_type == "video":
        target = video[:, None].expand(-1, cfg.num_moment_queries, -1)
    elif cfg.moment_query_type == "music":
        target = music[:, None].expand(-1, cfg.num_moment_queries, -1)
    elif cfg.moment_query_type == "xpool":
        target = pooled.mean(dim=1)[:, None].expand(-1, cfg.num_moment_queries, -1)
    else:
        target = None
    hs, mem = detr_transformer(fus, fus_mask, pos, target, P, cfg, drop)
    r["hs"], r["memory"] = hs, mem

    # retrieval loss (model_Uni.py:236-275)
    ls = P["logit_scale"]
    r["sims_dual"] = cos_sim(video, music)
    if pooled is not None:
        r["sims_single"] = sim_music_pooling(video, pooled)
    if with_losses:
        if cfg.vmr_loss == "dual":
            r["retrieval_loss"] = clip_loss(r["sims_dual"], ls) * cfg.dual_single_loss_weight
        elif cfg.vmr_loss == "single" and "XA" in cfg.vmr_fusion:
            single = torch.zeros_like(r["sims_dual"])
            if pooled is not None:
                single = single + r["sims_single"]
            if pooled_v is not None:
                r["sims_video_pooling"] = sim_video_pooling(pooled_v, music)
                single = single + r["sims_video_pooling"]
            r["retrieval_loss"] = clip_loss(single, ls) * cfg.dual_single_loss_weight
        elif cfg.vmr_loss == "dual_single_loss_fuse" and "XA" in cfg.vmr_fusion:
            # the reference calls InfoNCELoss(..., audio_id=None, ...) here (model_Uni.py:255), so its same-music branch
            # (modules/loss.py:90-114, restated as info_nce_same_music above) is never taken whatever --ignore_same_music says
            dual = clip_loss(r["sims_dual"], ls)
            r["retrieval_loss"] = dual + clip_loss(r["sims_single"], ls)
        elif cfg.vmr_loss == "dual_single_sim_fuse" and "XA" in cfg.vmr_fusion:
            r["retrieval_loss"] = clip_loss(r["sims_dual"] + r["sims_single"], ls) * cfg.dual_single_loss_weight
        elif cfg.vmr_loss == "dual_single_feature_fuse" and "XA" in cfg.vmr_fusion:
            fused = (pooled + music[:, None]) * 0.5
            r["retrieval_loss"] = clip_loss(sim_music_pooling(video, fused), ls) * cfg.dual_single_loss_weight
        else:
            raise ValueError(f"Error: vmr_loss={cfg.vmr_loss} and vmr_fusion={cfg.vmr_fusion} is not supported in VMR_model")

    wp = None
    if cfg.predict_center == 1:
        wp = (_t(v_duration).to(dt) / cfg.max_m_duration)[:, None, None].expand(-1, cfg.num_moment_queries, -1)
    if "regression" in cfg.mml_localization:
        # model_Uni.py:228-232,290-300: mean of the encoder memory over ALL L positions' values / number of valid ones
        # (padded positions are not masked out of the sum), 3-layer MLP, sigmoid; L1 * 20
        fusion = mem.sum(dim=1) / fus_mask.sum(dim=1, keepdim=True)
        coord = torch.sigmoid(mlp3(fusion, P, "reg_mlp"))[:, None, :]
        if cfg.predict_center == 1:
            coord = torch.cat([coord, wp], dim=-1)
        r["pred_spans"] = coord
        if with_losses:
            assert coord.shape == tg.shape, (coord.shape, tg.shape)
            l1 = (coord - tg).abs().mean()
            r["loss_dict"] = {"loss_span": l1, "loss_giou": 0, "loss_label": 0, "class_error": 0}
            r["localization_loss"] = l1 * 20
        return r
    out = calc_output(hs, frame, music, P, cfg, wp)
    r.update({k: v for k, v in out.items()})
    if with_losses:
        r["matcher_indices"] = hungarian_match(out["pred_logits"], out["pred_spans"], tg, cfg.foreground_label)
        ld = set_criterion(out, tg, P, cfg)
        wd = criterion_weight_dict(cfg)
        r["loss_dict"] = ld
        r["localization_loss"] = sum(ld[k] * wd[k] for k in ld if k in wd)
    return r


# ------------------------------------------------------------ retrieval (test-MaDe.py)
def retrieval_sim_matrix(P, cfg, video_embeds, segment_embeds, segment_masks, music_embeds,
                         chunk_v: int = 512) -> Tensor:
    """reference: test-MaDe.py:386-403 (vmr_loss dual_single_loss_fuse / _sim_fuse):
    sim[Nv,Nm] = sim_matrix_music_pooling(v, XA(v, seg, mask)) + cos(v, m).
    Evaluated in video chunks (pairs are independent), since the reference's un-chunked
    [Nm,Nv,D] tensor does not fit at dataset scale."""
    v, s = _t(video_embeds).float(), _t(segment_embeds).float()
    sm, m = _t(segment_masks).float(), _t(music_embeds).float()
    rows = []
    for a in range(0, v.shape[0], chunk_v):
        vc = v[a:a + chunk_v]
        pooled = xpool(vc, s, sm if cfg.fusion_mask == 1 else None, P)
        rows.append(sim_music_pooling(vc, pooled) + cos_sim(vc, m))
    return torch.cat(rows, dim=0)


# ------------------------------------------------------------ evaluation metrics (utils/util_test.py)
def recall_ranks_dedup(sim: np.ndarray, music_ids: Sequence) -> np.ndarray:
    """reference: utils/util_test.py:44-70 (Recall_metrics, dedup=True): walk each row in descending similarity, count the
    distinct music ids met before the ground-truth id."""
    sim = np.asarray(sim)
    order = np.argsort(sim, axis=1)[:, ::-1]
    ind = []
    for i, gt in enumerate(music_ids):
        seen = set()
        for j in order[i]:
            m = music_ids[j]
            if m not in seen:
                seen.add(m)
                if m == gt:
                    ind.append(len(seen) - 1)
                    break
    return np.asarray(ind)


def recall_ranks_plain(sim: np.ndarray) -> np.ndarray:
    """reference: utils/util_test.py:71-80 (dedup=False): position of the diagonal element in the sorted row."""
    sim = np.asarray(sim)
    srt = np.sort(sim, axis=1)[:, ::-1]
    return np.argmax((srt - np.diag(sim)[:, None]) == 0, axis=1)


def top_span_iou(pred_logits: Tensor, pred_spans: Tensor, gt_moment: Tensor, m_duration: Tensor, fg: int, max_m_duration: float) -> Tensor:
    """reference: test-MaDe.py:304-313 (ranked_preds[0]) + music_detr/span_utils.py:119-170 (detr_iou, individual_IoU_tensor)."""
    out = []
    prob = torch.softmax(pred_logits.float(), dim=-1)[:, :, fg]
    for i in range(pred_spans.shape[0]):
        se = span_cw_to_se(pred_spans[i].float()) * max_m_duration
        q = max(range(se.shape[0]), key=lambda k: (float(prob[i, k]), -k))
        ps, pe = se[q, 0].clamp(min=0), se[q, 1].clamp(max=max_m_duration)
        gs, ge = gt_moment[i].reshape(-1)[0].float(), gt_moment[i].reshape(-1)[1].float()
        if gs >= ge:
            out.append(torch.tensor(0.0)); continue
        ps, pe = ps.clamp(min=0), torch.minimum(pe, m_duration[i].float())
        inter = (torch.minimum(ge, pe) - torch.maximum(gs, ps)).clamp(min=0)
        union = (pe - ps) + (ge - gs) - inter
        out.append(inter / union if union > 0 else torch.tensor(0.0))
    return torch.stack(out)
