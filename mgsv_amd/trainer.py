"""Training executor of the MaDe hot path on MI355X: forward in train() mode + hand-written backward.

Mirrors one iteration of the reference's training loop up to the gradients (reference train-MaDe.py:337-371:
`model(..., is_train=True)`, `loss = retrieval_loss*w_r + localization_loss*w_l`, `loss.backward()`): every autograd node
on the path is a kernel of libmade_hip.so (include/made_hip.h, "Training path").  PyTorch provides device memory and the
stream; nothing falls back to ATen/autograd.

Layout
  * master parameters and their gradients live in two flat f32 buffers (`flat_param`, `flat_grad`), addressed through
    per-parameter views named like the reference's state_dict -- one memset clears all gradients, one RCCL all-reduce
    averages them over data-parallel ranks, one fused optimizer launch can walk them;
  * `repack()` derives the kernel-facing copies (compute dtype W for the forward / dX products' B operand W^T);
  * activations needed by the backward are kept per layer in the compute dtype (about 10 [B*L, D] tensors per layer);
    dropout masks are never stored: they are a pure function of (seed, site, element index) (mgsv_amd/dropout.py);
  * weight gradients are A^T B products over the token axis (made_gemm_tn) accumulated with f32 atomics.

Differences from the eval executor (engine.py): the decoder cross-attention still runs in memory space, but with the
per-head products written out (q' = W_k,h^T q_h and v_h = W_v,h pooled_h as batched GEMMs) instead of pre-folded weights,
so the gradients of in_proj / out_proj come out of ordinary Linear backward steps.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib, dropout as dr, ops, ops_train as tr, tape as _tape
from .config import MadeConfig
from .engine import MadeEngine
from .ops import Seg, round_up

Tensor = torch.Tensor
XA = "video_guided_to_music_pooling_cross_transformer"


# The two-column heads (class logits, span) hand a [rows, 2] gradient back through W^T: as a K = 2 product it falls to the general
# kernel (18.5 us on the backward's critical path, twice); zero-padded to K = 64 it is a tiny-M MFMA launch like its neighbours.
HEAD_PAD = 64


class MadeTrainer(MadeEngine):
    def __init__(self, cfg: MadeConfig, state_dict: Dict[str, object], device="cuda:0", dtype: str = "f32"):
        if dtype == "f32x3":
            raise ValueError('dtype "f32x3" (split-bf16 products) is an inference mode: it meets the forward\'s 1e-4 gate, its gradients do not meet '
                             'the f32 training path\'s (measured up to 3e-2 relative per tensor); train in "f32" or "bf16"')
        super().__init__(cfg, state_dict, device, dtype)
        self._check_train_supported()
        self._init_master(state_dict)
        self.repack()
        self._tws: Dict[tuple, Dict[str, Tensor]] = {}
        self.generation = 0                                  # bumped by every in-place update of the masters (optimizer_step)
        self.seed = 0
        self.training_dropout = True
        # bf16: the flash-attention forward stores its dropout decisions (one bit per score) and the two backward kernels test the bit
        # instead of re-drawing it (made_attention's keep_bits; MADE_ATTN_BITS=0: re-draw, for A/B measurements)
        self._bits = dtype == "bf16" and _lib.variant_env("MADE_ATTN_BITS", "1") != "0"

    # ------------------------------------------------------------------ support matrix
    def _check_train_supported(self):
        c = self.cfg
        bad = []
        if "concat" not in c.mml_fusion and "CA" not in c.mml_fusion:
            bad.append(f"mml_fusion={c.mml_fusion}")
        if c.moment_query_type not in ("video", "music", "zero", "random", "xpool"):
            bad.append(f"moment_query_type={c.moment_query_type}")
        if c.moment_query_type == "xpool" and "music" not in c.vmr_fusion:
            bad.append("moment_query_type=xpool without the music-pooling tower (the reference fails there too)")
        if c.vmr_fusion not in ("XA-music", "XA-video", "XA-video-music", "XA-music-video"):
            bad.append(f"vmr_fusion={c.vmr_fusion}")
        if c.vmr_loss == "dual_single_feature_fuse" and "music" not in c.vmr_fusion:
            bad.append("vmr_loss=dual_single_feature_fuse without the music-pooling tower (the reference fails there too)")
        if c.agg_module != "mlp" and c.with_cls_token and (c.video_transformer_depth < 1 or c.audio_transformer_depth < 1):
            bad.append("with_cls_token without a temporal block")
        if "detr" not in c.mml_localization and "regression" not in c.mml_localization:
            bad.append(f"mml_localization={c.mml_localization}")
        if c.audio_short_cut and c.contrastive_align_loss and c.contrastive_hdim != c.D:
            bad.append("audio_short_cut with contrastive_dim != D (the reference's own add would not broadcast)")
        if c.D not in (128, 256, 512):
            bad.append(f"dim_input={c.D} (the kernels are built for 128, 256 and 512; 128 trains through the chain of separate launches)")
        if bad:
            raise NotImplementedError("MadeTrainer (HIP training path) does not cover yet: " + "; ".join(bad))

    # ------------------------------------------------------------------ parameters
    def _towers(self) -> List[Tuple[str, str]]:
        t = []
        if "music" in self.cfg.vmr_fusion:
            t.append(("xa", XA))
        if "video" in self.cfg.vmr_fusion:
            t.append(("xav", "music_guided_to_video_pooling_cross_transformer"))
        return t

    def _mlp_towers(self):
        c = self.cfg
        return (("video_mlp", "Video_encoder_projection", c.max_v_frames), ("audio_mlp", "Music_encoder_projection", c.max_snippet_num))

    def _table(self) -> Tuple[List[Tuple[str, object]], List[Tuple[str, object]]]:
        """(kernel key, reference name(s)) for matrices and for vectors, mirroring MadeEngine.load_state_dict."""
        c = self.cfg
        mats: List[Tuple[str, object]] = []
        vecs: List[Tuple[str, object]] = []

        def lin(key, name):
            mats.append((key + ".w", name + ".weight")); vecs.append((key + ".b", name + ".bias"))

        def ln(key, name):
            vecs.append((key + ".g", name + ".weight")); vecs.append((key + ".b", name + ".bias"))

        lin("vit_proj", "vit_proj"); lin("ast_proj", "ast_proj")
        if c.with_cls_token and c.agg_module != "mlp":        # reference model/model_Base.py:314-321: [1, 1, D] learned tokens
            vecs.append(("cls_video", "video_cls_token")); vecs.append(("cls_audio", "audio_cls_token"))
        # one block for both towers when transformer_is_share (reference model/model_Base.py:300-302,322-331): both towers' keys view the
        # same masters and the same gradient ranges; every gradient kernel accumulates atomically, so the two towers' backward passes
        # (on two streams) simply add up there
        share = bool(c.transformer_is_share) and c.video_transformer_depth == c.audio_transformer_depth and c.video_transformer_depth > 0
        if c.agg_module == "mlp":                              # EmbeddingNet aggregators (reference model/model_Base.py:216-249,357-377)
            for key, mod, _ in self._mlp_towers():
                lin(key + ".0", mod + ".net.0"); lin(key + ".3", mod + ".net.3"); lin(key + ".6", mod + ".net.6")
                for bn in ("1", "4"):
                    vecs.append((f"{key}.{bn}.g", f"{mod}.net.{bn}.weight")); vecs.append((f"{key}.{bn}.beta", f"{mod}.net.{bn}.bias"))
        for mod, depth in (() if c.agg_module == "mlp" else (("video_transformer", c.video_transformer_depth), ("audio_transformer", c.audio_transformer_depth))):
            src = "share_transformer" if share else mod
            for l in range(depth):
                p, q = f"{mod}.layers.{l}", f"{src}.layers.{l}"
                ln(p + ".ln1", q + ".0")
                mats.append((p + ".in.w", q + ".1.in_proj_weight")); vecs.append((p + ".in.b", q + ".1.in_proj_bias"))
                lin(p + ".out", q + ".1.out_proj"); ln(p + ".ln2", q + ".2")
                lin(p + ".ff1", q + ".3.0"); lin(p + ".ff2", q + ".3.3")
            lin(mod + ".final", src + ".final_linear")
        # the X-Pool towers the configuration builds (reference model/model_Uni.py:24-27): video-guided music pooling ("xa"),
        # music-guided video pooling ("xav": the same block with the roles swapped)
        for key, xa in self._towers():
            ln(key + ".ln1", xa + ".layer_norm1"); ln(key + ".ln2", xa + ".layer_norm2"); ln(key + ".ln3", xa + ".layer_norm3")
            lin(key + ".q", xa + ".cross_attn.q_proj")
            mats.append((key + ".kv.w", (xa + ".cross_attn.k_proj.weight", xa + ".cross_attn.v_proj.weight")))
            vecs.append((key + ".kv.b", (xa + ".cross_attn.k_proj.bias", xa + ".cross_attn.v_proj.bias")))
            lin(key + ".out", xa + ".cross_attn.out_proj"); lin(key + ".lin", xa + ".linear_proj")
        vecs.append(("logit_scale", "logit_scale"))
        if "CA" in c.mml_fusion:                                      # reference model/model_Base.py:99-213 (bias-free q / kv)
            ca = "video_music_fusion_cross_transformer"
            mats.append(("ca.q.w", ca + ".layers.0.0.to_q.weight")); mats.append(("ca.kv.w", ca + ".layers.0.0.to_kv.weight"))
            lin("ca.out", ca + ".layers.0.0.to_out.0"); lin("ca.ff1", ca + ".layers.0.1.net.0"); lin("ca.ff2", ca + ".layers.0.1.net.3")
            ln("ca.lnq", ca + ".attention_query_layer_norms.0"); ln("ca.lnc", ca + ".attention_context_layer_norms.0")
            ln("ca.lnf", ca + ".ff_layer_norms.0"); lin("ca.final", ca + ".final_linear")
        for l in range(c.detr_enc_layers):
            p = f"detr_transformer.encoder.layers.{l}"
            mats.append((p + ".in.w", p + ".self_attn.in_proj_weight")); vecs.append((p + ".in.b", p + ".self_attn.in_proj_bias"))
            lin(p + ".out", p + ".self_attn.out_proj"); lin(p + ".ff1", p + ".linear1"); lin(p + ".ff2", p + ".linear2")
            ln(p + ".ln1", p + ".norm1"); ln(p + ".ln2", p + ".norm2")
        if c.detr_pre_norm and c.detr_enc_layers > 0:
            ln("enc.norm", "detr_transformer.encoder.norm")        # (exists with normalize_before only, music_detr/transformer.py:34)
        for l in range(c.detr_dec_layers):
            p = f"detr_transformer.decoder.layers.{l}"
            mats.append((p + ".sa.in.w", p + ".self_attn.in_proj_weight")); vecs.append((p + ".sa.in.b", p + ".self_attn.in_proj_bias"))
            lin(p + ".sa.out", p + ".self_attn.out_proj")
            mats.append((p + ".ca.in.w", p + ".multihead_attn.in_proj_weight")); vecs.append((p + ".ca.in.b", p + ".multihead_attn.in_proj_bias"))
            lin(p + ".ca.out", p + ".multihead_attn.out_proj")
            lin(p + ".ff1", p + ".linear1"); lin(p + ".ff2", p + ".linear2")
            ln(p + ".ln1", p + ".norm1"); ln(p + ".ln2", p + ".norm2"); ln(p + ".ln3", p + ".norm3")
        ln("dec.norm", "detr_transformer.decoder.norm")
        mats.append(("query_embed", "decoder_query_embed.weight"))
        if "regression" in c.mml_localization:                # reference model/model_Uni.py:66-69: no DETR heads in this variant
            for i in range(3):
                lin(f"reg_mlp.{i}", f"reg_mlp.layers.{i}")
            return mats, vecs
        lin("class_embed", "class_embed")
        for i in range(3):
            lin(f"span_embed.{i}", f"span_embed.layers.{i}")
        if c.contrastive_align_loss:
            lin("proj_q", "contrastive_align_projection_query"); lin("proj_v", "contrastive_align_projection_vid")
        if c.moment_loss:
            for i in range(3):
                lin(f"moment_embed.{i}", f"moment_embed.layers.{i}")
        return mats, vecs

    def _init_master(self, sd: Dict[str, object]):
        """Flat f32 master / gradient buffers; k_proj|v_proj of the X-Pool block are laid out back to back so the packed
        [2D, D] projection is one view."""
        dev = self.device
        BUF = (".running_mean", ".running_var", ".num_batches_tracked")      # BatchNorm buffers (agg_module = "mlp"): state, not parameters
        names = [k for k in sd if not k.endswith(".pe") and k != "criterion.empty_weight" and not k.endswith(BUF)]
        self.buffers: Dict[str, Tensor] = {}
        for k in sd:
            if k.endswith(BUF):
                t = sd[k].detach() if isinstance(sd[k], torch.Tensor) else torch.from_numpy(np.asarray(sd[k]))
                self.buffers[k] = t.to(dev, torch.int64 if k.endswith(".num_batches_tracked") else torch.float32).clone()
        XAV = "music_guided_to_video_pooling_cross_transformer"
        pair_after = {}
        for xa in (XA, XAV):
            pair_after[xa + ".cross_attn.k_proj.weight"] = xa + ".cross_attn.v_proj.weight"
            pair_after[xa + ".cross_attn.k_proj.bias"] = xa + ".cross_attn.v_proj.bias"
        # optimizer groups of the reference (model/model_Uni.py:73-114, train-MaDe.py:262-266) laid out as contiguous ranges
        def group_of(k: str) -> int:
            if k.startswith(("vit_proj.", "ast_proj.", "video_transformer.", "audio_transformer.", "share_transformer.",
                             "video_cls_token", "audio_cls_token", "Video_encoder_projection.", "Music_encoder_projection.")):
                return 0                                      # temporal (model_Base.py:378-401: projections + SA block / CLS tokens / EmbeddingNets)
            if k.startswith((XA + ".", XAV + ".")) or k == "logit_scale":
                return 1                                      # matching (both X-Pool towers: model_Uni.py:77-86)
            if "regression" in self.cfg.mml_localization:     # model_Uni.py:92-100: the regression variant optimises the CA fusion block and
                if k.startswith(("video_music_fusion_cross_transformer.", "reg_mlp.")):       # the regression MLP only; the DETR
                    return 2                                  # transformer gets gradients but is in no optimizer group
                return 3
            if k.startswith(("detr_transformer.", "span_embed.", "class_embed.", "contrastive_align_projection_", "moment_embed.",
                             "video_music_fusion_cross_transformer.")):
                return 2                                      # detection (the CA fusion block belongs here: model_Uni.py:95-97)
            return 3                                          # not optimised (decoder_query_embed, unused variants)
        names = sorted(names, key=group_of)                   # stable: keeps the state_dict order inside a group
        order: List[str] = []
        for k in names:
            if k in pair_after.values():
                continue
            order.append(k)
            if k in pair_after:
                order.append(pair_after[k])
        shapes = {k: tuple(np.asarray(sd[k].detach().cpu() if isinstance(sd[k], torch.Tensor) else sd[k]).shape) for k in order}
        offs, off = {}, 0
        for k in order:
            offs[k] = off
            n = int(np.prod(shapes[k])) if len(shapes[k]) else 1
            off += n if k in pair_after else round_up(n, 64)
            if k in pair_after:
                assert n % 64 == 0
        self.flat_param = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.master: Dict[str, Tensor] = {}
        self.grad: Dict[str, Tensor] = {}
        for k in order:
            n = int(np.prod(shapes[k])) if len(shapes[k]) else 1
            self.master[k] = self.flat_param[offs[k]:offs[k] + n].view(shapes[k])
            self.grad[k] = self.flat_grad[offs[k]:offs[k] + n].view(shapes[k])
            v = sd[k]
            t = v.detach() if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))
            self.master[k].copy_(t.to(dev, torch.float32))
        self.param_names = order
        self._offs = offs
        self.group_ranges = []
        for gi in range(3):
            ks = [k for k in order if group_of(k) == gi]
            b = offs[ks[0]]
            last = ks[-1]
            e = offs[last] + round_up(int(np.prod(shapes[last])) if len(shapes[last]) else 1, 64)
            self.group_ranges.append((b, e))
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.opt_step = 0
        self._norm_ws = torch.zeros(2 * 4 * (1 + 1024), device=dev, dtype=torch.float32)   # (two halves: a step applied in parts)

    def _view(self, store: Dict[str, Tensor], ref) -> Tensor:
        if isinstance(ref, tuple):                       # adjacent pair -> one view over both
            a, b = store[ref[0]], store[ref[1]]
            flat = self.flat_param if store is self.master else self.flat_grad
            o = self._offs[ref[0]]
            rows = a.shape[0] + b.shape[0]
            return flat[o:o + a.numel() + b.numel()].view((rows,) + tuple(a.shape[1:]))
        return store[ref]

    def repack(self, part: Optional[str] = None):
        """Kernel-facing parameter copies from the f32 masters: W (compute dtype), W^T (for dX = dY W) -- one launch
        (made_repack) into buffers allocated once, so the pointers the kernels (and a captured hipGraph) see never change.
        Vectors (biases, LayerNorm parameters) and, in f32 mode, the matrices themselves alias the masters.
        part: None = every matrix; "early" / "rest" = the matching + detection groups' / the temporal group's (optimizer_step)."""
        import ctypes as C
        if getattr(self, "_pack_desc", None) is None:
            P, tc, dev = self.P, self.tc, self.device
            mats, vecs = self._table()
            self.G: Dict[str, Tensor] = {}
            for key, ref in vecs:
                v = self._view(self.master, ref)
                P[key] = v.view(-1) if v.dim() != 1 else v
                g = self._view(self.grad, ref)
                self.G[key] = g.view(-1) if g.dim() != 1 else g
            descs = []
            tiles = 0
            cut = self.group_ranges[0][1]                     # masters below: the temporal group (final last in the backward pass)
            part_of = []
            for key, ref in mats:
                m = self._view(self.master, ref)
                self.G[key] = self._view(self.grad, ref)
                rows, cols = m.shape
                w = None
                if tc == torch.float32:
                    P[key] = m
                else:
                    w = torch.empty(rows, cols, device=dev, dtype=tc)
                    P[key] = w
                wt = None
                if key not in ("query_embed", "vit_proj.w", "ast_proj.w"):
                    base = key[:-2] if key.endswith(".w") else key
                    wt = torch.zeros(cols, max(rows, HEAD_PAD), device=dev, dtype=tc)  # tiny heads (N = 2): reduction dim zero-padded (see HEAD_PAD)
                    P[base + ".wt"] = wt
                if w is None and wt is None:
                    continue
                d = _lib.MadeRepackDesc()
                d.src, d.w, d.wt = m.data_ptr(), (w.data_ptr() if w is not None else None), (wt.data_ptr() if wt is not None else None)
                d.rows, d.cols, d.wt_ld, d.tile_begin = rows, cols, (wt.shape[1] if wt is not None else 0), tiles
                d.dtype = ops.dt_of(wt if wt is not None else w)
                tiles += ((rows + 63) // 64) * ((cols + 63) // 64)      # (made_repack: one workgroup per 64 x 64 tile)
                descs.append(d)
                part_of.append("rest" if (m.data_ptr() - self.flat_param.data_ptr()) // 4 < cut else "early")

            def pack(ds):
                t0, out = 0, []
                for d in ds:
                    e = _lib.MadeRepackDesc()
                    C.memmove(C.byref(e), C.byref(d), C.sizeof(e))
                    e.tile_begin = t0
                    t0 += ((e.rows + 63) // 64) * ((e.cols + 63) // 64)
                    out.append(e)
                if not out:
                    return None
                arr = (_lib.MadeRepackDesc * len(out))(*out)
                return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev), len(out), t0
            # the whole set, and the two parts of a step applied in parts (optimizer_step(part=...))
            self._pack_desc = {None: pack(descs), "early": pack([d for d, q in zip(descs, part_of) if q == "early"]),
                               "rest": pack([d for d, q in zip(descs, part_of) if q == "rest"])}
        pk = self._pack_desc[part]
        if pk is not None:
            _lib.check(_lib.lib().made_repack(pk[0].data_ptr(), pk[1], pk[2], torch.cuda.current_stream().cuda_stream), "made_repack")
        if self.cfg.agg_module == "mlp" and part != "early":  # the eval path's per-position affine (MadeEngine._encode_mlp) from the current
            P = self.P                                        # BatchNorm parameters and running buffers
            for key, mod, _ in self._mlp_towers():
                for bn in ("1", "4"):
                    sc, sh = P[f"{key}.{bn}.scale"], P[f"{key}.{bn}.shift"]
                    torch.add(self.buffers[f"{mod}.net.{bn}.running_var"], 1e-5, out=sc)
                    sc.rsqrt_().mul_(P[f"{key}.{bn}.g"])
                    torch.mul(self.buffers[f"{mod}.net.{bn}.running_mean"], sc, out=sh)
                    sh.neg_().add_(P[f"{key}.{bn}.beta"])

    def optimizer_step(self, lr_temporal: float, lr_matching: float, lr_detection: float, max_grad_norm: float = 1.0,
                       betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0, device_state: Optional[Tensor] = None,
                       part: Optional[str] = None) -> None:
        """Three-group gradient clipping + Adam (reference train-MaDe.py:262-266,375-381) on the flat buffers, then repack().
        device_state (a MadeAdamDeviceState in device memory, see TrainStepGraph): the step count and the learning rates are read
        there by the kernels -- the launch sequence no longer depends on host values and can be captured.
        part: the groups are clipped and updated independently of each other, so a step may be applied in two parts --
        "early": the matching + detection groups (~85 % of the parameters), whose gradients are final before the temporal encoders'
        backward starts (backward(early_opt=...) runs this on a third stream under it); then "rest": the temporal group.
        None: everything at once."""
        import ctypes as C
        assert part in (None, "early", "rest")
        if part != "rest":
            self.opt_step += 1
            self.generation += 1
        groups = (_lib.MadeAdamGroup * 3)()
        for i, lr in enumerate((lr_temporal, lr_matching, lr_detection)):
            groups[i].begin, groups[i].end = self.group_ranges[i]
            if (part == "early" and i == 0) or (part == "rest" and i != 0):
                groups[i].end = groups[i].begin             # (an empty group is skipped)
            groups[i].lr, groups[i].max_norm = float(lr), float(max_grad_norm)
        ws = self._norm_ws[:self._norm_ws.numel() // 2] if part != "early" else self._norm_ws[self._norm_ws.numel() // 2:]
        if device_state is not None:
            _lib.check(_lib.lib().made_adam_step_device(self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                                        self.exp_avg_sq.data_ptr(), self.flat_param.numel(), groups, 3, float(betas[0]),
                                                        float(betas[1]), float(eps), device_state.data_ptr(), 0 if part == "rest" else 1,
                                                        float(grad_scale), ws.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       "made_adam_step_device")
            self.repack(part)
            return
        _lib.check(_lib.lib().made_adam_step(self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                             self.exp_avg_sq.data_ptr(), self.flat_param.numel(), groups, 3, float(betas[0]), float(betas[1]),
                                             float(eps), self.opt_step, float(grad_scale), ws.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream), "made_adam_step")
        self.repack(part)

    def train_step(self, frame_feats, segment_feats, frame_masks, segment_masks, spans_target, seed: int, lrs=(1e-4, 1e-4, 1e-4),
                   max_grad_norm: float = 1.0, w_ret: Optional[Tensor] = None, w_loc: Optional[Tensor] = None, dist=None,
                   music_ids=None, v_duration: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """One iteration of the reference's loop body (train-MaDe.py:337-381): forward, backward, (data-parallel gradient
        average: one RCCL all-reduce of the flat buffer), clip + Adam, repack."""
        self._zero_grad_in_forward = True
        try:
            out = self.forward_train(frame_feats, segment_feats, frame_masks, segment_masks, spans_target, seed=seed, music_ids=music_ids,
                                     v_duration=v_duration)
        finally:
            self._zero_grad_in_forward = False
        scale = 1.0
        if dist is not None and dist.get_world_size() > 1:
            # two buckets of the flat f32 gradient buffer (RCCL all-reduce, sum; the 1/W goes into the optimizer's grad_scale): the
            # matching + detection ranges (~85 % of the bytes) are final before the temporal encoders' backward starts and travel
            # under it; the temporal range follows at the end
            cut = self.group_ranges[0][1]
            works = []
            self.backward(w_ret, w_loc, grad_sync=lambda: works.append(dist.all_reduce(self.flat_grad[cut:], async_op=True)))
            works.append(dist.all_reduce(self.flat_grad[:cut], async_op=True))
            for w in works:
                w.wait()
            scale = 1.0 / dist.get_world_size()
        elif self._early_opt_ok():
            # (opt-in) the matching + detection groups' gradients (~85 % of the parameters) are final before the temporal encoders'
            # backward starts -- their clip + Adam + repack run on a third stream under it, the temporal group's at the end
            self.backward(w_ret, w_loc, early_opt=lambda: self.optimizer_step(*lrs, max_grad_norm=max_grad_norm, part="early"))
            self.optimizer_step(*lrs, max_grad_norm=max_grad_norm, part="rest")
            return out
        else:
            self.backward(w_ret, w_loc)
        self.optimizer_step(*lrs, max_grad_norm=max_grad_norm, grad_scale=scale)
        return out

    def _early_opt_ok(self) -> bool:
        """the step applied in two parts (see optimizer_step): MADE_EARLY_OPT=1.  Off by default -- measured on MI355X (B = 64 headline
        step): 5.65 ms in two parts against 5.62 ms in one piece; the temporal encoders' backward is slowed down by as much as the
        optimizer's tail gets shorter (both are bound by memory traffic).  The regression variant has no decoder and another group
        layout: always one piece."""
        return _lib.variant_env("MADE_EARLY_OPT", "0") == "1" and "regression" not in self.cfg.mml_localization

    def capture_train_step(self, frame_feats, segment_feats, frame_masks, segment_masks, spans_target, *, max_grad_norm: float = 1.0,
                           music_ids=None, v_duration: Optional[Tensor] = None, dist=None, mode: str = "graph") -> "TrainStepGraph":
        """The whole iteration (forward, backward, clip + Adam, repack) as hipGraph(s) over fixed buffers -- SURVEY 8(f)2.  The
        given batch only shapes the buffers and warms the kernels up: the trainer's state is the same before and after."""
        return TrainStepGraph(self, frame_feats, segment_feats, frame_masks, segment_masks, spans_target, max_grad_norm=max_grad_norm,
                              music_ids=music_ids, v_duration=v_duration, dist=dist, mode=mode)

    def state_dict_numpy(self) -> Dict[str, np.ndarray]:
        sd = {k: v.detach().cpu().numpy().copy() for k, v in self.master.items()}
        sd.update({k: v.detach().cpu().numpy().copy() for k, v in self.buffers.items()})
        return sd

    def grads_numpy(self) -> Dict[str, np.ndarray]:
        torch.cuda.synchronize()
        return {k: v.detach().cpu().numpy().copy() for k, v in self.grad.items()}

    # ------------------------------------------------------------------ dropout sites
    def _drop(self, site: str, p: float):
        if not self.training_dropout or p <= 0.0:
            return None
        seed_dev = getattr(self, "_seed_dev", None)           # set while a TrainStepGraph is captured: the kernels read the seed there
        return (seed_dev if seed_dev is not None else self.seed, dr.site_id(site), float(p))

    # ------------------------------------------------------------------ training workspace
    def _train_buffers(self, B: int, Tv: int, Ta: int) -> Dict[str, Tensor]:
        key = (B, Tv, Ta)
        ws = self._tws.get(key)
        if ws is not None:
            return ws
        c, dev, tc = self.cfg, self.device, self.tc
        concat = "concat" in c.mml_fusion
        D, L, Q, H = c.D, (Tv + Ta if concat else Ta), c.num_moment_queries, c.detr_nheads
        Ft, Fd, nd, ne = c.temporal_ffn_dim, c.detr_dim_feedforward, c.detr_dec_layers, c.detr_enc_layers
        Hh = c.SA_temporal_heads
        f32 = torch.float32

        def E(*shape, dtype=None):
            return torch.empty(shape, device=dev, dtype=dtype or tc)

        def Z(*shape, dtype=None):
            return torch.zeros(shape, device=dev, dtype=dtype or tc)

        ws = {}
        for tag, T, Kin in ((("v", Tv, c.vit_dim), ("a", Ta, c.ast_dim)) if c.agg_module == "mlp" else ()):
            r, Fh = B * T, 1024                                 # EmbeddingNet: Linear(D, 1024) - BN - ReLU - Linear(1024, D) - BN - ReLU - Linear(D, D)
            if c.with_act_after_proj:
                ws[f"{tag}.zproj"] = E(r, D)
            ws.update({f"{tag}.xin": E(r, Kin), f"{tag}.x0": E(r, D), f"{tag}.h0": E(r, Fh), f"{tag}.h1": E(r, Fh), f"{tag}.y0": E(r, D),
                       f"{tag}.y1": E(r, D), f"{tag}.mean": E(B, D, dtype=f32), f"{tag}.bn": E(4, T, dtype=f32), f"{tag}.dl": E(r, D),
                       f"{tag}.gd1": E(r, D), f"{tag}.gd2": E(r, D), f"{tag}.gh1": E(r, Fh), f"{tag}.gh2": E(r, Fh)})
        for tag, T, Kin, depth in (() if c.agg_module == "mlp" else (("v", Tv, c.vit_dim, c.video_transformer_depth), ("a", Ta, c.ast_dim, c.audio_transformer_depth))):
            T1 = T + 1 if c.with_cls_token else T               # with_cls_token: the block runs on T + 1 positions
            r, r0 = B * T1, B * T
            if c.with_act_after_proj:
                ws[f"{tag}.zproj"] = E(r0, D)
            ws.update({f"{tag}.xin": E(r0, Kin), f"{tag}.xlast": E(r, D), f"{tag}.mean": E(B, D, dtype=f32),
                       f"{tag}.dl": Z(r, D), f"{tag}.g1": E(r, D), f"{tag}.g1b": E(r, D), f"{tag}.g2": E(r, D), f"{tag}.g3": E(r, D), f"{tag}.g3b": E(r, D),
                       f"{tag}.gqkv": E(r, 3 * D), f"{tag}.gffn": E(r, Ft), f"{tag}.delta": E(B * Hh * T1, dtype=f32)})
            if c.with_cls_token:
                ws.update({f"{tag}.mask1": E(B, T1, dtype=f32), f"{tag}.rows1": (torch.empty(r, device=dev, dtype=torch.int32), torch.zeros(1, device=dev, dtype=torch.int32)),
                           f"{tag}.order1": torch.empty(B, device=dev, dtype=torch.int32), f"{tag}.y1": E(r, D), f"{tag}.dtok": Z(B, D, dtype=f32),
                           f"{tag}.dproj": E(r0, D)})
                ws[f"{tag}.mean"].fill_(1.0)                      # neutral operands of pool_bwd (dtok = 0): mask * (in1 + in2)
            for l in range(depth):
                ws.update({f"{tag}.{l}.x0": E(r, D), f"{tag}.{l}.x1": E(r, D), f"{tag}.{l}.qkv": E(r, 3 * D), f"{tag}.{l}.att": E(r, D),
                           f"{tag}.{l}.lse": E(B * Hh * T1, dtype=f32), f"{tag}.{l}.kbits": E(*ops.attention_bits_shape(B, Hh, T1, T1), dtype=torch.int32), f"{tag}.{l}.x2": E(r, D), f"{tag}.{l}.x3": E(r, D),
                           f"{tag}.{l}.z1": E(r, Ft), f"{tag}.{l}.h": E(r, Ft)})
        rows = B * L
        Lp = round_up(L, 8)
        Sp = round_up(Ta, 8)
        HQ = H * Q
        ws.update(
            # X-Pool (in-batch: every video against every track)
            xv1=E(B, D), xq=E(B, D), xs1=E(B * Ta, D), xk=E(B * Ta, D), xu=E(B * Ta, D), xo=E(B * B, D), xa2=E(B * B, D),
            xa3=E(B * B, D), xy=E(B * B, D), xg1=E(B * B, D), xg2=E(B * B, D), xg3=E(B * B, D),
            xS=E(B * B, Sp, dtype=f32), xdP=E(B * B, Sp, dtype=f32), xP=E(B * B, Sp), xdS=E(B * B, Sp), xdSt=E(B, Ta, B),
            xdkv=E(B * Ta, 2 * D), xds1=E(B * Ta, D), xdseg=E(B * Ta, D), xdq32=E(B, D, dtype=f32), xdq=E(B, D), xdv1=E(B, D),
            vn=E(B, D, dtype=f32), mn=E(B, D, dtype=f32), dvn=E(B, D, dtype=f32), dmn=E(B, D, dtype=f32),
            sims_vp_t=E(B, B, dtype=f32), dsims_st=E(B, B, dtype=f32),
            xpooled=E(B * B, D, dtype=f32), xpool_q=E(B, D, dtype=f32), dxpool_q=E(B, D, dtype=f32),
            ones_bb=torch.ones(B, B, device=self.device, dtype=f32),
            dsims_s=E(B, B, dtype=f32), dsims_d=E(B, B, dtype=f32), dsims_dt=E(B, B, dtype=f32), clip_ws=E(2 * B, dtype=f32),
            sims_both=E(B, B, dtype=f32), sd_ws=E(16 * B * B, dtype=f32),
            dclip=E(2 * B * D + B * max(c.contrastive_hdim, 1), dtype=f32),   # dvideo | dmusic | dvid_sum: zeroed by ONE fill per backward
            # DETR encoder
            e_delta=E(B * H * L, dtype=f32), eg1=E(rows, D), eg2=E(rows, D), eg2b=E(rows, D), eg3=E(rows, D), eg3b=E(rows, D), eg3c=E(rows, D),
            egqkv=E(rows, 3 * D), egffn=E(rows, Fd),
            # a second set of the four gradients a layer's weight-gradient launch reads: that launch runs on the second stream while
            # the next layer's backward (odd / even layers alternate between the sets) already writes its own
            eg2b_2=E(rows, D), eg3_2=E(rows, D), egqkv_2=E(rows, 3 * D), egffn_2=E(rows, Fd),
            dfus=E(rows, D),
            # decoder (rows = B*Q)
            s_raw=E(B * HQ, dtype=f32), dds_raw=E(B * HQ, dtype=f32), gq_raw=E(B, HQ, D),
            d_tq0=E(B * Q, D), GQ=Z(B, 2, nd, HQ, D), PdS=Z(B, 2, nd, HQ, Lp), dS_S=E(nd, B * HQ, Lp, dtype=f32), dS_dP=E(B * HQ, Lp, dtype=f32),
            dSt=E(B, L, HQ), d_ds=E(B * Q, H, dtype=f32), d_delta=E(B * H * Q, dtype=f32),
            dg1=E(B * Q, D), dg2=E(B * Q, D), dg3=E(B * Q, D), dg4=E(B * Q, D), dgqkv=E(B * Q, 3 * D), dgffn=E(B * Q, Fd),
            dgq=E(B, HQ, D), dtgt=E(B * Q, D), dhs=E(nd * B * Q, D), dgN=E(nd * B * Q, D),
            # the fused backward chain's hand-off rows (residual-stream gradients between its stages): a buffer of their own per layer and
            # stage -- see backward()
            dchain=E(nd, 6, B * Q, D),
            # heads
            h1=E(nd * B * Q, D), h2=E(nd * B * Q, D), hg1=E(nd * B * Q, D), hg2=E(nd * B * Q, D),
            dlogsp=Z(2, nd * B * Q, HEAD_PAD, dtype=f32), dlogsp_c=Z(2, nd * B * Q, HEAD_PAD),       # (one buffer per dtype: one cast launch for both)
        )
        ws["dlog"], ws["dsp"], ws["dlog_c"], ws["dsp_c"] = ws["dlogsp"][0], ws["dlogsp"][1], ws["dlogsp_c"][0], ws["dlogsp_c"][1]
        ws["dvideo"], ws["dmusic"] = ws["dclip"][:B * D].view(B, D), ws["dclip"][B * D:2 * B * D].view(B, D)
        if "video" in c.vmr_fusion:                             # second X-Pool tower: music vectors attend to the frames ("y" = "x" with S = T_v)
            Svp = round_up(Tv, 8)
            ws.update(
                yv1=E(B, D), yq=E(B, D), ys1=E(B * Tv, D), yk=E(B * Tv, D), yu=E(B * Tv, D), yo=E(B * B, D), ya2=E(B * B, D),
                ya3=E(B * B, D), yy=E(B * B, D), yg1=E(B * B, D), yg2=E(B * B, D), yg3=E(B * B, D),
                yS=E(B * B, Svp, dtype=f32), ydP=E(B * B, Svp, dtype=f32), yP=E(B * B, Svp), ydS=E(B * B, Svp), ydSt=E(B, Tv, B),
                ydkv=E(B * Tv, 2 * D), yds1=E(B * Tv, D), ydseg=E(B * Tv, D), ydq32=E(B, D, dtype=f32), ydq=E(B, D), ydv1=E(B, D),
                dframe_sum=E(B * Tv, D))
        for l in range(ne):
            ws.update({f"e.{l}.src": E(rows, D), f"e.{l}.srcpos": E(rows, D), f"e.{l}.qkv": E(rows, 3 * D), f"e.{l}.att": E(rows, D),
                       f"e.{l}.lse": E(B * H * L, dtype=f32), f"e.{l}.kbits": E(*ops.attention_bits_shape(B, H, L, L), dtype=torch.int32), f"e.{l}.x": E(rows, D), f"e.{l}.s1": E(rows, D), f"e.{l}.h": E(rows, Fd),
                       f"e.{l}.x2": E(rows, D)})
        ws.update(mem=E(rows, D), mempos=E(rows, D))
        if not concat:                                          # CA fusion block (query = segments, context = frames)
            inner, ra, rv = c.ca_heads * c.ca_dim_head, B * Ta, B * Tv
            ws.update(c_nx=E(ra, D), c_nc=E(rv, D), c_q=E(ra, inner), c_kv=E(rv, 2 * inner), c_att=E(ra, inner), c_lse=E(B * c.ca_heads * Ta, dtype=f32),
                      c_ax=E(ra, D), c_nf=E(ra, D), c_z1=E(ra, c.ca_ffn_dim), c_h=E(ra, c.ca_ffn_dim), c_y=E(ra, D),
                      c_g1=E(ra, D), c_g2=E(ra, D), c_g3=E(ra, D), c_gf=E(ra, c.ca_ffn_dim), c_gq=E(ra, inner), c_gkv=E(rv, 2 * inner),
                      c_gatt=E(ra, inner), c_delta=E(B * c.ca_heads * Ta, dtype=f32), c_dseg=E(ra, D), c_dframe=E(rv, D), c_gnc=E(rv, D))
        i32 = torch.int32
        ws.update(rows_v=(E(B * Tv, dtype=i32), E(1, dtype=i32)), rows_a=(E(B * Ta, dtype=i32), E(1, dtype=i32)), rows_f=(E(rows, dtype=i32), E(1, dtype=i32)),
                  order_v=E(B, dtype=i32), order_a=E(B, dtype=i32), order_f=E(B, dtype=i32))
        # decoder: saved activations and per-layer output gradients as [nd, ...] stacks (uniform layer stride), so the weight
        # gradients of all 6 layers are a handful of layer-batched products after the loop instead of 60 tiny launches inside it
        BQ = B * Q
        stacks = dict(tgt=E(nd + 1, BQ, D), tq=E(nd + 1, BQ, D), qkv=E(nd, BQ, 3 * D), att=E(nd, BQ, D), t_a=E(nd, BQ, D), t1=E(nd, BQ, D),
                      t1q=E(nd, BQ, D), qc=E(nd, BQ, D), pooled=E(nd, BQ, H * D), attc=E(nd, BQ, D), t_b=E(nd, BQ, D), t2=E(nd, BQ, D),
                      h=E(nd, BQ, Fd), t_c=E(nd, BQ, D),
                      **({"n1": E(nd, BQ, D)} if c.detr_pre_norm else {}),      # pre-norm: LN1(tgt), the self-attention's input
                      g_ffn=E(nd, BQ, D), g_z=E(nd, BQ, Fd), g_ca=E(nd, BQ, D), g_attc=E(nd, BQ, D), g_q=E(nd, B, HQ, D), g_qc=E(nd, BQ, D),
                      g_sa=E(nd, BQ, D), g_qkv=E(nd, BQ, 3 * D), dt1q=E(nd, BQ, D))
        ws["dstack"] = stacks
        ws["s_stack"] = E(nd, BQ, H, dtype=f32)               # sums of the dropped cross-attention weights, one row per (layer, query)
        ws["ca_lse"] = E(nd, B * HQ, dtype=f32)               # log-sum-exp of the memory-space attention's scaled scores (fused backward)
        for l in range(nd):
            ws.update({f"d.{l}.{k}": v[l] for k, v in stacks.items() if not k.startswith("g_")})
            ws.update({f"d.{l}.lse": E(B * H * Q, dtype=f32), f"d.{l}.s": ws["s_stack"][l], f"d.{l}.t3": stacks["tgt"][l + 1]})
        if c.contrastive_align_loss:
            Dc = c.contrastive_hdim
            if c.audio_short_cut:                             # normalize(normalize(p) + music), a second time for the auxiliary layers
                ws.update(pq_n0=E(nd * B * Q, Dc, dtype=f32), pq_s1=E(nd * B * Q, Dc, dtype=f32), pq_s2=E(nd * B * Q, Dc, dtype=f32),
                          dpq_s=E(nd * B * Q, Dc, dtype=f32), dpq_s2=E(nd * B * Q, Dc, dtype=f32))
            ws.update(dpq=E(nd * B * Q, Dc, dtype=f32), dvid_sum=ws["dclip"][2 * B * D:2 * B * D + B * Dc].view(B, Dc), dpq_raw=E(nd * B * Q, Dc), dpv_raw=E(B * Tv, Dc),
                      dframe_x=E(B * Tv, D))
        self._tws[key] = ws
        return ws

    # ================================================================== forward (train mode)
    @torch.no_grad()
    def same_music_exclusion(self, music_ids) -> Optional[Tensor]:
        """[B, B] f32, 1 where another sample shares this row's music track: the negatives modules/loss.py:90-114 leaves out of
        the video -> music direction of the dual loss.  OPT-IN (pass music_ids to forward_train / train_step): the reference's
        own forward hands that loss audio_id=None (model_Uni.py:255), so the drop-in module and the drivers never use it."""
        c = self.cfg
        if music_ids is None or c.vmr_loss != "dual_single_loss_fuse":
            return None
        ids = list(music_ids)
        code = {}
        idx = torch.tensor([code.setdefault(str(i), len(code)) for i in ids])
        ex = (idx[:, None] == idx[None, :]).float()
        ex.fill_diagonal_(0.0)
        return ex.to(self.device)

    def forward_train(self, frame_feats: Tensor, segment_feats: Tensor, frame_masks: Tensor, segment_masks: Tensor,
                      spans_target: Tensor, seed: int = 0, music_ids=None, v_duration: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """reference model/model_Uni.py:177-322 under model.train(): same outputs as MadeEngine.forward plus everything
        the backward needs, kept in the training workspace."""
        c, P = self.cfg, self.P
        self._set_products()
        if c.predict_center == 1 and v_duration is None:
            raise ValueError("predict_center=1 needs v_duration (reference model/model_Uni.py:280-282)")
        self.seed = int(seed)
        B, Tv, _ = frame_feats.shape
        Ta = segment_feats.shape[1]
        concat = "concat" in c.mml_fusion
        D, L, Q, nd, H = c.D, (Tv + Ta if concat else Ta), c.num_moment_queries, c.detr_dec_layers, c.detr_nheads
        ws, tw = self._buffers(B, Tv, Ta), self._train_buffers(B, Tv, Ta)
        fm, sm = frame_masks.contiguous(), segment_masks.contiguous()
        pd = float(c.detr_dropout)
        self._shape = (B, Tv, Ta)
        self._inputs = (frame_feats.contiguous(), segment_feats.contiguous(), fm, sm, spans_target.contiguous())

        fus, fus_mask = ws["fus"], ws["fus_mask"]
        # valid-token lists: every large GEMM (forward, dX and dW) gathers the valid rows only, so padding costs nothing; issue order of
        # the attention workgroups: longest sample first (a padded batch otherwise waits on whichever long sample happens to start
        # last).  These are single-workgroup scans of ~10 us each at the very head of the step: only the audio branch's two stay on the
        # main stream, the other five (+ the position embedding) go to the second stream with the video branch.
        # The video branch (B*T_v rows: launches far smaller than the chip) runs on a second HIP stream beside the audio branch;
        # so does the X-Pool / similarity / retrieval-loss branch beside the DETR stack (joined at the end of the step)
        self._groups = {}
        cur, side = torch.cuda.current_stream(), self._side_stream()
        side.wait_stream(cur)
        self._rows = {sm.data_ptr(): ops.row_index(sm, out=tw["rows_a"])}
        self._order = {sm.data_ptr(): ops.batch_order(sm, out=tw["order_a"])}
        with torch.cuda.stream(side):
            ops.concat_cols(fm if concat else None, sm, fus_mask)
            self._rows[fm.data_ptr()] = ops.row_index(fm, out=tw["rows_v"])
            self._order[fm.data_ptr()] = ops.batch_order(fm, out=tw["order_v"])
            self._rows[fus_mask.data_ptr()] = ops.row_index(fus_mask, out=tw["rows_f"])
            self._order[fus_mask.data_ptr()] = ops.batch_order(fus_mask, out=tw["order_f"])
            pos = ops.sine_pe(fus_mask, P["dim_t"], out=ws["pos"])
            self._encode_train(self._inputs[0], fm, "video", ws, tw, 0)
        self._encode_train(self._inputs[1], sm, "audio", ws, tw, Tv)
        cur.wait_stream(side)
        rows_f = self._rows[fus_mask.data_ptr()]
        if concat:
            frame, seg = fus[:, :Tv], fus[:, Tv:]
        else:
            frame, seg = ws["frame_buf"], ws["seg_buf"]
            self._ca_fusion_train(ws, tw, frame, seg, fm, sm, B, Tv, Ta)
        self._views = (frame, seg)
        video, music = ws["video"], ws["music"]
        out: Dict[str, Tensor] = dict(video_feats=video, music_feats=music, frame_feats=frame, segment_feats=seg)

        # ---- the decoder's query side (defined here: layer 0's runs on the second stream beside the DETR encoder, see below)
        qp = P["query_embed"]
        hd = D // H
        ca_scale = 1.0 / math.sqrt(hd)
        GQ = tw.get("GQ")                                    # [B, 2, nd, H*Q, D]: part 1 holds the q' rows of every layer
        regression = "regression" in c.mml_localization

        def dec_fill_tgt() -> None:
            tgt = tw["d.0.tgt"]
            if c.moment_query_type in ("video", "music", "xpool"):
                if c.moment_query_type == "xpool":
                    cur.wait_stream(side)                    # the X-Pool branch (second stream) produces the query
                src_vec = video if c.moment_query_type == "video" else (music if c.moment_query_type == "music" else tw["xpool_q"])
                if Q == 1:
                    tr.add3(tgt, src_vec)                     # (f32 clip vector -> compute dtype; no framework kernel inside the step)
                else:
                    tgt.view(B, Q, D).copy_(src_vec[:, None, :].expand(B, Q, D))
            else:                                            # "zero" / "random": reference music_detr/transformer.py:73-74
                _tape.zero_(tgt)
            if Q > 1 and not c.detr_pre_norm:                # (a single query's q / k projections are never formed: see the loop)
                tr.add3(tw["d.0.tq"], tgt, qp, b_mod=Q * D)

        # bf16, one moment query: the chain's LayerNorms run in the prologue of the Linear that consumes them (made_dec_stage, as on
        # the eval path; dropout and the saved GEMM input in its epilogue / prologue), the value bias of the
        # memory-space attention is the per-head Linear's epilogue: 10 launches per layer instead of 14
        stage = self._dec_stage_chain()

        def dec_query_side(l: int) -> None:
            """the part of decoder layer l in front of its cross-attention: self-attention block, LayerNorm, cross-attention query and
            its per-head fold -- it reads the layer's input and the weights, nothing the DETR encoder produces"""
            p, d = f"detr_transformer.decoder.layers.{l}", f"d.{l}"
            tgt, tq = tw[d + ".tgt"], tw[d + ".tq"]
            qkv = tw[d + ".qkv"]
            Win, bin_ = P[p + ".ca.in.w"], P[p + ".ca.in.b"]
            Wt = P[p + ".ca.in.wt"]                           # [D, 3D] = in_proj^T
            if c.detr_pre_norm:
                # reference music_detr/transformer.py:246-257 (forward_pre): tgt is the un-normalised stream; the self-attention reads
                # n1 = LN1(tgt) (q = k = n1 + query_pos, v = n1) and always runs; the cross-attention's query is LN2(stream) + query_pos.
                # Buffers keep their roles: .t_a = stream after the self-attention, .t1 / .t1q = the cross-attention query's input.
                n1 = tw[d + ".n1"]
                qp_rows = qp.expand(B * Q, D) if Q == 1 else qp.repeat(B, 1)
                if Q == 1:                                    # one query, one key: the value path only (see the post-norm branch below)
                    Wsa, bsa = P[p + ".sa.in.w"], P[p + ".sa.in.b"]
                    ops.layernorm(tgt, P[p + ".ln1.g"], P[p + ".ln1.b"], out=n1)
                    ops.linear(n1, Wsa[2 * D:], bsa[2 * D:], out=tw[d + ".att"], drop=self._drop(f"dec.{l}" + ".sa_attn", pd), drop_ld=H, drop_col_div=hd)
                else:
                    ops.layernorm_add(tgt, P[p + ".ln1.g"], P[p + ".ln1.b"], qp_rows, n1, tq)
                    ops.linear(n1, P[p + ".sa.in.w"], P[p + ".sa.in.b"], A2=tq, a2_replace=True,
                               segs=[Seg(out=qkv, col_begin=0, use_a2=True), Seg(out=qkv[:, 2 * D:], col_begin=2 * D, ldo=qkv.stride(0))])
                    q3 = qkv.view(B, Q, 3 * D)
                    ops.attention(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], tw[d + ".att"].view(B, Q, D), H, lse=tw[d + ".lse"],
                                  drop=self._drop(f"dec.{l}" + ".sa_attn", pd))
                ta = ops.linear(tw[d + ".att"], P[p + ".sa.out.w"], P[p + ".sa.out.b"], R=tgt, out=tw[d + ".t_a"], drop=self._drop(f"dec.{l}" + ".drop1", pd))
                ops.layernorm_add(ta, P[p + ".ln2.g"], P[p + ".ln2.b"], qp_rows, tw[d + ".t1"], tw[d + ".t1q"])
                qc = ops.linear(tw[d + ".t1q"], Win[:D], bin_[:D], out=tw[d + ".qc"])
            elif stage:
                Wsa, bsa = P[p + ".sa.in.w"], P[p + ".sa.in.b"]
                sa_drop = dict(drop=self._drop(f"dec.{l}" + ".sa_attn", pd), drop_ld=H, drop_col_div=hd)
                if l == 0:
                    ops.dec_stage(tgt, Wsa[2 * D:], bsa[2 * D:], tw[d + ".att"], **sa_drop)
                else:                                         # the previous layer's norm 3 in the prologue: t3 = this layer's input
                    q_ = f"detr_transformer.decoder.layers.{l - 1}"
                    ops.dec_stage(tw[f"d.{l - 1}.t_c"], Wsa[2 * D:], bsa[2 * D:], tw[d + ".att"], ln=(P[q_ + ".ln3.g"], P[q_ + ".ln3.b"]),
                                  x_out=tgt, **sa_drop)
                ta = ops.linear(tw[d + ".att"], P[p + ".sa.out.w"], P[p + ".sa.out.b"], R=tgt, out=tw[d + ".t_a"], drop=self._drop(f"dec.{l}" + ".drop1", pd))
                qc = ops.dec_stage(ta, Win[:D], bin_[:D], tw[d + ".qc"], ln=(P[p + ".ln1.g"], P[p + ".ln1.b"]), add=qp,
                                   x_out=tw[d + ".t1"], a_out=tw[d + ".t1q"])
            else:
                if Q == 1:
                    # one query, one key: the softmax weight is 1, so the block is the value path; its attention-weight dropout is one
                    # draw per (sample, head) (element index (b*H + h)*1*1), and q / k get no gradient
                    # (the draw is the value Linear's epilogue: the undropped value itself is needed by nobody)
                    Wsa, bsa = P[p + ".sa.in.w"], P[p + ".sa.in.b"]
                    ops.linear(tgt, Wsa[2 * D:], bsa[2 * D:], out=tw[d + ".att"], drop=self._drop(f"dec.{l}" + ".sa_attn", pd), drop_ld=H, drop_col_div=hd)
                else:
                    ops.linear(tgt, P[p + ".sa.in.w"], P[p + ".sa.in.b"], A2=tq, a2_replace=True,
                               segs=[Seg(out=qkv, col_begin=0, use_a2=True), Seg(out=qkv[:, 2 * D:], col_begin=2 * D, ldo=qkv.stride(0))])
                    q3 = qkv.view(B, Q, 3 * D)
                    ops.attention(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], tw[d + ".att"].view(B, Q, D), H, lse=tw[d + ".lse"],
                                  drop=self._drop(f"dec.{l}" + ".sa_attn", pd))
                ta = ops.linear(tw[d + ".att"], P[p + ".sa.out.w"], P[p + ".sa.out.b"], R=tgt, out=tw[d + ".t_a"], drop=self._drop(f"dec.{l}" + ".drop1", pd))
                t1 = tw[d + ".t1"]
                ops.layernorm_add(ta, P[p + ".ln1.g"], P[p + ".ln1.b"], qp.expand(B * Q, D) if Q == 1 else qp.repeat(B, 1), t1, tw[d + ".t1q"])
                qc = ops.linear(tw[d + ".t1q"], Win[:D], bin_[:D], out=tw[d + ".qc"])
            qprime = GQ[:, 1, l]                              # [B, H*Q, D] view; q'_h = W_k,h^T qc_h  (b_k shifts all keys alike)
            ops.linear(qc[:, :hd], Wt[:, D:D + hd], None, M=B * Q, N=D, K=hd, batch=H, a_z_stride=hd, w_z_stride=hd,
                       segs=[Seg(out=qprime, ldo=D, rows_per_batch=Q, out_batch_stride=qprime.stride(0), out_z_stride=Q * D)])

        dec_early = None
        # ---- X-Pool (in-batch) + similarities + retrieval loss
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if getattr(self, "_zero_grad_in_forward", False):
                # train_step: the 120 MB fill of the gradient buffer that opens the backward pass runs here, on the second stream beside
                # the DETR encoder (the previous step's optimizer is behind us on the main stream; nothing writes a gradient before the
                # streams join at the end of the forward)
                _tape.zero_(self.flat_grad)
                self._grads_zeroed = True
            if (Q == 1 and not regression and c.moment_query_type != "xpool" and c.detr_enc_layers > 0
                    and _lib.variant_env("MADE_DEC_EARLY", "1") != "0"):
                # the query side of decoder layer 0 depends on the clip-level vector and the weights only: here, beside the DETR encoder,
                # instead of at the head of the decoder's chain of dependent launches (as MadeEngine does for the eval path).
                # (MADE_DEC_EARLY=0: at the head of the chain.  While the raw barriers of the LDS-DMA GEMM kernels lacked their
                # lgkmcnt(0) -- csrc/linear.hip, linear_ring_kernel -- this placement made one stage of the chain come out a few ulps off in
                # a fifth of the steps; 0 of 80 since: tools/race_probe3.py)
                dec_fill_tgt()
                dec_query_side(0)
                dec_early = torch.cuda.Event()
                dec_early.record(side)
            xmask = sm if c.fusion_mask == 1 else None
            if "music" in c.vmr_fusion:
                self._xpool_train(video, seg, xmask, ws, tw, B, Ta)
            if "video" in c.vmr_fusion and c.vmr_loss == "single":
                # music-guided video pooling gives sims[m, v]; the reference adds its transpose to the music-pooling similarities in
                # the `single` loss and uses it nowhere else (model_Uni.py:203,247-251)
                self._xpool_train(music, frame, fm if c.fusion_mask == 1 else None, ws, tw, B, Tv, key="xav", pre="y", sims_out=tw["sims_vp_t"])
                if "music" in c.vmr_fusion:
                    ws["sims_single"].add_(tw["sims_vp_t"].t())
                else:
                    ws["sims_single"].copy_(tw["sims_vp_t"].t())
                out["sims_video_pooling"] = tw["sims_vp_t"].t()
            ops.l2norm_rows(video, out_f32=tw["vn"]); ops.l2norm_rows(music, out_f32=tw["mn"])
            if B % 4 == 0 and B <= 256:                      # one output tile: split K over workgroups (exact-f32 MFMA either way)
                split = max(2, min(16, D // 32))
                ops.linear_splitk(tw["vn"], tw["mn"], None, tw["sd_ws"][:split * B * B], split, out=ws["sims_dual"])
            else:
                ops.linear(tw["vn"], tw["mn"], None, out=ws["sims_dual"])
            static_ex = getattr(self, "_static_exclusion", None)  # TrainStepGraph: a fixed buffer the host refills before each replay
            self._row_exclude = static_ex if static_ex is not None else self.same_music_exclusion(music_ids)
            self._retrieval_loss(ws, video, music, row_exclude=self._row_exclude,
                                 pooled=tw["xpooled"] if c.vmr_loss == "dual_single_feature_fuse" else None)
        out.update(sims_single=ws["sims_single"], sims_dual=ws["sims_dual"], retrieval_loss=ws["ret_loss"])

        # ---- DETR encoder
        rows = B * L
        fskip = fus_mask.view(-1)
        qskip = fus_mask
        regression = "regression" in c.mml_localization
        if regression:
            # the regression head sums the memory over ALL positions, padded ones included (reference model_Uni.py:229): the encoder
            # computes them too (keys stay masked)
            fskip = qskip = rows_f = None
        pos2 = pos.view(rows, D)
        src = fus.view(rows, D)
        if c.detr_enc_layers == 0:                            # no encoder: memory = the fused sequence itself
            ops.layernorm_add(src, None, None, pos2, tw["mem"], tw["mempos"], row_skip=fskip)
        else:
            srcpos = tw["e.0.srcpos"]
            if not c.detr_pre_norm:
                ops.layernorm_add(src, None, None, pos2, None, srcpos, row_skip=fskip)
        pre = bool(c.detr_pre_norm)
        xin = src                                             # pre-norm: the un-normalised residual stream entering the layer
        for l in range(c.detr_enc_layers):
            p, e = f"detr_transformer.encoder.layers.{l}", f"e.{l}"
            if pre:
                # reference music_detr/transformer.py:170-189 (forward_pre): norm 1 in front of the attention (q = k = LN1(x) + pos, v = LN1(x)), norm 2 in
                # front of the FFN, both branches added to the un-normalised stream.  Buffers keep their roles: .src / .srcpos = the attention's
                # input (+ pos), .x = stream after the attention, .s1 = the FFN's input, .x2 = stream after the FFN (= the next layer's input).
                src, srcpos = tw[e + ".src"], tw[e + ".srcpos"]
                ops.layernorm_add(xin, P[p + ".ln1.g"], P[p + ".ln1.b"], pos2, src, srcpos, row_skip=fskip)
            elif l > 0:
                src, srcpos = tw[e + ".src"], tw[e + ".srcpos"]
            qkv = tw[e + ".qkv"]
            ops.linear(src, P[p + ".in.w"], P[p + ".in.b"], A2=srcpos, a2_replace=True, rows=rows_f,
                       segs=[Seg(out=qkv, col_begin=0, use_a2=True), Seg(out=qkv[:, 2 * D:], col_begin=2 * D, ldo=qkv.stride(0))])
            q3 = qkv.view(B, L, 3 * D)
            att = tw[e + ".att"]
            ops.attention(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], att.view(B, L, D), H, key_mask=fus_mask,
                          q_skip_mask=qskip, lse=tw[e + ".lse"], drop=self._drop(f"enc.{l}" + ".attn", pd),
                          order=self._order[fus_mask.data_ptr()], keep_bits=tw[e + ".kbits"] if self._bits else None)
            x = ops.linear(att, P[p + ".out.w"], P[p + ".out.b"], R=xin if pre else src, out=tw[e + ".x"], rows=rows_f,
                           drop=self._drop(f"enc.{l}" + ".drop1", pd))
            s1 = ops.layernorm(x, P[p + (".ln2.g" if pre else ".ln1.g")], P[p + (".ln2.b" if pre else ".ln1.b")], out=tw[e + ".s1"], row_skip=fskip)
            h = ops.linear(s1, P[p + ".ff1.w"], P[p + ".ff1.b"], act=ops.ACT_RELU, out=tw[e + ".h"], rows=rows_f,
                           drop=self._drop(f"enc.{l}" + ".ffn_act", pd))
            x2 = ops.linear(h, P[p + ".ff2.w"], P[p + ".ff2.b"], R=x if pre else s1, out=tw[e + ".x2"], rows=rows_f,
                            drop=self._drop(f"enc.{l}" + ".drop2", pd))
            last = l == c.detr_enc_layers - 1
            if pre:
                xin = x2
                if last:                                      # the encoder's own norm behind the last layer (:33-35,107-108)
                    ops.layernorm_add(x2, P["enc.norm.g"], P["enc.norm.b"], pos2, tw["mem"], tw["mempos"], row_skip=fskip)
                continue
            nsrc, nsp = (tw["mem"], tw["mempos"]) if last else (tw[f"e.{l + 1}.src"], tw[f"e.{l + 1}.srcpos"])
            ops.layernorm_add(x2, P[p + ".ln2.g"], P[p + ".ln2.b"], pos2, nsrc, nsp, row_skip=fskip)
        memory, mempos = tw["mem"], tw["mempos"]
        out["memory"] = memory.view(B, L, D)

        if regression:
            return self._regression_train(out, ws, tw, memory.view(B, L, D), fus_mask, v_duration, cur, side)

        # ---- DETR decoder (memory-space cross-attention, per-head products written out)
        mem3, mempos3 = memory.view(B, L, D), mempos.view(B, L, D)
        if dec_early is None:
            dec_fill_tgt()
        hs = ws["hs"]
        GQ = tw["GQ"]                                        # [B, 2, nd, H*Q, D]: part 1 holds the q' rows of every layer
        n_split = int(_lib.variant_env("MADE_WIDE_NSPLIT", "0")) or max(1, min(8, 256 // max(B, 1)))   # few queries, long memory: keys split over workgroups (knob for measurements)
        t3_stack = tw["dstack"]["tgt"][1:]                   # [nd, B*Q, D]: slot l + 1 = layer l's output (t3)
        for l in range(nd):
            p, d = f"detr_transformer.decoder.layers.{l}", f"d.{l}"
            tgt, tq = tw[d + ".tgt"], tw[d + ".tq"]
            qkv = tw[d + ".qkv"]
            if l == 0 and dec_early is not None:
                cur.wait_event(dec_early)                      # layer 0's query side ran beside the DETR encoder (second stream)
            else:
                dec_query_side(l)
            Win, bin_ = P[p + ".ca.in.w"], P[p + ".ca.in.b"]
            qprime, t1 = GQ[:, 1, l], tw[d + ".t1"]
            pooled = tw[d + ".pooled"]                        # [B*Q, H*D]: row (b, q), head-major columns
            s_out = tw[d + ".s"] if Q == 1 else tw["s_raw"]   # the kernel numbers its rows (b, h, q); the Linears below (b, q, h)
            ops.attention_wide(qprime.view(B, H, Q, D), mempos3, mem3, pooled.view(B, Q, H, D).permute(0, 2, 1, 3), scale=ca_scale,
                               key_mask=fus_mask, drop=self._drop(f"dec.{l}" + ".ca_attn", pd), sum_out=s_out,
                               n_split=n_split, part_o=ws["part_o"], part_ml=ws["part_ml"],
                               lse_out=tw["ca_lse"][l] if stage else None)
            if Q > 1:
                tw[d + ".s"].view(B, Q, H).copy_(s_out.view(B, H, Q).permute(0, 2, 1))
            attc = tw[d + ".attc"]
            if stage:                                         # the value bias s_h * b_v,h is the per-head Linear's epilogue
                ops.linear(pooled[:, :D], Win[2 * D:2 * D + hd], bin_[2 * D:], M=B * Q, N=hd, K=D, batch=H, a_z_stride=D, w_z_stride=hd * D,
                           segs=[Seg(out=attc, ldo=D, out_z_stride=hd)], bias_row_scale=tw[d + ".s"], bias_z_stride=hd)
            else:
                ops.linear(pooled[:, :D], Win[2 * D:2 * D + hd], None, M=B * Q, N=hd, K=D, batch=H, a_z_stride=D, w_z_stride=hd * D,
                           segs=[Seg(out=attc, ldo=D, out_z_stride=hd)])
                tr.head_bias(attc, tw[d + ".s"], bin_[2 * D:], H)
            if c.detr_pre_norm:
                # (:258-271) the cross-attention and the FFN (input LN3(stream)) are added to the un-normalised stream; the layer's output is the
                # stream itself (slot l + 1 of the tgt stack; decoder.norm of it -> hs[l] after the loop, as on the post-norm path)
                tb = ops.linear(attc, P[p + ".ca.out.w"], P[p + ".ca.out.b"], R=tw[d + ".t_a"], out=tw[d + ".t_b"], drop=self._drop(f"dec.{l}" + ".drop2", pd))
                t2 = ops.layernorm(tb, P[p + ".ln3.g"], P[p + ".ln3.b"], out=tw[d + ".t2"])
                h = ops.linear(t2, P[p + ".ff1.w"], P[p + ".ff1.b"], act=ops.ACT_RELU, out=tw[d + ".h"], drop=self._drop(f"dec.{l}" + ".ffn_act", pd))
                ops.linear(h, P[p + ".ff2.w"], P[p + ".ff2.b"], R=tb, out=tw[d + ".t3"], drop=self._drop(f"dec.{l}" + ".drop3", pd))
                continue
            tb = ops.linear(attc, P[p + ".ca.out.w"], P[p + ".ca.out.b"], R=t1, out=tw[d + ".t_b"], drop=self._drop(f"dec.{l}" + ".drop2", pd))
            if stage:
                h = ops.dec_stage(tb, P[p + ".ff1.w"], P[p + ".ff1.b"], tw[d + ".h"], ln=(P[p + ".ln2.g"], P[p + ".ln2.b"]), x_out=tw[d + ".t2"],
                                  act=ops.ACT_RELU, drop=self._drop(f"dec.{l}" + ".ffn_act", pd))
                t2 = tw[d + ".t2"]
            else:
                t2 = ops.layernorm(tb, P[p + ".ln2.g"], P[p + ".ln2.b"], out=tw[d + ".t2"])
                h = ops.linear(t2, P[p + ".ff1.w"], P[p + ".ff1.b"], act=ops.ACT_RELU, out=tw[d + ".h"], drop=self._drop(f"dec.{l}" + ".ffn_act", pd))
            tcx = ops.linear(h, P[p + ".ff2.w"], P[p + ".ff2.b"], R=t2, out=tw[d + ".t_c"], drop=self._drop(f"dec.{l}" + ".drop3", pd))
            t3 = tw[d + ".t3"]                                # = the content query of layer l + 1 (slot l + 1 of the tgt stack)
            if stage:
                if l + 1 == nd:                               # (inside the chain norm 3 is the next layer's prologue; the last one has no consumer)
                    ops.layernorm(tcx, P[p + ".ln3.g"], P[p + ".ln3.b"], out=t3)
            elif l + 1 < nd:
                ops.layernorm_add(tcx, P[p + ".ln3.g"], P[p + ".ln3.b"], qp.expand(B * Q, D) if Q == 1 else qp.repeat(B, 1), t3, tw[f"d.{l + 1}.tq"])
            else:
                ops.layernorm(tcx, P[p + ".ln3.g"], P[p + ".ln3.b"], out=t3)
        # the shared output norm of every layer feeds the heads only, not the next layer: one launch over the [nd, B*Q] stack
        # of layer outputs after the chain instead of one inside every layer
        ops.layernorm(t3_stack.reshape(nd * B * Q, D), P["dec.norm.g"], P["dec.norm.b"], out=hs.view(nd * B * Q, D))
        out["hs"] = hs.view(nd, B, Q, D)

        # ---- heads
        hs2 = hs.view(nd * B * Q, D)
        logits, spans = ws["logits"], ws["spans"]
        ops.linear(hs2, P["class_embed.w"], P["class_embed.b"], out=logits.view(-1, 2))
        h1 = ops.linear(hs2, P["span_embed.0.w"], P["span_embed.0.b"], act=ops.ACT_RELU, out=tw["h1"])
        h2 = ops.linear(h1, P["span_embed.1.w"], P["span_embed.1.b"], act=ops.ACT_RELU, out=tw["h2"])
        if c.predict_center == 1:
            # the head predicts the centre only; the width is the video's share of the longest track, a constant of the batch
            # (reference model/model_Uni.py:135-136,280-282): no gradient reaches it
            ops.linear(h2, P["span_embed.2.w"], P["span_embed.2.b"], act=ops.ACT_SIGMOID, segs=[Seg(out=spans.view(-1, 2), ldo=2)])
            spans[..., 1] = (v_duration.to(self.device, torch.float32) / c.max_m_duration).view(1, B, 1)
        else:
            ops.linear(h2, P["span_embed.2.w"], P["span_embed.2.b"], act=ops.ACT_SIGMOID, out=spans.view(-1, 2))
        out.update(pred_logits=logits[-1], pred_spans=spans[-1], logits_all=logits, spans_all=spans)
        if c.moment_loss:                                    # reference model_Uni.py:152-159: outputs only, no loss reads them, so
            last = hs[nd - 1]                                # moment_embed gets no gradient (nor does the reference's)
            m1 = ops.linear(last, P["moment_embed.0.w"], P["moment_embed.0.b"], act=ops.ACT_RELU)
            m2 = ops.linear(m1, P["moment_embed.1.w"], P["moment_embed.1.b"], act=ops.ACT_RELU)
            m3 = ops.linear(m2, P["moment_embed.2.w"], P["moment_embed.2.b"], out_dtype=torch.float32)
            mf = ops.l2norm_rows(m3)
            if c.audio_short_cut:
                mq = music if Q == 1 else music[:, None, :].expand(B, Q, D).contiguous()
                tr.add3(m3, mf, mq, b_mod=B * Q * D)
                mf = ops.l2norm_rows(m3)
            out["moment_feats"] = mf.view(B, Q, D)
        pq = vid_sum = None
        if c.contrastive_align_loss:
            ops.linear(hs2, P["proj_q.w"], P["proj_q.b"], out=ws["pq_raw"])
            pq = ws["pq"]
            pq2 = pq.view(nd * B * Q, -1)
            if c.audio_short_cut:
                # reference model/model_Uni.py:142-145,166-169: normalize(normalize(proj) + music); the auxiliary layers get the
                # short-cut a second time when their output dicts are built.  Every stage is kept for the backward.
                mq = music if Q == 1 else music[:, None, :].expand(B, Q, D).contiguous()
                ops.l2norm_rows(ws["pq_raw"], out_f32=tw["pq_n0"])
                tr.add3(tw["pq_s1"], tw["pq_n0"], mq, b_mod=B * Q * D)
                ops.l2norm_rows(tw["pq_s1"], out_f32=pq2)
                if c.aux_loss and nd > 1:
                    n_aux = (nd - 1) * B * Q
                    tr.add3(tw["pq_s2"][:n_aux], pq2[:n_aux], mq, b_mod=B * Q * D)
                    ops.l2norm_rows(tw["pq_s2"][:n_aux], out_f32=pq2[:n_aux])
            else:
                ops.l2norm_rows(ws["pq_raw"], out_f32=pq2)
            self._frame_rows_linear(frame, P["proj_v.w"], P["proj_v.b"], ws["pv_raw"], B, Tv)
            pv = ws["pv"]
            ops.l2norm_rows(ws["pv_raw"], out_f32=pv.view(B * Tv, pq.shape[-1]))
            vid_sum = ops.masked_mean(pv, None, out=ws["vid_sum"])
            out.update(proj_queries=pq[-1], proj_vid_mem=pv, proj_queries_all=pq)
        tg = self._inputs[4]
        pi, ti, cnt, status, cost = ops.hungarian_match(logits.view(nd * B, Q, 2), spans.view(nd * B, Q, 2), tg, c.foreground_label)
        losses, total = ops.set_criterion(logits, spans, tg, pi, ti, cnt, pq, vid_sum, P["empty_weight"], c.foreground_label, P["crit_weights"])
        self._match = (pi, ti, cnt)
        out.update(matcher_pred_idx=pi.view(nd, B, -1), matcher_tgt_idx=ti.view(nd, B, -1), matcher_count=cnt.view(nd, B),
                   matcher_status=status, criterion_losses=losses, localization_loss=total)
        cur.wait_stream(side)
        return out

    def _opt_stream(self):
        st = getattr(self, "_opt_st", None)
        if st is None:
            st = self._opt_st = torch.cuda.Stream(device=self.device)
        return st

    def _dec_stage_chain(self) -> bool:
        """The training decoder's fused chain (made_dec_stage with the training options, in-launch merge of the memory-space attention,
        fused attention backward): bf16, one moment query, D = 256 / 512.  MADE_DEC_STAGE=0 keeps round 2's chain (A/B measurements);
        f32 and Q > 1 always take it."""
        c = self.cfg
        return (self.tc == torch.bfloat16 and c.num_moment_queries == 1 and c.D in (256, 512) and not c.detr_pre_norm
                and _lib.variant_env("MADE_DEC_STAGE", "1") != "0")

    def _encode_train(self, feats: Tensor, mask: Tensor, which: str, ws, tw, row_off: int) -> None:
        """reference model/model_Base.py:544-617 in train mode (dropout 0.8 inside the temporal block)."""
        c, P = self.cfg, self.P
        if c.agg_module == "mlp":
            return self._encode_train_mlp(feats, mask, which, ws, tw, row_off)
        B, T, Kin = feats.shape
        D, Hh = c.D, c.SA_temporal_heads
        proj, mod, pe, depth, tag = (("vit_proj", "video_transformer", "pe_video", c.video_transformer_depth, "v") if which == "video"
                                     else ("ast_proj", "audio_transformer", "pe_audio", c.audio_transformer_depth, "a"))
        cls = bool(c.with_cls_token)
        T1 = T + 1 if cls else T
        if P[pe].shape[0] < T1:
            raise ValueError(f"{which} position table holds {P[pe].shape[0]} positions < {T1}")
        nrow = B * T
        mflat = mask.reshape(-1)
        pt = dr.P_TEMPORAL
        name = "video" if which == "video" else "audio"
        act = ops.ACT_QUICKGELU if c.with_act_after_proj else ops.ACT_NONE       # reference model_Base.py:559-561
        zp = tw[tag + ".zproj"] if c.with_act_after_proj else None             # pre-activation, for the backward gate
        x = tw[f"{tag}.0.x0"]
        if cls:
            # reference model_Base.py:527-530: the learned token goes first with mask entry 1; the block then runs on T + 1 positions
            mask1 = tw[tag + ".mask1"]
            mask1[:, 0] = 1.0; mask1[:, 1:] = mask
            self._rows[mask1.data_ptr()] = ops.row_index(mask1, out=tw[tag + ".rows1"])
            self._order[mask1.data_ptr()] = ops.batch_order(mask1, out=tw[tag + ".order1"])
            x3 = x.view(B, T1, D)
            x3[:, 0] = (P["cls_" + name].float() + P[pe][0].float()).to(self.tc)
            lin_out = dict(segs=[Seg(out=x3[:, 1:], ldo=D, rows_per_batch=T, out_batch_stride=T1 * D)])
            pe_rows = P[pe][1:T1]
        else:
            mask1 = mask
            lin_out = dict(out=x, rows=self._rw(mflat))
            pe_rows = P[pe][:T]
        mflat1 = mask1.reshape(-1)
        rws = self._rw(mflat1)
        if self.tc == torch.bfloat16:
            xin = ops.cast_mask_rows(feats.view(nrow, Kin), mflat, tw[tag + ".xin"])
            ops.linear(xin, P[proj + ".w"], P[proj + ".b"], act=act, Zout=zp, R=pe_rows, r_row_mod=T, **lin_out)
        else:
            ops.linear(feats.view(nrow, Kin), P[proj + ".w"], P[proj + ".b"], a_row_mask=mflat, act=act, Zout=zp, R=pe_rows, r_row_mod=T, **lin_out)
        for l in range(depth):
            p, t = f"{mod}.layers.{l}", f"{tag}.{l}"
            x1 = ops.layernorm(x, P[p + ".ln1.g"], P[p + ".ln1.b"], out=tw[t + ".x1"], row_skip=mflat1)
            qkv = ops.linear(x1, P[p + ".in.w"], P[p + ".in.b"], out=tw[t + ".qkv"], rows=rws)
            q3 = qkv.view(B, T1, 3 * D)
            att = tw[t + ".att"]
            ops.attention(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], att.view(B, T1, D), Hh, key_mask=mask1, q_skip_mask=mask1,
                          lse=tw[t + ".lse"], drop=self._drop(f"{name}.{l}.attn", pt), order=self._order[mask1.data_ptr()],
                          keep_bits=tw[t + ".kbits"] if self._bits else None)
            x2 = ops.linear(att, P[p + ".out.w"], P[p + ".out.b"], R=x1, out=tw[t + ".x2"], rows=rws)
            x3_ = ops.layernorm(x2, P[p + ".ln2.g"], P[p + ".ln2.b"], out=tw[t + ".x3"], row_skip=mflat1)
            h = ops.linear(x3_, P[p + ".ff1.w"], P[p + ".ff1.b"], act=ops.ACT_GELU, out=tw[t + ".h"], Zout=tw[t + ".z1"], rows=rws,
                           drop=self._drop(f"{name}.{l}.ffn_act", pt))
            nxt = tw[f"{tag}.{l + 1}.x0"] if l + 1 < depth else tw[tag + ".xlast"]
            x = ops.linear(h, P[p + ".ff2.w"], P[p + ".ff2.b"], R=x3_, out=nxt, rows=rws, drop=self._drop(f"{name}.{l}.ffn_out", pt))
        if "concat" in c.mml_fusion:
            local = ws["fus"][:, row_off:row_off + T]
        else:
            local = ws["frame_buf"] if which == "video" else ws["seg_buf"]
        vec = ws["video"] if which == "video" else ws["music"]
        if cls:                                              # reference model_Base.py:572-574: clip vector = the token's output
            y3 = ops.linear(x, P[mod + ".final.w"], P[mod + ".final.b"], out_row_mask=mflat1, out=tw[tag + ".y1"]).view(B, T1, D)
            ops.l2norm_rows(y3[:, 0], out_f32=vec)
            local.copy_(y3[:, 1:])
            return
        ops.linear(x, P[mod + ".final.w"], P[mod + ".final.b"], out_row_mask=mflat, tile_skip_mask=mflat,
                   segs=[Seg(out=local, ldo=local.stride(1), rows_per_batch=T, out_batch_stride=local.stride(0))])
        ops.masked_mean(local, mask, out=tw[tag + ".mean"])
        ops.l2norm_rows(tw[tag + ".mean"], out_f32=vec)

    def _xpool_train(self, video: Tensor, seg: Tensor, seg_mask: Optional[Tensor], ws, tw, B: int, S: int, key: str = "xa", pre: str = "x",
                     sims_out: Optional[Tensor] = None) -> None:
        """reference modules/transformer.py:156-180 (+ :87-123) over the batch's B x B pairs, dropout 0.3 on linear_out.  `video` = the
        B query vectors, `seg` [B, S, D] the sequences they pool; tower "xav" (buffers "y*") is the same block with the roles swapped."""
        P, D = self.P, self.cfg.D
        skip = seg_mask.reshape(-1) if seg_mask is not None else None
        v1 = ops.layernorm(video, P[key + ".ln1.g"], P[key + ".ln1.b"], out=tw[pre + "v1"])
        q = ops.linear(v1, P[key + ".q.w"], P[key + ".q.b"], out=tw[pre + "q"])
        s1 = ops.layernorm(seg, P[key + ".ln1.g"], P[key + ".ln1.b"], out=tw[pre + "s1"], row_skip=skip)
        ops.linear(s1, P[key + ".kv.w"], P[key + ".kv.b"], rows=self._rw(skip), tile_skip_mask=skip if self._rw(skip) is None else None,
                   segs=[Seg(out=tw[pre + "k"], col_begin=0), Seg(out=tw[pre + "u"], col_begin=D)])
        # 64 videos x 64 tracks: one workgroup per track would leave three quarters of the chip idle on a kernel that streams 1 MB of
        # K / U per track at one CU's rate -- the keys are split over workgroups (up to 256 of them), a second launch merges the slices
        # (tools/xpool_qk_bench.py, profiles/r03_xpool_qk_microbench.txt: 56.8 us unsplit, 34.9 us split four ways)
        inbatch = (self.tc == torch.bfloat16 and B <= 64 and S <= 512 and S * D >= 65536 and D in (256, 512) and _lib.variant_env("MADE_XPOOL_INBATCH", "1") != "0")   # (shorter / narrower tracks: one launch of made_attention_wide is faster -- 10.9 vs 12.3 us at S = 96, D = 256)
        if inbatch:
            # round 4: scores per (track, 128 segments), then P.V per (track, 128 value columns) -- two launches of one workgroup per CU, only the
            # bf16 probabilities between them (made_xpool_inbatch; profiles/r04_*xpool_qk_microbench.txt)
            if tw.get(pre + "xib_ws") is None or tw[pre + "xib_ws"].numel() < ops.xpool_inbatch_ws_bytes(B, S):    # (one per tower: they may run on two streams)
                tw[pre + "xib_ws"] = torch.zeros(ops.xpool_inbatch_ws_bytes(B, S), device=self.device, dtype=torch.uint8)
            ops.xpool_inbatch(q, tw[pre + "k"].view(B, S, D), tw[pre + "u"].view(B, S, D), seg_mask, tw[pre + "o"].view(B, B, D),
                              scale=1.0 / math.sqrt(D), ws=tw[pre + "xib_ws"])
        xsplit = int(_lib.variant_env("MADE_XPOOL_NSPLIT", "0")) or (max(1, min(4, 256 // max(B, 1), S // 64)) if (B <= 64 and self.tc == torch.bfloat16) else 1)
        if inbatch:
            pass
        elif xsplit > 1:
            need = B * xsplit * B
            if tw.get("xpart_o") is None or tw["xpart_o"].numel() < need * D:
                tw["xpart_o"] = torch.empty(need * D, device=self.device, dtype=torch.float32)
                tw["xpart_ml"] = torch.empty(need * 4, device=self.device, dtype=torch.float32)
        if not inbatch:
            ops.attention_wide(q.view(1, B, 1, D), tw[pre + "k"].view(B, S, D), tw[pre + "u"].view(B, S, D), tw[pre + "o"].view(B, B, 1, D),
                               scale=1.0 / math.sqrt(D), key_mask=seg_mask, shared_q=True, n_split=xsplit,
                               part_o=tw.get("xpart_o") if xsplit > 1 else None, part_ml=tw.get("xpart_ml") if xsplit > 1 else None)
        a2 = ops.linear(tw[pre + "o"], P[key + ".out.w"], P[key + ".out.b"], out=tw[pre + "a2"])
        a3 = ops.layernorm(a2, P[key + ".ln2.g"], P[key + ".ln2.b"], out=tw[pre + "a3"])
        # (the oracle / reference masks name the site after the block class, not the tower)
        y = ops.linear(a3, P[key + ".lin.w"], P[key + ".lin.b"], R=a3, out=tw[pre + "y"], drop=self._drop("xa.linear_out", dr.P_XPOOL))
        want_pooled = key == "xa" and (self.cfg.moment_query_type == "xpool" or self.cfg.vmr_loss == "dual_single_feature_fuse")
        ops.xpool_tail(y, P[key + ".ln3.g"], P[key + ".ln3.b"], video, ws["sims_single"] if sims_out is None else sims_out, B, B,
                       pooled_out=tw["xpooled"] if want_pooled else None)
        if want_pooled and self.cfg.moment_query_type == "xpool":   # reference model_Uni.py:222-223: a track's pooled vectors, averaged over the videos
            ops.masked_mean(tw["xpooled"].view(B, B, D), tw["ones_bb"], out=tw["xpool_q"])       # (a NULL mask would give the sum)

    def _ca_fusion_train(self, ws, tw, frame: Tensor, seg: Tensor, fm: Tensor, sm: Tensor, B: int, Tv: int, Ta: int) -> None:
        """reference model/model_Base.py:194-213 (+ :130-167, :22-45) in train mode, then the masked_fill of model_Uni.py:211:
        pre-LN cross-attention (query = segments, context = frames, 8 heads x 128, bias-free q / kv, key mask before the softmax,
        query mask after it), residual, pre-LN GELU FFN with residual, final Linear -> ws["fus"]; dropout 0.8 at three sites."""
        c, P = self.cfg, self.P
        D, Hc = c.D, c.ca_heads
        inner = Hc * c.ca_dim_head
        pt = dr.P_TEMPORAL
        sflat, fflat = sm.reshape(-1), fm.reshape(-1)
        ra, rv = self._rw(sflat), self._rw(fflat)
        x = seg.reshape(B * Ta, D)
        nx = ops.layernorm(x, P["ca.lnq.g"], P["ca.lnq.b"], out=tw["c_nx"], row_skip=sflat)
        nc = ops.layernorm(frame.reshape(B * Tv, D), P["ca.lnc.g"], P["ca.lnc.b"], out=tw["c_nc"], row_skip=fflat)
        q = ops.linear(nx, P["ca.q.w"], None, out=tw["c_q"], rows=ra)
        kv = ops.linear(nc, P["ca.kv.w"], None, out=tw["c_kv"], rows=rv)
        kv3 = kv.view(B, Tv, 2 * inner)
        ops.attention(q.view(B, Ta, inner), kv3[:, :, :inner], kv3[:, :, inner:], tw["c_att"].view(B, Ta, inner), Hc, key_mask=fm, q_mask=sm,
                      q_skip_mask=sm, scale=c.ca_dim_head ** -0.5, lse=tw["c_lse"], order=self._order[sm.data_ptr()])
        ax = ops.linear(tw["c_att"], P["ca.out.w"], P["ca.out.b"], R=x, out=tw["c_ax"], rows=ra, drop=self._drop("ca.attn_out", pt))
        nf = ops.layernorm(ax, P["ca.lnf.g"], P["ca.lnf.b"], out=tw["c_nf"], row_skip=sflat)
        h = ops.linear(nf, P["ca.ff1.w"], P["ca.ff1.b"], act=ops.ACT_GELU, out=tw["c_h"], Zout=tw["c_z1"], rows=ra, drop=self._drop("ca.ffn_act", pt))
        y = ops.linear(h, P["ca.ff2.w"], P["ca.ff2.b"], R=ax, out=tw["c_y"], rows=ra, drop=self._drop("ca.ffn_out", pt))
        ops.linear(y, P["ca.final.w"], P["ca.final.b"], out_row_mask=sflat, tile_skip_mask=sflat, out=ws["fus"].view(B * Ta, D))

    def _ca_fusion_bwd(self, ws, tw, dfus: Tensor, frame: Tensor, seg: Tensor, fm: Tensor, sm: Tensor, B: int, Tv: int, Ta: int) -> None:
        """gradients of the CA fusion block: -> tw["c_dseg"] (w.r.t. the segment features) and tw["c_dframe"] (frame features)."""
        c, P, G = self.cfg, self.P, self.G
        D, Hc = c.D, c.ca_heads
        inner = Hc * c.ca_dim_head
        pt = dr.P_TEMPORAL
        sflat, fflat = sm.reshape(-1), fm.reshape(-1)
        g1, g2, g3, gf = tw["c_g1"], tw["c_g2"], tw["c_g3"], tw["c_gf"]
        # fus = mask(final(y)); y = ax + drop(ffn2(h)): d y raw (residual) and dropped (branch)
        dff = self._lin_bwd(dfus, tw["c_y"], "ca.final", dx_out=g1, row_mask=sflat, Zout=g2, drop=self._drop("ca.ffn_out", pt))
        dz1 = self._lin_bwd(dff, tw["c_h"], "ca.ff2", dx_out=gf, row_mask=sflat, gate=_lib.GATE_GELU_Z, G=tw["c_z1"], drop=self._drop("ca.ffn_act", pt))
        dnf = self._lin_bwd(dz1, tw["c_nf"], "ca.ff1", dx_out=g3, row_mask=sflat)
        # ax = x + drop(to_out(att)): LN_f backward + the residual d y
        tr.layernorm_bwd(tw["c_ax"], P["ca.lnf.g"], dnf, g1, dgamma=G["ca.lnf.g"], dbeta=G["ca.lnf.b"], add=g2, dx_drop=g3,
                         drop=self._drop("ca.attn_out", pt), row_skip=sflat)
        datt = self._lin_bwd(g3, tw["c_att"], "ca.out", dx_out=tw["c_gatt"], row_mask=sflat)
        q, kv = tw["c_q"], tw["c_kv"]
        kv3, gkv3 = kv.view(B, Tv, 2 * inner), tw["c_gkv"].view(B, Tv, 2 * inner)
        tr.attention_bwd(q.view(B, Ta, inner), kv3[:, :, :inner], kv3[:, :, inner:], tw["c_att"].view(B, Ta, inner), datt.view(B, Ta, inner),
                         tw["c_gq"].view(B, Ta, inner), gkv3[:, :, :inner], gkv3[:, :, inner:], tw["c_lse"], tw["c_delta"], Hc,
                         key_mask=fm, q_skip_mask=sm, scale=c.ca_dim_head ** -0.5, order=self._order[sm.data_ptr()])
        dnx = self._lin_bwd(tw["c_gq"], tw["c_nx"], "ca.q", dx_out=g2, row_mask=sflat)
        tr.layernorm_bwd(seg, P["ca.lnq.g"], dnx, tw["c_dseg"], dgamma=G["ca.lnq.g"], dbeta=G["ca.lnq.b"], add=g1, row_skip=sflat)
        dnc = self._lin_bwd(tw["c_gkv"], tw["c_nc"], "ca.kv", dx_out=tw["c_gnc"], row_mask=fflat)
        tr.layernorm_bwd(frame, P["ca.lnc.g"], dnc, tw["c_dframe"], dgamma=G["ca.lnc.g"], dbeta=G["ca.lnc.b"], row_skip=fflat)

    # ================================================================== backward
    def _groupable(self, dz: Tensor, x: Tensor, gw: Tensor) -> bool:
        return (self.tc == torch.bfloat16 and dz.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and gw.shape[0] % 128 == 0
                and gw.shape[1] % 128 == 0 and dz.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and dz.shape[0] >= 256
                and not getattr(self, "no_grouped_dw", False))

    def _flush_dw(self, pending: list, row_mask: Optional[Tensor]) -> None:
        """The weight gradients a layer's backward has queued (they reduce over the same rows) in one launch (made_gemm_tn_grouped)."""
        rows = self._rw(row_mask)
        # the launch's workspace (tile partials: ops_train.gemm_tn_grouped): one for the trainer's second stream, one for whatever stream the step
        # runs on -- the launches of a stream are ordered, so they can share one
        lane = "side" if torch.cuda.current_stream() == getattr(self, "_side", None) else "main"

        def workspace(nbytes: int) -> Tensor:
            wsd = self.__dict__.setdefault("_tn_ws", {})
            if lane not in wsd or wsd[lane].numel() < nbytes:
                if _tape.recording() or torch.cuda.is_current_stream_capturing():
                    raise _lib.MadeError("the weight-gradient workspace has to exist before the step is recorded (the warm-up run allocates it)")
                wsd[lane] = tr.gemm_tn_grouped_workspace(self.device, nbytes)
            return wsd[lane]
        for i in range(0, len(pending), 8):
            tr.gemm_tn_grouped(pending[i:i + 8], rows=rows, workspace=workspace)
        pending.clear()

    def _lin_bwd(self, dz: Tensor, x: Tensor, key: str, *, dx_out: Optional[Tensor] = None, row_mask: Optional[Tensor] = None,
                 skip: Optional[Tensor] = None, gw: Optional[Tensor] = None, gb: Optional[Tensor] = None, wt: Optional[Tensor] = None,
                 defer: Optional[list] = None, later: Optional[list] = None, **kw) -> Optional[Tensor]:
        """Linear backward: dW += dz^T x, db += colsum(dz), dx = dz W (with the epilogue options of ops.linear).
        later: a list that receives the weight-gradient launch as a closure instead (the caller runs it when and where it likes --
        dz and x must stay untouched until then)."""
        rows = self._rw(row_mask)                             # the mask's valid-row list: gather instead of masking
        gw_ = self.G[key + ".w"] if gw is None else gw
        gb_ = self.G.get(key + ".b") if gb is None else gb
        if later is not None:
            later.append(lambda: tr.gemm_tn(dz, x, gw_, accumulate=True, colsum=gb_, row_mask=row_mask if rows is None else None,
                                            row_groups=self._rg(row_mask), rows=rows))
        elif defer is not None and self._groupable(dz, x, gw_):
            defer.append((dz, x, gw_, gb_))                   # launched with the layer's other weight gradients (_flush_dw): the caller
        else:                                                 # keeps dz and x untouched until then
            tr.gemm_tn(dz, x, gw_, accumulate=True, colsum=gb_,
                       row_mask=row_mask if rows is None else None, row_groups=self._rg(row_mask), rows=rows)
        if dx_out is None:
            return None
        return ops.linear(dz, self.P[key + ".wt"] if wt is None else wt, None, out=dx_out, tile_skip_mask=skip if rows is None else None,
                          rows=rows, **kw)

    def _rg(self, row_mask: Optional[Tensor]) -> Optional[Tensor]:
        return self._groups.get(row_mask.data_ptr()) if row_mask is not None else None

    def _rw(self, mask: Optional[Tensor]):
        """(row_index, n_rows) of a token mask computed in forward_train, or None."""
        return self._rows.get(mask.data_ptr()) if mask is not None else None

    @torch.no_grad()
    def backward(self, g_ret: Optional[Tensor] = None, g_loc: Optional[Tensor] = None, zero_grad: bool = True, grad_sync=None,
                 early_opt=None) -> None:
        """Gradients of g_ret * retrieval_loss + g_loc * localization_loss (device scalars, default 1) into `flat_grad`.
        grad_sync(): called once, at the point where every gradient except the temporal group's (the first range of the flat
        buffer) is final -- data-parallel training starts that part's all-reduce there, under the encoders' backward.
        early_opt(): called at the same point on a THIRD stream that waits for everything queued so far on the two others (neither
        of them waits for it until the end of the backward pass): optimizer_step(part="early") there applies the matching + detection
        groups' update under the temporal encoders' backward, which reads none of their weights (opt-in, see _early_opt_ok)."""
        self._set_products()
        c, P, G = self.cfg, self.P, self.G
        B, Tv, Ta = self._shape
        ws, tw = self._buffers(B, Tv, Ta), self._train_buffers(B, Tv, Ta)
        feats_v, feats_a, fm, sm, tg = self._inputs
        concat = "concat" in c.mml_fusion
        D, L, Q, nd, ne, H = c.D, (Tv + Ta if concat else Ta), c.num_moment_queries, c.detr_dec_layers, c.detr_enc_layers, c.detr_nheads
        hd, HQ = D // H, H * Q
        pd = float(c.detr_dropout)
        inv_keep = 1.0 / (1.0 - pd) if (self.training_dropout and pd > 0) else 1.0
        fus, fus_mask = ws["fus"], ws["fus_mask"]
        fskip = fus_mask.view(-1)
        qskip = fus_mask
        regression = "regression" in c.mml_localization
        if regression:
            fskip = qskip = None
        rows = B * L
        if zero_grad and not getattr(self, "_grads_zeroed", False):
            _tape.zero_(self.flat_grad)
        self._grads_zeroed = False
        dvideo, dmusic = tw["dvideo"], tw["dmusic"]
        _tape.zero_(tw["dclip"])                              # dvideo, dmusic and the contrastive head's dvid_sum in one fill
        video, music = ws["video"], ws["music"]
        frame, seg_view = self._views
        # the X-Pool / similarity branch is independent of the DETR stack until the temporal encoders: its (latency-bound)
        # backward runs on the second stream beside the decoder's
        cur, side = torch.cuda.current_stream(), self._side_stream()
        xq = c.moment_query_type == "xpool" and not regression
        side.wait_stream(cur)
        ret_done = None
        ret_gen = None
        if not xq:                                           # (with an xpool query it follows the decoder's backward: see below)
            # MADE_RET_BWD_MAIN=1: on the main stream, in front of the heads and the decoder's chain (tools/race_probe3.py: beside the
            # first decoder layers of the backward chain these launches are what makes one element of a chain product come out one
            # bf16 ulp off in ~15 % of the first steps -- 0 of 60 with them on the main stream or with one stream only)
            ret_main = _lib.variant_env("MADE_RET_BWD_MAIN", "0") == "1"
            # MADE_RET_SPLIT (default 1): the branch's backward is issued in two parts -- up to the X-Pool tower's batched score / dP
            # products now, those and everything behind them on the second stream BEHIND the decoder's chain (see below)
            ret_split = (_lib.variant_env("MADE_RET_SPLIT", "1") != "0" and not ret_main and not regression
                         and _lib.variant_env("MADE_RET_HANDOFF", "") == "" and _lib.variant_env("MADE_RET_HANDOFF_REV", "") == "")
            with torch.cuda.stream(cur if ret_main else side):
                self._ret_main_stream = cur
                try:
                    self._ret_ck(0)
                    if ret_split:
                        ret_gen = self._retrieval_bwd_gen(ws, tw, g_ret, B, Ta, sm)
                        if next(ret_gen, "done") == "done":          # (a configuration without the tower: nothing left for later)
                            ret_gen = None
                    else:
                        self._retrieval_bwd(ws, tw, g_ret, B, Ta, sm)
                    on_main = torch.cuda.current_stream() == cur
                finally:
                    self._ret_main_stream = None
                if ret_gen is None:
                    ret_done = torch.cuda.Event()
                    ret_done.record(cur if on_main else side)

        if regression:
            dmem, dtgt0 = self._regression_bwd(ws, tw, g_loc, B, L), None
        else:
            # ---------------- criterion + heads
            logits, spans = ws["logits"], ws["spans"]
            pi, ti, cnt = self._match
            pq = ws["pq"] if c.contrastive_align_loss else None
            vid_sum = ws["vid_sum"] if c.contrastive_align_loss else None
            tr.set_criterion_bwd(logits, spans, tg, pi, ti, cnt, pq, vid_sum, P["empty_weight"], c.foreground_label, P["crit_weights"], g_loc,
                                 tw["dlog"], tw["dsp"], tw["dpq"] if pq is not None else None, tw["dvid_sum"] if pq is not None else None,
                                 ld_out=HEAD_PAD, through_sigmoid=True)
            hs2 = ws["hs"].view(nd * B * Q, D)
            dhs = tw["dhs"]
            dlog, dsp = tw["dlog"], tw["dsp"]
            if self.tc != torch.float32:                          # same dtype as the activations for the A^T B products
                tr.add3(tw["dlogsp_c"], tw["dlogsp"])             # both casts in one launch
                dlog, dsp = tw["dlog_c"], tw["dsp_c"]
            # the heads' weight gradients wait for nobody on the main stream: they are launched with the decoder's (second stream,
            # after the loop below); their operands (dlog / dsp / hg1 / hg2 / h1 / h2 / hs) are not touched in between
            heads_dw: list = []
            heads_dw.append(lambda: tr.gemm_tn(dlog[:, :2], hs2, G["class_embed.w"], accumulate=True, colsum=G["class_embed.b"]))
            ops.linear(dlog, P["class_embed.wt"], None, out=dhs)
            n_span = 1 if c.predict_center == 1 else 2             # predict_center: the width column is a constant, its gradient is dropped
            heads_dw.append(lambda: tr.gemm_tn(dsp[:, :n_span], tw["h2"], G["span_embed.2.w"], accumulate=True, colsum=G["span_embed.2.b"]))
            dz2 = ops.linear(dsp, P["span_embed.2.wt"], None, out=tw["hg1"], gate=_lib.GATE_RELU_OUT, G=tw["h2"])
            dz1 = self._lin_bwd(dz2, tw["h1"], "span_embed.1", dx_out=tw["hg2"], gate=_lib.GATE_RELU_OUT, G=tw["h1"], later=heads_dw)
            self._lin_bwd(dz1, hs2, "span_embed.0", dx_out=dhs, R=dhs, later=heads_dw)
            if c.contrastive_align_loss:
                Dc = pq.shape[-1]
                dn = tw["dpq"]
                if c.audio_short_cut:                             # back through normalize(. + music), aux layers twice; music collects the sums
                    if c.aux_loss and nd > 1:
                        n_aux = (nd - 1) * B * Q
                        ds2 = tw["dpq_s2"][:n_aux]
                        tr.l2norm_bwd(tw["pq_s2"][:n_aux], dn[:n_aux], dx=ds2)
                        dmusic.add_(ds2.view(nd - 1, B, Q, D).sum((0, 2)))
                        dn[:n_aux].copy_(ds2)
                    tr.l2norm_bwd(tw["pq_s1"], dn, dx=tw["dpq_s"])
                    dmusic.add_(tw["dpq_s"].view(nd, B, Q, D).sum((0, 2)))
                    dn = tw["dpq_s"]
                tr.l2norm_bwd(ws["pq_raw"], dn, dx_alt=tw["dpq_raw"])
                self._lin_bwd(tw["dpq_raw"], hs2, "proj_q", dx_out=dhs, R=dhs, later=heads_dw)
                # proj_vid_mem: every frame (padded ones too) receives d vid_sum (reference loss_detr.py:118)
                tr.l2norm_bwd(ws["pv_raw"], tw["dvid_sum"], dx_alt=tw["dpv_raw"], dy_rows_per=Tv)
                # (its weight gradient is nobody's input: with the heads' other weight gradients on the second stream, after the decoder's
                #  chain -- it was 52 us at the head of the main stream's backward)
                heads_dw.append(lambda: tr.gemm_tn(tw["dpv_raw"][:Tv], frame[0], G["proj_v.w"], accumulate=True, colsum=G["proj_v.b"], batch=(B, 1),
                                                   a_zs=(Tv * Dc, 0), b_zs=(frame.stride(0), 0), colsum_zs=(0, 0)))
                ops.linear(tw["dpv_raw"], P["proj_v.wt"], None, out=tw["dframe_x"])

            # ---------------- decoder, last layer first.  Inside the loop only the data-gradient chain runs; every output gradient a
            # weight gradient needs is kept per layer (the g_* stacks) and the weight-gradient products of all layers are batched after it.
            qp = P["query_embed"]
            mem3, mempos3 = tw["mem"].view(B, L, D), tw["mempos"].view(B, L, D)
            Lp = tw["PdS"].shape[-1]
            GQ, PdS = tw["GQ"], tw["PdS"]
            ca_scale = 1.0 / math.sqrt(hd)
            st = tw["dstack"]
            dtgt = None
            # the scores of every layer's memory-space attention depend on forward values only (q', memory + pos): ONE batched
            # product for all layers ([nd, B, H*Q, L] rows) ahead of the dependent chain instead of one launch inside every layer
            stage = self._dec_stage_chain()                    # fused chain: see forward_train
            pre = bool(c.detr_pre_norm)                        # pre-norm layers (reference music_detr/transformer.py:246-271): separate launches
            S_all = tw["dS_S"]
            if not stage:
                ops.linear(GQ[0, 1].reshape(nd * HQ, D), mempos3[0], None, M=nd * HQ, N=L, K=D, batch=B, a_z_stride=GQ.stride(0), w_z_stride=L * D,
                           segs=[Seg(out=S_all, ldo=Lp, rows_per_batch=HQ, out_batch_stride=B * HQ * Lp, out_z_stride=HQ * Lp)])
            n_split_b = int(_lib.variant_env("MADE_WIDE_NSPLIT", "0")) or max(1, min(8, 256 // max(B, 1)))
            if stage:
                # the shared output norm's backward depends on the heads only: all layers in ONE launch ahead of the dependent chain
                gN = tw["dgN"]
                tr.layernorm_bwd(tw["dstack"]["tgt"][1:].reshape(nd * B * Q, D), P["dec.norm.g"], dhs, gN, dgamma=G["dec.norm.g"], dbeta=G["dec.norm.b"])
            for l in range(nd - 1, -1, -1):
                p, d = f"detr_transformer.decoder.layers.{l}", f"d.{l}"
                g1, g2, g4 = tw["dg1"], tw["dg2"], tw["dg4"]
                if stage and _lib.variant_env("MADE_CHAIN_BUFS", "1") != "0":
                    # every hand-off of the chain gets rows of its own instead of three scratch buffers rewritten and re-read a few launches
                    # apart (MADE_CHAIN_BUFS=0: the shared buffers) -- a leftover of the hunt for the chain's one-ulp deviation, which
                    # turned out to need two launches of the retrieval branch beside the chain (MADE_RET_SPLIT above);
                    # kept: tools/race_probe3.py reads the hand-offs from here
                    ch = tw["dchain"][l]
                    g1a, g1b, g2a, g2b, g2c, dt_out = ch[0], ch[1], ch[2], ch[3], ch[4], ch[5]
                else:
                    g1a = g1b = g1
                    g2a = g2b = g2c = g2
                    dt_out = tw["dtgt"]
                Win, Wt = P[p + ".ca.in.w"], P[p + ".ca.in.wt"]
                g_ffn, g_z, g_ca, g_attc, g_q, g_qc, g_sa, gqkv = (st[k][l] for k in ("g_ffn", "g_z", "g_ca", "g_attc", "g_q", "g_qc", "g_sa", "g_qkv"))
                if stage:
                    # hs_l = dec.norm(t3) (its backward: one launch for all layers ahead of the chain), t3 = LN3(t2 + drop3(ffn)) also feeds
                    # the next layer: norm 3's backward of (d hs_l through the output norm + d tgt_{l+1}) in the prologue of the FFN's second
                    # dX product
                    tr.dec_stage_bwd(tw[d + ".t_c"], P[p + ".ln3.g"], gN[l * B * Q:(l + 1) * B * Q], P[p + ".ff2.wt"], g_z,
                                     dgamma_a=G[p + ".ln3.g"], dbeta_a=G[p + ".ln3.b"], add=dtgt, dx_out=g2a, a_out=g_ffn,
                                     drop_a=self._drop(f"dec.{l}" + ".drop3", pd), G=tw[d + ".h"], gate_scale=inv_keep)
                elif pre:
                    # hs_l = dec.norm(stream); the stream (slot l + 1 of the tgt stack) also feeds the next layer: g1 = d stream
                    tr.layernorm_bwd(tw[d + ".t3"], P["dec.norm.g"], dhs[l * B * Q:(l + 1) * B * Q], g1, dgamma=G["dec.norm.g"], dbeta=G["dec.norm.b"], add=dtgt)
                    # stream = t_b + drop3(ffn(LN3(t_b)))
                    tr.gate_rows(g1, g_ffn, drop=self._drop(f"dec.{l}" + ".drop3", pd))
                else:
                    # hs_l = dec.norm(t3); t3 also feeds the next layer
                    tr.layernorm_bwd(tw[d + ".t3"], P["dec.norm.g"], dhs[l * B * Q:(l + 1) * B * Q], g1, dgamma=G["dec.norm.g"], dbeta=G["dec.norm.b"], add=dtgt)
                    # t3 = LN3(t2 + drop3(ffn))
                    tr.layernorm_bwd(tw[d + ".t_c"], P[p + ".ln3.g"], g1, g2, dgamma=G[p + ".ln3.g"], dbeta=G[p + ".ln3.b"],
                                     dx_drop=g_ffn, drop=self._drop(f"dec.{l}" + ".drop3", pd))
                if not stage:
                    ops.linear(g_ffn, P[p + ".ff2.wt"], None, out=g_z, gate=_lib.GATE_RELU_OUT, G=tw[d + ".h"], gate_scale=inv_keep)
                if pre:
                    dn3 = ops.linear(g_z, P[p + ".ff1.wt"], None, out=g4)
                    # t_b = t_a + drop2(cross-attention): d t_b = LN3'(dn3) + d stream -> g2; its dropped copy feeds the out-projection
                    tr.layernorm_bwd(tw[d + ".t_b"], P[p + ".ln3.g"], dn3, g2, dgamma=G[p + ".ln3.g"], dbeta=G[p + ".ln3.b"], add=g1,
                                     dx_drop=g_ca, drop=self._drop(f"dec.{l}" + ".drop2", pd))
                    dattc = ops.linear(g_ca, P[p + ".ca.out.wt"], None, out=g_attc)
                else:
                    dt2 = ops.linear(g_z, P[p + ".ff1.wt"], None, out=g1a, R=g2a)
                # t2 = LN2(t1 + drop2(cross-attention))
                if pre:
                    pass
                elif stage:                                       # norm 2's backward in the prologue of the out-projection's dX product
                    dattc = tr.dec_stage_bwd(tw[d + ".t_b"], P[p + ".ln2.g"], dt2, P[p + ".ca.out.wt"], g_attc, dgamma_a=G[p + ".ln2.g"],
                                             dbeta_a=G[p + ".ln2.b"], dx_out=g2b, a_out=g_ca, drop_a=self._drop(f"dec.{l}" + ".drop2", pd))
                else:
                    tr.layernorm_bwd(tw[d + ".t_b"], P[p + ".ln2.g"], dt2, g2, dgamma=G[p + ".ln2.g"], dbeta=G[p + ".ln2.b"],
                                     dx_drop=g_ca, drop=self._drop(f"dec.{l}" + ".drop2", pd))
                    dattc = ops.linear(g_ca, P[p + ".ca.out.wt"], None, out=g_attc)
                dpooled = GQ[:, 0, l]                             # [B, H*Q, D] slice of the concatenated buffer
                qprime = GQ[:, 1, l]
                if stage:
                    # v_h = W_v,h pooled_h : dpooled_h = dattc_h W_v,h
                    ops.linear(dattc[:, :hd], Wt[:, 2 * D:2 * D + hd], None, M=B * Q, N=D, K=hd, batch=H, a_z_stride=hd, w_z_stride=hd,
                               segs=[Seg(out=dpooled, ldo=D, rows_per_batch=Q, out_batch_stride=dpooled.stride(0), out_z_stride=Q * D)])
                    # the memory-space attention's backward in ONE launch: scores and probabilities recomputed from the saved lse, the
                    # value-bias term reduced from dattc, Pd / dS written for the memory-gradient product after the loop, dq' = dS (mem + pos)
                    # (the value bias' own gradient needs nothing of the chain: after the loop, second stream)
                    tr.attention_wide_bwd(qprime, dpooled, tw[d + ".pooled"].view(B, HQ, D), mempos3, mem3, tw["ca_lse"][l].view(B, HQ),
                                          PdS[:, 0, l], PdS[:, 1, l], g_q, scale=ca_scale, key_mask=fus_mask, ssum=tw[d + ".s"].view(B, HQ),
                                          dattc=dattc, vbias=P[p + ".ca.in.b"][2 * D:], hd=hd, drop=self._drop(f"dec.{l}" + ".ca_attn", pd),
                                          n_split=n_split_b, part_dq=ws["part_o"])
                else:
                    tr.head_bias_bwd(dattc, tw[d + ".s"], P[p + ".ca.in.b"][2 * D:], G[p + ".ca.in.b"][2 * D:], tw["d_ds"], H)
                    d_ds = tw["d_ds"]                                 # rows (b, q), columns h; the softmax backward numbers its rows (b, h, q)
                    if Q > 1:
                        tw["dds_raw"].view(B, H, Q).copy_(d_ds.view(B, Q, H).permute(0, 2, 1))
                        d_ds = tw["dds_raw"]
                    # v_h = W_v,h pooled_h : dpooled_h = dattc_h W_v,h
                    ops.linear(dattc[:, :hd], Wt[:, 2 * D:2 * D + hd], None, M=B * Q, N=D, K=hd, batch=H, a_z_stride=hd, w_z_stride=hd,
                               segs=[Seg(out=dpooled, ldo=D, rows_per_batch=Q, out_batch_stride=dpooled.stride(0), out_z_stride=Q * D)])
                    # scores and dPd of the memory-space attention (few rows per sample: materialised)
                    S, dP = S_all[l], tw["dS_dP"]
                    ops.linear(dpooled[0], mem3[0], None, M=HQ, N=L, K=D, batch=B, a_z_stride=dpooled.stride(0), w_z_stride=L * D,
                               segs=[Seg(out=dP, ldo=Lp, out_z_stride=HQ * Lp)])
                    tr.softmax_bwd(S, dP, fus_mask, HQ, ca_scale, PdS[:, 0, l], PdS[:, 1, l], tw["dSt"], HQ, L, extra=d_ds.view(-1),
                                   drop=self._drop(f"dec.{l}" + ".ca_attn", pd), ldo=Lp, ldt=HQ, out_batch_stride=PdS.stride(0))
                    # dq'[b] = dS[b] (mem + pos)[b]
                    gq_out = g_q if Q == 1 else tw["gq_raw"]         # [B, (h, q), D] out of the product; [B, (q, h), D] for the Linears
                    tr.gemm_tn(tw["dSt"][0], mempos3[0], gq_out[0], batch=(B, 1), a_zs=(L * HQ, 0), b_zs=(L * D, 0), c_zs=(HQ * D, 0),
                               row_mask=fus_mask, mask_zs=(L, 0))
                    if Q > 1:
                        g_q.view(B, Q, H, D).copy_(gq_out.view(B, H, Q, D).permute(0, 2, 1, 3))
                # q'_h = W_k,h^T qc_h : dqc_h = dq'_h W_k,h^T
                dq2 = g_q.view(B * Q, H * D)
                ops.linear(dq2[:, :D], Win[D:D + hd], None, M=B * Q, N=hd, K=D, batch=H, a_z_stride=D, w_z_stride=hd * D,
                           segs=[Seg(out=g_qc, ldo=D, out_z_stride=hd)])
                # qc = W_q (t1 + qp) + b_q
                # dt1 = residual path + query path in the Linear's epilogue; the query path alone (the pre-residual value) is kept for
                # the query embedding's gradient
                dt1q = st["dt1q"][l]                               # (summed over the batch into the query embedding's gradient after the loop)
                if pre:
                    # qc = W_q (LN2(t_a) + qp) + b_q: d (LN2(t_a) + qp) -> dt1q; d t_a = LN2'(dt1q) + d t_b -> g1, its dropped copy feeds the
                    # self-attention's out-projection
                    ops.linear(g_qc, Wt[:, :D], None, out=dt1q)
                    tr.layernorm_bwd(tw[d + ".t_a"], P[p + ".ln2.g"], dt1q, g1, dgamma=G[p + ".ln2.g"], dbeta=G[p + ".ln2.b"], add=g2,
                                     dx_drop=g_sa, drop=self._drop(f"dec.{l}" + ".drop1", pd))
                    if Q == 1:                                    # value path only (see the forward)
                        ops.linear(g_sa, P[p + ".sa.out.wt"], None, segs=[Seg(out=gqkv[:, 2 * D:], ldo=gqkv.stride(0))],
                                   drop=self._drop(f"dec.{l}" + ".sa_attn", pd), drop_ld=H, drop_col_div=hd)
                        dn1 = ops.linear(gqkv[:, 2 * D:], P[p + ".sa.in.wt"][:, 2 * D:], None, out=g4)
                    else:
                        datt = ops.linear(g_sa, P[p + ".sa.out.wt"], None, out=g4)
                        qkv = tw[d + ".qkv"]
                        q3, g3v = qkv.view(B, Q, 3 * D), gqkv.view(B, Q, 3 * D)
                        tr.attention_bwd(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], tw[d + ".att"].view(B, Q, D), datt.view(B, Q, D),
                                         g3v[:, :, :D], g3v[:, :, D:2 * D], g3v[:, :, 2 * D:], tw[d + ".lse"], tw["d_delta"], H,
                                         drop=self._drop(f"dec.{l}" + ".sa_attn", pd))
                        dn1 = ops.linear(gqkv, P[p + ".sa.in.wt"], None, out=g2)       # q, k and v all read n1 (+ qp)
                        dqk = ops.linear(gqkv[:, :2 * D], P[p + ".sa.in.wt"][:, :2 * D], None, out=g4)
                        tr.colsum(dqk.view(B, Q * D), G["query_embed"].view(-1))        # the query embedding enters through q, k
                    # n1 = LN1(tgt): d tgt = LN1'(dn1) + d t_a
                    dtgt = tr.layernorm_bwd(tw[d + ".tgt"], P[p + ".ln1.g"], dn1, tw["dtgt"], dgamma=G[p + ".ln1.g"], dbeta=G[p + ".ln1.b"], add=g1)
                    continue
                ops.linear(g_qc, Wt[:, :D], None, out=g1b, R=g2b, Zout=dt1q)
                # t1 = LN1(tgt + drop1(self-attention))
                if stage:
                    # norm 1's backward in the prologue of the self-attention out-projection's dX product (value path only: dv = datt under
                    # the same per-head mask, drawn in the epilogue)
                    tr.dec_stage_bwd(tw[d + ".t_a"], P[p + ".ln1.g"], g1b, P[p + ".sa.out.wt"], gqkv[:, 2 * D:], dgamma_a=G[p + ".ln1.g"],
                                     dbeta_a=G[p + ".ln1.b"], dx_out=g2c, a_out=g_sa, drop_a=self._drop(f"dec.{l}" + ".drop1", pd),
                                     drop_o=self._drop(f"dec.{l}" + ".sa_attn", pd), drop_o_ld=H, drop_o_col_div=hd)
                    dtgt = ops.linear(gqkv[:, 2 * D:], P[p + ".sa.in.wt"][:, 2 * D:], None, R=g2c, out=dt_out)
                    continue
                tr.layernorm_bwd(tw[d + ".t_a"], P[p + ".ln1.g"], g1, g2, dgamma=G[p + ".ln1.g"], dbeta=G[p + ".ln1.b"],
                                 dx_drop=g_sa, drop=self._drop(f"dec.{l}" + ".drop1", pd))
                if Q == 1:                                        # value path only (see the forward): dv = datt under the same per-head mask,
                    ops.linear(g_sa, P[p + ".sa.out.wt"], None, segs=[Seg(out=gqkv[:, 2 * D:], ldo=gqkv.stride(0))],       # drawn in the epilogue
                               drop=self._drop(f"dec.{l}" + ".sa_attn", pd), drop_ld=H, drop_col_div=hd)
                    dtgt = ops.linear(gqkv[:, 2 * D:], P[p + ".sa.in.wt"][:, 2 * D:], None, R=g2, out=tw["dtgt"])
                else:
                    datt = ops.linear(g_sa, P[p + ".sa.out.wt"], None, out=g4)
                    qkv = tw[d + ".qkv"]
                    q3, g3v = qkv.view(B, Q, 3 * D), gqkv.view(B, Q, 3 * D)
                    tr.attention_bwd(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], tw[d + ".att"].view(B, Q, D), datt.view(B, Q, D),
                                     g3v[:, :, :D], g3v[:, :, D:2 * D], g3v[:, :, 2 * D:], tw[d + ".lse"], tw["d_delta"], H,
                                     drop=self._drop(f"dec.{l}" + ".sa_attn", pd))
                    dtgt = ops.linear(gqkv, P[p + ".sa.in.wt"], None, R=g2, out=tw["dtgt"])
                    # the query embedding also enters through q,k of the self-attention
                    dqk = ops.linear(gqkv[:, :2 * D], P[p + ".sa.in.wt"][:, :2 * D], None, out=g4)
                    tr.colsum(dqk.view(B, Q * D), G["query_embed"].view(-1))
            dtgt0 = dtgt                                          # gradient of the decoder's content query = the clip-level vector
            # gradient of the memory: values path (Pd^T dpooled) + keys path (dS^T q'), all layers in one product per sample -- the
            # one thing the DETR encoder's backward waits for
            dmem = tw["eg1"]
            tr.gemm_tn(PdS.view(B, 2 * nd * HQ, Lp)[0, :, :L], GQ.view(B, 2 * nd * HQ, D)[0], dmem.view(B, L, D)[0], batch=(B, 1),
                       a_zs=(PdS.stride(0), 0), b_zs=(GQ.stride(0), 0), c_zs=(L * D, 0))

            # ---- weight gradients of all decoder layers: one layer-batched product per parameter (the layers' parameters, and
            # the [nd, ...] stacks, are equally spaced).  Nothing on the main stream needs them before the optimizer: they go to the
            # second stream (a dozen launches of modest size, ~0.35 ms of kernel time) and run beside the DETR encoder's backward;
            # the join in front of grad_sync below covers them.
            dw_side = _lib.variant_env("MADE_DEC_DW_SIDE", "1") != "0"       # (knob for A/B measurements)
            if dw_side:
                side.wait_stream(cur)
            if ret_gen is not None:
                # second part of the retrieval branch's backward: behind the chain (second stream), in front of the weight gradients
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    for _ in ret_gen:
                        pass
                    ret_done = torch.cuda.Event()
                    ret_done.record(side)
                ret_gen = None
            dec_dw = torch.cuda.stream(side if dw_side else cur)
            dec_dw.__enter__()
            for launch in heads_dw:
                launch()
            tr.colsum(st["dt1q"].view(nd * B, Q * D), G["query_embed"].view(-1))     # every layer's query path at once
            if stage:                                          # the value bias' gradient d b_v,h = sum_b s[b, h] dattc[b, h] of every layer
                for l in range(nd):
                    pl = f"detr_transformer.decoder.layers.{l}"
                    tr.head_bias_bwd(st["g_attc"][l], tw[f"d.{l}.s"], P[pl + ".ca.in.b"][2 * D:], G[pl + ".ca.in.b"][2 * D:], tw["d_ds"], H)
            p0, p1 = "detr_transformer.decoder.layers.0", "detr_transformer.decoder.layers.1"
            BQ = B * Q

            def lstride(key):
                return (G[p1 + key].data_ptr() - G[p0 + key].data_ptr()) // 4 if nd > 1 else 0

            def batched(a_stack, b_stack, wkey, bkey, a_cols=None, b_cols=None, w_rows=None):
                A, Bm = a_stack[0], b_stack[0]
                if a_cols is not None:
                    A = A[:, a_cols[0]:a_cols[1]]
                if b_cols is not None:
                    Bm = Bm[:, b_cols[0]:b_cols[1]]
                gw, gb = G[p0 + wkey], (G[p0 + bkey] if bkey is not None else None)
                if w_rows is not None:
                    gw = gw[w_rows[0]:w_rows[1]]
                    gb = gb[w_rows[0]:w_rows[1]] if gb is not None else None
                tr.gemm_tn(A, Bm, gw, accumulate=True, colsum=gb, batch=(nd, 1), a_zs=(a_stack.stride(0), 0), b_zs=(b_stack.stride(0), 0),
                           c_zs=(lstride(wkey), 0), colsum_zs=(lstride(bkey) if bkey is not None else 0, 0))

            batched(st["g_ffn"], st["h"], ".ff2.w", ".ff2.b")
            batched(st["g_z"], st["t2"], ".ff1.w", ".ff1.b")
            batched(st["g_ca"], st["attc"], ".ca.out.w", ".ca.out.b")
            batched(st["g_qc"], st["t1q"], ".ca.in.w", ".ca.in.b", w_rows=(0, D))
            batched(st["g_sa"], st["att"], ".sa.out.w", ".sa.out.b")
            if Q > 1:                                             # (a single query's q / k projections get no gradient)
                batched(st["g_qkv"], st["tq"], ".sa.in.w", ".sa.in.b", a_cols=(0, 2 * D), w_rows=(0, 2 * D))
            batched(st["g_qkv"], st["n1"] if pre else st["tgt"], ".sa.in.w", ".sa.in.b", a_cols=(2 * D, 3 * D), w_rows=(2 * D, 3 * D))
            # per-head products of the memory-space cross-attention, batched over (layer, head)
            gWin0 = G[p0 + ".ca.in.w"]
            ls = lstride(".ca.in.w")
            # v_h = W_v,h pooled_h : dW_v,h += dattc_h^T pooled_h
            tr.gemm_tn(st["g_attc"][0][:, :hd], st["pooled"][0][:, :D], gWin0[2 * D:2 * D + hd], accumulate=True, batch=(nd, H),
                       a_zs=(st["g_attc"].stride(0), hd), b_zs=(st["pooled"].stride(0), D), c_zs=(ls, hd * D))
            # q'_h = W_k,h^T qc_h : dW_k,h += qc_h^T dq'_h
            gq2 = st["g_q"].view(nd, BQ, H * D)
            tr.gemm_tn(st["qc"][0][:, :hd], gq2[0][:, :D], gWin0[D:D + hd], accumulate=True, batch=(nd, H),
                       a_zs=(st["qc"].stride(0), hd), b_zs=(gq2.stride(0), D), c_zs=(ls, hd * D))
            dec_dw.__exit__(None, None, None)


        # ---------------- DETR encoder
        dsrc = dmem
        # the encoder layers' grouped weight-gradient launches: on the main stream since round 6 (the single-pass attention backward shortened the
        # main stream's chain; inside the step 4.80 against 4.88 ms with these and the audio tower's on the main stream: gpurun_out/ab_dw_side.txt,
        # profiles/r06_ab_dw_side.txt).  MADE_ENC_DW_SIDE=1 (measurement knob): on the second stream, as rounds 3-5 had them
        enc_dw_side = _lib.variant_env("MADE_ENC_DW_SIDE", "0") != "0"
        enc_dw_done = [None, None]
        for l in range(ne - 1, -1, -1):
            p, e = f"detr_transformer.encoder.layers.{l}", f"e.{l}"
            # (every gradient that feeds a weight-gradient product keeps a buffer of its own until the layer's grouped launch)
            g2, g2b, g3, g3b, g3c, gq, gf = tw["eg2"], tw["eg2b"], tw["eg3"], tw["eg3b"], tw["eg3c"], tw["egqkv"], tw["egffn"]
            if enc_dw_side:
                if l & 1:
                    g2b, g3, gq, gf = tw["eg2b_2"], tw["eg3_2"], tw["egqkv_2"], tw["egffn_2"]
                if enc_dw_done[l & 1] is not None:               # (more than two layers: the set's previous weight-gradient launch)
                    cur.wait_event(enc_dw_done[l & 1])
            pend: list = []
            if c.detr_pre_norm:
                # reference music_detr/transformer.py:170-189 (forward_pre), backwards.  dsrc = gradient of the layer's output stream x2.
                n1, n1pos = tw[e + ".src"], tw[e + ".srcpos"]                # the attention's input LN1(x_in) (+ pos)
                xin = fus.view(rows, D) if l == 0 else tw[f"e.{l - 1}.x2"]
                if l == ne - 1:                                # the encoder's own norm behind the last layer: dsrc = d memory here
                    dsrc = tr.layernorm_bwd(tw[e + ".x2"], P["enc.norm.g"], dsrc, g2, dgamma=G["enc.norm.g"], dbeta=G["enc.norm.b"], row_skip=fskip)
                # x2 = x + drop2(ffn(LN2(x)))
                tr.gate_rows(dsrc, g3, drop=self._drop(f"enc.{l}" + ".drop2", pd), row_skip=fskip)
                dz = self._lin_bwd(g3, tw[e + ".h"], p + ".ff2", dx_out=gf, row_mask=fskip, skip=fskip, gate=_lib.GATE_RELU_OUT, G=tw[e + ".h"],
                                   gate_scale=inv_keep, defer=pend)
                dn2 = self._lin_bwd(dz, tw[e + ".s1"], p + ".ff1", dx_out=g3b, row_mask=fskip, skip=fskip, defer=pend)
                # x = x_in + drop1(attention(LN1(x_in))): d x = LN2'(dn2) + d x2; its dropped copy feeds the out-projection
                dx = tw["eg1"] if dsrc is not tw["eg1"] else tw["dfus"]
                tr.layernorm_bwd(tw[e + ".x"], P[p + ".ln2.g"], dn2, dx, dgamma=G[p + ".ln2.g"], dbeta=G[p + ".ln2.b"], add=dsrc, dx_drop=g2b,
                                 drop=self._drop(f"enc.{l}" + ".drop1", pd), row_skip=fskip)
                datt = self._lin_bwd(g2b, tw[e + ".att"], p + ".out", dx_out=g3c, row_mask=fskip, skip=fskip, defer=pend)
                qkv = tw[e + ".qkv"]
                q3, gq3 = qkv.view(B, L, 3 * D), gq.view(B, L, 3 * D)
                tr.attention_bwd(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], tw[e + ".att"].view(B, L, D), datt.view(B, L, D),
                                 gq3[:, :, :D], gq3[:, :, D:2 * D], gq3[:, :, 2 * D:], tw[e + ".lse"], tw["e_delta"], H,
                                 key_mask=fus_mask, q_skip_mask=qskip, drop=self._drop(f"enc.{l}" + ".attn", pd),
                                 order=self._order[fus_mask.data_ptr()], keep_bits=tw[e + ".kbits"] if self._bits else None)
                gW, gb = G[p + ".in.w"], G[p + ".in.b"]
                if self._groupable(gq, n1pos, gW[:2 * D]) and self._rw(fskip) is not None:
                    pend.append((gq[:, :2 * D], n1pos, gW[:2 * D], gb[:2 * D]))
                    pend.append((gq[:, 2 * D:], n1, gW[2 * D:], gb[2 * D:]))
                else:
                    tr.gemm_tn(gq[:, :2 * D], n1pos, gW[:2 * D], accumulate=True, colsum=gb[:2 * D], rows=self._rw(fskip))
                    tr.gemm_tn(gq[:, 2 * D:], n1, gW[2 * D:], accumulate=True, colsum=gb[2 * D:], rows=self._rw(fskip))
                dn1 = ops.linear(gq, P[p + ".in.wt"], None, out=g2, rows=self._rw(fskip))      # q, k and v all read LN1(x_in) (+ pos)
                nxt = tw["dfus"] if dx is tw["eg1"] else tw["eg1"]
                dsrc = tr.layernorm_bwd(xin, P[p + ".ln1.g"], dn1, nxt, dgamma=G[p + ".ln1.g"], dbeta=G[p + ".ln1.b"], add=dx, row_skip=fskip)
                if enc_dw_side:
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        self._flush_dw(pend, fskip)
                        enc_dw_done[l & 1] = torch.cuda.Event()
                        enc_dw_done[l & 1].record(side)
                else:
                    self._flush_dw(pend, fskip)
                continue
            src = fus.view(rows, D) if l == 0 else tw[e + ".src"]
            srcpos = tw[e + ".srcpos"]
            # src_{l+1} = LN2(s1 + drop2(ffn))
            tr.layernorm_bwd(tw[e + ".x2"], P[p + ".ln2.g"], dsrc, g2, dgamma=G[p + ".ln2.g"], dbeta=G[p + ".ln2.b"], dx_drop=g3,
                             drop=self._drop(f"enc.{l}" + ".drop2", pd), row_skip=fskip)
            dz = self._lin_bwd(g3, tw[e + ".h"], p + ".ff2", dx_out=gf, row_mask=fskip, skip=fskip, gate=_lib.GATE_RELU_OUT, G=tw[e + ".h"],
                               gate_scale=inv_keep, defer=pend)
            ds1 = self._lin_bwd(dz, tw[e + ".s1"], p + ".ff1", dx_out=g3b, row_mask=fskip, skip=fskip, R=g2, defer=pend)
            # s1 = LN1(src + drop1(attention))
            dx = tw["eg1"] if dsrc is not tw["eg1"] else tw["dfus"]
            tr.layernorm_bwd(tw[e + ".x"], P[p + ".ln1.g"], ds1, dx, dgamma=G[p + ".ln1.g"], dbeta=G[p + ".ln1.b"], dx_drop=g2b,
                             drop=self._drop(f"enc.{l}" + ".drop1", pd), row_skip=fskip)
            datt = self._lin_bwd(g2b, tw[e + ".att"], p + ".out", dx_out=g3c, row_mask=fskip, skip=fskip, defer=pend)
            qkv = tw[e + ".qkv"]
            q3, gq3 = qkv.view(B, L, 3 * D), gq.view(B, L, 3 * D)
            tr.attention_bwd(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], tw[e + ".att"].view(B, L, D), datt.view(B, L, D),
                             gq3[:, :, :D], gq3[:, :, D:2 * D], gq3[:, :, 2 * D:], tw[e + ".lse"], tw["e_delta"], H,
                             key_mask=fus_mask, q_skip_mask=qskip, drop=self._drop(f"enc.{l}" + ".attn", pd),
                             order=self._order[fus_mask.data_ptr()], keep_bits=tw[e + ".kbits"] if self._bits else None)
            gW, gb = G[p + ".in.w"], G[p + ".in.b"]
            if self._groupable(gq, srcpos, gW[:2 * D]) and self._rw(fskip) is not None:
                pend.append((gq[:, :2 * D], srcpos, gW[:2 * D], gb[:2 * D]))
                pend.append((gq[:, 2 * D:], src, gW[2 * D:], gb[2 * D:]))
            else:
                tr.gemm_tn(gq[:, :2 * D], srcpos, gW[:2 * D], accumulate=True, colsum=gb[:2 * D], rows=self._rw(fskip))
                tr.gemm_tn(gq[:, 2 * D:], src, gW[2 * D:], accumulate=True, colsum=gb[2 * D:], rows=self._rw(fskip))
            nxt = tw["dfus"] if dx is tw["eg1"] else tw["eg1"]
            dsrc = ops.linear(gq, P[p + ".in.wt"], None, R=dx, out=nxt, rows=self._rw(fskip))
            if enc_dw_side:
                # the layer's grouped weight-gradient launch (~170 us of MFMA work) needs nothing but this layer's gradients and
                # nobody on the main stream needs it: second stream, beside the next layer's (bandwidth- and VALU-heavy) data chain
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    self._flush_dw(pend, fskip)
                    enc_dw_done[l & 1] = torch.cuda.Event()
                    enc_dw_done[l & 1].record(side)
            else:
                self._flush_dw(pend, fskip)
        dfus = dsrc.view(B, L, D)
        if concat:
            dl_v, dl_a = dfus[:, :Tv], dfus[:, Tv:]
        else:                                                 # through the CA fusion block back to the two encoders' outputs
            self._ca_fusion_bwd(ws, tw, dsrc, frame, seg_view, fm, sm, B, Tv, Ta)
            dl_v, dl_a = tw["c_dframe"].view(B, Tv, D), tw["c_dseg"].view(B, Ta, D)

        # ---------------- join the X-Pool / similarity branch, merge the gradients of the clip-level vectors
        # (one process: the main stream needs the retrieval branch's results here, not the weight-gradient products queued behind it on
        #  the second stream -- joining the whole stream left it idle behind the last DETR layer's grouped product, 0.2 ms per step;
        #  data-parallel jobs start their first all-reduce below and need every gradient of its range complete)
        if ret_done is not None and grad_sync is None and _lib.variant_env("MADE_BWD_EVENT_JOIN", "1") != "0":
            cur.wait_event(ret_done)
        else:
            cur.wait_stream(side)
        if xq:
            # the decoder's content query was the mean over the videos of each track's pooled vectors: its gradient enters the
            # X-Pool tail as dpool[m] / N_v for every video n
            dq = tw["dxpool_q"]
            tr.add3(dq, dtgt0.view(B, D) if Q == 1 else dtgt0.view(B, Q, D).float().sum(dim=1).contiguous())
            self._retrieval_bwd(ws, tw, g_ret, B, Ta, sm, dpool=dq)
        if c.moment_query_type in ("video", "music") and not regression:       # (a zero content query has no gradient to hand on)
            dq_vec = dvideo if c.moment_query_type == "video" else dmusic
            tr.add3(dq_vec, dq_vec, dtgt0.view(B, D) if Q == 1 else dtgt0.view(B, Q, D).float().sum(dim=1))    # the vector was repeated Q times

        if grad_sync is not None:
            grad_sync()
        opt_st = None
        if early_opt is not None:
            opt_st = self._opt_stream()
            e_main, e_side = torch.cuda.Event(), torch.cuda.Event()
            e_main.record(cur); e_side.record(side)            # (the last DETR layer's weight-gradient launch sits on the second stream)
            opt_st.wait_event(e_main); opt_st.wait_event(e_side)
            with torch.cuda.stream(opt_st):
                early_opt()
        # ---------------- temporal encoders (video on the second stream)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            # extra gradients of the frame features: the contrastive-align projection and the second X-Pool tower's pooled sequences
            dxv = tw["dframe_x"] if (c.contrastive_align_loss and not regression) else None
            if "video" in c.vmr_fusion:
                dxv = tw["ydseg"] if dxv is None else tr.add3(tw["dframe_sum"], dxv, tw["ydseg"])
            self._encode_bwd("video", ws, tw, dl_v, dxv.view(B, Tv, D) if dxv is not None else None, dvideo, fm, feats_v)
        # (the audio tower's weight-gradient products stay on the main stream since round 6 -- it idles at the end of the step while the second
        # stream is still in the video tower's chain; MADE_AUDIO_DW_SIDE=1: behind the video tower on the second stream, as before)
        self._encode_bwd("audio", ws, tw, dl_a, tw["xdseg"].view(B, Ta, D), dmusic, sm, feats_a,
                         dw_stream=side if _lib.variant_env("MADE_AUDIO_DW_SIDE", "0") != "0" else None)
        cur.wait_stream(side)
        if opt_st is not None:
            cur.wait_stream(opt_st)

    def _regression_train(self, out, ws, tw, mem3: Tensor, fus_mask: Tensor, v_duration, cur, side) -> Dict[str, Tensor]:
        """reference model/model_Uni.py:228-232,290-300 in train mode: memory summed over all L positions / number of valid ones ->
        3-layer ReLU MLP -> sigmoid; loss = 20 * L1 (mean over every element of the [B, 1, 2] spans).  No decoder on this path."""
        c, P = self.cfg, self.P
        B = mem3.shape[0]
        cnt = fus_mask.sum(dim=1, keepdim=True)
        fx = (ops.masked_mean(mem3, None) / cnt).to(self.tc)
        h1 = ops.linear(fx, P["reg_mlp.0.w"], P["reg_mlp.0.b"], act=ops.ACT_RELU)
        h2 = ops.linear(h1, P["reg_mlp.1.w"], P["reg_mlp.1.b"], act=ops.ACT_RELU)
        spans = torch.empty(B, 2, device=self.device, dtype=torch.float32)
        if c.predict_center == 1:
            ops.linear(h2, P["reg_mlp.2.w"], P["reg_mlp.2.b"], act=ops.ACT_SIGMOID, segs=[Seg(out=spans, ldo=2)])
            spans[:, 1] = v_duration.to(self.device, torch.float32) / c.max_m_duration
        else:
            ops.linear(h2, P["reg_mlp.2.w"], P["reg_mlp.2.b"], act=ops.ACT_SIGMOID, out=spans)
        tg = self._inputs[4].to(torch.float32)
        assert tg.shape == (B, 1, 2), f"spans_target.shape {tuple(tg.shape)} must equal to src_spans.shape {(B, 1, 2)}"
        l1 = (spans.view(B, 1, 2) - tg).abs().mean()
        self._reg = (fx, h1, h2, spans, cnt, tg)
        out.update(pred_spans=spans.view(B, 1, 2), regression_loss_span=l1, localization_loss=(l1 * 20).view(1))
        cur.wait_stream(side)
        return out

    def _regression_bwd(self, ws, tw, g_loc: Optional[Tensor], B: int, L: int) -> Tensor:
        """Gradient of g_loc * 20 * mean|spans - target| back through the regression MLP to the encoder memory: every one of the L
        positions of a sample receives d fusion / (number of valid positions).  Returns the memory gradient buffer."""
        c, P, G, D = self.cfg, self.P, self.G, self.cfg.D
        fx, h1, h2, spans, cnt, tg = self._reg
        n = 1 if c.predict_center == 1 else 2                  # predict_center: the width column is a constant of the batch
        scale = 20.0 / (B * 2)
        g = torch.sign(spans - tg.view(B, 2)) * scale
        if g_loc is not None:
            g = g * g_loc.to(torch.float32).view(())
        dz = torch.zeros(B, HEAD_PAD, device=self.device, dtype=self.tc)  # through the sigmoid; reduction dim padded like reg_mlp.2.wt
        dz[:, :n] = (g * spans * (1.0 - spans))[:, :n].to(self.tc)
        tr.gemm_tn(dz[:, :n], h2, G["reg_mlp.2.w"], accumulate=True, colsum=G["reg_mlp.2.b"])
        dh2 = ops.linear(dz, P["reg_mlp.2.wt"], None, out=torch.empty_like(h2), gate=_lib.GATE_RELU_OUT, G=h2)
        dh1 = self._lin_bwd(dh2, h1, "reg_mlp.1", dx_out=torch.empty_like(h1), gate=_lib.GATE_RELU_OUT, G=h1)
        dfx = self._lin_bwd(dh1, fx, "reg_mlp.0", dx_out=torch.empty(B, D, device=self.device, dtype=self.tc))
        dmem = tw["eg1"]
        dmem.view(B, L, D).copy_((dfx.float() / cnt)[:, None, :].expand(B, L, D))
        return dmem

    def _retrieval_bwd(self, *a, **k) -> None:
        for _ in self._retrieval_bwd_gen(*a, **k):
            pass

    def _retrieval_bwd_gen(self, ws, tw, g_ret: Optional[Tensor], B: int, S: int, sm: Tensor, dpool: Optional[Tensor] = None):
        c, P, G, D = self.cfg, self.P, self.G, self.cfg.D
        video, music = ws["video"], ws["music"]
        dvideo, dmusic = tw["dvideo"], tw["dmusic"]
        ls, gls = P["logit_scale"], G["logit_scale"]
        wgt = float(c.dual_single_loss_weight)
        ds_s, ds_d, ds_dt, cws = tw["dsims_s"], tw["dsims_d"], tw["dsims_dt"], tw["clip_ws"]
        ds_st = tw["dsims_st"] if "video" in c.vmr_fusion else None      # d sims_single^T: the second tower scores [track, video]
        single = dual = False
        fuse_dz = None
        if c.vmr_loss == "dual":
            tr.clip_loss_bwd(ws["sims_dual"], ls, wgt, g_ret, cws, ds_d, ds_dt, gls); dual = True
        elif c.vmr_loss == "single":
            tr.clip_loss_bwd(ws["sims_single"], ls, wgt, g_ret, cws, ds_s, ds_st, gls); single = True
        elif c.vmr_loss == "dual_single_loss_fuse":
            tr.clip_loss_bwd(ws["sims_dual"], ls, 1.0, g_ret, cws, ds_d, ds_dt, gls, row_exclude=getattr(self, "_row_exclude", None))
            tr.clip_loss_bwd(ws["sims_single"], ls, 1.0, g_ret, cws, ds_s, ds_st, gls)
            single = dual = True
        elif c.vmr_loss == "dual_single_feature_fuse":
            # sims[n, m] = <f^[m, n], v^[n]> with f = pooled[m, n] + music[m] (reference model_Uni.py:268-273).  Rare variant: the glue
            # between the kernels (outer products of the similarity gradient) is plain torch
            fused, fn, vn = ws["ff_fused"], ws["ff_fn"], ws["ff_vn"]
            tr.clip_loss_bwd(ws["sims_fused"], ls, wgt, g_ret, cws, ds_s, None, gls)
            dfn = (ds_s.t().reshape(B * B, 1) * vn.repeat(B, 1)).contiguous()              # rows (m, n): ds[n, m] v^[n]
            dvn = torch.einsum("nm,mnd->nd", ds_s, fn.view(B, B, D)).contiguous()
            dfused = torch.empty(B * B, D, device=self.device, dtype=torch.float32)
            tr.l2norm_bwd(fused, dfn, dfused)
            tr.l2norm_bwd(video, dvn, dvideo, accumulate=True)
            dmusic.add_(dfused.view(B, B, D).sum(dim=1))
            fuse_dz = dfused                                 # = the gradient of the pooled rows (LayerNorm3's output)
        else:                                                # dual_single_sim_fuse: one loss on the summed similarities
            both = tr.add3(tw["sims_both"], ws["sims_dual"], ws["sims_single"])
            tr.clip_loss_bwd(both, ls, wgt, g_ret, cws, ds_d, ds_dt, gls)
            _tape.copy_(ds_s, ds_d)
            if ds_st is not None:
                _tape.copy_(ds_st, ds_dt)
            single = dual = True
        frame, seg = self._views
        if (single or dpool is not None or fuse_dz is not None) and "music" in c.vmr_fusion:
            if not single:
                _tape.zero_(ds_s)                            # the tower feeds only the decoder's query / the fused similarity
            yield from self._xpool_bwd_gen(tw, "xa", "x", ds_s, video, dvideo, seg, sm if c.fusion_mask == 1 else None, B, S, dpool=dpool, dz=fuse_dz)
        else:
            _tape.zero_(tw["xdseg"])
        if "video" in c.vmr_fusion:
            if single and c.vmr_loss == "single":            # queries = the music vectors, pooled sequences = the frames
                self._xpool_bwd(tw, "xav", "y", ds_st, music, dmusic, frame, self._inputs[2] if c.fusion_mask == 1 else None, B, frame.shape[1])
            else:
                _tape.zero_(tw["ydseg"])
        self._ret_ck(9)
        if dual:
            tr.gemm_tn(ds_dt, tw["mn"], tw["dvn"])            # d vhat = dsims mhat
            tr.gemm_tn(ds_d, tw["vn"], tw["dmn"])             # d mhat = dsims^T vhat
            tr.l2norm_bwd(video, tw["dvn"], dvideo, accumulate=True)
            tr.l2norm_bwd(music, tw["dmn"], dmusic, accumulate=True)

    def _ret_ck(self, i: int) -> None:
        """measurement knob (MADE_RET_HANDOFF=i): from checkpoint i of the retrieval branch's backward on, its launches go to the MAIN
        stream (the ones before stay on the second stream) -- which of them disturbs the decoder's backward chain"""
        if getattr(self, "_ret_main_stream", None) is None:
            return
        main = self._ret_main_stream
        if _lib.variant_env("MADE_RET_HANDOFF", "") == str(i) and torch.cuda.current_stream() != main:
            main.wait_stream(torch.cuda.current_stream())
            torch.cuda.set_stream(main)
        # MADE_RET_HANDOFF_REV=i: the other way round -- the launches in front of checkpoint i on the main stream, the rest on the second
        rev = _lib.variant_env("MADE_RET_HANDOFF_REV", "")
        if rev != "":
            side = self._side_stream()
            if i == 0 and int(rev) > 0 and torch.cuda.current_stream() != main:
                main.wait_stream(torch.cuda.current_stream())
                torch.cuda.set_stream(main)
            elif i == int(rev) and i > 0 and torch.cuda.current_stream() == main:
                side.wait_stream(main)
                torch.cuda.set_stream(side)

    def _xpool_bwd(self, *a, **k) -> None:
        for _ in self._xpool_bwd_gen(*a, **k):
            pass

    def _xpool_bwd_gen(self, tw, key: str, pre: str, ds: Tensor, qvec: Tensor, dqvec: Tensor, seg: Tensor, seg_mask: Optional[Tensor], B: int, S: int,
                       dpool: Optional[Tensor] = None, dz: Optional[Tensor] = None):
        """backward of one X-Pool tower (_xpool_train): ds = d loss / d sims [query, sequence]; accumulates the gradient of the query
        vectors into dqvec and writes the gradient of the pooled sequences to tw[pre + "dseg"].  A generator: it yields once, in front
        of the two batched score / dP products (64 problems of 64 x 512 x 512, 2 x 33 MB streamed), so that backward() can issue what
        follows behind the decoder's chain (MADE_RET_SPLIT)."""
        P, G, D = self.P, self.G, self.cfg.D
        skip = seg_mask.reshape(-1) if seg_mask is not None else None
        g1, g2, g3 = tw[pre + "g1"], tw[pre + "g2"], tw[pre + "g3"]
        if dz is not None:
            # the pooled rows were consumed outside the fused tail (dual_single_feature_fuse): plain LayerNorm3 backward of their gradient,
            # plus the decoder query's share when it is an xpool query
            if dpool is not None:
                dz = dz + (dpool / B)[:, None, :].expand(B, B, dpool.shape[-1]).reshape(B * B, -1)
            tr.layernorm_bwd(tw[pre + "y"], P[key + ".ln3.g"], dz.contiguous(), g1, dgamma=G[key + ".ln3.g"], dbeta=G[key + ".ln3.b"], dx_drop=g2,
                             drop=self._drop("xa.linear_out", dr.P_XPOOL), drop_ld=D)
        else:
            tr.xpool_tail_bwd(tw[pre + "y"], P[key + ".ln3.g"], P[key + ".ln3.b"], qvec, ds, g1, B, B, dy_drop=g2, drop=self._drop("xa.linear_out", dr.P_XPOOL),
                              dgamma=G[key + ".ln3.g"], dbeta=G[key + ".ln3.b"], dvideo=dqvec, dpool=dpool, dpool_scale=1.0 / B)
        self._ret_ck(1)
        da3 = self._lin_bwd(g2, tw[pre + "a3"], key + ".lin", dx_out=g3, R=g1)
        tr.layernorm_bwd(tw[pre + "a2"], P[key + ".ln2.g"], da3, g1, dgamma=G[key + ".ln2.g"], dbeta=G[key + ".ln2.b"])
        self._ret_ck(2)
        do = self._lin_bwd(g1, tw[pre + "o"], key + ".out", dx_out=g2)
        Sp = tw[pre + "S"].shape[1]
        yield
        self._ret_ck(3)
        xk, xu, q = tw[pre + "k"], tw[pre + "u"], tw[pre + "q"]
        ops.linear(q, xk[:S], None, M=B, N=S, K=D, batch=B, a_z_stride=0, w_z_stride=S * D, segs=[Seg(out=tw[pre + "S"], ldo=Sp, out_z_stride=B * Sp)])
        ops.linear(do[:B], xu[:S], None, M=B, N=S, K=D, batch=B, a_z_stride=B * D, w_z_stride=S * D, segs=[Seg(out=tw[pre + "dP"], ldo=Sp, out_z_stride=B * Sp)])
        self._ret_ck(4)
        tr.softmax_bwd(tw[pre + "S"], tw[pre + "dP"], seg_mask, B, 1.0 / math.sqrt(D), tw[pre + "P"], tw[pre + "dS"], tw[pre + "dSt"], B, S, ldo=Sp, ldt=B)
        dkv = tw[pre + "dkv"]
        self._ret_ck(5)
        # dU[m] = P[m]^T dO[m];  dK[m] = dS[m]^T q;  dq = sum_m dS[m] K[m]
        tr.gemm_tn(tw[pre + "P"][:B, :S], do[:B], dkv[:S, D:], batch=(B, 1), a_zs=(B * Sp, 0), b_zs=(B * D, 0), c_zs=(S * 2 * D, 0))
        tr.gemm_tn(tw[pre + "dS"][:B, :S], q, dkv[:S, :D], batch=(B, 1), a_zs=(B * Sp, 0), b_zs=(0, 0), c_zs=(S * 2 * D, 0))
        _tape.zero_(tw[pre + "dq32"])
        tr.gemm_tn(tw[pre + "dSt"][0], xk[:S], tw[pre + "dq32"], batch=(B, 1), a_zs=(S * B, 0), b_zs=(S * D, 0), c_zs=(0, 0), accumulate=True,
                   row_mask=seg_mask, mask_zs=(S, 0))
        self._ret_ck(6)
        ds1 = self._lin_bwd(dkv, tw[pre + "s1"], key + ".kv", dx_out=tw[pre + "ds1"], row_mask=skip, skip=skip)
        self._ret_ck(7)
        tr.layernorm_bwd(seg, P[key + ".ln1.g"], ds1, tw[pre + "dseg"], dgamma=G[key + ".ln1.g"], dbeta=G[key + ".ln1.b"], row_skip=skip)
        self._ret_ck(8)
        dq = tr.add3(tw[pre + "dq"], tw[pre + "dq32"])
        dv1 = self._lin_bwd(dq, tw[pre + "v1"], key + ".q", dx_out=tw[pre + "dv1"])
        tr.layernorm_bwd(qvec, P[key + ".ln1.g"], dv1, dqvec, dgamma=G[key + ".ln1.g"], dbeta=G[key + ".ln1.b"], add=dqvec)

    def _encode_train_mlp(self, feats: Tensor, mask: Tensor, which: str, ws, tw, row_off: int) -> None:
        """agg_module = "mlp" (reference model/model_Base.py:567-570,606-609 with EmbeddingNet :216-249).  In train mode both
        BatchNorm1d layers normalise every position with the statistics of the batch's B * F values there (padded samples
        included: every row is computed, none gathered away) and move their running buffers; with `training_dropout` off
        (= the reference's model.eval()) they use the running statistics."""
        c, P = self.cfg, self.P
        B, T, Kin = feats.shape
        D = c.D
        tag, proj = ("v", "vit_proj") if which == "video" else ("a", "ast_proj")
        key, mod, Tbn = self._mlp_towers()[0 if which == "video" else 1]
        if T != Tbn:
            raise ValueError(f"agg_module=mlp: {which} sequence length {T} != {Tbn} positions of its BatchNorm1d (the reference fails there too)")
        nrow = B * T
        mflat = mask.reshape(-1)
        act = ops.ACT_QUICKGELU if c.with_act_after_proj else ops.ACT_NONE
        zp = tw[tag + ".zproj"] if c.with_act_after_proj else None
        if self.tc == torch.bfloat16:
            xin = ops.cast_mask_rows(feats.view(nrow, Kin), mflat, tw[tag + ".xin"])
            x0 = ops.linear(xin, P[proj + ".w"], P[proj + ".b"], act=act, Zout=zp, out=tw[tag + ".x0"])
        else:
            x0 = ops.linear(feats.view(nrow, Kin), P[proj + ".w"], P[proj + ".b"], a_row_mask=mflat, act=act, Zout=zp, out=tw[tag + ".x0"])
        stats = bool(self.training_dropout)
        bn, buf = tw[tag + ".bn"], self.buffers
        h0 = ops.linear(x0, P[key + ".0.w"], P[key + ".0.b"], out=tw[tag + ".h0"])
        h1 = tr.posbn_relu(h0.view(B, T, -1), P[key + ".1.g"], P[key + ".1.beta"], buf[mod + ".net.1.running_mean"], buf[mod + ".net.1.running_var"],
                           0.1, stats, bn[0], bn[1], out=tw[tag + ".h1"].view(B, T, -1))
        y0 = ops.linear(h1.view(nrow, -1), P[key + ".3.w"], P[key + ".3.b"], out=tw[tag + ".y0"])
        y1 = tr.posbn_relu(y0.view(B, T, D), P[key + ".4.g"], P[key + ".4.beta"], buf[mod + ".net.4.running_mean"], buf[mod + ".net.4.running_var"],
                           0.99, stats, bn[2], bn[3], out=tw[tag + ".y1"].view(B, T, D))      # momentum 0.99: model_Base.py:228
        if stats:
            buf[mod + ".net.1.num_batches_tracked"] += 1; buf[mod + ".net.4.num_batches_tracked"] += 1
            self.generation += 1                              # state moved in place: whoever caches derived weights (Uni_model's engines) reloads
        if "concat" in c.mml_fusion:
            local = ws["fus"][:, row_off:row_off + T]
        else:
            local = ws["frame_buf"] if which == "video" else ws["seg_buf"]
        ops.linear(y1.view(nrow, D), P[key + ".6.w"], P[key + ".6.b"], out_row_mask=mflat,
                   segs=[Seg(out=local, ldo=local.stride(1), rows_per_batch=T, out_batch_stride=local.stride(0))])
        vec = ws["video"] if which == "video" else ws["music"]
        ops.masked_mean(local, mask, out=tw[tag + ".mean"])
        ops.l2norm_rows(tw[tag + ".mean"], out_f32=vec)

    def _encode_bwd_mlp(self, which: str, ws, tw, d_local: Tensor, d_extra: Optional[Tensor], dvec: Tensor, mask: Tensor, feats: Tensor) -> None:
        c, P, G = self.cfg, self.P, self.G
        B, T, Kin = feats.shape
        D = c.D
        tag, proj = ("v", "vit_proj") if which == "video" else ("a", "ast_proj")
        key, mod, _ = self._mlp_towers()[0 if which == "video" else 1]
        nrow = B * T
        mflat = mask.reshape(-1)
        stats = bool(self.training_dropout)
        bn = tw[tag + ".bn"]
        dl, gd1, gd2, gh1, gh2 = tw[tag + ".dl"], tw[tag + ".gd1"], tw[tag + ".gd2"], tw[tag + ".gh1"], tw[tag + ".gh2"]
        tr.pool_bwd(tw[tag + ".mean"], dvec, mask, dl.view(B, T, D), in1=d_local, in2=d_extra)        # zero at padded tokens
        dy1 = self._lin_bwd(dl, tw[tag + ".y1"], key + ".6", dx_out=gd1)
        dy0 = tr.posbn_relu_bwd(tw[tag + ".y0"].view(B, T, D), tw[tag + ".y1"].view(B, T, D), dy1.view(B, T, D), P[key + ".4.g"], bn[2], bn[3], stats,
                                gd2.view(B, T, D), G[key + ".4.g"], G[key + ".4.beta"])
        dh1 = self._lin_bwd(dy0.view(nrow, D), tw[tag + ".h1"], key + ".3", dx_out=gh1)
        dh0 = tr.posbn_relu_bwd(tw[tag + ".h0"].view(B, T, -1), tw[tag + ".h1"].view(B, T, -1), dh1.view(B, T, -1), P[key + ".1.g"], bn[0], bn[1], stats,
                                gh2.view(B, T, -1), G[key + ".1.g"], G[key + ".1.beta"])
        dx = self._lin_bwd(dh0.view(nrow, -1), tw[tag + ".x0"], key + ".0", dx_out=gd1)
        if c.with_act_after_proj:
            dx = tr.gate_rows(dx, gd2, G=tw[tag + ".zproj"], gate=_lib.GATE_QUICKGELU_Z)
        # with batch statistics the padded tokens (projection of a zero row = the bias) shape the normalisation: they send a
        # gradient to the projection's bias, none to its weight
        xin = tw[tag + ".xin"] if self.tc == torch.bfloat16 else feats.view(nrow, Kin)
        tr.gemm_tn(dx, xin, G[proj + ".w"], accumulate=True, rows=self._rw(mflat))
        tr.colsum(dx, G[proj + ".b"])

    def _encode_bwd(self, which: str, ws, tw, d_local: Tensor, d_extra: Optional[Tensor], dvec: Tensor, mask: Tensor, feats: Tensor,
                    dw_stream=None) -> None:
        """dw_stream: run the weight-gradient products there instead of on the current stream (a one-layer block only: a deeper stack
        reuses the gradient buffers they read from layer to layer)."""
        c, P, G = self.cfg, self.P, self.G
        if c.agg_module == "mlp":
            return self._encode_bwd_mlp(which, ws, tw, d_local, d_extra, dvec, mask, feats)
        B, T, Kin = feats.shape
        D, Hh = c.D, c.SA_temporal_heads
        proj, mod, depth, tag = (("vit_proj", "video_transformer", c.video_transformer_depth, "v") if which == "video"
                                 else ("ast_proj", "audio_transformer", c.audio_transformer_depth, "a"))
        name = "video" if which == "video" else "audio"
        cls = bool(c.with_cls_token)
        T1 = T + 1 if cls else T
        rows = B * T
        mask0, mflat0 = mask, mask.reshape(-1)
        if cls:
            mask = tw[tag + ".mask1"]
        mflat = mask.reshape(-1)
        pt = dr.P_TEMPORAL
        dl = tw[tag + ".dl"]
        if cls:                                              # vec = l2norm(y[:, 0]); local = y[:, 1:]
            dl3, y3 = dl.view(B, T1, D), tw[tag + ".y1"].view(B, T1, D)
            tr.l2norm_bwd(y3[:, 0], dvec, None, dx_alt=dl3[:, 0])
            tr.pool_bwd(tw[tag + ".mean"], tw[tag + ".dtok"], mask0, dl3[:, 1:], in1=d_local, in2=d_extra)    # mask * (d_local + d_extra)
        else:
            tr.pool_bwd(tw[tag + ".mean"], dvec, mask, dl.view(B, T, D), in1=d_local, in2=d_extra)
        g1, g1b, g2, g3, g3b, gq, gf = (tw[tag + ".g1"], tw[tag + ".g1b"], tw[tag + ".g2"], tw[tag + ".g3"], tw[tag + ".g3b"], tw[tag + ".gqkv"],
                                        tw[tag + ".gffn"])
        pend: list = []                                      # this layer's weight gradients: one grouped launch (see the DETR encoder)
        # local = mask(final(x4)); x4 = x3 + drop(ffn2): the product dl W_f is needed raw (residual) and dropped (branch)
        x_last = tw[tag + ".xlast"]
        dx = None
        proj_grouped = False
        for l in range(depth - 1, -1, -1):
            p, t = f"{mod}.layers.{l}", f"{tag}.{l}"
            x4 = x_last if l == depth - 1 else tw[f"{tag}.{l + 1}.x0"]
            if l == depth - 1:
                df = self._lin_bwd(dl, x4, mod + ".final", dx_out=g1, row_mask=mflat, skip=mflat, Zout=g2,
                                   drop=self._drop(f"{name}.{l}.ffn_out", pt), defer=pend)
                dx4 = g2
            else:                                            # deeper stacks: x0_{l+1} = x4_l, its gradient arrives from LN1 of layer l+1
                dx4 = dx
                df = tr.gate_rows(dx4, g1, drop=self._drop(f"{name}.{l}.ffn_out", pt), drop_ld=D, row_skip=mflat)
            dz1 = self._lin_bwd(df, tw[t + ".h"], p + ".ff2", dx_out=gf, row_mask=mflat, skip=mflat, gate=_lib.GATE_GELU_Z, G=tw[t + ".z1"],
                                drop=self._drop(f"{name}.{l}.ffn_act", pt), defer=pend)
            dx3 = self._lin_bwd(dz1, tw[t + ".x3"], p + ".ff1", dx_out=g3, row_mask=mflat, skip=mflat, R=dx4, defer=pend)
            tr.layernorm_bwd(tw[t + ".x2"], P[p + ".ln2.g"], dx3, g1b, dgamma=G[p + ".ln2.g"], dbeta=G[p + ".ln2.b"], row_skip=mflat)
            datt = self._lin_bwd(g1b, tw[t + ".att"], p + ".out", dx_out=g2, row_mask=mflat, skip=mflat, defer=pend)
            qkv = tw[t + ".qkv"]
            q3, gq3 = qkv.view(B, T1, 3 * D), gq.view(B, T1, 3 * D)
            tr.attention_bwd(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], tw[t + ".att"].view(B, T1, D), datt.view(B, T1, D),
                             gq3[:, :, :D], gq3[:, :, D:2 * D], gq3[:, :, 2 * D:], tw[t + ".lse"], tw[tag + ".delta"], Hh,
                             key_mask=mask, q_skip_mask=mask, drop=self._drop(f"{name}.{l}.attn", pt), order=self._order[mask.data_ptr()],
                             keep_bits=tw[t + ".kbits"] if self._bits else None)
            dx1 = self._lin_bwd(gq, tw[t + ".x1"], p + ".in", dx_out=g3b, row_mask=mflat, skip=mflat, R=g1b, defer=pend)
            dx = tr.layernorm_bwd(tw[t + ".x0"], P[p + ".ln1.g"], dx1, g2, dgamma=G[p + ".ln1.g"], dbeta=G[p + ".ln1.b"], row_skip=mflat)
            if (l == 0 and not cls and not c.with_act_after_proj and self.tc == torch.bfloat16 and len(pend) < 8 and Kin % 256 == 0 and D % 256 == 0
                    and self._rw(mflat) is not None and self._groupable(dx, tw[tag + ".xin"], G[proj + ".w"])):
                # the input projection's weight gradient reduces over the same rows as the first layer's: one more problem of its grouped launch
                # instead of a launch of its own (round 6: 63 us at the very end of the main stream for the audio tower)
                pend.append((dx, tw[tag + ".xin"], G[proj + ".w"], G[proj + ".b"]))
                proj_grouped = True
            if dw_stream is not None and depth == 1 and not cls and not c.with_act_after_proj:
                dw_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(dw_stream):
                    self._flush_dw(pend, mflat)
                    if not proj_grouped:
                        xin_ = tw[tag + ".xin"] if self.tc == torch.bfloat16 else feats.view(rows, Kin)
                        tr.gemm_tn(dx, xin_, G[proj + ".w"], accumulate=True, colsum=G[proj + ".b"], rows=self._rw(mflat0))
                return
            self._flush_dw(pend, mflat)
        if proj_grouped:
            return
        xin = tw[tag + ".xin"] if self.tc == torch.bfloat16 else feats.view(rows, Kin)
        if cls:                                              # x0 = [token + pe_0 ; proj(x) + pe_1..T]
            dx3d = dx.view(B, T1, D)
            G["cls_" + name].view(-1).add_(dx3d[:, 0].float().sum(0))
            dx = tw[tag + ".dproj"]
            dx.view(B, T, D).copy_(dx3d[:, 1:])
        if c.with_act_after_proj:                            # x0 = quickgelu(z) + pe: gradient w.r.t. z (the input itself needs none)
            dx = tr.gate_rows(dx, g1[:rows], G=tw[tag + ".zproj"], gate=_lib.GATE_QUICKGELU_Z, row_skip=mflat0)
        tr.gemm_tn(dx, xin, G[proj + ".w"], accumulate=True, colsum=G[proj + ".b"], rows=self._rw(mflat0))

    # ================================================================== convenience
    def loss_and_grads(self, inp: dict, seed: int = 0, w_ret: float = 1.0, w_loc: float = 1.0) -> dict:
        """numpy in, numpy out (synchronises): losses and all parameter gradients of w_ret*retrieval + w_loc*localization."""
        dev = self.device
        t = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
        o = self.forward_train(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=seed,
                               music_ids=inp.get("music_ids"), v_duration=t.get("v_duration"))
        gr = torch.tensor([w_ret], device=dev) if w_ret != 1.0 else None
        gl = torch.tensor([w_loc], device=dev) if w_loc != 1.0 else None
        self.backward(gr, gl)
        torch.cuda.synchronize()
        return dict(retrieval_loss=float(o["retrieval_loss"].cpu()), localization_loss=float(o["localization_loss"].cpu()),
                    grads=self.grads_numpy(), loss_dict={k: float(v.cpu()) if torch.is_tensor(v) else float(v) for k, v in self.loss_dict(o).items()})


class TrainStepGraph:
    """One training iteration of the reference's loop body (train-MaDe.py:337-381) captured once and replayed: about 600 kernel
    launches cost the host one graph launch.  Everything that changes from step to step lives in device memory the host
    refreshes before a replay: the batch (fixed input buffers), the dropout seed (MadeDropout.seed_device), the Adam step count
    and the three learning rates of the LambdaLR schedule (MadeAdamDeviceState), the same-music exclusion matrix.
    Data-parallel jobs replay three graphs with the two bucketed RCCL all-reduces of the flat gradient buffer between them,
    the large bucket under the encoders' backward exactly as the eager path has it."""

    def __init__(self, trainer: MadeTrainer, frame_feats, segment_feats, frame_masks, segment_masks, spans_target, *,
                 max_grad_norm: float = 1.0, music_ids=None, v_duration: Optional[Tensor] = None, dist=None, mode: str = "graph"):
        """mode = "graph": hipGraph capture (torch.cuda.CUDAGraph); mode = "tape": the library's own launch tape (mgsv_amd/tape.py,
        made_tape_*): the step is EXECUTED once while every launch is recorded, then replayed from one C loop onto the same two
        streams -- no framework kernel may run inside the step (none does), single process only."""
        assert mode in ("graph", "tape")
        self.mode = mode
        t = self.trainer = trainer
        dev = t.device
        self.dist = dist if (dist is not None and dist.get_world_size() > 1) else None
        self.inputs = {k: v.to(dev).contiguous().clone() for k, v in (("frame_feats", frame_feats), ("segment_feats", segment_feats),
                                                                    ("frame_masks", frame_masks), ("segment_masks", segment_masks),
                                                                    ("spans_target", spans_target))}
        self.v_duration = v_duration.to(dev).contiguous().clone() if v_duration is not None else None
        self.seed_dev = torch.zeros(1, device=dev, dtype=torch.int64)
        # MadeAdamDeviceState {int64 step; float lr[4]; float bc1, bc2_sqrt} in device memory, laid out by the ctypes mirror (which
        # tests/test_abi_cpu.py holds to the header): the step count is word 0, the learning rates start at the `lr` field
        import ctypes as C
        S = _lib.MadeAdamDeviceState
        assert C.sizeof(S) % 8 == 0 and S.step.offset == 0 and S.lr.offset % 4 == 0
        self.adam_state = torch.zeros(C.sizeof(S) // 8, device=dev, dtype=torch.int64)
        lr0 = S.lr.offset // 4
        self._lr_view = self.adam_state.view(torch.float32)[lr0:lr0 + 3]
        self._lrs: Optional[tuple] = None                   # what the device words hold (refilled only when the schedule moves)
        ex = t.same_music_exclusion(music_ids)
        self.exclusion = ex.clone() if ex is not None else None
        scale = 1.0 / self.dist.get_world_size() if self.dist is not None else 1.0
        i = self.inputs

        early = self.dist is None and t._early_opt_ok()       # (opt-in: the step applied in two parts, see MadeTrainer.train_step)

        def fwd_bwd():
            t._zero_grad_in_forward = True
            try:
                out = t.forward_train(i["frame_feats"], i["segment_feats"], i["frame_masks"], i["segment_masks"], i["spans_target"], seed=0,
                                      v_duration=self.v_duration)
            finally:
                t._zero_grad_in_forward = False
            if early:
                t.backward(None, None, early_opt=lambda: t.optimizer_step(0.0, 0.0, 0.0, max_grad_norm=max_grad_norm, grad_scale=scale,
                                                                           device_state=self.adam_state, part="early"))
            else:
                t.backward(None, None)
            return out

        def opt():
            t.optimizer_step(0.0, 0.0, 0.0, max_grad_norm=max_grad_norm, grad_scale=scale, device_state=self.adam_state,
                             part="rest" if early else None)

        # warm-up run (loads the kernels, allocates the workspaces, creates the side streams) on a copy of the state
        keep = [x.clone() for x in (t.flat_param, t.exp_avg, t.exp_avg_sq)]
        keep_buf = {k: v.clone() for k, v in t.buffers.items()}
        keep_step, keep_gen = t.opt_step, t.generation
        t._seed_dev, t._static_exclusion = self.seed_dev, self.exclusion
        try:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                fwd_bwd(); opt()
            cur.wait_stream(side)
            torch.cuda.synchronize()
            for dst, src in zip((t.flat_param, t.exp_avg, t.exp_avg_sq), keep):
                dst.copy_(src)
            for k, v in keep_buf.items():
                t.buffers[k].copy_(v)
            t.opt_step, t.generation = keep_step, keep_gen
            t.repack()
            self.adam_state[0] = keep_step
            torch.cuda.synchronize()
            self.graphs = []
            if mode == "tape":
                keep2 = [x.clone() for x in (t.flat_param, t.exp_avg, t.exp_avg_sq)]
                if self.dist is not None:
                    # data parallel: the two bucketed all-reduces of the flat gradient buffer (RCCL calls of the framework) ride the tape
                    # as host callbacks at the points where the eager step makes them -- the large bucket inside backward's grad_sync
                    # hook (both streams joined, the temporal encoders' backward still to come), the temporal range behind backward,
                    # then the wait that orders the optimizer's launches behind both (reference train-MaDe.py:238-241,371)
                    dist_, works = self.dist, []
                    cut = t.group_ranges[0][1]

                    def fwd_bwd():                              # noqa: F811 (the data-parallel form of the step's first part)
                        t._zero_grad_in_forward = True
                        try:
                            out = t.forward_train(i["frame_feats"], i["segment_feats"], i["frame_masks"], i["segment_masks"], i["spans_target"], seed=0,
                                                  v_duration=self.v_duration)
                        finally:
                            t._zero_grad_in_forward = False
                        t.backward(None, None, grad_sync=lambda: _tape.callback(lambda: works.append(dist_.all_reduce(t.flat_grad[cut:], async_op=True))))
                        _tape.callback(lambda: works.append(dist_.all_reduce(t.flat_grad[:cut], async_op=True)))

                        def wait_all():
                            for w in works:
                                w.wait()
                            works.clear()
                        _tape.callback(wait_all)
                        return out
                # the tape replays only the library's launches: a step that still runs a framework kernel (some non-headline
                # configurations do: the BatchNorm affine of agg_module='mlp', Q > 1 copies, the regression head ...) is refused here
                # (ForeignKernelError names the operators) instead of silently skipping that work on every replay
                # The recording runs the step for real (in data-parallel mode: with its all-reduces), so the state is put back afterwards
                # -- also when the recording is refused (ForeignKernelError on leaving the `with`) or fails: the caller's fallback
                # (mode='graph' / the eager step) must start from the parameters, Adam moments and step count it had before.
                try:
                    with _tape.LaunchTape.record(check=os.environ.get("MADE_TAPE_CHECK", "1") != "0") as tp:
                        self.out = fwd_bwd(); opt()
                    torch.cuda.synchronize()
                    self.tape = tp
                    tw_ = int(_lib.variant_env("MADE_TAPE_INTERLEAVE", "2"))   # 0: replay in program order (knob for A/B measurements)
                    if tw_ > 0:
                        tp.interleave(tw_)
                finally:
                    torch.cuda.synchronize()
                    for dst, src in zip((t.flat_param, t.exp_avg, t.exp_avg_sq), keep2):
                        dst.copy_(src)
                    for k, v in keep_buf.items():
                        t.buffers[k].copy_(v)
                    t.opt_step, t.generation = keep_step, keep_gen
                    t.repack()
                    self.adam_state[0] = keep_step
                    torch.cuda.synchronize()
            elif self.dist is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.out = fwd_bwd(); opt()
                self.graphs.append(g)
            else:
                # three graphs cut where the eager path starts its collectives: [forward .. backward of everything but the temporal
                # encoders] | all-reduce of the matching + detection ranges starts | [encoders' backward] | all-reduce of the
                # temporal range | [optimizer].  The capture is switched from the first graph to the second inside backward's
                # grad_sync hook (both streams are joined there).
                ga, gb, gc_ = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                ctx = [torch.cuda.graph(ga)]
                ctx[0].__enter__()

                def cut():
                    ctx[0].__exit__(None, None, None)
                    ctx[0] = torch.cuda.graph(gb, pool=ga.pool())
                    ctx[0].__enter__()
                try:
                    self.out = t.forward_train(i["frame_feats"], i["segment_feats"], i["frame_masks"], i["segment_masks"], i["spans_target"],
                                               seed=0, v_duration=self.v_duration)
                    t.backward(None, None, grad_sync=cut)
                finally:
                    ctx[0].__exit__(None, None, None)
                with torch.cuda.graph(gc_, pool=ga.pool()):
                    opt()
                self.graphs += [ga, gb, gc_]
            t.opt_step, t.generation = keep_step, keep_gen        # capturing runs the Python side of optimizer_step, not the kernels
            self._dev_step = keep_step
        finally:
            t._seed_dev, t._static_exclusion = None, None

    def close(self) -> None:
        """Waits for the last replay, then releases the tape / graphs and every buffer they reference.  Dropping a TrainStepGraph without
        this is safe too (LaunchTape.close synchronises before it frees), this just makes the point in time explicit."""
        torch.cuda.synchronize()
        if getattr(self, "tape", None) is not None:
            self.tape.close()
            self.tape = None
        self.graphs = []

    @staticmethod
    def _store_words(dst: Tensor, words) -> None:
        """dst's first len(words) 32-bit words = words, stream-ordered (the values travel as kernel arguments: made_store_words)"""
        import ctypes as C
        arr = (C.c_uint32 * len(words))(*[int(w) & 0xFFFFFFFF for w in words])
        _lib.check(_lib.lib().made_store_words(dst.data_ptr(), arr, len(words), torch.cuda.current_stream().cuda_stream), "made_store_words")

    def step(self, frame_feats, segment_feats, frame_masks, segment_masks, spans_target, seed: int, lrs=(1e-4, 1e-4, 1e-4),
             music_ids=None, v_duration: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """Same contract as MadeTrainer.train_step; the returned tensors are the graph's fixed output buffers."""
        t, i = self.trainer, self.inputs
        for k, v in (("frame_feats", frame_feats), ("segment_feats", segment_feats), ("frame_masks", frame_masks),
                     ("segment_masks", segment_masks), ("spans_target", spans_target)):
            if v.data_ptr() != i[k].data_ptr():
                if tuple(v.shape) != tuple(i[k].shape):
                    raise ValueError(f"TrainStepGraph: {k} has shape {tuple(v.shape)}, captured for {tuple(i[k].shape)}")
                i[k].copy_(v, non_blocking=True)
        if (v_duration is None) != (self.v_duration is None):
            raise ValueError("TrainStepGraph: v_duration must be given exactly when it was at capture")
        if v_duration is not None:
            self.v_duration.copy_(v_duration, non_blocking=True)
        if self.exclusion is not None:
            ex = t.same_music_exclusion(music_ids)
            if ex is None:
                raise ValueError("TrainStepGraph: captured with music_ids; pass them on every step")
            self.exclusion.copy_(ex, non_blocking=True)
        # the per-step scalars travel as kernel arguments of tiny fill launches (stream-ordered; a pinned staging word would be
        # overwritten by the host while earlier replays are still queued)
        # (made_store_words: no framework kernel inside or around the replayed step)
        sd = int(seed) & 0x7FFFFFFFFFFFFFFF
        self._store_words(self.seed_dev, (sd & 0xFFFFFFFF, sd >> 32))
        lrs = tuple(float(x) for x in lrs)
        if lrs != self._lrs:
            import struct
            self._store_words(self._lr_view, tuple(struct.unpack("<I", struct.pack("<f", x))[0] for x in lrs))
            self._lrs = lrs
        if t.opt_step != self._dev_step:                      # eager optimizer steps in between: realign the device-side count
            st_ = int(t.opt_step)
            self._store_words(self.adam_state[0:1], (st_ & 0xFFFFFFFF, (st_ >> 32) & 0xFFFFFFFF))
        t.seed = int(seed)
        if self.mode == "tape":
            self.tape.replay()
        else:
            self.graphs[0].replay()
        if self.dist is not None and self.mode != "tape":         # (the tape makes the all-reduces itself: host callbacks)
            cut = t.group_ranges[0][1]
            w1 = self.dist.all_reduce(t.flat_grad[cut:], async_op=True)      # travels under the encoders' backward
            self.graphs[1].replay()
            w2 = self.dist.all_reduce(t.flat_grad[:cut], async_op=True)
            w1.wait(); w2.wait()
            self.graphs[2].replay()
        t.opt_step += 1
        t.generation += 1
        self._dev_step = t.opt_step
        return self.out
