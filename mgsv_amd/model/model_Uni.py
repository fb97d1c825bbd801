"""Drop-in `Uni_model` for the reference's drivers (reference model/model_Uni.py:14-322).

Same constructor `Uni_model(args, device=None, logger=None)`, same `forward(...)` signature and
5-dict return value, same parameter-group getters, same attributes the drivers poke
(`criterion.foreground_label`, `criterion.weight_dict`, `video_guided_to_music_pooling_cross_transformer`)
and the same `state_dict()` key layout, so a reference checkpoint loads (the frozen `vit_model.*` /
`ast_model.*` tensors a real checkpoint also carries are ignored: they are never used on the feature
path).  Underneath, every op runs as a hand-written HIP kernel through
libmade_hip.so (mgsv_amd/engine.py); there is no ATen fallback -- on a box without the library or a
GPU the model raises.

Inference (`model.eval()` / `torch.no_grad()`): mgsv_amd/engine.py.  Training (`model.train()` with autograd on):
mgsv_amd/trainer.py -- the two losses come back attached to the autograd tape through one custom Function whose backward
runs the hand-written backward pass and points every parameter's `.grad` at its slice of the trainer's flat gradient
buffer, so the reference's loop body works unchanged (train-MaDe.py:337-381: `loss.backward()`, the three
`clip_grad_norm_` calls, `optimizer.step()`, `optimizer.zero_grad()`).  Parameters alias the trainer's flat f32 master
buffer, so in-place optimizer updates are seen by the next forward (the bf16 copies are re-derived when a parameter's
version counter moved).  Configurations the HIP training path does not cover raise NotImplementedError.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import synth
from ..config import MadeConfig
from ..engine import MadeEngine


class _TrainStep(torch.autograd.Function):
    """(retrieval_loss, localization_loss) = forward_train(batch); backward = MadeTrainer.backward.
    The parameters are inputs only so that autograd schedules the node; their gradients are written straight into
    `param.grad` (views of the trainer's flat buffer, accumulated if a gradient is already there)."""

    @staticmethod
    def forward(ctx, owner, batch, seed, *params):
        trn = owner._trainer
        out = trn.forward_train(*batch, seed=seed, v_duration=getattr(owner, "_train_vdur", None))
        ctx.owner = owner
        owner._last_train_out = out
        return out["retrieval_loss"].clone().view(()), out["localization_loss"].clone().view(())

    @staticmethod
    def backward(ctx, g_ret, g_loc):
        owner = ctx.owner
        trn = owner._trainer
        dev = trn.device
        zero = torch.zeros(1, device=dev)
        gr = g_ret.reshape(1).to(dev, torch.float32) if g_ret is not None else zero
        gl = g_loc.reshape(1).to(dev, torch.float32) if g_loc is not None else zero
        trn.backward(gr, gl)
        with torch.no_grad():
            for name, p in owner.named_parameters():
                g = trn.grad.get(name)
                if g is None:
                    continue
                if p.grad is None:
                    p.grad = g
                elif p.grad.data_ptr() != g.data_ptr():
                    p.grad.add_(g)
        return (None, None, None) + tuple(None for _ in range(len(ctx.needs_input_grad) - 3))

_FROZEN_PREFIXES = ("vit_model.", "ast_model.")


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer: bool) -> None:
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))


class _Criterion(nn.Module):
    """Holds what the drivers read from `model.criterion` (reference music_detr/loss_detr.py:36-57)."""

    def __init__(self, cfg: MadeConfig):
        super().__init__()
        from ..music_detr.loss_detr import weight_dict
        self.foreground_label = cfg.foreground_label
        self.background_label = cfg.background_label
        self.weight_dict = weight_dict(cfg)


class _XPoolModule(nn.Module):
    """`model.video_guided_to_music_pooling_cross_transformer`: the drivers call it directly on the
    whole split (reference train-MaDe.py:588-591, test-MaDe.py:392-395).  `.cpu()` / `.to()` are accepted
    and ignored: the computation always runs on the GPU in track chunks."""

    def __init__(self, owner: "Uni_model"):
        super().__init__()
        object.__setattr__(self, "_owner", owner)

    def forward(self, video_embeds, music_embeds, music_mask=None):
        eng = self._owner._engine_ready()
        dev = eng.device
        v = video_embeds.to(dev, torch.float32).contiguous()
        s = music_embeds.to(dev).to(eng.tc).contiguous()
        m = music_mask.to(dev, torch.float32).contiguous() if music_mask is not None else None
        Nm, Nv, D = s.shape[0], v.shape[0], v.shape[1]
        pooled = torch.empty(Nm * Nv, D, device=dev, dtype=torch.float32)
        eng.xpool_sims(v, s, m, pooled_out=pooled)
        return pooled.view(Nm, Nv, D).to(video_embeds.device)

    def cpu(self):
        return self

    def to(self, *a, **k):
        return self


class Uni_model(nn.Module):
    def __init__(self, args, device=None, logger=None, compute_dtype: Optional[str] = None):
        super().__init__()
        self.args = args
        self.device = torch.device(device) if device is not None else torch.device("cuda:0")
        self.logger = logger
        assert args.hidden_dim == args.dim_input, "hidden_dim must equal to dim_input"
        self.cfg = MadeConfig.from_args(args)
        self.dim_input = self.cfg.D
        self.num_moment_queries = self.cfg.num_moment_queries
        self.aux_loss = self.cfg.aux_loss
        self.compute_dtype = compute_dtype or getattr(args, "compute_dtype", "f32")
        # parameters and persistent buffers with the reference's names and shapes, initialised module by module as the reference
        # does (mgsv_amd/model/init.py) from args.seed -- every rank of a data-parallel run draws the same weights from the same
        # seed, so no broadcast is needed.  (mgsv_amd.synth's N(0, 1/fan_in) weights are for tests and bench.py only.)
        from .init import reference_init
        sd = reference_init(self.cfg, seed=int(getattr(args, "seed", 0) or 0))
        xa = "video_guided_to_music_pooling_cross_transformer"
        xpool = _XPoolModule(self)
        for name, arr in sd.items():
            t = torch.from_numpy(arr.copy())
            is_buf = name.endswith((".pe", ".running_mean", ".running_var", ".num_batches_tracked")) or name == "criterion.empty_weight"
            if name.startswith("criterion."):
                continue
            target, rel = (xpool, name[len(xa) + 1:]) if name.startswith(xa + ".") else (self, name)
            _attach(target, rel, t, is_buf)
        if "music" in self.cfg.vmr_fusion:
            self.add_module(xa, xpool)
        if "detr" in self.cfg.mml_localization:                    # the regression variant builds no criterion (model_Uni.py:66-69)
            self.criterion = _Criterion(self.cfg)
            self.criterion.register_buffer("empty_weight", torch.from_numpy(sd["criterion.empty_weight"].copy()))
        self._engine: Optional[MadeEngine] = None
        self._engine_stamp = None
        self._lane_engines: Dict[int, MadeEngine] = {}              # extra engines (own workspace) for batches in flight
        self._lane_stamps: Dict[int, object] = {}
        self._trainer = None
        self._trainer_stamp = None
        self._train_seed = int(getattr(args, "seed", 0)) << 20

    # ---- parameter groups (reference model/model_Uni.py:73-114, model_Base.py:379-404)
    def _params(self, prefixes) -> List[nn.Parameter]:
        return [p for n, p in self.named_parameters() if n.startswith(tuple(prefixes))]

    def get_temporal_parameter(self):
        # reference model/model_Base.py:379-404 (projection + SA parameters)
        return self._params(["vit_proj.", "ast_proj.", "video_transformer.", "audio_transformer.", "share_transformer.", "video_cls_token",
                             "audio_cls_token", "Video_encoder_projection.", "Music_encoder_projection."])

    def get_matching_parameter(self):
        return self._params(["video_guided_to_music_pooling_cross_transformer.", "music_guided_to_video_pooling_cross_transformer."]) + [self.logit_scale]

    def get_detection_parameter(self):
        # reference model/model_Uni.py:92-114: the CA fusion block, then the DETR stack + heads -- or, for the regression
        # variant, the regression MLP only (the DETR transformer is then in no optimizer group)
        if "regression" in self.cfg.mml_localization:
            return self._params(["video_music_fusion_cross_transformer.", "reg_mlp."])
        return self._params(["video_music_fusion_cross_transformer.", "detr_transformer.", "span_embed.", "class_embed.", "moment_embed.",
                             "contrastive_align_projection_"])

    # ---- state handling
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        filtered = {k: v for k, v in state_dict.items() if not k.startswith(_FROZEN_PREFIXES)}
        res = super().load_state_dict(filtered, strict=strict, **kw)
        self._engine_stamp = None
        return res

    def _stamp(self):
        # (the fused optimizer step updates the masters in place through raw pointers: neither data_ptr nor _version moves, so the
        # trainer counts its own updates)
        gen = self._trainer.generation if self._trainer is not None else 0
        return tuple((p.data_ptr(), p._version) for p in self.parameters()) + (gen,)

    def _engine_ready(self, lane: int = 0) -> MadeEngine:
        """The forward executor with this module's current weights.  lane > 0: a further engine with its own workspace, so that an
        evaluation loop can keep several independent batches in flight on different streams (mgsv_amd/driver.py: eval_epoch)."""
        stamp = self._stamp()
        if lane > 0:
            eng = self._lane_engines.get(lane)
            if eng is None:
                eng = self._lane_engines[lane] = MadeEngine(self.cfg, self.state_dict(), device=self.device, dtype=self.compute_dtype)
            elif self._lane_stamps.get(lane) != stamp:
                eng.load_state_dict(self.state_dict())
            self._lane_stamps[lane] = stamp
            return eng
        if self._engine is None:
            self._engine = MadeEngine(self.cfg, self.state_dict(), device=self.device, dtype=self.compute_dtype)
        elif stamp != self._engine_stamp:
            self._engine.load_state_dict(self.state_dict())
        self._engine_stamp = stamp
        return self._engine

    def _trainer_ready(self):
        """MadeTrainer whose flat f32 master buffer the module's parameters alias."""
        from ..trainer import MadeTrainer
        if self._trainer is None:
            self._trainer = MadeTrainer(self.cfg, self.state_dict(), device=self.device, dtype=self.compute_dtype)
            with torch.no_grad():
                for name, p in self.named_parameters():
                    if name in self._trainer.master:
                        p.data = self._trainer.master[name]
                # registered buffers (the BatchNorm running statistics of agg_module = "mlp") share the trainer's storage the same way:
                # the statistics a train step moves are the ones eval (state_dict -> MadeEngine) and save_model see
                for name, _ in list(self.named_buffers()):
                    t = self._trainer.buffers.get(name)
                    if t is not None:
                        owner = self
                        *path, leaf = name.split(".")
                        for part in path:
                            owner = owner._modules[part]
                        owner._buffers[leaf] = t
            self._trainer_stamp = self._stamp()
        elif self._stamp() != self._trainer_stamp:                 # an optimizer (or load_state_dict) touched the masters
            with torch.no_grad():
                for name, p in self.named_parameters():
                    m = self._trainer.master.get(name)
                    if m is not None and p.data_ptr() != m.data_ptr():
                        m.copy_(p.data)
                        p.data = m
            self._trainer.repack()
            self._trainer_stamp = self._stamp()
        return self._trainer

    def _maps(self, o, eng, frame_masks, segment_masks, video_ids, music_ids):
        nd, cfg = self.cfg.detr_dec_layers, self.cfg
        if "regression" in cfg.mml_localization:                   # reference model/model_Uni.py:290-300: spans only, no matcher
            feat_map = {"video_feats": o["video_feats"], "music_feats": o["music_feats"],
                        "frame_feats": o["frame_feats"].float(), "segment_feats": o["segment_feats"].float()}
            return ({"pred_spans": o["pred_spans"]}, feat_map, {"frame_masks": frame_masks, "segment_masks": segment_masks},
                    {"video_ids": video_ids, "music_ids": music_ids})
        output_map: Dict[str, object] = {"pred_logits": o["pred_logits"], "pred_spans": o["pred_spans"]}
        if cfg.contrastive_align_loss:
            output_map.update(proj_queries=o["proj_queries"], proj_vid_mem=o["proj_vid_mem"])
        if cfg.aux_loss:
            aux = []
            for i in range(nd - 1):
                d = {"pred_logits": o["logits_all"][i], "pred_spans": o["spans_all"][i]}
                if cfg.contrastive_align_loss:
                    d.update(proj_queries=o["proj_queries_all"][i], proj_vid_mem=o["proj_vid_mem"])
                aux.append(d)
            output_map["aux_outputs"] = aux
        feat_map = {"video_feats": o["video_feats"], "music_feats": o["music_feats"],
                    "frame_feats": o["frame_feats"].float(), "segment_feats": o["segment_feats"].float()}
        mask_map = {"frame_masks": frame_masks, "segment_masks": segment_masks}
        id_map = {"video_ids": video_ids, "music_ids": music_ids}
        self.last_matcher = {k: o[k] for k in ("matcher_pred_idx", "matcher_tgt_idx", "matcher_count", "matcher_status")}
        return output_map, feat_map, mask_map, id_map

    # ---- forward (reference model/model_Uni.py:177-322)
    def forward(self, frame_feats, segment_feats, frame_masks, segment_masks, spans_target, v_duration=None,
                video_ids=None, music_ids=None, is_train=False, lane: int = 0):
        if self.cfg.vmr_loss not in ("dual", "single", "dual_single_loss_fuse", "dual_single_sim_fuse", "dual_single_feature_fuse") or "XA" not in self.cfg.vmr_fusion:
            raise ValueError(f"Error: vmr_loss={self.cfg.vmr_loss} and vmr_fusion={self.cfg.vmr_fusion} is not supported in VMR_model")
        if self.training:
            trn = self._trainer_ready()
            dev, f32 = trn.device, torch.float32
            batch = (frame_feats.to(dev, f32), segment_feats.to(dev, f32), frame_masks.to(dev, f32), segment_masks.to(dev, f32),
                     spans_target.to(dev, f32))
            self._train_seed += 1
            self._train_vdur = v_duration.to(dev, f32) if torch.is_tensor(v_duration) else v_duration
            if torch.is_grad_enabled():
                params = [p for _, p in self.named_parameters()]
                ret, loc = _TrainStep.apply(self, batch, self._train_seed, *params)
                o = self._last_train_out
            else:
                o = trn.forward_train(*batch, seed=self._train_seed, v_duration=self._train_vdur)
                ret, loc = o["retrieval_loss"][0], o["localization_loss"][0]
            output_map, feat_map, mask_map, id_map = self._maps(o, trn, frame_masks, segment_masks, video_ids, music_ids)
            loss_map = {"retrieval_loss": ret, "localization_loss": loc, "localization_loss_dict": trn.loss_dict(o)}
            return output_map, loss_map, feat_map, mask_map, id_map
        if self.cfg.vmr_loss not in ("dual", "single", "dual_single_loss_fuse", "dual_single_sim_fuse", "dual_single_feature_fuse") or "XA" not in self.cfg.vmr_fusion:
            raise ValueError(f"Error: vmr_loss={self.cfg.vmr_loss} and vmr_fusion={self.cfg.vmr_fusion} is not supported in VMR_model")
        eng = self._engine_ready(lane)
        dev = eng.device
        f32 = torch.float32
        o = eng.forward(frame_feats.to(dev, f32), segment_feats.to(dev, f32), frame_masks.to(dev, f32),
                        segment_masks.to(dev, f32), spans_target.to(dev, f32),
                        v_duration=v_duration.to(dev, f32) if torch.is_tensor(v_duration) else v_duration)
        nd, cfg = self.cfg.detr_dec_layers, self.cfg
        feat_map = {"video_feats": o["video_feats"], "music_feats": o["music_feats"],
                    "frame_feats": o["frame_feats"].float(), "segment_feats": o["segment_feats"].float()}
        mask_map = {"frame_masks": frame_masks, "segment_masks": segment_masks}
        id_map = {"video_ids": video_ids, "music_ids": music_ids}
        if "regression" in cfg.mml_localization:                   # reference model/model_Uni.py:228-232,290-300
            loss_map = {"retrieval_loss": o["retrieval_loss"][0], "localization_loss": o["localization_loss"][0],
                        "localization_loss_dict": {"loss_span": o["regression_loss_span"], "loss_giou": 0, "loss_label": 0, "class_error": 0}}
            return {"pred_spans": o["pred_spans"]}, loss_map, feat_map, mask_map, id_map
        output_map: Dict[str, object] = {"pred_logits": o["pred_logits"], "pred_spans": o["pred_spans"]}
        if cfg.contrastive_align_loss:
            output_map.update(proj_queries=o["proj_queries"], proj_vid_mem=o["proj_vid_mem"])
        if cfg.moment_loss:                                        # reference model/model_Uni.py:152-159
            output_map.update(moment_feats=o["moment_feats"], video_feats=o["video_feats"])
        if cfg.aux_loss:
            aux = []
            for i in range(nd - 1):
                d = {"pred_logits": o["logits_all"][i], "pred_spans": o["spans_all"][i]}
                if cfg.contrastive_align_loss:
                    d.update(proj_queries=o["proj_queries_all"][i], proj_vid_mem=o["proj_vid_mem"])
                aux.append(d)
            output_map["aux_outputs"] = aux
        loss_map = {"retrieval_loss": o["retrieval_loss"][0], "localization_loss": o["localization_loss"][0],
                    "localization_loss_dict": eng.loss_dict(o)}
        feat_map = {"video_feats": o["video_feats"], "music_feats": o["music_feats"],
                    "frame_feats": o["frame_feats"].float(), "segment_feats": o["segment_feats"].float()}
        mask_map = {"frame_masks": frame_masks, "segment_masks": segment_masks}
        id_map = {"video_ids": video_ids, "music_ids": music_ids}
        self.last_matcher = {k: o[k] for k in ("matcher_pred_idx", "matcher_tgt_idx", "matcher_count", "matcher_status")}
        return output_map, loss_map, feat_map, mask_map, id_map

    # ---- retrieval assembly (reference test-MaDe.py:386-403) as one call
    @torch.no_grad()
    def retrieval_sim_matrix(self, video_embeds, segment_embeds, segment_masks, music_embeds):
        eng = self._engine_ready()
        dev = eng.device
        return eng.retrieval_sim_matrix(video_embeds.to(dev, torch.float32).contiguous(), segment_embeds.to(dev).contiguous(),
                                        segment_masks.to(dev, torch.float32).contiguous(),
                                        music_embeds.to(dev, torch.float32).contiguous())
