"""Configuration of the MaDe hot path.

The reference passes one flat argparse namespace into every module
(reference: train-MaDe.py:27-173).  `MadeConfig` holds only the fields the
hot path reads and can be built from such a namespace
with `MadeConfig.from_args`, so the reference's drivers keep working.

Defaults are the *script* values (reference: scripts/train_kuai_all_feature.sh)
and not the parser defaults, because the scripts are the configuration the
reference is actually run with.
"""
from __future__ import annotations

import argparse
from dataclasses import dataclass, asdict, fields


@dataclass
class MadeConfig:
    # dims
    dim_input: int = 256            # D  (hidden_dim == detr_hidden_dim == dim_input, train-MaDe.py:155-156)
    vit_dim: int = 512              # model_Base.py:287
    ast_dim: int = 768              # model_Base.py:275
    temporal_ffn_dim: int = 1024    # model_Base.py:294
    SA_temporal_heads: int = 8
    video_transformer_depth: int = 1
    audio_transformer_depth: int = 1
    video_attention_seqlen: int = 250   # PE table length, video (flag)
    audio_attention_seqlen: int = 300   # PE table length, audio (hard-coded model_Base.py:293)
    with_act_after_proj: int = 0
    agg_module: str = "transf"          # "transf" | "mlp" (EmbeddingNet, model_Base.py:216-249,357-377)
    transformer_is_share: int = 0       # one temporal block for both towers (model_Base.py:300-302,322-331)
    with_cls_token: int = 0             # learned token prepended, clip vector = its output (model_Base.py:527-530,572-574)
    with_last_token: int = 0            # only read when agg_module contains "cal": unreachable through the parsers
    # sequence bounds
    max_v_frames: int = 50
    max_snippet_num: int = 96       # int(max_m_duration / stride)
    max_m_duration: int = 240
    # matching
    vmr_fusion: str = "XA-music"
    vmr_loss: str = "dual_single_loss_fuse"
    fusion_mask: int = 1
    dual_single_loss_weight: float = 1.0
    temperature_init_value: float = 0.03
    ignore_same_music: int = 1
    # detection
    mml_fusion: str = "concat"      # "concat" | "CA"
    mml_localization: str = "detr"
    detr_nheads: int = 8
    detr_dim_feedforward: int = 1024
    detr_enc_layers: int = 2
    detr_dec_layers: int = 6
    detr_dropout: float = 0.1
    detr_pre_norm: bool = False     # pre-norm DETR layers + a final encoder norm (music_detr/transformer.py:33-35,170-189,246-271)
    num_moment_queries: int = 1
    moment_query_type: str = "video"
    predict_center: int = 0
    fb_label: str = "01"
    span_loss_type: str = "l1"
    # losses
    l1_loss: int = 1
    aux_loss: int = 1
    contrastive_align_loss: int = 1
    moment_loss: int = 0
    audio_short_cut: int = 0
    contrastive_dim: int = 256
    # CA fusion block (model_Uni.py:33-43)
    ca_heads: int = 8
    ca_dim_head: int = 128
    ca_ffn_dim: int = 1024

    @property
    def D(self) -> int:
        return self.dim_input

    @property
    def contrastive_hdim(self) -> int:
        # model_Uni.py:57-60
        return self.dim_input if self.audio_short_cut else self.contrastive_dim

    @property
    def foreground_label(self) -> int:
        return 0 if self.fb_label == "01" else 1

    @property
    def background_label(self) -> int:
        return 1 - self.foreground_label

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_args(cls, args) -> "MadeConfig":
        """Build from the reference's argparse namespace (extra fields ignored)."""
        kw = {}
        for f in fields(cls):
            if hasattr(args, f.name):
                kw[f.name] = getattr(args, f.name)
        cfg = cls(**kw)
        # flags that select code this build does not have must fail loudly, not be ignored
        bad = []
        if getattr(args, "span_loss_type", "l1") != "l1":
            # the reference cannot run it either: its matcher views the [B, Q, 2] spans as [B * Q, 2, snippet_num] and indexes with float
            # targets (music_detr/matcher.py:83-86: RuntimeError / IndexError on the first iteration)
            bad.append(f"span_loss_type={args.span_loss_type} (only the l1 span head / matcher cost exists; the reference's ce branch fails at its first matcher call)")
        if getattr(args, "position_embedding", "sine") not in ("sine", "v2"):
            # the reference refuses it too, with this error (music_detr/position_encoding.py:98-105: the 'learned' branch is commented out)
            raise ValueError(f"not supported {args.position_embedding}")
        if bad:
            raise NotImplementedError("MaDe HIP path does not cover: " + "; ".join(bad))
        if hasattr(args, "max_m_duration") and hasattr(args, "stride") and not hasattr(args, "max_snippet_num"):
            cfg.max_snippet_num = int(args.max_m_duration / args.stride)
        return cfg

    def to_args(self, **extra) -> argparse.Namespace:
        """The namespace the reference's `Uni_model(args, ...)` ctor expects."""
        d = self.to_dict()
        d.update(
            hidden_dim=self.dim_input,
            detr_hidden_dim=self.dim_input,
            detr_pre_norm=bool(self.detr_pre_norm),
            decoder_SA=0,
            position_embedding="sine",
            input_dropout=0.5,
            local_rank=-1,
            name="made",
            audio_encoder_type="AST",
            video_encoder_type="ViT",
            music_frozen_feature_path="ast_feature2p5",
            frame_frozen_feature_path="vit_feature1",
        )
        d.update(extra)
        return argparse.Namespace(**d)


# BASELINE.json configs (SURVEY.md section 8 notation)
def cfg_plumbing() -> MadeConfig:      # cfg 1: CPU plumbing, B=2, T_v=30, T_a=200, D=256
    return MadeConfig(dim_input=256, max_v_frames=30, max_snippet_num=200)


def cfg_headline() -> MadeConfig:      # cfg 2/3: B=64, T_v=30, T_a=512, D=512
    return MadeConfig(dim_input=512, max_v_frames=30, max_snippet_num=512,
                      audio_attention_seqlen=512, contrastive_dim=256)


def cfg_native() -> MadeConfig:        # scripts' shape: T_v=50, T_a=96, D=256
    return MadeConfig()
