"""Fresh-weight initialisation of the drop-in `Uni_model`, module by module as the reference does it, seeded from `args.seed`.

  * `nn.Linear` default (kaiming_uniform(a=sqrt 5) = U(+-1/sqrt fan_in) for weight and bias): vit_proj / ast_proj, the temporal
    blocks' FFN and final Linear, the heads (class_embed, span_embed, contrastive projections, reg_mlp, moment_embed), the final
    Linear of the CA block -- reference model/model_Base.py:64-80,286-296; model/model_Uni.py:51-72.
  * `nn.MultiheadAttention`: in_proj_weight xavier_uniform, in_proj_bias = 0, out_proj.bias = 0 (torch's _reset_parameters);
    out_proj.weight: Linear default in the temporal blocks; xavier_uniform inside the DETR transformer.
  * DETR transformer: every parameter with more than one dimension xavier_uniform (reference music_detr/transformer.py:46-49);
    linear1 / linear2 biases keep the Linear default.
  * X-Pool blocks: every `*proj*` / `*linear*` weight = identity, bias = 0 (reference modules/transformer.py:148-154).
  * CA fusion block and EmbeddingNet: Linear weights xavier_normal (`init_method = "xavier"`, model_Base.py:297), biases 0.01
    (model_Base.py:35-42,120-128,237-245).
  * LayerNorm 1 / 0, BatchNorm 1 / 0 (running stats 0 / 1), class tokens trunc_normal(0.02) (model_Base.py:316,321),
    decoder_query_embed N(0, 1) (nn.Embedding), logit_scale = ln(1 / temperature) (model_Uni.py:29).
Persistent buffers (position tables, criterion.empty_weight) are deterministic and come from mgsv_amd.synth.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from .. import synth
from ..config import MadeConfig


def reference_init(cfg: MadeConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    template = synth.make_state_dict(cfg, seed=0)              # names, shapes and the deterministic buffers
    g = torch.Generator().manual_seed(int(seed))
    out: Dict[str, np.ndarray] = {}

    def uniform(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def xavier_uniform(shape):
        fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
        return uniform(shape, math.sqrt(6.0 / (fan_in + fan_out)))

    def xavier_normal(shape):
        fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
        return torch.randn(shape, generator=g) * math.sqrt(2.0 / (fan_in + fan_out))

    def fan_in_of(name: str) -> int:
        w = name[:-len("bias")] + "weight"
        return int(template[w].shape[1])

    def is_norm(name: str) -> bool:
        parts = name.split(".")
        if "norm" in name:                                      # DETR norms, X-Pool layer norms, the CA block's *_layer_norms
            return True
        if parts[0] in ("video_transformer", "audio_transformer", "share_transformer") and len(parts) == 5 and parts[3] in ("0", "2"):
            return True                                          # Transformer_enhancement: layers.<l>.0 / layers.<l>.2 are the LayerNorms
        if parts[0] in ("Video_encoder_projection", "Music_encoder_projection") and parts[2] in ("1", "4"):
            return True                                          # EmbeddingNet: net.1 / net.4 are BatchNorm1d
        return False

    for name, arr in template.items():
        shape = tuple(arr.shape)
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith(".pe") or name.startswith("criterion."):
            t = torch.from_numpy(np.asarray(arr).copy())       # deterministic buffers
        elif "running_" in name or "num_batches" in name:      # a fresh BatchNorm1d: mean 0, variance 1, no batches seen
            t = torch.ones(shape) if "running_var" in name else torch.zeros(shape)
        elif name == "logit_scale":
            t = torch.tensor(math.log(1.0 / cfg.temperature_init_value), dtype=torch.float32)
        elif name == "decoder_query_embed.weight":
            t = torch.randn(shape, generator=g)
        elif name.endswith("_cls_token"):
            t = torch.nn.init.trunc_normal_(torch.empty(shape), std=0.02, generator=g)
        elif "pooling_cross_transformer." in name:
            if "proj" in name or "linear" in name:
                t = torch.eye(shape[0], shape[1]) if leaf == "weight" else torch.zeros(shape)
            else:                                              # layer norms
                t = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
        elif is_norm(name):                                     # LayerNorm / BatchNorm affine parameters
            t = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
        elif leaf == "in_proj_weight":
            t = xavier_uniform(shape)
        elif leaf == "in_proj_bias" or name.endswith("out_proj.bias"):
            t = torch.zeros(shape)
        elif name.startswith("detr_transformer.") and len(shape) > 1:
            t = xavier_uniform(shape)
        elif name.startswith(("video_music_fusion_cross_transformer.layers.", "Video_encoder_projection.", "Music_encoder_projection.")):
            t = xavier_normal(shape) if leaf == "weight" else torch.full(shape, 0.01)
        elif leaf == "weight":                                  # nn.Linear default
            t = uniform(shape, 1.0 / math.sqrt(shape[1]))
        elif leaf == "bias":
            t = uniform(shape, 1.0 / math.sqrt(fan_in_of(name)))
        else:
            raise KeyError(f"reference_init: no rule for parameter {name}")
        out[name] = t.to(torch.float32).numpy().reshape(shape).copy()
    return out
