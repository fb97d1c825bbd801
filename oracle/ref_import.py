"""Import the reference (xxayt/MGSV) from /root/reference on CPU  --  TEST INFRASTRUCTURE.

Only usable in the build container: /root/reference does not exist on the GPU box and
nothing that runs there (pytest -m gpu, smoke(), bench.py) may call this.  Used by
oracle/validate_against_reference.py and tests/golden/make_golden.py to pin the oracle
and to emit golden vectors.  Nothing from the reference is copied: it is imported in
place with four stubs.
"""
from __future__ import annotations

import logging
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MADE_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "model", "model_Uni.py"))


def import_reference():
    """Returns the reference's `model.model_Uni` module (and leaves `modules.*`,
    `music_detr.*` importable).  Stubs: `clip`, `model.ast_models`, DDP, torch.load of
    the AST checkpoint (reference: model/model_Base.py:8,10,277-289)."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True          # never drop __pycache__ into the read-only tree
    import torch
    import torch.nn as nn

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    clip_stub = types.ModuleType("clip")
    clip_stub.load = lambda *a, **k: (nn.Module(), None)
    sys.modules["clip"] = clip_stub

    ast_stub = types.ModuleType("model.ast_models")

    class ASTModel(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    ast_stub.ASTModel = ASTModel
    sys.modules["model.ast_models"] = ast_stub

    class _IdentityDDP(nn.Module):
        def __init__(self, module, *a, **k):
            super().__init__()
            self.module = module

        def forward(self, *a, **k):
            return self.module(*a, **k)

    torch.nn.parallel.DistributedDataParallel = _IdentityDDP

    _orig_load = torch.load

    def _load(path, *a, **k):
        if isinstance(path, str) and path.endswith("audioset_0.4593.pth"):
            return {}
        return _orig_load(path, *a, **k)

    torch.load = _load

    import importlib
    mod = importlib.import_module("model.model_Uni")
    return mod


def build_reference_model(cfg, state_dict_np, T_a_max=None):
    """Instantiate the reference's Uni_model for `cfg` and load our seeded weights."""
    import numpy as np
    import torch

    mod = import_reference()
    args = cfg.to_args(local_rank=-1)
    logger = logging.getLogger("ref")
    model = mod.Uni_model(args, device=torch.device("cpu"), logger=logger).float()
    # audio PE table is hard-coded to 300 positions (model_Base.py:293); rebuild when longer
    from model.model_Base import PositionalEncoding
    need = cfg.audio_attention_seqlen
    if hasattr(model, "audio_position_embedding") and model.audio_position_embedding.pe.shape[1] != need:
        model.audio_position_embedding = PositionalEncoding(seq_len=need, dim_model=cfg.D)
    sd = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in state_dict_np.items()}
    ref_keys = set(model.state_dict().keys())
    ours = set(sd.keys())
    missing = sorted(ref_keys - ours)
    extra = sorted(ours - ref_keys)
    if missing or extra:
        raise RuntimeError(f"state_dict key mismatch vs reference: missing={missing[:8]} extra={extra[:8]}")
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model


def reference_grads(ref, inp, train: bool, schedule=None, seed: int = 1234, dtype=None):
    """Run the reference forward + backward of (retrieval_loss + localization_loss) and return
    (loss_map, {param name: grad}).  train=True puts the model in train() and, for this call only, replaces its dropout
    draws (torch.nn.functional.dropout and the fused attention's dropout_p) by the build's stateless masks
    (mgsv_amd/dropout.py), consumed in the order given by `schedule` = [(site, p, logical shape), ...] as recorded from
    the oracle -- so a dropout the oracle places differently from the reference shows up as a mismatch."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from mgsv_amd import dropout as dr

    dtype = dtype or torch.float32
    tin = {k: torch.from_numpy(v).to(dtype) for k, v in inp.items() if isinstance(v, np.ndarray)}
    orig_dropout, orig_sdpa = F.dropout, F.scaled_dot_product_attention
    schedule = list(schedule or [])
    cursor = [0]

    def patched_dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        site, p0, shape = schedule[cursor[0]]
        cursor[0] += 1
        assert abs(p0 - p) < 1e-12, (site, p0, p)
        n = int(np.prod(shape))
        assert x.numel() == n, (site, shape, tuple(x.shape))
        keep = torch.from_numpy(dr.keep_mask(seed, dr.site_id(site), p, n).reshape(shape)).to(x.dtype)
        if tuple(x.shape) == shape:
            m = keep
        elif len(shape) == 3 and tuple(x.shape) == (shape[1], shape[0], shape[2]):
            m = keep.permute(1, 0, 2)                    # the reference runs sequence-first
        else:
            m = keep.reshape(x.shape)                    # attention weights [B*H, Lq, Lk]
        return x * m * (1.0 / (1.0 - p))

    def patched_sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
        sc = (q.shape[-1] ** -0.5) if scale is None else scale
        s_ = (q @ k.transpose(-1, -2)) * sc
        if attn_mask is not None:
            s_ = s_.masked_fill(~attn_mask, float("-inf")) if attn_mask.dtype == torch.bool else s_ + attn_mask
        return patched_dropout(torch.softmax(s_, dim=-1), dropout_p, True) @ v

    try:
        if train:
            ref.train()
            F.dropout, F.scaled_dot_product_attention = patched_dropout, patched_sdpa
        for p_ in ref.parameters():
            p_.grad = None
        om, lm, fm, mm, im = ref(tin["frame_feats"].clone(), tin["segment_feats"].clone(), tin["frame_masks"].clone(),
                                 tin["segment_masks"].clone(), tin["spans_target"].clone(), v_duration=tin["v_duration"],
                                 video_ids=inp["video_ids"], music_ids=inp["music_ids"], is_train=train)
        (lm["retrieval_loss"] + lm["localization_loss"]).backward()
    finally:
        F.dropout, F.scaled_dot_product_attention = orig_dropout, orig_sdpa
        ref.eval()
    if train:
        assert cursor[0] == len(schedule), (cursor[0], len(schedule))
    grads = {n: p_.grad.detach().clone() for n, p_ in ref.named_parameters() if p_.grad is not None}
    return lm, grads
