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
    if model.audio_position_embedding.pe.shape[1] != need:
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
