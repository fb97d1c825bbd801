"""Retrieval losses -- everything `from modules.loss import *` gives the reference's drivers (reference modules/loss.py:5-123;
train-MaDe.py:20), same names, arguments and return values, computed by libmade_hip.so (made_clip_loss, made_l2norm_rows +
made_linear, made_scale_exp); no CPU fallback.  Forward values only: the training path differentiates these losses inside
MadeTrainer (made_clip_loss_bwd), not through autograd on these functions."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib, ops

__all__ = ["CLIPLoss", "cal_distance", "InfoNCELoss"]


def _dev():
    if not torch.cuda.is_available():
        raise _lib.MadeError("mgsv_amd.modules.loss needs a GPU (the MaDe hot path has no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _scalar(logit_scale, dev) -> torch.Tensor:
    t = logit_scale if isinstance(logit_scale, torch.Tensor) else torch.tensor(float(logit_scale))
    return t.detach().to(dev, torch.float32).reshape(1).contiguous()


def _square(sims, dev) -> torch.Tensor:
    s = sims if isinstance(sims, torch.Tensor) else torch.as_tensor(np.asarray(sims))
    s = s.detach().to(dev, torch.float32).contiguous()
    assert s.dim() == 2 and s.shape[0] == s.shape[1], "the contrastive losses take a square similarity matrix"
    return s


def CLIPLoss(sims, logit_scale):
    """reference modules/loss.py:5-24: symmetric cross entropy of `sims * exp(logit_scale)` against the diagonal."""
    dev = _dev()
    s, ls = _square(sims, dev), _scalar(logit_scale, dev)
    out = torch.empty(1, device=dev, dtype=torch.float32)
    ops.clip_loss(s, ls, out)
    return out[0].to(sims.device if isinstance(sims, torch.Tensor) else dev)


def cal_distance(x, y, distance_type="COS"):
    """reference modules/loss.py:30-62.  COS: cosine similarity of all (x, y) rows, [bs_x, bs_y] (float64 numpy for numpy inputs,
    as the reference returns).  L2 is not used by any script of the reference and is not on the HIP path."""
    assert x.shape[1] == y.shape[1], "The second dimension of x and y must be the same."
    if distance_type != "COS":
        raise NotImplementedError("cal_distance: only distance_type='COS' (the reference's scripts) runs on the HIP path")
    dev = _dev()
    as_np = isinstance(x, np.ndarray)
    xt = torch.as_tensor(x).detach().to(dev, torch.float32).contiguous()
    yt = torch.as_tensor(y).detach().to(dev, torch.float32).contiguous()
    d = ops.linear(ops.l2norm_rows(xt), ops.l2norm_rows(yt), None, out_dtype=torch.float32)      # exact-f32 MFMA
    if as_np:
        return d.cpu().numpy().astype(np.float64)
    return d.to(x.device)


def InfoNCELoss(output, logit_scale, audio_id=None, distance_type="COS", args=None, is_train=False):
    """reference modules/loss.py:66-123: (loss, logits_per_video, logits_per_audio).  With `audio_id` given, `is_train` and
    `args.ignore_same_music == 0` the video -> music direction leaves out the other samples that share the row's track (:90-114)."""
    dev = _dev()
    s, ls = _square(output, dev), _scalar(logit_scale, dev)
    n = s.shape[0]
    exclude = None
    if audio_id is not None and is_train and args is not None and getattr(args, "ignore_same_music", 1) == 0:
        code = {}
        idx = torch.tensor([code.setdefault(str(i), len(code)) for i in list(audio_id)])
        ex = (idx[:, None] == idx[None, :]).float()
        ex.fill_diagonal_(0.0)
        exclude = ex.to(dev).contiguous()
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    ops.clip_loss(s, ls, loss, row_exclude=exclude)
    logits = torch.empty(n, n, device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().made_scale_exp(s.data_ptr(), ls.data_ptr(), logits.data_ptr(), n * n, torch.cuda.current_stream().cuda_stream),
               "made_scale_exp")
    odev = output.device if isinstance(output, torch.Tensor) else dev
    return loss[0].to(odev), logits.to(odev), logits.t().to(odev)
