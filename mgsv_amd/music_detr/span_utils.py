"""Span helpers of the drivers and the matcher (reference music_detr/span_utils.py:4-170; train-MaDe.py:22), same names,
arguments and results.  Tensors may live on the CPU or the GPU (the reference's drivers call these on CPU tensors); the
arithmetic runs in libmade_hip.so (made_span_convert / made_span_pairwise / made_span_iou_se) and the result returns on the
input's device.  No CPU fallback."""
from __future__ import annotations

import torch

from .. import _lib


def _dev():
    if not torch.cuda.is_available():
        raise _lib.MadeError("mgsv_amd.music_detr.span_utils needs a GPU (the MaDe hot path has no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _f32(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(dev, torch.float32).contiguous()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _convert(spans: torch.Tensor, mode: int) -> torch.Tensor:
    assert spans.dim() == 2 and spans.shape[1] == 2, "spans: [#windows, 2]"
    dev = _dev()
    s = _f32(spans, dev)
    out = torch.empty_like(s)
    _lib.check(_lib.lib().made_span_convert(s.data_ptr(), out.data_ptr(), s.shape[0], mode, _stream()), "made_span_convert")
    return out.to(spans.device).to(spans.dtype if spans.dtype.is_floating_point else torch.float32)


def span_se_to_cw(se_spans):
    """reference span_utils.py:4-13: (start, end) -> (centre, width)."""
    return _convert(se_spans, 1)


def span_cw_to_se(cw_spans):
    """reference span_utils.py:15-24: (centre, width) -> (start, end)."""
    return _convert(cw_spans, 0)


def _pairwise(spans1, spans2, want):
    assert spans1.dim() == 2 and spans2.dim() == 2 and spans1.shape[1] == 2 and spans2.shape[1] == 2
    dev = _dev()
    a, b = _f32(spans1, dev), _f32(spans2, dev)
    N, M = a.shape[0], b.shape[0]
    bufs = {k: torch.empty(N, M, device=dev, dtype=torch.float32) for k in want}
    ptr = lambda k: bufs[k].data_ptr() if k in bufs else None
    _lib.check(_lib.lib().made_span_pairwise(a.data_ptr(), b.data_ptr(), ptr("iou"), ptr("union"), ptr("giou"), ptr("iop"), N, M, _stream()),
               "made_span_pairwise")
    return {k: v.to(spans1.device) for k, v in bufs.items()}


def temporal_iou(spans1, spans2):
    """reference span_utils.py:39-66: (iou [N, M], union [N, M]).

    >>> temporal_iou(torch.Tensor([[0, 0.2], [0.5, 1.0]]), torch.Tensor([[0, 0.3], [0., 1.0]]))   # doctest: +SKIP
    (tensor([[0.6667, 0.2000], [0.0000, 0.5000]]), tensor([[0.3000, 1.0000], [0.8000, 1.0000]]))
    """
    r = _pairwise(spans1, spans2, ("iou", "union"))
    return r["iou"], r["union"]


def temporal_intersection_over_pred(gt_spans, pred_spans):
    """reference span_utils.py:69-83: intersection over the second input's spans, [N, M]."""
    return _pairwise(gt_spans, pred_spans, ("iop",))["iop"]


def generalized_temporal_iou(spans1, spans2):
    """reference span_utils.py:86-115 (asserts end >= start on both inputs, as the reference does)."""
    spans1, spans2 = spans1.float(), spans2.float()
    assert (spans1[:, 1] >= spans1[:, 0]).all()
    assert (spans2[:, 1] >= spans2[:, 0]).all()
    return _pairwise(spans1, spans2, ("giou",))["giou"]


def _iou_se(pred_se, gt, dur, max_dur: float, clamp_max: bool, discounted: bool) -> torch.Tensor:
    dev = _dev()
    p, g, d = _f32(pred_se, dev), _f32(gt, dev), _f32(dur, dev)
    N = p.shape[0]
    out = torch.empty(N, device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().made_span_iou_se(p.data_ptr(), g.data_ptr(), d.data_ptr(), N, float(max_dur), 1 if clamp_max else 0,
                                           1 if discounted else 0, out.data_ptr(), _stream()), "made_span_iou_se")
    return out


def individual_IoU_tensor(gt_st, gt_ed, gt_m_duration, pred_st, pred_ed, discounted=False):
    """reference span_utils.py:119-145: IoU of one prediction with one ground-truth moment (0-d tensors or floats)."""
    t = lambda x: torch.as_tensor(x, dtype=torch.float32).reshape(1)
    odev = gt_st.device if isinstance(gt_st, torch.Tensor) else torch.device("cpu")
    out = _iou_se(torch.stack([t(pred_st), t(pred_ed)], dim=-1), torch.stack([t(gt_st), t(gt_ed)], dim=-1), t(gt_m_duration), 0.0, False,
                  bool(discounted))
    return out[0].to(odev)


def detr_iou(args, mr_results_list):
    """reference span_utils.py:147-170: list of per-sample IoUs of the top-ranked prediction (one launch for the whole list)."""
    n = len(mr_results_list)
    if n == 0:
        return []
    pred = torch.stack([torch.as_tensor(d["ranked_preds"][0], dtype=torch.float32)[:2].reshape(2) for d in mr_results_list])
    gt = torch.stack([torch.as_tensor(d["gt_moment"], dtype=torch.float32).reshape(-1)[:2] for d in mr_results_list])
    dur = torch.stack([torch.as_tensor(d["m_duration"], dtype=torch.float32).reshape(()) for d in mr_results_list])
    iou = _iou_se(pred, gt, dur, float(args.max_m_duration), True, False).cpu()
    return [iou[i] for i in range(n)]
