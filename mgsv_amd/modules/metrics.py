"""Similarity matrices of the X-Pool towers -- the free functions the reference's drivers import
(reference modules/metrics.py:10-57; train-MaDe.py:16, test-MaDe.py:16), same names, arguments and results.

Inputs may live on the CPU or the GPU (the reference's drivers call these on CPU tensors after `.cpu()`); the arithmetic
always runs in libmade_hip.so (made_pooled_cosine), there is no CPU fallback, and the result comes back on the device of
`video_embeds` / `music_embeds`.  At dataset scale use `Uni_model.retrieval_sim_matrix` instead: it never materialises the
[bs_m, bs_v, dim] pooled tensor these functions take.
"""
from __future__ import annotations

import torch

from .. import _lib
from ..ops import dt_of


def _dev():
    if not torch.cuda.is_available():
        raise _lib.MadeError("mgsv_amd.modules.metrics needs a GPU (the MaDe hot path has no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _pooled_cosine(anchor: torch.Tensor, pooled: torch.Tensor, anchor_major: bool) -> torch.Tensor:
    """anchor [A, D], pooled [P, A, D] -> cos(anchor[a], pooled[p, a]) as [A, P] (anchor_major) or [P, A]."""
    assert anchor.dim() == 2 and pooled.dim() == 3 and pooled.shape[1] == anchor.shape[0] and pooled.shape[2] == anchor.shape[1], \
        f"Shape mismatch: {tuple(anchor.shape)} vs {tuple(pooled.shape)}"
    dev = _dev()
    a = anchor.detach().to(dev, torch.float32).contiguous()
    p = pooled.detach().to(dev)
    if p.dtype not in (torch.float32, torch.bfloat16):
        p = p.to(torch.float32)
    p = p.contiguous()
    A, D = a.shape
    P = p.shape[0]
    out = torch.empty((A, P) if anchor_major else (P, A), device=dev, dtype=torch.float32)
    sa, sp = (P, 1) if anchor_major else (1, A)
    _lib.check(_lib.lib().made_pooled_cosine(a.data_ptr(), a.stride(0), p.data_ptr(), dt_of(p), out.data_ptr(), sa, sp, A, P, D,
                                             torch.cuda.current_stream().cuda_stream), "made_pooled_cosine")
    return out


def sim_matrix_music_pooling(video_embeds, music_embeds_pooled):
    """reference modules/metrics.py:10-24.  video_embeds [bs_v, dim], music_embeds_pooled [bs_m, bs_v, dim] -> sims [bs_v, bs_m]."""
    return _pooled_cosine(video_embeds, music_embeds_pooled, True).to(video_embeds.device)


def sim_matrix_video_pooling(video_embeds_pooled, music_embeds):
    """reference modules/metrics.py:26-41.  video_embeds_pooled [bs_v, bs_m, dim], music_embeds [bs_m, dim] -> sims [bs_v, bs_m]."""
    return _pooled_cosine(music_embeds, video_embeds_pooled, False).to(music_embeds.device)


def sim_matrix_both_pooling(video_embeds_pooled, music_embeds_pooled):
    """reference modules/metrics.py:43-57 (reachable only through vmr_fusion values the reference's own forward rejects): the mean over
    the middle index of <v_pooled[v, j], m_pooled[m', v]> ... -- kept for API completeness, computed per (v, m) pair on the GPU."""
    assert video_embeds_pooled.dim() == 3 and music_embeds_pooled.dim() == 3
    bs_v, bs_m = video_embeds_pooled.shape[0], music_embeds_pooled.shape[0]
    dev = _dev()
    from .. import ops
    vp = video_embeds_pooled.detach().to(dev, torch.float32).contiguous()
    mp = music_embeds_pooled.detach().to(dev, torch.float32).contiguous()
    vn = ops.l2norm_rows(vp.view(bs_v * bs_m, -1)).view(bs_v, bs_m, -1)
    mn = ops.l2norm_rows(mp.view(bs_m * bs_v, -1)).view(bs_m, bs_v, -1)
    # sims[v] = mean_j( vn[v, j, :] @ mn[:, v, :]^T )[m]  =  (mean_j vn[v, j, :]) @ mn[:, v, :]^T   (the mean commutes with the product)
    vbar = ops.masked_mean(vn, torch.ones(bs_v, bs_m, device=dev))
    sims = _pooled_cosine_raw(vbar, mn)
    assert sims.shape == (bs_v, bs_m), f"Shape mismatch: {sims.shape} != {(bs_v, bs_m)}"
    return sims.to(video_embeds_pooled.device)


def _pooled_cosine_raw(anchor: torch.Tensor, pooled_unit: torch.Tensor) -> torch.Tensor:
    """<anchor[a], pooled_unit[p, a]> for unit-norm pooled rows: cos(.) * |anchor[a]| (made_pooled_cosine + the anchors' norms)."""
    from .. import ops
    cos = _pooled_cosine(anchor, pooled_unit, True)                                # [A, P]
    A, D = anchor.shape
    # |anchor[a]| = <anchor[a], anchor[a] / |anchor[a]|>: a batched 1 x D times D x 1 product on the f32 MFMA path
    an = ops.l2norm_rows(anchor)
    norms = torch.empty(A, 1, device=anchor.device, dtype=torch.float32)
    ops.linear(anchor, an, None, M=1, N=1, K=D, batch=A, a_z_stride=anchor.stride(0), w_z_stride=an.stride(0),
               segs=[ops.Seg(out=norms, ldo=1, out_z_stride=1)])
    ops.row_affine(cos, norms.view(-1), torch.zeros(A, device=anchor.device))       # cos[a, :] *= |anchor[a]|
    return cos
