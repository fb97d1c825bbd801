"""The free functions the reference's drivers import (reference train-MaDe.py:16,20,22) over the HIP kernels:
mgsv_amd.modules.metrics, mgsv_amd.modules.loss, mgsv_amd.music_detr.span_utils -- against the reference's own formulas evaluated with
plain torch on the CPU, the reference's doctest known answers, and the golden fixtures the reference produced."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mgsv_amd.modules import loss as L, metrics as Mx  # noqa: E402
from mgsv_amd.music_detr import span_utils as S  # noqa: E402


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def test_sim_matrices_match_reference_formulas(golden_dir):
    v, pooled = rnd(9, 64, seed=1), rnd(5, 9, 64, seed=2)              # [bs_v, D], [bs_m, bs_v, D]
    vn, pn = v / v.norm(dim=-1, keepdim=True), pooled / pooled.norm(dim=-1, keepdim=True)
    ref = torch.bmm(vn.unsqueeze(1), pn.permute(1, 2, 0)).squeeze(1)   # reference modules/metrics.py:19-23
    got = Mx.sim_matrix_music_pooling(v, pooled)
    assert got.device == v.device and got.shape == (9, 5)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-6)
    m, vp = rnd(5, 64, seed=3), rnd(9, 5, 64, seed=4)                  # [bs_m, D], [bs_v, bs_m, D]
    mn, vpn = m / m.norm(dim=-1, keepdim=True), vp / vp.norm(dim=-1, keepdim=True)
    ref2 = torch.bmm(mn.unsqueeze(1), vpn.permute(1, 2, 0)).squeeze(1).t()     # :35-40
    np.testing.assert_allclose(Mx.sim_matrix_video_pooling(vp, m).numpy(), ref2.numpy(), atol=2e-6)
    ref3 = torch.bmm(vpn, pn.permute(1, 2, 0)).mean(dim=1)             # :53-55
    np.testing.assert_allclose(Mx.sim_matrix_both_pooling(vp, pooled).numpy(), ref3.numpy(), atol=5e-6)
    # the reference's own outputs: pooled track vectors and the similarity matrix it derived from them
    fix = np.load(os.path.join(golden_dir, "retrieval.npz"))
    if "music_embeds_pooled" in fix.files and "single_sim_matrix" in fix.files:
        got = Mx.sim_matrix_music_pooling(torch.from_numpy(fix["video_embeds"]), torch.from_numpy(fix["music_embeds_pooled"]))
        np.testing.assert_allclose(got.numpy(), fix["single_sim_matrix"], atol=1e-5)
    # bf16 pooled vectors straight from the engine
    got_bf = Mx.sim_matrix_music_pooling(v.cuda(), pooled.cuda().bfloat16())
    assert got_bf.is_cuda
    np.testing.assert_allclose(got_bf.cpu().numpy(), ref.numpy(), atol=1e-2)


def test_losses_match_reference_formulas():
    sims = torch.tanh(rnd(12, 12, seed=5))
    ls = torch.tensor(np.log(1 / 0.03), dtype=torch.float32)
    logits = sims * ls.exp()
    lab = torch.arange(12)
    ref_clip = (F.cross_entropy(logits, lab) + F.cross_entropy(logits.t(), lab)) / 2          # reference modules/loss.py:12-24
    np.testing.assert_allclose(float(L.CLIPLoss(sims, ls)), float(ref_clip), rtol=2e-5)
    loss, lv, la = L.InfoNCELoss(sims, ls)                                                     # :116-123 (audio_id None)
    np.testing.assert_allclose(float(loss), float(ref_clip), rtol=2e-5)
    np.testing.assert_allclose(lv.numpy(), logits.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(la.numpy(), logits.t().numpy(), rtol=1e-5, atol=1e-5)
    # same-music-aware branch (:90-114)
    ids = ["a", "b", "a", "c", "d", "b", "e", "f", "g", "a", "h", "i"]
    args = SimpleNamespace(ignore_same_music=0)
    tot = 0.0
    for i in range(12):
        neg = [j for j in range(12) if ids[j] != ids[i]]
        row = torch.cat([logits[i, i].view(1), logits[i, neg]]).view(1, -1)
        tot = tot + F.cross_entropy(row, torch.zeros(1, dtype=torch.long))
    ref_nce = (tot / 12 + F.cross_entropy(logits.t(), lab)) / 2
    got, _, _ = L.InfoNCELoss(sims, ls, audio_id=ids, args=args, is_train=True)
    np.testing.assert_allclose(float(got), float(ref_nce), rtol=2e-5)
    # cal_distance (:52-61), tensors and numpy
    x, y = rnd(7, 32, seed=6), rnd(11, 32, seed=7)
    ref = (x / x.norm(dim=1, keepdim=True)) @ (y / y.norm(dim=1, keepdim=True)).t()
    np.testing.assert_allclose(L.cal_distance(x, y).numpy(), ref.numpy(), atol=2e-6)
    d = L.cal_distance(x.numpy(), y.numpy())
    assert d.dtype == np.float64
    np.testing.assert_allclose(d, ref.numpy(), atol=2e-6)
    with pytest.raises(NotImplementedError):
        L.cal_distance(x, y, "L2")


def test_span_utils_known_answers_and_formulas():
    s1, s2 = torch.Tensor([[0, 0.2], [0.5, 1.0]]), torch.Tensor([[0, 0.3], [0., 1.0]])
    iou, union = S.temporal_iou(s1, s2)                                                       # reference span_utils.py:48-54
    np.testing.assert_allclose(iou.numpy(), [[0.6667, 0.2], [0.0, 0.5]], atol=1e-4)
    np.testing.assert_allclose(union.numpy(), [[0.3, 1.0], [0.8, 1.0]], atol=1e-6)
    np.testing.assert_allclose(S.generalized_temporal_iou(s1, s2).numpy(), [[0.6667, 0.2], [-0.2, 0.5]], atol=1e-4)   # :99-103
    cw = torch.rand(17, 2, generator=torch.Generator().manual_seed(8))
    se = S.span_cw_to_se(cw)
    np.testing.assert_allclose(se.numpy(), torch.stack([cw[:, 0] - 0.5 * cw[:, 1], cw[:, 0] + 0.5 * cw[:, 1]], -1).numpy(), atol=1e-7)
    np.testing.assert_allclose(S.span_se_to_cw(se).numpy(), cw.numpy(), atol=1e-6)
    a = torch.sort(torch.rand(6, 2, generator=torch.Generator().manual_seed(9)), dim=1).values
    b = torch.sort(torch.rand(4, 2, generator=torch.Generator().manual_seed(10)), dim=1).values
    left, right = torch.max(a[:, None, 0], b[:, 0]), torch.min(a[:, None, 1], b[:, 1])
    inter = (right - left).clamp(min=0)
    np.testing.assert_allclose(S.temporal_intersection_over_pred(a, b).numpy(), (inter / (b[:, 1] - b[:, 0])).numpy(), atol=1e-6)
    with pytest.raises(AssertionError):
        S.generalized_temporal_iou(torch.Tensor([[0.5, 0.2]]), s2)
    # individual_IoU_tensor / detr_iou (:119-170), including the clamps, the empty ground truth and the discounted form
    def ref_iou(gs, ge, dur, ps, pe, disc=False):
        if gs >= ge:
            return 0.0
        ps, pe = max(ps, 0.0), min(pe, dur)
        inter = max(min(ge, pe) - max(gs, ps), 0.0)
        uni = (pe - ps) + (ge - gs) - inter
        if uni <= 0:
            return 0.0
        v = inter / uni
        return v * (1 - abs(gs - ps) / dur) * (1 - abs(ge - pe) / dur) if disc else v
    cases = [(10.0, 30.0, 120.0, 12.0, 28.0), (10.0, 30.0, 120.0, -5.0, 400.0), (30.0, 30.0, 90.0, 0.0, 10.0), (5.0, 50.0, 40.0, 20.0, 60.0),
             (100.0, 130.0, 200.0, 10.0, 20.0)]
    for gs, ge, dur, ps, pe in cases:
        for disc in (False, True):
            got = S.individual_IoU_tensor(torch.tensor(gs), torch.tensor(ge), torch.tensor(dur), torch.tensor(ps), torch.tensor(pe), discounted=disc)
            np.testing.assert_allclose(float(got), ref_iou(gs, ge, dur, ps, pe, disc), atol=1e-6)
    args = SimpleNamespace(max_m_duration=240)
    lst = [dict(gt_moment=torch.tensor([[gs, ge]]), m_duration=torch.tensor(dur), ranked_preds=torch.tensor([[ps, pe, 0.9], [0.0, 1.0, 0.1]]))
           for gs, ge, dur, ps, pe in cases]
    got = S.detr_iou(args, lst)
    assert len(got) == len(cases)
    for g, (gs, ge, dur, ps, pe) in zip(got, cases):
        np.testing.assert_allclose(float(g), ref_iou(gs, ge, dur, max(ps, 0.0), min(pe, 240.0)), atol=1e-6)
