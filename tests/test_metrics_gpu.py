"""GPU: evaluation metrics on the device (made_recall_ranks, made_span_iou) behind the reference's util_test API, against the
values the reference's own functions produced (tests/golden/metrics.npz) and against the oracle on larger random cases."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_metrics_match_reference_fixture(golden_dir):
    from mgsv_amd.utils.util_test import Composite_metrics, IoU_metrics, Recall_metrics, detr_iou_device
    fix = np.load(os.path.join(golden_dir, "metrics.npz"))
    ids = [str(x) for x in fix["ids"]]
    sim = torch.from_numpy(fix["sim"]).cuda()
    met, ind, res = Recall_metrics(sim, dedup=True, all_music_ids_list=ids)
    assert ind.tolist() == fix["ind"].tolist()
    assert [r["topk_music_ids"][0] for r in res] == [str(x) for x in fix["top1"]]
    assert [r["rank"] for r in res] == (fix["ind"] + 1).tolist()
    for k in ("R1", "R3", "R5", "R10", "R20", "R25", "R50", "R100", "MedianR", "MeanR", "MRR"):
        assert abs(float(met[k]) - float(fix["ret." + k])) <= 1e-9, k
    iou, pred = detr_iou_device(torch.from_numpy(fix["logits"]).cuda(), torch.from_numpy(fix["spans"]).cuda(), torch.from_numpy(fix["gt"]).cuda(),
                                torch.from_numpy(fix["dur"]).cuda(), 0, 240.0)
    assert float((iou.cpu() - torch.from_numpy(fix["iou"])).abs().max()) <= 1e-5
    loc = IoU_metrics(iou.cpu().tolist())
    for k in ("mIoU", "IoU@0.3", "IoU@0.5", "IoU@0.7"):
        assert abs(loc[k] - float(fix["loc." + k])) <= 1e-5, k
    com = Composite_metrics(ind, iou.cpu().tolist(), None, ids, ids)
    for k, v in com.items():
        assert abs(v - float(fix["com." + k])) <= 1e-5, k


def test_recall_ranks_large_random_vs_oracle():
    from oracle import made_oracle as O
    from mgsv_amd.utils.util_test import Recall_metrics
    rng = np.random.default_rng(3)
    Nv = Nm = 700
    ids = [int(x) for x in rng.integers(0, 500, size=Nm)]
    sim = rng.standard_normal((Nv, Nm)).astype(np.float32)
    met, ind, _ = Recall_metrics(torch.from_numpy(sim).cuda(), dedup=True, all_music_ids_list=ids)
    assert ind.tolist() == O.recall_ranks_dedup(sim, ids).tolist()
    met2, ind2, _ = Recall_metrics(sim, dedup=False)
    assert ind2.tolist() == O.recall_ranks_plain(sim).tolist()
