"""Evaluation metrics of the drivers (reference utils/util_test.py:10-199), same names, arguments and return values.

`Recall_metrics` accepts the similarity matrix as a numpy array (host path, for small matrices) or as a CUDA tensor: then the
de-duplicated ranks are computed where the matrix lives (made_recall_ranks) and only one int per video crosses PCIe -- at
dataset scale the reference moves an 848 MB matrix to the host and walks it with Python loops.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import _lib, ops


def calc_similarity(video_feat_list, audio_feat_list, distance_type: str = "COS"):
    """reference utils/util_test.py:10-29: cosine similarity of all (video, music) vectors -> [val_len, val_len] float32
    (CUDA tensor).  Inputs: lists of [bs, dim] arrays / tensors."""
    if distance_type != "COS":
        raise ValueError("only the COS distance of the reference's scripts is on the HIP path")
    dev = torch.device("cuda")
    v = torch.cat([torch.as_tensor(x) for x in video_feat_list], 0).to(dev, torch.float32).contiguous()
    a = torch.cat([torch.as_tensor(x) for x in audio_feat_list], 0).to(dev, torch.float32).contiguous()
    return ops.linear(ops.l2norm_rows(v), ops.l2norm_rows(a), None, out_dtype=torch.float32)


def _summarise(ind: np.ndarray) -> dict:
    """reference utils/util_test.py:82-96."""
    n = len(ind)
    m = {}
    for k in (1, 3, 5, 10, 20, 25, 50, 100):
        m[f"R{k}"] = float(np.sum(ind < k)) * 100 / n if k > 1 else float(np.sum(ind == 0)) * 100 / n
    m["MedianR"] = np.median(ind) + 1
    m["MeanR"] = np.mean(ind) + 1
    m["cols"] = [int(i) for i in list(ind)]
    m["MRR"] = np.mean(1.0 / (ind + 1))
    return m


def recall_ranks_device(sim: torch.Tensor, group_id: Sequence[int], gt_group: Sequence[int]):
    """ranks [Nv] int32 and top-1 columns [Nv] int32 (CUDA tensors) of a CUDA similarity matrix."""
    assert sim.is_cuda and sim.dtype == torch.float32 and sim.dim() == 2 and sim.stride(1) == 1
    Nv, Nm = sim.shape
    dev = sim.device
    gid = torch.as_tensor(np.asarray(group_id, dtype=np.int32)).to(dev)
    gt = torch.as_tensor(np.asarray(gt_group, dtype=np.int32)).to(dev)
    G = int(np.max(group_id)) + 1
    rank = torch.empty(Nv, device=dev, dtype=torch.int32)
    top1 = torch.empty(Nv, device=dev, dtype=torch.int32)
    _lib.check(_lib.lib().made_recall_ranks(sim.data_ptr(), sim.stride(0), gid.data_ptr(), gt.data_ptr(), Nv, Nm, G, rank.data_ptr(),
                                            top1.data_ptr(), torch.cuda.current_stream().cuda_stream), "made_recall_ranks")
    return rank, top1


def Recall_metrics(sim_matrix, distance_type: str = "COS", dedup: bool = False, all_music_ids_list: Optional[List] = None):
    """reference utils/util_test.py:32-96.  Returns (metrics, ind, ret_results_list)."""
    ids = list(all_music_ids_list) if all_music_ids_list is not None else []
    sim = sim_matrix if isinstance(sim_matrix, torch.Tensor) else torch.as_tensor(np.asarray(sim_matrix, dtype=np.float32))
    if not sim.is_cuda:
        sim = sim.to("cuda")
    sim = sim.to(torch.float32).contiguous()
    Nv, Nm = sim.shape
    ret_results_list = []
    if dedup and len(ids) > 0:
        lut = {}
        gid = [lut.setdefault(m, len(lut)) for m in ids]      # group = music id, numbered in order of first appearance
        rank, top1 = recall_ranks_device(sim, gid, gid)
        ind = rank.cpu().numpy().astype(np.int64)
        top1 = top1.cpu().numpy()
        for i, m in enumerate(ids):
            ret_results_list.append(dict(music_id=m, rank=int(ind[i]) + 1, topk_music_ids=[ids[int(top1[i])]]))
    else:                                                      # position of the diagonal element in the sorted row
        rank, _ = recall_ranks_device(sim, list(range(Nm)), list(range(Nv)))
        ind = rank.cpu().numpy().astype(np.int64)
    return _summarise(ind), ind, ret_results_list


def IoU_metrics(IoU_list):
    """reference utils/util_test.py:101-111."""
    v = np.asarray([float(x) for x in IoU_list], dtype=np.float64)
    n = len(v)
    return {"mIoU": float(v.sum() / n), "IoU@0.3": float((v > 0.3).sum() * 100 / n), "IoU@0.5": float((v > 0.5).sum() * 100 / n),
            "IoU@0.7": float((v > 0.7).sum() * 100 / n)}


def Composite_metrics(ret_rank_list, IoU_list, mr_results_list=None, all_video_ids_list=None, all_music_ids_list=None):
    """reference utils/util_test.py:142-199: localisation quality among the videos whose music was retrieved within the top k."""
    rank = np.asarray(ret_rank_list, dtype=np.int64) + 1
    iou = np.asarray([float(x) for x in IoU_list], dtype=np.float64)
    n = len(rank)
    out = {}
    for k in (1, 10, 50, 100):
        sel = rank <= k if k > 1 else rank == 1
        out[f"R{k}_iou0.5"] = float((iou[sel] > 0.5).sum()) / n * 100
        out[f"R{k}_iou0.7"] = float((iou[sel] > 0.7).sum()) / n * 100
        out[f"R{k}_miou"] = float(iou[sel].sum() / n / sel.sum()) if sel.sum() > 0 else 0.0
    # key order of the reference
    return {k: out[k] for k in ("R1_iou0.5", "R10_iou0.5", "R50_iou0.5", "R100_iou0.5", "R1_iou0.7", "R10_iou0.7", "R50_iou0.7", "R100_iou0.7",
                                "R1_miou", "R10_miou", "R50_miou", "R100_miou")}


def detr_iou_device(pred_logits: torch.Tensor, pred_spans: torch.Tensor, gt_moment: torch.Tensor, m_duration: torch.Tensor,
                    foreground_label: int, max_m_duration: float):
    """IoU of the top-scoring predicted span of every sample (reference test-MaDe.py:304-313 + music_detr/span_utils.py:146-170).
    pred_logits / pred_spans [N,Q,2], gt_moment [N,1,2] or [N,2] seconds, m_duration [N] -> (iou [N], ranked_pred0 [N,3])."""
    dev = pred_logits.device
    N, Q = pred_logits.shape[0], pred_logits.shape[1]
    lg = pred_logits.to(torch.float32).contiguous()
    sp = pred_spans.to(torch.float32).contiguous()
    gt = gt_moment.to(dev, torch.float32).reshape(N, -1)[:, :2].contiguous()
    dur = m_duration.to(dev, torch.float32).contiguous()
    iou = torch.empty(N, device=dev, dtype=torch.float32)
    pred = torch.empty(N, 3, device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().made_span_iou(lg.data_ptr(), sp.data_ptr(), gt.data_ptr(), dur.data_ptr(), N, Q, int(foreground_label),
                                        float(max_m_duration), iou.data_ptr(), pred.data_ptr(), torch.cuda.current_stream().cuda_stream),
               "made_span_iou")
    return iou, pred
