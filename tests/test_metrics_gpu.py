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


def test_eval_metrics_of_a_synthetic_split_match_the_oracle_end_to_end():
    """north_star asks for retrieval R@k and mIoU within +-0.1 of the reference.  There is no dataset here, so the whole evaluation
    pipeline (per-batch forward -> all-pairs similarity matrix -> de-duplicated recall ranks, top-span IoU -> metrics) is run on
    a synthetic split through the HIP path (f32 engine) and through the oracle: every rank and every IoU must agree, hence every
    metric; the bf16 engine's mIoU stays within 0.02 and its similarity rows rank the tracks like the oracle's (Spearman >= 0.98)."""
    import torch
    from mgsv_amd import synth
    from mgsv_amd.config import cfg_native
    from mgsv_amd.engine import MadeEngine
    from mgsv_amd.utils.util_test import IoU_metrics, Recall_metrics, detr_iou_device
    from oracle import made_oracle as O

    cfg = cfg_native()
    sd = synth.make_state_dict(cfg, seed=0)
    P = O.to_torch_params(sd)
    N, B = 96, 32
    rng = np.random.Generator(np.random.PCG64(11))
    mids = [f"m{int(i)}" for i in rng.integers(0, 40, size=N)]        # several videos share a track: the de-duplication matters
    m_dur = rng.uniform(60.0, 240.0, size=N).astype(np.float32)
    gs = rng.uniform(0.0, 0.6, size=N).astype(np.float32) * m_dur
    gt = np.stack([gs, np.minimum(gs + rng.uniform(5.0, 45.0, size=N).astype(np.float32), m_dur)], 1).astype(np.float32)
    batches = [synth.make_inputs(cfg, B, 50, 96, seed=100 + i) for i in range(N // B)]

    def run_oracle():
        V, M, S, SM, LG, SP = [], [], [], [], [], []
        with torch.no_grad():
            for inp in batches:
                r = O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"], inp["spans_target"],
                              v_duration=inp["v_duration"], with_losses=False)
                V.append(r["video_feats"]); M.append(r["music_feats"]); S.append(r["segment_feats"]); SM.append(torch.from_numpy(inp["segment_masks"]))
                LG.append(r["pred_logits"]); SP.append(r["pred_spans"])
            sim = O.retrieval_sim_matrix(P, cfg, torch.cat(V), torch.cat(S), torch.cat(SM), torch.cat(M))
            iou = O.top_span_iou(torch.cat(LG), torch.cat(SP), torch.from_numpy(gt), torch.from_numpy(m_dur), cfg.foreground_label, float(cfg.max_m_duration))
        return sim.numpy(), O.recall_ranks_dedup(sim.numpy(), mids), iou.numpy()

    def run_hip(dtype):
        eng = MadeEngine(cfg, sd, dtype=dtype)
        dev = eng.device
        V, M, S, SM, LG, SP = [], [], [], [], [], []
        for inp in batches:
            t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
            o = eng.forward(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], with_losses=False)
            torch.cuda.synchronize()
            V.append(o["video_feats"].clone()); M.append(o["music_feats"].clone()); S.append(o["segment_feats"].float().clone()); SM.append(t["segment_masks"])
            LG.append(o["pred_logits"].clone()); SP.append(o["pred_spans"].clone())
        sim = eng.retrieval_sim_matrix(torch.cat(V), torch.cat(S), torch.cat(SM), torch.cat(M))
        met, ranks, _ = Recall_metrics(sim, dedup=True, all_music_ids_list=mids)
        iou, _ = detr_iou_device(torch.cat(LG), torch.cat(SP), torch.from_numpy(gt).to(dev), torch.from_numpy(m_dur).to(dev),
                                 cfg.foreground_label, float(cfg.max_m_duration))
        torch.cuda.synchronize()
        return sim.cpu().numpy(), ranks, iou.cpu().numpy(), met

    sim_o, rank_o, iou_o = run_oracle()
    sim_h, rank_h, iou_h, met_h = run_hip("f32")
    np.testing.assert_allclose(sim_h, sim_o, atol=1e-4, rtol=0)
    # a rank may only differ where two similarities of that row are closer than the f32 tolerance
    for i in np.nonzero(rank_h != rank_o)[0]:
        srt = np.sort(sim_o[i])
        assert np.min(np.diff(srt)) < 2e-4, (i, rank_h[i], rank_o[i])
    assert np.mean(rank_h != rank_o) <= 0.03
    np.testing.assert_allclose(iou_h, iou_o, atol=2e-4, rtol=0)
    r1_o, r1_h = 100.0 * np.mean(rank_o < 1), met_h["R1"]
    assert abs(r1_o - r1_h) <= 100.0 / N + 1e-6 and abs(IoU_metrics(iou_h.tolist())["mIoU"] - IoU_metrics(iou_o.tolist())["mIoU"]) <= 1e-3
    # bf16 engine: same pipeline within its stated tolerance class
    sim_b, rank_b, iou_b, met_b = run_hip("bf16")
    assert abs(float(np.mean(iou_b)) - float(np.mean(iou_o))) <= 0.02
    def spearman(a, b):
        ra, rb = np.argsort(np.argsort(a)), np.argsort(np.argsort(b))
        return float(np.corrcoef(ra, rb)[0, 1])
    rho = np.mean([spearman(sim_b[i], sim_o[i]) for i in range(N)])
    assert rho >= 0.98, rho
