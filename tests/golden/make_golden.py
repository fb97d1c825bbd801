"""Generate golden vectors from the reference  --  build container only.

Imports the reference from /root/reference (oracle/ref_import.py; nothing is copied),
feeds it the seeded synthetic weights/inputs of mgsv_amd/synth.py and stores its OUTPUTS
as small .npz fixtures.  Inputs are not stored: tests regenerate them from the seeds
recorded in each fixture.  Large tensors are stored as strided sub-samples.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from mgsv_amd.config import cfg_headline, MadeConfig, cfg_plumbing, cfg_native  # noqa: E402
from mgsv_amd import synth  # noqa: E402
from oracle import ref_import  # noqa: E402

SUB = (slice(None), slice(None, None, 7), slice(None, None, 5))     # sub-sampling of [B,T,D] tensors



def provenance():
    """What the fixtures' floating-point results depend on besides the sources: library versions, the host CPU and its vector ISA."""
    import platform
    import scipy
    model, flags = platform.processor() or "unknown", ""
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        m = re.search(r"model name\s*:\s*(.+)", txt)
        if m:
            model = m.group(1).strip()
        fl = re.search(r"flags\s*:\s*(.+)", txt)
        if fl:
            have = set(fl.group(1).split())
            flags = " ".join(x for x in ("sse4_2", "avx", "avx2", "fma", "avx512f", "avx512bw", "avx512vl", "avx512_vnni", "avx512_bf16") if x in have)
    except OSError:
        pass
    return {"meta_torch": torch.__version__, "meta_numpy": np.__version__, "meta_scipy": scipy.__version__, "meta_cpu_model": model, "meta_cpu_isa": flags}

def forward_fixture(cfg: MadeConfig, B: int, T_v: int, T_a: int, name: str, cfg_overrides: dict):
    sd = synth.make_state_dict(cfg, seed=0)
    inp = synth.make_inputs(cfg, B, T_v, T_a, seed=1)
    ref = ref_import.build_reference_model(cfg, sd)
    t = {k: torch.from_numpy(v) for k, v in inp.items() if isinstance(v, np.ndarray)}
    with torch.no_grad():
        om, lm, fm, mm, im = ref(t["frame_feats"].clone(), t["segment_feats"].clone(), t["frame_masks"].clone(),
                                 t["segment_masks"].clone(), t["spans_target"].clone(), v_duration=t["v_duration"],
                                 video_ids=inp["video_ids"], music_ids=inp["music_ids"], is_train=False)
        fix = dict(
            meta_B=B, meta_T_v=T_v, meta_T_a=T_a, meta_weight_seed=0, meta_data_seed=1,
            meta_cfg_overrides=np.array(repr(sorted(cfg_overrides.items()))),
            pred_logits=om["pred_logits"].numpy(), pred_spans=om["pred_spans"].numpy(),
            proj_queries=om["proj_queries"].numpy(), proj_vid_mem=om["proj_vid_mem"].numpy()[SUB],
            video_feats=fm["video_feats"].numpy(), music_feats=fm["music_feats"].numpy(),
            frame_feats_sub=fm["frame_feats"].numpy()[SUB], segment_feats_sub=fm["segment_feats"].numpy()[SUB],
            retrieval_loss=np.float32(lm["retrieval_loss"]), localization_loss=np.float32(lm["localization_loss"]),
        )
        for i, aux in enumerate(om["aux_outputs"]):
            fix[f"aux{i}_pred_logits"] = aux["pred_logits"].numpy()
            fix[f"aux{i}_pred_spans"] = aux["pred_spans"].numpy()
        for k, v in lm["localization_loss_dict"].items():
            fix["loss_" + k] = np.float32(v)
        pooled = ref.video_guided_to_music_pooling_cross_transformer(
            fm["video_feats"], fm["segment_feats"], mm["segment_masks"] if cfg.fusion_mask == 1 else None)
        from modules.metrics import sim_matrix_music_pooling
        from modules.loss import cal_distance
        fix["music_feats_pooled_sub"] = pooled.numpy()[:, :, ::5]
        fix["sims_single"] = sim_matrix_music_pooling(fm["video_feats"], pooled).numpy()
        fix["sims_dual"] = cal_distance(fm["video_feats"], fm["music_feats"]).numpy()
        fus_mask = torch.cat([mm["frame_masks"], mm["segment_masks"]], 1) if "concat" in cfg.mml_fusion else mm["segment_masks"]
        fix["detr_pos_sub"] = ref.music_position_embedding(torch.zeros(1), fus_mask).numpy()[SUB]
        idx = ref.criterion.matcher({"pred_logits": om["pred_logits"], "pred_spans": om["pred_spans"]}, t["spans_target"])
        fix["matcher_pred_idx"] = np.stack([i.numpy() for i, _ in idx])
        fix["matcher_tgt_idx"] = np.stack([j.numpy() for _, j in idx])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fix)
    print(name, {k: getattr(v, "shape", None) for k, v in fix.items() if not k.startswith("meta")})


VARIANTS = {      # name -> cfg overrides (on cfg_native): the option variants of SURVEY section 8(f) item 4
    "xa_music_video_single": {"vmr_fusion": "XA-music-video", "vmr_loss": "single"},
    "xa_video_single": {"vmr_fusion": "XA-video", "vmr_loss": "single"},
    "predict_center": {"predict_center": 1},
    "audio_short_cut_fb10": {"audio_short_cut": 1, "fb_label": "10"},
    "audio_short_cut_Q3": {"audio_short_cut": 1, "num_moment_queries": 3},
    "xpool_query": {"moment_query_type": "xpool"},
    "moment_embedding": {"moment_loss": 1, "audio_short_cut": 1},
    "feature_fuse": {"vmr_loss": "dual_single_feature_fuse"},
    "regression": {"mml_localization": "regression"},
    "regression_center_CA": {"mml_localization": "regression", "predict_center": 1, "mml_fusion": "CA"},
    "shared_temporal_block": {"transformer_is_share": 1},
    "cls_token": {"with_cls_token": 1},
    "agg_mlp": {"agg_module": "mlp", "video_transformer_depth": 0, "audio_transformer_depth": 0},
    "pre_norm": {"detr_pre_norm": True},                      # round 5: pre-norm DETR layers (music_detr/transformer.py:170-189,246-271)
    "pre_norm_Q3_CA": {"detr_pre_norm": True, "num_moment_queries": 3, "mml_fusion": "CA"},
}


def variants_fixture(B: int = 4, T_v: int = 50, T_a: int = 96):
    """One small fixture for all option variants: what the reference's forward returns for each."""
    fix = dict(meta_B=B, meta_T_v=T_v, meta_T_a=T_a, meta_weight_seed=0, meta_data_seed=1)
    for name, ov in VARIANTS.items():
        cfg = cfg_native()
        for k, v in ov.items():
            setattr(cfg, k, v)
        sd = synth.make_state_dict(cfg, seed=0)
        inp = synth.make_inputs(cfg, B, T_v, T_a, seed=1)
        ref = ref_import.build_reference_model(cfg, sd)
        t = {k: torch.from_numpy(v) for k, v in inp.items() if isinstance(v, np.ndarray)}
        with torch.no_grad():
            om, lm, fm, mm, im = ref(t["frame_feats"].clone(), t["segment_feats"].clone(), t["frame_masks"].clone(),
                                     t["segment_masks"].clone(), t["spans_target"].clone(), v_duration=t["v_duration"],
                                     video_ids=inp["video_ids"], music_ids=inp["music_ids"], is_train=False)
        for k in ("pred_logits", "pred_spans", "proj_queries", "moment_feats"):
            if k in om:
                fix[f"{name}.{k}"] = om[k].numpy()
        fix[f"{name}.video_feats"] = fm["video_feats"].numpy()
        fix[f"{name}.music_feats"] = fm["music_feats"].numpy()
        fix[f"{name}.retrieval_loss"] = np.float32(lm["retrieval_loss"])
        fix[f"{name}.localization_loss"] = np.float32(lm["localization_loss"])
        for k, v in lm["localization_loss_dict"].items():
            fix[f"{name}.loss_{k}"] = np.float32(v)
    np.savez_compressed(os.path.join(HERE, "variants.npz"), **fix)
    print("variants:", len(fix), "entries for", list(VARIANTS))


def matcher_fixture():
    """Known-answer test held by the reference (music_detr/test_matcher.py:15-29, expected
    output in its comment at :28) plus seeded random Q x G cases with ties, zero-width
    targets and both label conventions, answered by the reference's HungarianMatcher."""
    ref_import.import_reference()
    from music_detr.matcher import build_matcher, HungarianMatcher
    import argparse
    fix = {}
    # KAT: call adapted only as far as the current signature requires
    kat_logits = torch.tensor([[[0.8, 0.2], [0.5, 0.5], [0.1, 0.9]]])
    kat_spans = torch.tensor([[[0.6, 0.15], [0.8, 0.05], [0.2, 0.1]]])
    kat_tg = torch.tensor([[[0.3, 0.1], [0.7, 0.2]]])
    for fb in ("01", "10"):
        m = HungarianMatcher(argparse.Namespace(fb_label=fb))
        (i, j), = m({"pred_logits": kat_logits, "pred_spans": kat_spans}, kat_tg)
        assert i.tolist() == [0, 2] and j.tolist() == [1, 0], (i, j)       # the comment's answer
    fix["kat_logits"], fix["kat_spans"], fix["kat_targets"] = kat_logits.numpy(), kat_spans.numpy(), kat_tg.numpy()
    fix["kat_pred_idx"], fix["kat_tgt_idx"] = np.array([0, 2]), np.array([1, 0])
    rng = np.random.default_rng(11)
    cases = []
    for t in range(48):
        B, Q, G = int(rng.integers(1, 5)), int(rng.integers(1, 9)), int(rng.integers(1, 7))
        logits = rng.standard_normal((B, Q, 2)).astype(np.float32)
        c = rng.uniform(0.1, 0.9, (B, Q, 1)); w = rng.uniform(0.01, 0.5, (B, Q, 1))
        spans = np.concatenate([c, w], -1).astype(np.float32)
        tc = rng.uniform(0.1, 0.9, (B, G, 1)); tw = rng.uniform(0.02, 0.4, (B, G, 1))
        tg = np.concatenate([tc, tw], -1).astype(np.float32)
        if t % 3 == 1:                      # zero-width targets are dropped (matcher.py:59-61)
            drop = rng.random((B, G)) < 0.35
            drop[:, 0] = False
            tg[..., 1][drop] = 0.0
        if t % 4 == 2:                      # exact ties: duplicate predictions / coarse grid
            spans = np.round(spans * 4) / 4 + np.float32(0.125)
            logits = np.round(logits)
            spans[:, -1] = spans[:, 0]; logits[:, -1] = logits[:, 0]
        fb = "01" if t % 2 == 0 else "10"
        args = argparse.Namespace(fb_label=fb, span_loss_type="l1", max_snippet_num=96)
        m = build_matcher(args)
        res = m({"pred_logits": torch.from_numpy(logits), "pred_spans": torch.from_numpy(spans)}, torch.from_numpy(tg))
        pi = -np.ones((B, max(Q, G)), dtype=np.int64); tj = -np.ones((B, max(Q, G)), dtype=np.int64)
        for b, (i, j) in enumerate(res):
            pi[b, :len(i)] = i.numpy(); tj[b, :len(j)] = j.numpy()
        cases.append((logits, spans, tg, fb, pi, tj))
    fix["n_cases"] = len(cases)
    for n, (l, s, tg, fb, pi, tj) in enumerate(cases):
        fix[f"c{n}_logits"], fix[f"c{n}_spans"], fix[f"c{n}_targets"] = l, s, tg
        fix[f"c{n}_fg"] = np.int64(0 if fb == "01" else 1)
        fix[f"c{n}_pred_idx"], fix[f"c{n}_tgt_idx"] = pi, tj
    # span_utils doctests (music_detr/span_utils.py:48-54, :99-103)
    from music_detr.span_utils import temporal_iou, generalized_temporal_iou
    s1 = torch.Tensor([[0, 0.2], [0.5, 1.0]]); s2 = torch.Tensor([[0, 0.3], [0., 1.0]])
    iou, union = temporal_iou(s1, s2)
    assert np.allclose(iou.numpy(), [[0.6667, 0.2], [0.0, 0.5]], atol=1e-4)
    assert np.allclose(union.numpy(), [[0.3, 1.0], [0.8, 1.0]], atol=1e-4)
    g = generalized_temporal_iou(s1, s2)
    assert np.allclose(g.numpy(), [[0.6667, 0.2], [-0.2, 0.5]], atol=1e-4)
    fix["doc_spans1"], fix["doc_spans2"] = s1.numpy(), s2.numpy()
    fix["doc_iou"], fix["doc_union"], fix["doc_giou"] = iou.numpy(), union.numpy(), g.numpy()
    # provenance: the class probabilities behind the costs go through torch's CPU softmax, whose exp is a 1 - 2 ulp approximation that
    # moves with the torch build and the host's vector ISA -- an exact tie of the fixture can then fall the other way.
    prov = provenance()
    old_path = os.path.join(HERE, "matcher.npz")
    if os.path.isfile(old_path):
        old = np.load(old_path)
        if "meta_torch" in old.files:
            was = {k: str(old[k]) for k in ("meta_torch", "meta_numpy", "meta_scipy", "meta_cpu_model", "meta_cpu_isa")}
            if was != prov:
                print("matcher.npz: REGENERATING UNDER ANOTHER ENVIRONMENT -- recorded", was, "now", prov,
                      "(exact ties of the fixture may move: re-check MATCHER_FIXTURE_TIE_SAMPLES in tests/test_ops_gpu.py)")
            same = all(np.array_equal(old[k], fix[k]) for k in fix if k in old.files)
            assert same or was != prov, "the same environment must reproduce the committed fixture bit for bit"
    fix.update({k: np.array(v) for k, v in prov.items()})
    np.savez_compressed(os.path.join(HERE, "matcher.npz"), **fix)
    print("matcher.npz", len(cases), "cases")


def retrieval_fixture():
    cfg = cfg_native()
    sd = synth.make_state_dict(cfg, seed=0)
    ref = ref_import.build_reference_model(cfg, sd)
    from modules.metrics import sim_matrix_music_pooling
    from modules.loss import cal_distance
    fix = {}
    for tag, (N_v, N_m, S) in {"a": (96, 80, 96), "b": (33, 17, 40)}.items():
        ri = synth.make_retrieval_inputs(N_v, N_m, S, cfg.D, seed=2)
        with torch.no_grad():
            v = torch.from_numpy(ri["video_embeds"]); s = torch.from_numpy(ri["segment_embeds"])
            m = torch.from_numpy(ri["segment_masks"]); mu = torch.from_numpy(ri["music_embeds"])
            pooled = ref.video_guided_to_music_pooling_cross_transformer(v, s, m)
            single = sim_matrix_music_pooling(v, pooled)
            dual = cal_distance(v, mu)
        fix[f"{tag}_shape"] = np.array([N_v, N_m, S, cfg.D])
        fix[f"{tag}_single"], fix[f"{tag}_dual"] = single.numpy(), dual.numpy()
        fix[f"{tag}_sim"] = (single + dual).numpy()
    np.savez_compressed(os.path.join(HERE, "retrieval.npz"), **fix)
    print("retrieval.npz")


def train_fixture(cfg: MadeConfig, B: int, T_v: int, T_a: int, name: str, cfg_overrides: dict, seed: int = 1234):
    """Losses and parameter gradients of the reference in train() mode (with the build's dropout masks, see
    oracle/ref_import.reference_grads) and in eval() mode, computed in float64.  Per parameter: L2 norm, sum and a
    strided sample of 512 entries."""
    from oracle import made_oracle as O
    from oracle.validate_against_reference import _RecordingDrop
    sd = synth.make_state_dict(cfg, seed=0)
    inp = synth.make_inputs(cfg, B, T_v, T_a, seed=1)
    ref = ref_import.build_reference_model(cfg, sd).double()
    P = {k: (v.double() if v.is_floating_point() else v) for k, v in O.to_torch_params(sd).items()}
    drop = _RecordingDrop(seed)
    drop.p_detr = cfg.detr_dropout
    with torch.no_grad():                                       # only to learn the order / shapes of the dropout calls
        O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"],
                  inp["spans_target"], v_duration=inp["v_duration"], drop=drop)
    fix = dict(meta_B=B, meta_T_v=T_v, meta_T_a=T_a, meta_weight_seed=0, meta_data_seed=1, meta_dropout_seed=seed,
               meta_cfg_overrides=np.array(repr(sorted(cfg_overrides.items()))))
    for mode, train in (("train", True), ("eval", False)):
        lm, grads = ref_import.reference_grads(ref, inp, train, schedule=drop.calls if train else None, seed=seed,
                                               dtype=torch.float64)
        fix[f"{mode}.retrieval_loss"] = np.float64(lm["retrieval_loss"].detach())
        fix[f"{mode}.localization_loss"] = np.float64(lm["localization_loss"].detach())
        for k, v in lm["localization_loss_dict"].items():
            fix[f"{mode}.loss_{k}"] = np.float64(v.detach())
        for n, g in grads.items():
            flat = g.reshape(-1).numpy()
            step = max(1, flat.size // 512)
            fix[f"{mode}.gnorm.{n}"] = np.float64(np.sqrt((flat ** 2).sum()))
            fix[f"{mode}.gsum.{n}"] = np.float64(flat.sum())
            fix[f"{mode}.gsample.{n}"] = flat[::step][:512].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fix)
    print("wrote", name, len(fix), "entries")


TRAIN_VARIANTS = {   # round-2 training variants: overrides on cfg_native, B = 3, T_v = 20, T_a = 40
    "cls": {"with_cls_token": 1},
    "mlp": {"agg_module": "mlp", "video_transformer_depth": 0, "audio_transformer_depth": 0, "max_v_frames": 20, "max_snippet_num": 40},
    "tower2": {"vmr_fusion": "XA-video-music", "vmr_loss": "single"},
    "xpool_query": {"moment_query_type": "xpool"},
    "feature_fuse": {"vmr_loss": "dual_single_feature_fuse"},
    "pre_norm": {"detr_pre_norm": True},
    "pre_norm_Q2": {"detr_pre_norm": True, "num_moment_queries": 2},
}


def train_variants_fixture(seed: int = 1234, sample: int = 48):
    """The reference's train()-mode losses and float64 parameter gradients (norm + a strided sample per parameter) for the
    round-2 training variants, one file; for the mlp aggregator also its BatchNorm buffers after the step."""
    from oracle import made_oracle as O
    from oracle.validate_against_reference import _RecordingDrop
    fix = dict(meta_B=3, meta_T_v=20, meta_T_a=40, meta_dropout_seed=seed, meta_sample=sample)
    for tag, ov in TRAIN_VARIANTS.items():
        cfg = cfg_native()
        for k, v in ov.items():
            setattr(cfg, k, v)
        sd = synth.make_state_dict(cfg, seed=0)
        inp = synth.make_inputs(cfg, 3, 20, 40, seed=1)
        ref = ref_import.build_reference_model(cfg, sd).double()
        P = {k: (v.double() if v.is_floating_point() else v) for k, v in O.to_torch_params(sd).items()}
        drop = _RecordingDrop(seed)
        drop.p_detr = cfg.detr_dropout
        with torch.no_grad():
            O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"],
                      inp["spans_target"], v_duration=inp["v_duration"], drop=drop)
        lm, grads = ref_import.reference_grads(ref, inp, True, schedule=drop.calls, seed=seed, dtype=torch.float64)
        fix[f"{tag}.overrides"] = np.array(repr(sorted(ov.items())))
        fix[f"{tag}.retrieval_loss"] = np.float64(lm["retrieval_loss"].detach())
        fix[f"{tag}.localization_loss"] = np.float64(lm["localization_loss"].detach())
        for n, g in grads.items():
            flat = g.reshape(-1).numpy()
            step = max(1, flat.size // sample)
            fix[f"{tag}.gnorm.{n}"] = np.float64(np.sqrt((flat ** 2).sum()))
            fix[f"{tag}.gsample.{n}"] = flat[::step][:sample].astype(np.float32)
        for k, v in ref.state_dict().items():
            if k.endswith((".running_mean", ".running_var")):
                fix[f"{tag}.buffer.{k}"] = v.double().numpy()
        print("variant", tag, len(grads), "gradients")
    np.savez_compressed(os.path.join(HERE, "train_variants_B3.npz"), **fix)
    print("wrote train_variants_B3", len(fix), "entries")


def metrics_fixture():
    """Evaluation metrics of the reference's drivers on seeded inputs (utils/util_test.py, music_detr/span_utils.py)."""
    import argparse
    ref_import.import_reference()
    from utils.util_test import Recall_metrics, IoU_metrics, Composite_metrics
    from music_detr.span_utils import detr_iou, span_cw_to_se
    rng = np.random.default_rng(11)
    N = 60
    ids = [f"m{int(x)}" for x in rng.integers(0, 23, size=N)]          # many videos share a music id
    sim = rng.standard_normal((N, N)).astype(np.float32)
    for i in range(N):                                                   # make the ground truth competitive
        sim[i, i] += 1.5
    met, ind, res = Recall_metrics(sim, dedup=True, all_music_ids_list=ids)
    Q = 3
    logits = rng.standard_normal((N, Q, 2)).astype(np.float32)
    spans = np.stack([rng.uniform(0.1, 0.9, (N, Q)), rng.uniform(0.02, 0.6, (N, Q))], -1).astype(np.float32)
    gt = np.sort(rng.uniform(0, 200, (N, 1, 2)), axis=-1).astype(np.float32)
    gt[5, 0, 1] = gt[5, 0, 0]                                            # degenerate moment -> IoU 0
    dur = rng.uniform(150, 240, N).astype(np.float32)
    args = argparse.Namespace(max_m_duration=240)
    mr = []
    prob = torch.softmax(torch.from_numpy(logits), -1)[:, :, 0]
    for i in range(N):
        se = span_cw_to_se(torch.from_numpy(spans[i])) * 240
        rp = torch.cat((se, prob[i].unsqueeze(-1)), dim=-1)
        rp = sorted(rp, key=lambda x: x[2], reverse=True)
        mr.append(dict(gt_moment=torch.from_numpy(gt[i]), m_duration=torch.tensor(dur[i]), ranked_preds=rp))
    iou = [float(x) for x in detr_iou(args, mr)]
    loc = IoU_metrics(iou)
    com = Composite_metrics(ind, iou, mr, ids, ids)
    fix = dict(sim=sim, ids=np.array(ids), ind=np.asarray(ind), logits=logits, spans=spans, gt=gt, dur=dur, iou=np.asarray(iou, dtype=np.float32),
               top1=np.array([r["topk_music_ids"][0] for r in res]))
    for k, v in met.items():
        if k != "cols":
            fix["ret." + k] = np.float64(v)
    for k, v in loc.items():
        fix["loc." + k] = np.float64(v)
    for k, v in com.items():
        fix["com." + k] = np.float64(v)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **fix)
    print("wrote metrics", len(fix), "entries; R1 =", met["R1"], "mIoU =", loc["mIoU"])


def bench_shape_fixtures():
    """Gradients of the reference at the shapes bench.py times: BASELINE configs[2] (D = 512, T_a = 512) and the per-GPU shape of
    configs[4] (T_a = 1024: the audio position table is rebuilt for 1024 positions on both sides), B = 2."""
    train_fixture(cfg_headline(), 2, 30, 512, "train_cfg2_B2", {"_cfg": "headline"})
    c = cfg_headline(); c.max_snippet_num = 1024; c.audio_attention_seqlen = 1024
    train_fixture(c, 2, 30, 1024, "train_cfg4_Ta1024_B2", {"_cfg": "headline", "max_snippet_num": 1024, "audio_attention_seqlen": 1024})


def main():
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "matcher":          # the matcher fixture alone (its provenance record: round 6)
        ref_import.import_reference()
        return matcher_fixture()
    if len(sys.argv) > 1 and sys.argv[1] == "bench_shapes":
        return bench_shape_fixtures()
    if len(sys.argv) > 1 and sys.argv[1] == "train_variants":
        return train_variants_fixture()
    variants_fixture()
    metrics_fixture()
    train_fixture(cfg_native(), 3, 20, 40, "train_native_B3", {})
    train_variants_fixture()
    forward_fixture(cfg_plumbing(), 2, 30, 200, "forward_cfg1_B2", {})
    c = cfg_native(); c.num_moment_queries = 3
    forward_fixture(c, 4, 50, 96, "forward_native_Q3_B4", {"num_moment_queries": 3})
    matcher_fixture()
    retrieval_fixture()


if __name__ == "__main__":
    main()
