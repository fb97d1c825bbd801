"""CPU: the oracle (oracle/made_oracle.py) against the golden vectors the reference produced
(tests/golden/*.npz, made by tests/golden/make_golden.py) and the reference's own known answers."""
import ast
import os

import numpy as np
import pytest
import torch

from mgsv_amd import synth
from mgsv_amd.config import cfg_native, cfg_plumbing
from oracle import made_oracle as O

SUB = (slice(None), slice(None, None, 7), slice(None, None, 5))
TOL = 2e-5      # float32, torch CPU vs torch CPU (oracle/validate_against_reference.py measured <= 1.3e-5)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _cfg_for(name, fix):
    cfg = cfg_plumbing() if "cfg1" in name else cfg_native()
    for k, v in ast.literal_eval(str(fix["meta_cfg_overrides"])):
        setattr(cfg, k, v)
    return cfg


@pytest.mark.parametrize("name", ["forward_cfg1_B2", "forward_native_Q3_B4"])
def test_forward_matches_reference_golden(golden_dir, name):
    fix = _load(golden_dir, name)
    cfg = _cfg_for(name, fix)
    B, T_v, T_a = int(fix["meta_B"]), int(fix["meta_T_v"]), int(fix["meta_T_a"])
    P = O.to_torch_params(synth.make_state_dict(cfg, seed=int(fix["meta_weight_seed"])))
    inp = synth.make_inputs(cfg, B, T_v, T_a, seed=int(fix["meta_data_seed"]))
    with torch.no_grad():
        r = O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"],
                      inp["spans_target"], v_duration=inp["v_duration"], music_ids=inp["music_ids"])
    for k in ("pred_logits", "pred_spans", "proj_queries", "video_feats", "music_feats", "sims_single", "sims_dual"):
        np.testing.assert_allclose(r[k].numpy(), fix[k], atol=TOL, rtol=0, err_msg=k)
    np.testing.assert_allclose(r["proj_vid_mem"].numpy()[SUB], fix["proj_vid_mem"], atol=TOL, rtol=0)
    np.testing.assert_allclose(r["frame_feats"].numpy()[SUB], fix["frame_feats_sub"], atol=TOL, rtol=0)
    np.testing.assert_allclose(r["segment_feats"].numpy()[SUB], fix["segment_feats_sub"], atol=TOL, rtol=0)
    np.testing.assert_allclose(r["music_feats_pooled"].numpy()[:, :, ::5], fix["music_feats_pooled_sub"], atol=TOL, rtol=0)
    np.testing.assert_allclose(r["detr_pos"].numpy()[SUB], fix["detr_pos_sub"], atol=TOL, rtol=0)
    for i, aux in enumerate(r["aux_outputs"]):
        np.testing.assert_allclose(aux["pred_logits"].numpy(), fix[f"aux{i}_pred_logits"], atol=TOL, rtol=0)
        np.testing.assert_allclose(aux["pred_spans"].numpy(), fix[f"aux{i}_pred_spans"], atol=TOL, rtol=0)
    loss_keys = [k for k in fix.files if k.startswith("loss_")]
    assert len(loss_keys) == 5 * cfg.detr_dec_layers            # 30 scalars at dec=6
    for k in loss_keys:
        np.testing.assert_allclose(float(r["loss_dict"][k[len("loss_"):]]), float(fix[k]), atol=5e-5, rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(float(r["retrieval_loss"]), float(fix["retrieval_loss"]), atol=5e-5)
    np.testing.assert_allclose(float(r["localization_loss"]), float(fix["localization_loss"]), atol=2e-4, rtol=1e-5)
    for b, (i, j) in enumerate(r["matcher_indices"]):            # bit-exact
        assert i.tolist() == fix["matcher_pred_idx"][b].tolist()
        assert j.tolist() == fix["matcher_tgt_idx"][b].tolist()


def test_matcher_known_answer_and_random_cases(golden_dir):
    fix = _load(golden_dir, "matcher")
    # the reference's own example: music_detr/test_matcher.py:15-29, answer in its comment at :28
    for fg in (0, 1):
        (i, j), = O.hungarian_match(torch.from_numpy(fix["kat_logits"]), torch.from_numpy(fix["kat_spans"]),
                                    torch.from_numpy(fix["kat_targets"]), fg)
        assert i.tolist() == [0, 2] and j.tolist() == [1, 0]
    for n in range(int(fix["n_cases"])):
        res = O.hungarian_match(torch.from_numpy(fix[f"c{n}_logits"]), torch.from_numpy(fix[f"c{n}_spans"]),
                                torch.from_numpy(fix[f"c{n}_targets"]), int(fix[f"c{n}_fg"]))
        for b, (i, j) in enumerate(res):
            assert i.tolist() == fix[f"c{n}_pred_idx"][b][:len(i)].tolist(), n
            assert j.tolist() == fix[f"c{n}_tgt_idx"][b][:len(j)].tolist(), n
            assert (fix[f"c{n}_pred_idx"][b][len(i):] == -1).all()


def test_span_doctests(golden_dir):
    # music_detr/span_utils.py:48-54 and :99-103
    fix = _load(golden_dir, "matcher")
    s1, s2 = torch.from_numpy(fix["doc_spans1"]), torch.from_numpy(fix["doc_spans2"])
    iou, union = O.temporal_iou(s1, s2)
    assert np.array_equal(iou.numpy(), fix["doc_iou"]) and np.array_equal(union.numpy(), fix["doc_union"])
    assert np.array_equal(O.generalized_temporal_iou(s1, s2).numpy(), fix["doc_giou"])
    np.testing.assert_allclose(fix["doc_giou"], [[0.6667, 0.2], [-0.2, 0.5]], atol=1e-4)


def test_lsap_matches_scipy_and_raises():
    scipy_opt = pytest.importorskip("scipy.optimize")
    rng = np.random.default_rng(3)
    for t in range(300):
        nr, nc = int(rng.integers(1, 10)), int(rng.integers(1, 10))
        c = rng.integers(0, 4, (nr, nc)).astype(np.float64) if t % 2 else rng.standard_normal((nr, nc))
        a, b = scipy_opt.linear_sum_assignment(c)
        oa, ob = O.lsap(c)
        assert a.tolist() == oa.tolist() and b.tolist() == ob.tolist()
    for bad in (np.array([[np.nan, 1.0]]), np.array([[np.inf, np.inf]]), np.array([[-np.inf, 0.0]])):
        with pytest.raises(ValueError):
            O.lsap(bad)
    a, b = O.lsap(np.zeros((0, 3)))
    assert len(a) == 0 and len(b) == 0


def test_retrieval_matches_reference_golden(golden_dir):
    fix = _load(golden_dir, "retrieval")
    cfg = cfg_native()
    P = O.to_torch_params(synth.make_state_dict(cfg, seed=0))
    for tag in ("a", "b"):
        N_v, N_m, S, D = [int(x) for x in fix[f"{tag}_shape"]]
        ri = synth.make_retrieval_inputs(N_v, N_m, S, D, seed=2)
        with torch.no_grad():
            sim = O.retrieval_sim_matrix(P, cfg, ri["video_embeds"], ri["segment_embeds"], ri["segment_masks"],
                                         ri["music_embeds"], chunk_v=29)
        np.testing.assert_allclose(sim.numpy(), fix[f"{tag}_sim"], atol=TOL, rtol=0)


def test_state_dict_layout_counts():
    # SURVEY 5.4 [probe]: 199 tensors / 10.68 M elements at the native config (incl. two PE buffers)
    sd = synth.make_state_dict(cfg_native(), seed=0)
    assert len(sd) == 199
    n = sum(int(np.prod(v.shape)) for v in sd.values())
    assert abs(n - 10.68e6) < 0.02e6


# ------------------------------------------------------------------ training path: losses + gradients
def _oracle_grads(golden_dir, mode: str, double: bool):
    import torch
    from mgsv_amd.config import cfg_native
    fix = _load(golden_dir, "train_native_B3")
    cfg = cfg_native()
    sd = synth.make_state_dict(cfg, seed=int(fix["meta_weight_seed"]))
    inp = synth.make_inputs(cfg, int(fix["meta_B"]), int(fix["meta_T_v"]), int(fix["meta_T_a"]), seed=int(fix["meta_data_seed"]))
    P = O.to_torch_params(sd)
    if double:
        P = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    names = [k[len(mode) + 7:] for k in fix.files if k.startswith(mode + ".gnorm.")]
    for n in names:
        P[n].requires_grad_(True)
    drop = O.Drop(int(fix["meta_dropout_seed"]), p_detr=cfg.detr_dropout) if mode == "train" else None
    r = O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"],
                  inp["spans_target"], v_duration=inp["v_duration"], drop=drop)
    (r["retrieval_loss"] + r["localization_loss"]).backward()
    return fix, names, P, r


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_oracle_gradients_match_reference_fixture(golden_dir, mode):
    """float64 oracle autograd == float64 reference autograd (train mode: same stateless dropout masks)."""
    fix, names, P, r = _oracle_grads(golden_dir, mode, double=True)
    assert abs(float(r["retrieval_loss"]) - float(fix[f"{mode}.retrieval_loss"])) < 1e-6
    assert abs(float(r["localization_loss"]) - float(fix[f"{mode}.localization_loss"])) < 1e-6
    assert len(names) > 150
    for n in names:
        g = P[n].grad.reshape(-1).numpy()
        step = max(1, g.size // 512)
        ref = fix[f"{mode}.gsample.{n}"].astype(np.float64)
        scale = max(float(np.abs(ref).max()), 1e-4)
        assert np.abs(g[::step][:512] - ref).max() / scale < 1e-5, n
        assert abs(np.sqrt((g ** 2).sum()) - float(fix[f"{mode}.gnorm.{n}"])) <= 1e-5 * max(float(fix[f"{mode}.gnorm.{n}"]), 1e-4), n


def test_dropout_mask_statistics_and_determinism():
    from mgsv_amd import dropout as dr
    for p in (0.1, 0.3, 0.8):
        k = dr.keep_mask(77, dr.site_id("enc.0.attn"), p, 1 << 18)
        assert abs(k.mean() - (1 - p)) < 4e-3
        assert (k == dr.keep_mask(77, dr.site_id("enc.0.attn"), p, 1 << 18)).all()
        assert (k[1000:2000] == dr.keep_mask(77, dr.site_id("enc.0.attn"), p, 1000, offset=1000)).all()
        k2 = dr.keep_mask(77, dr.site_id("enc.1.attn"), p, 1 << 18)
        assert abs((k == k2).mean() - (p * p + (1 - p) ** 2)) < 6e-3          # independent sites
    big = dr.rng_mix(5, 9, np.array([(1 << 32) + 7, 7], dtype=np.uint64))    # the high index word matters
    assert big[0] != big[1]


def test_eval_metrics_oracle_matches_reference_fixture(golden_dir):
    """oracle restatement of Recall_metrics(dedup=True) / detr_iou against values the reference's own functions produced."""
    import torch
    fix = _load(golden_dir, "metrics")
    ids = [str(x) for x in fix["ids"]]
    ind = O.recall_ranks_dedup(fix["sim"], ids)
    assert ind.tolist() == fix["ind"].tolist()
    iou = O.top_span_iou(torch.from_numpy(fix["logits"]), torch.from_numpy(fix["spans"]), torch.from_numpy(fix["gt"]),
                         torch.from_numpy(fix["dur"]), 0, 240.0)
    assert float((iou - torch.from_numpy(fix["iou"])).abs().max()) <= 1e-6
    plain = O.recall_ranks_plain(fix["sim"])
    assert plain.tolist() == [(fix["sim"][i] > fix["sim"][i, i]).sum() for i in range(len(ids))]


VARIANTS = {      # the same table as tests/golden/make_golden.py
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


def variant_case(fix, name):
    cfg = cfg_native()
    for k, v in VARIANTS[name].items():
        setattr(cfg, k, v)
    B, Tv, Ta = int(fix["meta_B"]), int(fix["meta_T_v"]), int(fix["meta_T_a"])
    sd = synth.make_state_dict(cfg, seed=int(fix["meta_weight_seed"]))
    inp = synth.make_inputs(cfg, B, Tv, Ta, seed=int(fix["meta_data_seed"]))
    return cfg, sd, inp


def check_variant(fix, name, got, tol=1e-4):
    """got: dict with pred_spans, (pred_logits, proj_queries), retrieval_loss, localization_loss, loss_dict."""
    for k in ("pred_logits", "pred_spans", "proj_queries", "moment_feats", "video_feats", "music_feats"):
        if f"{name}.{k}" in fix.files and k in got:
            np.testing.assert_allclose(np.asarray(got[k]), fix[f"{name}.{k}"], atol=tol, rtol=0, err_msg=f"{name}.{k}")
    np.testing.assert_allclose(float(got["retrieval_loss"]), float(fix[f"{name}.retrieval_loss"]), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(float(got["localization_loss"]), float(fix[f"{name}.localization_loss"]), rtol=2e-4, atol=5e-4)
    keys = [k[len(name) + 6:] for k in fix.files if k.startswith(name + ".loss_")]
    assert set(keys) == set(got["loss_dict"]), (sorted(keys), sorted(got["loss_dict"]))
    for k in keys:
        np.testing.assert_allclose(float(got["loss_dict"][k]), float(fix[f"{name}.loss_{k}"]), rtol=2e-4, atol=2e-4, err_msg=f"{name}.{k}")


@pytest.mark.parametrize("name", list(VARIANTS))
def test_option_variants_match_reference_fixture(golden_dir, name):
    """Oracle against what the reference's forward returned for each option variant (tests/golden/variants.npz)."""
    fix = _load(golden_dir, "variants")
    cfg, sd, inp = variant_case(fix, name)
    with torch.no_grad():
        r = O.forward(O.to_torch_params(sd), cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"],
                      inp["spans_target"], v_duration=inp["v_duration"], music_ids=inp["music_ids"])
    got = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in r.items()
           if k in ("pred_logits", "pred_spans", "proj_queries", "moment_feats", "video_feats", "music_feats")}
    got.update(retrieval_loss=r["retrieval_loss"], localization_loss=r["localization_loss"], loss_dict=r["loss_dict"])
    check_variant(fix, name, got, tol=2e-5)


TRAIN_VARIANT_TAGS = ["cls", "mlp", "tower2", "xpool_query", "feature_fuse", "pre_norm", "pre_norm_Q2"]


def _variant_setup(fix, tag):
    cfg = cfg_native()
    for k, v in ast.literal_eval(str(fix[f"{tag}.overrides"])):
        setattr(cfg, k, v)
    sd = synth.make_state_dict(cfg, seed=0)
    inp = synth.make_inputs(cfg, int(fix["meta_B"]), int(fix["meta_T_v"]), int(fix["meta_T_a"]), seed=1)
    names = [k[len(tag) + 7:] for k in fix.files if k.startswith(tag + ".gnorm.")]
    return cfg, sd, inp, names


@pytest.mark.parametrize("tag", TRAIN_VARIANT_TAGS)
def test_oracle_train_variants_match_reference_fixture(golden_dir, tag):
    """Round-2 training variants (CLS token, mlp aggregator with train-mode BatchNorm, second X-Pool tower, xpool query, feature fuse):
    float64 oracle autograd in train mode == the reference's (tests/golden/train_variants_B3.npz, made by make_golden.py with the
    reference's dropout replaced by the build's masks), and the BatchNorm running buffers after the step."""
    fix = _load(golden_dir, "train_variants_B3")
    cfg, sd, inp, names = _variant_setup(fix, tag)
    P = {k: (v.double() if v.is_floating_point() else v) for k, v in O.to_torch_params(sd).items()}
    for n in names:
        P[n].requires_grad_(True)
    r = O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"], inp["spans_target"],
                  v_duration=inp["v_duration"], drop=O.Drop(int(fix["meta_dropout_seed"]), p_detr=cfg.detr_dropout))
    (r["retrieval_loss"] + r["localization_loss"]).backward()
    assert abs(float(r["retrieval_loss"]) - float(fix[f"{tag}.retrieval_loss"])) < 1e-6
    assert abs(float(r["localization_loss"]) - float(fix[f"{tag}.localization_loss"])) < 1e-6
    assert len(names) > 150
    sample = int(fix["meta_sample"])
    for n in names:
        g = (P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])).reshape(-1).numpy()
        step = max(1, g.size // sample)
        ref = fix[f"{tag}.gsample.{n}"].astype(np.float64)
        nr = float(fix[f"{tag}.gnorm.{n}"])
        assert np.abs(g[::step][:sample] - ref).max() <= 1e-5 * max(float(np.abs(ref).max()), 1e-4) + 1e-9, n
        assert abs(np.sqrt((g ** 2).sum()) - nr) <= 1e-5 * max(nr, 1e-4), n
    bufs = [k for k in fix.files if k.startswith(tag + ".buffer.")]
    assert (len(bufs) == 8) == (tag == "mlp")
    for k in bufs:
        assert np.abs(r["buffer_updates"][k[len(tag) + 8:]].numpy() - fix[k]).max() < 1e-9, k


def test_the_reference_tree_holds_no_bytecode():
    """Every script that imports the reference goes through oracle/ref_import.import_reference(), which sets sys.dont_write_bytecode first
    (SURVEY.md section 8(c) step 1): an ad-hoc `import` from /root/reference drops __pycache__ into a tree this repository must not write to.
    Runs where the reference exists (the build container); nothing to check on the GPU box."""
    import os
    from oracle import ref_import
    if not ref_import.reference_available():
        pytest.skip("no reference tree here")
    found = [os.path.join(d, n) for d, sub, files in os.walk(ref_import.REFERENCE_ROOT) for n in list(sub) + files
             if n == "__pycache__" or n.endswith(".pyc")]
    assert not found, f"bytecode under the reference tree (import it through oracle/ref_import.import_reference()): {found[:5]}"
