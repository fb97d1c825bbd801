"""GPU: the training path (forward in train mode + hand-written backward) against the oracle's autograd.

The oracle (oracle/made_oracle.py) is pinned to the reference's autograd in float64 (tests/golden/VALIDATION.json,
tests/golden/train_native_B3.npz); here every parameter gradient of the HIP path is compared with it on the same seeded
inputs and the same stateless dropout masks.  f32 path: relative L2 error per tensor <= 5e-3 (a ReLU input within rounding
noise of 0 may take the other side, which moves a whole row; measured errors are ~1e-5) and losses to 1e-4.  bf16 path:
cosine similarity of every gradient tensor >= 0.99 (0.97 with dropout on) and losses within 2e-2 relative."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(B, Tv, Ta, overrides=None):
    from mgsv_amd import synth
    from mgsv_amd.config import cfg_native
    cfg = cfg_native()
    for k, v in (overrides or {}).items():
        setattr(cfg, k, v)
    sd = synth.make_state_dict(cfg, seed=0)
    inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
    return cfg, sd, inp


def _oracle(cfg, sd, inp, seed, dropout, names, double=True):
    from oracle import made_oracle as O
    P = O.to_torch_params(sd)
    if double:
        P = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    for n in names:
        P[n].requires_grad_(True)
    drop = O.Drop(seed, p_detr=cfg.detr_dropout) if dropout else None
    r = O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"], inp["spans_target"],
                  v_duration=inp["v_duration"], drop=drop)
    (r["retrieval_loss"] + r["localization_loss"]).backward()
    return r, {n: (P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])) for n in names}


_RATIOS = []
BF16_NORM_RATIO_DROPOUT, BF16_NORM_RATIO_PLAIN = 0.09, 0.06    # 2 x the measured worst case over the ten configurations (0.0435 / 0.0299)


def _compare(res, r, grads, rel_tol, loss_tol, cos_min=None, norm_ratio_tol=0.5):
    assert abs(res["retrieval_loss"] - float(r["retrieval_loss"])) <= loss_tol * max(1.0, abs(float(r["retrieval_loss"])))
    assert abs(res["localization_loss"] - float(r["localization_loss"])) <= loss_tol * max(1.0, abs(float(r["localization_loss"])))
    gmax = max(float(g.abs().max()) for g in grads.values())
    worst = []
    for n, g in grads.items():
        ref = g.double().numpy().reshape(-1)
        got = res["grads"][n].astype(np.float64).reshape(-1)
        assert np.isfinite(got).all(), n
        nr = np.linalg.norm(ref)
        if nr < 1e-6 * gmax * np.sqrt(ref.size):            # theoretically zero gradients (e.g. the key bias): absolute check
            zero_tol = 1e-4 if cos_min is None else 5e-4      # bf16 mode: a sum of rounded terms that cancel exactly in f32
            assert np.abs(got).max() <= zero_tol * gmax, (n, np.abs(got).max())
            continue
        if cos_min is not None:
            cos = float(got @ ref / (np.linalg.norm(got) * nr + 1e-30))
            worst.append((1 - cos, n))
            assert cos >= cos_min, (n, cos)
            # the tensor's length too (a gradient scaled by 0.9 has cosine 1): bound = 2 x the measured worst case
            # (profiles/r06_bf16_grad_norm_ratio.txt)
            ratio = float(np.linalg.norm(got) / nr)
            _RATIOS.append(abs(ratio - 1))
            assert abs(ratio - 1) <= norm_ratio_tol, (n, ratio)
        else:
            rel = float(np.linalg.norm(got - ref) / nr)
            worst.append((rel, n))
            assert rel <= rel_tol, (n, rel)
    return sorted(worst)[-3:]


_NARROW = {"dim_input": 128, "SA_temporal_heads": 4, "detr_nheads": 4}      # heads of 32: the widths without the D = 256 / 512 specialisations
_MLP = {"agg_module": "mlp", "video_transformer_depth": 0, "audio_transformer_depth": 0, "max_v_frames": 20, "max_snippet_num": 40}


@pytest.mark.parametrize("dropout", [False, True])
@pytest.mark.parametrize("overrides", [{}, {"vmr_loss": "dual_single_sim_fuse", "moment_query_type": "music"}, {"mml_fusion": "CA"},
                                       {"with_act_after_proj": 1, "moment_query_type": "zero"},
                                       {"_shape": (1, 3, 5)}, {"_shape": (5, 33, 67), "mml_fusion": "CA"},
                                       {"num_moment_queries": 2}, {"num_moment_queries": 4, "mml_fusion": "CA", "moment_query_type": "music"},
                                       {"video_transformer_depth": 2, "audio_transformer_depth": 2, "with_act_after_proj": 1},
                                       {"mml_fusion": "CA", "detr_enc_layers": 0, "vmr_loss": "single"},
                                       {"predict_center": 1}, {"moment_loss": 1}, {"audio_short_cut": 1},
                                       {"audio_short_cut": 1, "num_moment_queries": 3, "moment_loss": 1},
                                       {"mml_localization": "regression"},
                                       {"mml_localization": "regression", "predict_center": 1, "mml_fusion": "CA"},
                                       {"transformer_is_share": 1, "_shape": (4, 18, 36)},
                                       {"vmr_fusion": "XA-video-music", "vmr_loss": "single"}, {"vmr_fusion": "XA-video", "vmr_loss": "single"},
                                       {"vmr_fusion": "XA-music-video", "vmr_loss": "dual_single_loss_fuse", "mml_fusion": "CA"},
                                       {"moment_query_type": "xpool"}, {"moment_query_type": "xpool", "vmr_loss": "dual", "num_moment_queries": 2},
                                       {"vmr_loss": "dual_single_feature_fuse"}, {"vmr_loss": "dual_single_feature_fuse", "moment_query_type": "xpool", "mml_fusion": "CA"},
                                       {"with_cls_token": 1}, {"with_cls_token": 1, "with_act_after_proj": 1, "video_transformer_depth": 2, "audio_transformer_depth": 2},
                                       {"with_cls_token": 1, "mml_fusion": "CA", "transformer_is_share": 1},
                                       dict(_MLP), dict(_MLP, with_act_after_proj=1, mml_fusion="CA"),
                                       dict(_NARROW),
                                       # round 5: pre-norm DETR layers (reference music_detr/transformer.py:170-189,246-271)
                                       {"detr_pre_norm": True}, {"detr_pre_norm": True, "num_moment_queries": 3, "mml_fusion": "CA"},
                                       {"detr_pre_norm": True, "moment_query_type": "xpool", "_shape": (5, 33, 67)},
                                       {"detr_pre_norm": True, "mml_localization": "regression"}, {"detr_pre_norm": True, "detr_enc_layers": 0, "mml_fusion": "CA"},
                                       # (dropout seeds 1 and 3: with 1234 one ReLU input of decoder layer 0's FFN -- unit 344, one of 6 rows -- lies within the
                                       #  forward's 1e-5 of zero and takes the other side: 6e-2 on that layer's tensors at this width and, through the gradient
                                       #  that flows on from there, a little on EVERY upstream tensor -- so masking the flipped unit's row and column out of
                                       #  the comparison (tried in round 6) does not repair the case; every other tensor and these two seeds sit at 3e-5.
                                       #  tools/train_variant_probe.py with DUMP= shows the one element)
                                       dict(_NARROW, mml_fusion="CA", num_moment_queries=2, vmr_fusion="XA-video-music", vmr_loss="single", _seed=1),
                                       dict(_NARROW, mml_fusion="CA", num_moment_queries=2, vmr_fusion="XA-video-music", vmr_loss="single", _seed=3)])
def test_f32_gradients_match_oracle_autograd(dropout, overrides):
    from mgsv_amd.trainer import MadeTrainer
    overrides = dict(overrides)
    shape = overrides.pop("_shape", (3, 20, 40))
    seed = overrides.pop("_seed", 1234)
    cfg, sd, inp = _setup(*shape, overrides)
    trn = MadeTrainer(cfg, sd, dtype="f32")
    trn.training_dropout = dropout
    res = trn.loss_and_grads(inp, seed=seed)
    r, grads = _oracle(cfg, sd, inp, seed, dropout, trn.param_names)
    print(_compare(res, r, grads, rel_tol=5e-3, loss_tol=1e-4))


def test_trainer_refuses_the_split_product_mode():
    """dtype "f32x3" (f32 storage, every product as three bf16 products on split operands) meets the forward's 1e-4 gate (tests/test_engine_gpu.py)
    but not the gradient gate of the f32 training path (5e-3 per tensor: measured up to 3.2e-2 under the p = 0.8 dropout, 8e-3 without --
    a near-zero ReLU input takes the other side): it is an inference mode, and the trainer says so instead of training at an unstated precision."""
    from mgsv_amd.trainer import MadeTrainer
    cfg, sd, _ = _setup(2, 20, 40)
    with pytest.raises(ValueError, match="f32x3"):
        MadeTrainer(cfg, sd, dtype="f32x3")


def test_f32_gradients_match_reference_fixture(golden_dir, dtype="f32"):
    """straight against the reference's own autograd (float64 fixture made by tests/golden/make_golden.py)."""
    from mgsv_amd.trainer import MadeTrainer
    fix = np.load(os.path.join(golden_dir, "train_native_B3.npz"))
    cfg, sd, inp = _setup(int(fix["meta_B"]), int(fix["meta_T_v"]), int(fix["meta_T_a"]))
    trn = MadeTrainer(cfg, sd, dtype=dtype)
    for mode, dropout in (("train", True), ("eval", False)):
        trn.training_dropout = dropout
        res = trn.loss_and_grads(inp, seed=int(fix["meta_dropout_seed"]))
        assert abs(res["retrieval_loss"] - float(fix[f"{mode}.retrieval_loss"])) <= 1e-4
        assert abs(res["localization_loss"] - float(fix[f"{mode}.localization_loss"])) <= 2e-4
        names = [k[len(mode) + 7:] for k in fix.files if k.startswith(mode + ".gnorm.")]
        gmax = max(float(fix[f"{mode}.gnorm.{n}"]) / np.sqrt(res["grads"][n].size) for n in names)
        for n in names:
            g = res["grads"][n].reshape(-1).astype(np.float64)
            step = max(1, g.size // 512)
            ref = fix[f"{mode}.gsample.{n}"].astype(np.float64)
            nr = float(fix[f"{mode}.gnorm.{n}"])
            if nr < 1e-6 * gmax * np.sqrt(g.size):
                continue
            assert abs(np.linalg.norm(g) - nr) <= 2e-3 * nr, (mode, n)
            s = g[::step][:512]
            assert np.linalg.norm(s - ref) <= 5e-3 * max(np.linalg.norm(ref), 1e-3 * nr), (mode, n)


@pytest.mark.parametrize("dropout", [False, True])
@pytest.mark.parametrize("overrides", [{}, {"with_cls_token": 1}, {"vmr_fusion": "XA-video-music", "vmr_loss": "single", "moment_query_type": "xpool"},
                                       dict(_NARROW), {"detr_pre_norm": True}])
def test_bf16_gradients_close_to_oracle(dropout, overrides):
    from mgsv_amd.trainer import MadeTrainer
    cfg, sd, inp = _setup(4, 20, 40, overrides)
    trn = MadeTrainer(cfg, sd, dtype="bf16")
    trn.training_dropout = dropout
    res = trn.loss_and_grads(inp, seed=77)
    r, grads = _oracle(cfg, sd, inp, 77, dropout, trn.param_names, double=False)
    _RATIOS.clear()
    print(_compare(res, r, grads, rel_tol=None, loss_tol=2e-2, cos_min=0.97 if dropout else 0.99,      # p = 0.8 dropout amplifies bf16 rounding 5x
                   norm_ratio_tol=BF16_NORM_RATIO_DROPOUT if dropout else BF16_NORM_RATIO_PLAIN), "worst |norm ratio - 1|", max(_RATIOS, default=0.0))


def test_optimizer_step_matches_torch_adam_with_group_clipping():
    """made_adam_step == three nn.utils.clip_grad_norm_ + torch.optim.Adam (reference train-MaDe.py:262-266,375-381)."""
    from mgsv_amd.trainer import MadeTrainer, XA
    cfg, sd, inp = _setup(3, 20, 40)
    trn = MadeTrainer(cfg, sd, dtype="bf16")
    names = trn.param_names

    def group(k):
        if k.startswith(("vit_proj.", "ast_proj.", "video_transformer.", "audio_transformer.")):
            return 0
        if k.startswith(XA + ".") or k == "logit_scale":
            return 1
        if k.startswith(("detr_transformer.", "span_embed.", "class_embed.", "contrastive_align_projection_")):
            return 2
        return 3
    params = {k: torch.nn.Parameter(trn.master[k].detach().clone()) for k in names}
    groups = [[params[k] for k in names if group(k) == g] for g in range(3)]
    lrs = (1e-3, 2e-3, 5e-4)
    opt = torch.optim.Adam([{"params": groups[g], "lr": lrs[g]} for g in range(3)])
    for it in range(3):
        trn.loss_and_grads(inp, seed=10 + it)
        for k in names:
            params[k].grad = trn.grad[k].detach().clone() * 0.5           # grad_scale = 0.5 (a 2-rank average)
        for g in range(3):
            torch.nn.utils.clip_grad_norm_(groups[g], 0.7)
        opt.step()
        trn.optimizer_step(*lrs, max_grad_norm=0.7, grad_scale=0.5)
        torch.cuda.synchronize()
        for k in names:
            ref, got = params[k].detach(), trn.master[k]
            if group(k) == 3:
                assert torch.equal(got, torch.from_numpy(np.asarray(sd[k])).to(got.device).view_as(got)), k    # untouched
                continue
            assert float((got - ref).abs().max()) <= 2e-6 + 1e-5 * float(ref.abs().max()), (it, k)
        # keep both sides on identical parameters so rounding does not accumulate through the next forward
        for k in names:
            trn.master[k].copy_(params[k].detach())
        trn.repack()
    # repack: compute-dtype copies and transposes follow the masters
    P = trn.P
    w = trn.master["detr_transformer.encoder.layers.0.linear1.weight"]
    assert torch.equal(P["detr_transformer.encoder.layers.0.ff1.w"], w.to(torch.bfloat16))
    assert torch.equal(P["detr_transformer.encoder.layers.0.ff1.wt"], w.t().contiguous().to(torch.bfloat16))
    assert torch.equal(P["class_embed.wt"][:, :2], trn.master["class_embed.weight"].t().to(torch.bfloat16))
    kv = torch.cat([trn.master[XA + ".cross_attn.k_proj.weight"], trn.master[XA + ".cross_attn.v_proj.weight"]], 0)
    assert torch.equal(P["xa.kv.w"], kv.to(torch.bfloat16))


@pytest.mark.parametrize("device_state", [False, True])
def test_optimizer_step_applied_in_two_parts_is_the_same_step(device_state):
    """optimizer_step(part="early") (matching + detection groups) followed by part="rest" (temporal group) == one optimizer_step, bit
    for bit -- masters, both Adam moments, the compute-dtype copies and the step count (device-side state included)."""
    import ctypes as C
    from mgsv_amd import _lib
    from mgsv_amd.trainer import MadeTrainer
    cfg, sd, inp = _setup(3, 20, 40)
    a, b = MadeTrainer(cfg, sd, dtype="bf16"), MadeTrainer(cfg, sd, dtype="bf16")
    lrs = (1e-3, 2e-3, 5e-4)
    st = []
    for _ in range(2):
        t_ = torch.zeros(C.sizeof(_lib.MadeAdamDeviceState) // 8, device="cuda", dtype=torch.int64)
        t_.view(torch.float32)[_lib.MadeAdamDeviceState.lr.offset // 4:][:3] = torch.tensor(lrs)
        st.append(t_ if device_state else None)
    for it in range(3):
        a.loss_and_grads(inp, seed=10 + it)
        b.flat_grad.copy_(a.flat_grad)
        a.optimizer_step(*lrs, max_grad_norm=0.7, grad_scale=0.5, device_state=st[0])
        b.optimizer_step(*lrs, max_grad_norm=0.7, grad_scale=0.5, device_state=st[1], part="early")
        b.optimizer_step(*lrs, max_grad_norm=0.7, grad_scale=0.5, device_state=st[1], part="rest")
        torch.cuda.synchronize()
        assert a.opt_step == b.opt_step == it + 1 and a.generation == b.generation
        assert torch.equal(a.flat_param, b.flat_param) and torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
        for k in a.P:
            if isinstance(a.P[k], torch.Tensor):
                assert torch.equal(a.P[k], b.P[k]), k
        if device_state:
            assert torch.equal(st[0], st[1]) and int(st[0][0]) == it + 1
    # every step moved every group
    assert float((a.flat_param[:a.group_ranges[0][1]] - MadeTrainer(cfg, sd, dtype="bf16").flat_param[:a.group_ranges[0][1]]).abs().max()) > 0


def test_train_step_with_the_early_optimizer_part_tracks_the_one_piece_step(monkeypatch):
    """MADE_EARLY_OPT=1: train_step applies the matching + detection groups' update on a third stream under the temporal encoders'
    backward; the default applies the step in one piece at the end: the same losses step after step (within what the f32 atomics of
    the weight-gradient products let two runs differ by)."""
    from mgsv_amd.trainer import MadeTrainer
    cfg, sd, inp = _setup(8, 20, 40)
    t = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
    curves = []
    for early in ("1", "0"):
        monkeypatch.setenv("MADE_EARLY_OPT", early)
        trn = MadeTrainer(cfg, sd, dtype="bf16")
        losses = []
        for it in range(6):
            o = trn.train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=it,
                               lrs=(3e-4, 3e-4, 3e-4))
            losses.append(float(o["retrieval_loss"]) + float(o["localization_loss"]))
        torch.cuda.synchronize()
        assert trn.opt_step == 6
        curves.append(losses)
    assert np.isfinite(curves).all()
    early, one = (np.asarray(c) for c in curves)
    # Two runs of the SAME schedule differ: the weight gradients are f32 atomic sums over workgroups (arrival order; csrc/gemm_tn*.hip), and six
    # steps at this learning rate on eight samples amplify a last-place difference.  FIXED bounds = 2 x the largest pairwise difference of
    # sixteen runs (eight per schedule; tools/early_opt_spread_probe.py, profiles/r06_early_opt_spread.txt: 0, 0, 8.8e-4, 1.6e-2, 1.1e-2,
    # 3.8e-2 per step): the first three steps -- where a wrong schedule would show first: step 1 already reads every updated group -- agree
    # tightly, the later ones within the chaos of the curve itself.
    bound = np.asarray([1e-4, 1e-4, 2e-3, 3.2e-2, 3.2e-2, 7.6e-2])
    diff = np.abs(early - one) / np.maximum(np.abs(one), 1.0)
    assert np.all(diff <= bound), (curves, diff.tolist())


def test_training_reduces_the_loss():
    """a few optimiser steps on one batch (dropout on): the total loss goes down."""
    from mgsv_amd.trainer import MadeTrainer
    cfg, sd, inp = _setup(8, 20, 40)
    trn = MadeTrainer(cfg, sd, dtype="bf16")
    t = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
    losses = []
    for it in range(12):
        o = trn.train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=it,
                           lrs=(3e-4, 3e-4, 3e-4))
        losses.append(float(o["retrieval_loss"]) + float(o["localization_loss"]))
    assert np.isfinite(losses).all()
    assert np.mean(losses[-3:]) < 0.9 * np.mean(losses[:3]), losses


def test_mlp_aggregator_running_buffers_and_eval_after_a_step():
    """agg_module = "mlp" in train mode: the BatchNorm running buffers move as torch.nn.BatchNorm1d moves them (oracle
    `buffer_updates`, pinned against the reference in VALIDATION.json), and the eval path then normalises with the moved buffers."""
    from mgsv_amd.trainer import MadeTrainer
    from oracle import made_oracle as O
    cfg, sd, inp = _setup(3, 20, 40, _MLP)
    trn = MadeTrainer(cfg, sd, dtype="f32")
    trn.loss_and_grads(inp, seed=5)
    P = O.to_torch_params(sd)
    r = O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"], inp["spans_target"],
                  v_duration=inp["v_duration"], drop=O.Drop(5, p_detr=cfg.detr_dropout))
    assert len(r["buffer_updates"]) == 12
    now = trn.state_dict_numpy()
    for k, v in r["buffer_updates"].items():
        ref = v.detach().numpy()
        assert np.abs(now[k] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k
    # eval forward of the trainer (kernel-facing affine refreshed by repack) against the oracle on the updated state
    trn.repack()
    P2 = O.to_torch_params({k: now[k] if k in now else v for k, v in sd.items()})
    e = O.forward(P2, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"], inp["spans_target"],
                  v_duration=inp["v_duration"])
    t = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
    out = trn.forward(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], v_duration=t["v_duration"])
    torch.cuda.synchronize()
    assert float((out["video_feats"].cpu() - e["video_feats"]).abs().max()) <= 2e-5
    assert float((out["music_feats"].cpu() - e["music_feats"]).abs().max()) <= 2e-5


def _batch(cfg, B, Tv, Ta, seed):
    from mgsv_amd import synth
    inp = synth.make_inputs(cfg, B, Tv, Ta, seed=seed)
    t = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
    return (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])


@pytest.mark.parametrize("mode", ["graph", "tape"])
@pytest.mark.parametrize("dtype,overrides", [("f32", None), ("bf16", None), ("bf16", {"detr_pre_norm": True}), ("bf16", dict(_NARROW))])
def test_captured_train_step_follows_the_eager_steps(dtype, overrides, mode):
    """the iteration as one hipGraph.  The replay reads the dropout seed, the Adam step count and the learning rates
    from device memory: three graph steps on three batches with three seeds and a moving schedule must land where three eager
    train_step calls land (same kernels, same order: the only freedom is the order of the f32 atomic gradient sums).
    Also on the two option paths of round 5 (pre-norm DETR layers, dim_input = 128: the decoder's chain of separate launches)."""
    from mgsv_amd.trainer import MadeTrainer
    cfg, sd, _ = _setup(4, 20, 40, overrides)
    eager, graph = MadeTrainer(cfg, sd, dtype=dtype), MadeTrainer(cfg, sd, dtype=dtype)
    batches = [_batch(cfg, 4, 20, 40, seed=10 + i) for i in range(3)]
    g = graph.capture_train_step(*batches[2], mode=mode)
    assert graph.opt_step == 0
    for k, v in eager.master.items():                        # capturing (and its warm-up step) left the state alone
        assert torch.equal(v, graph.master[k]), k
    tol = 2e-5 if dtype == "f32" else 2e-2
    for i, b in enumerate(batches):
        lrs = (1e-3 * (i + 1), 5e-4, 2e-3 / (i + 1))
        oe = eager.train_step(*b, seed=100 + i, lrs=lrs)
        le = (float(oe["retrieval_loss"]), float(oe["localization_loss"]))
        og = g.step(*b, seed=100 + i, lrs=lrs)
        lg = (float(og["retrieval_loss"]), float(og["localization_loss"]))
        assert abs(le[0] - lg[0]) <= tol * max(1.0, abs(le[0])) and abs(le[1] - lg[1]) <= tol * max(1.0, abs(le[1])), (i, le, lg)
    assert graph.opt_step == eager.opt_step == 3
    assert int(g.adam_state[0]) == 3
    # Adam moves every weight by at most ~lr per step whatever the gradient's size: a wrong step count, learning rate or mask would
    # show as a difference of that order; atomic-order noise in the gradients moves the update by a small fraction of it
    num = den = 0.0
    for k, v in eager.master.items():
        num += float((v - graph.master[k]).double().pow(2).sum()); den += float((v - torch.from_numpy(np.asarray(sd[k])).cuda()).double().pow(2).sum())
    assert num <= (1e-3 if dtype == "f32" else 0.3) * den, (num, den)


def test_refused_tape_recording_leaves_the_trainer_state_untouched(monkeypatch):
    """The tape records by running the step for real (optimizer included).  When the recording is refused -- a framework kernel inside the
    step: ForeignKernelError on leaving the recorder -- the caller falls back to mode='graph' or the eager step (bench.py does), and that
    fallback must start from the parameters, Adam moments and step count it had: nothing of the recorded step may remain."""
    from mgsv_amd import tape as T
    from mgsv_amd.trainer import MadeTrainer
    cfg, sd, _ = _setup(4, 20, 40)
    ref, trn = MadeTrainer(cfg, sd, dtype="bf16"), MadeTrainer(cfg, sd, dtype="bf16")
    b = _batch(cfg, 4, 20, 40, seed=12)
    trn.train_step(*b, seed=3, lrs=(1e-3, 1e-3, 1e-3)); ref.train_step(*b, seed=3, lrs=(1e-3, 1e-3, 1e-3))   # moments and step count are non-trivial
    torch.cuda.synchronize()
    keep = [x.clone() for x in (trn.flat_param, trn.exp_avg, trn.exp_avg_sq)]
    keep_buf = {k: v.clone() for k, v in trn.buffers.items()}
    orig = trn.optimizer_step

    def with_a_framework_kernel(*a, **kw):
        if T.recording():
            trn.flat_grad.mul_(1.0)                                              # an ATen kernel the replay would skip
        return orig(*a, **kw)
    monkeypatch.setattr(trn, "optimizer_step", with_a_framework_kernel)
    with pytest.raises(T.ForeignKernelError, match="aten::mul"):
        trn.capture_train_step(*b, mode="tape")
    torch.cuda.synchronize()
    assert trn.opt_step == 1 and trn.generation == ref.generation
    for a, c in zip((trn.flat_param, trn.exp_avg, trn.exp_avg_sq), keep):
        assert torch.equal(a, c)
    for k, v in keep_buf.items():
        assert torch.equal(trn.buffers[k], v), k
    monkeypatch.setattr(trn, "optimizer_step", orig)
    g = trn.capture_train_step(*b, mode="graph")                                 # the fallback, from the untouched state
    og = g.step(*b, seed=4, lrs=(1e-3, 1e-3, 1e-3))
    oe = ref.train_step(*b, seed=4, lrs=(1e-3, 1e-3, 1e-3))
    torch.cuda.synchronize()
    assert torch.equal(og["hs"], oe["hs"]) and trn.opt_step == ref.opt_step == 2
    moved = float((ref.flat_param - trn.flat_param).double().norm())
    assert moved <= 0.05 * float((ref.flat_param - keep[0]).double().norm()), moved


def test_captured_train_step_reads_seed_and_learning_rates_from_the_device():
    from mgsv_amd.trainer import MadeTrainer
    cfg, sd, _ = _setup(4, 20, 40)
    trn = MadeTrainer(cfg, sd, dtype="f32")
    b = _batch(cfg, 4, 20, 40, seed=3)
    g = trn.capture_train_step(*b)
    before = trn.flat_param.clone()
    l1 = float(g.step(*b, seed=1, lrs=(0.0, 0.0, 0.0))["localization_loss"])
    assert torch.equal(trn.flat_param, before)               # zero learning rates: the replay leaves the weights alone
    l1b = float(g.step(*b, seed=1, lrs=(0.0, 0.0, 0.0))["localization_loss"])
    l2 = float(g.step(*b, seed=2, lrs=(0.0, 0.0, 0.0))["localization_loss"])
    assert abs(l1 - l1b) <= 1e-5 * abs(l1) and abs(l1 - l2) > 1e-4 * abs(l1), (l1, l1b, l2)     # same seed, same masks; new seed, new masks
    g.step(*b, seed=3, lrs=(1e-3, 0.0, 0.0))
    r0 = trn.group_ranges[0]
    moved = (trn.flat_param != before)
    assert bool(moved[r0[0]:r0[1]].any()) and not bool(moved[r0[1]:].any())                         # only the temporal group has a rate


@pytest.mark.parametrize("tag", ["cls", "mlp", "tower2", "xpool_query", "feature_fuse", "pre_norm", "pre_norm_Q2"])
def test_f32_variant_gradients_match_reference_fixture(golden_dir, tag):
    """The round-2 training variants straight against the reference's own train()-mode autograd (float64 fixture
    tests/golden/train_variants_B3.npz): losses, every parameter gradient's norm and a strided sample, the BatchNorm buffers."""
    import ast
    from mgsv_amd.trainer import MadeTrainer
    fix = np.load(os.path.join(golden_dir, "train_variants_B3.npz"))
    ov = dict(ast.literal_eval(str(fix[f"{tag}.overrides"])))
    cfg, sd, inp = _setup(int(fix["meta_B"]), int(fix["meta_T_v"]), int(fix["meta_T_a"]), ov)
    trn = MadeTrainer(cfg, sd, dtype="f32")
    res = trn.loss_and_grads(inp, seed=int(fix["meta_dropout_seed"]))
    assert abs(res["retrieval_loss"] - float(fix[f"{tag}.retrieval_loss"])) <= 1e-4
    assert abs(res["localization_loss"] - float(fix[f"{tag}.localization_loss"])) <= 2e-4
    names = [k[len(tag) + 7:] for k in fix.files if k.startswith(tag + ".gnorm.")]
    sample = int(fix["meta_sample"])
    gmax = max(float(fix[f"{tag}.gnorm.{n}"]) / np.sqrt(res["grads"][n].size) for n in names)
    for n in names:
        g = res["grads"][n].reshape(-1).astype(np.float64)
        nr = float(fix[f"{tag}.gnorm.{n}"])
        if nr < 1e-6 * gmax * np.sqrt(g.size):
            continue
        assert abs(np.linalg.norm(g) - nr) <= 2e-3 * nr, (tag, n)
        step = max(1, g.size // sample)
        ref = fix[f"{tag}.gsample.{n}"].astype(np.float64)
        assert np.linalg.norm(g[::step][:sample] - ref) <= 5e-3 * max(np.linalg.norm(ref), 1e-3 * nr), (tag, n)
    now = trn.state_dict_numpy()
    for k in [k for k in fix.files if k.startswith(tag + ".buffer.")]:
        assert np.abs(now[k[len(tag) + 8:]] - fix[k]).max() <= 1e-5 * max(1.0, np.abs(fix[k]).max()), k


def test_headline_training_step_properties_at_full_size():
    """The leg bench.py times (BASELINE configs[2]: B = 64, T_v = 30, T_a = 512, D = 512, bf16, dropout on), through size-independent
    properties: every output and gradient finite; the same seed reproduces the forward bit for bit and the gradients up to the order of
    the f32 atomic sums; a new seed draws new masks; the captured iteration (TrainStepGraph) computes the eager step's forward bit for
    bit and lands on its parameters."""
    from mgsv_amd import synth
    from mgsv_amd.config import cfg_headline
    from mgsv_amd.trainer import MadeTrainer
    cfg = cfg_headline()
    B, Tv, Ta = 64, 30, 512
    sd = synth.make_state_dict(cfg, seed=0)
    b = _batch(cfg, B, Tv, Ta, seed=1)
    trn = MadeTrainer(cfg, sd, dtype="bf16")
    keys = ("retrieval_loss", "localization_loss", "hs", "pred_logits", "pred_spans", "sims_single", "sims_dual", "matcher_pred_idx", "criterion_losses")

    def run(seed):
        o = trn.forward_train(*b, seed=seed)
        trn.backward()
        torch.cuda.synchronize()
        return {k: o[k].clone() for k in keys}, trn.flat_grad.clone(), int(o["matcher_status"].cpu())
    f1, g1, st = run(7)
    assert st == 0
    for k in keys:
        assert bool(torch.isfinite(f1[k].float()).all()), k
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    assert float(f1["pred_spans"].min()) >= 0.0 and float(f1["pred_spans"].max()) <= 1.0
    f2, g2, _ = run(7)
    for k in keys:
        assert torch.equal(f1[k], f2[k]), f"{k}: same seed, different forward"
    # f32 atomics reorder the weight-gradient sums, and a few bf16 roundings downstream of atomically summed values flip with them
    assert float((g1 - g2).norm()) <= 1e-3 * float(g1.norm()), float((g1 - g2).norm()) / float(g1.norm())
    f3, g3, _ = run(8)
    assert not torch.equal(f1["hs"], f3["hs"]) and float((g1 - g3).norm()) > 1e-3 * float(g1.norm())
    # one optimizer step, eager against the captured iteration, from the same state
    eager = MadeTrainer(cfg, sd, dtype="bf16")
    oe = eager.train_step(*b, seed=7, lrs=(1e-4, 1e-4, 1e-4))
    le = {k: oe[k].clone() for k in keys}
    step = float((eager.flat_param - trn.flat_param).double().norm())           # trn never stepped: |one Adam step|
    for mode in ("graph", "tape"):                                               # hipGraph capture / the library's launch tape
        graph = MadeTrainer(cfg, sd, dtype="bf16")
        g = graph.capture_train_step(*b, mode=mode)
        og = g.step(*b, seed=7, lrs=(1e-4, 1e-4, 1e-4))
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(le[k], og[k]), f"{k}: captured ({mode}) forward differs from the eager one"
            assert torch.equal(le[k], f1[k]), k
        moved = float((eager.flat_param - graph.flat_param).double().norm())
        assert step > 0 and moved <= 0.05 * step, (mode, moved, step)
        if mode == "tape":
            nk, nw, no = g.tape.counts()
            assert nk > 300 and no >= 20, (nk, nw, no)            # kernels; event records / waits, fills and copies
            og2 = g.step(*b, seed=8, lrs=(1e-4, 1e-4, 1e-4))                     # a second replay: new masks, a moved state
            torch.cuda.synchronize()
            assert not torch.equal(og2["hs"], le["hs"]) and bool(torch.isfinite(og2["localization_loss"]).all())
        del g, graph


@pytest.mark.parametrize("dropout", [False, True])
def test_fused_decoder_chain_matches_the_separate_launches(dropout, monkeypatch):
    """bf16, one moment query: the fused training chain (LayerNorms in the consumers' prologues, key slices merged in the attention's
    launch, value bias in the per-head Linear, fused memory-space attention backward) against round 2's chain of separate launches
    (MADE_DEC_STAGE=0) on the same weights, batch and dropout masks: same losses and decoder states to bf16 rounding, gradients
    aligned tensor by tensor."""
    from mgsv_amd.trainer import MadeTrainer
    from mgsv_amd.config import cfg_headline
    from mgsv_amd import synth
    res = {}
    for cfg, shape in ((cfg_native_(), (6, 20, 40)), (cfg_headline(), (8, 30, 512))):
        sd = synth.make_state_dict(cfg, seed=0)
        b = _batch(cfg, *shape, seed=5)
        for mode in ("0", "1"):
            monkeypatch.setenv("MADE_DEC_STAGE", mode)
            trn = MadeTrainer(cfg, sd, dtype="bf16")
            trn.training_dropout = dropout
            assert trn._dec_stage_chain() == (mode == "1")
            o = trn.forward_train(*b, seed=13)
            trn.backward()
            torch.cuda.synchronize()
            res[mode] = (float(o["localization_loss"]), float(o["retrieval_loss"]), o["hs"].float().clone(), trn.grads_numpy())
        a, f = res["0"], res["1"]
        assert abs(a[0] - f[0]) <= 2e-2 * abs(a[0]) and abs(a[1] - f[1]) <= 1e-3 * abs(a[1]), (a[:2], f[:2])
        err = (a[2] - f[2]).abs() - 2.0 ** -6 * a[2].abs()
        assert float(err.max()) <= 6e-2, float(err.max())
        gmax = max(float(np.abs(g).max()) for g in a[3].values())
        for k, g0 in a[3].items():
            g1 = f[3][k]
            n0 = np.linalg.norm(g0)
            if n0 < 1e-5 * gmax * np.sqrt(g0.size):
                continue
            cos = float((g0.reshape(-1) @ g1.reshape(-1)) / (n0 * np.linalg.norm(g1) + 1e-30))
            assert cos >= 0.97, (k, cos)


def cfg_native_():
    from mgsv_amd.config import cfg_native
    return cfg_native()
