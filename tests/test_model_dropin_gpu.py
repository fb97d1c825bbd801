"""GPU: the drop-in `Uni_model` (same ctor / forward signature / state_dict layout as the reference's
model/model_Uni.py) against the reference's golden vectors."""
import ast
import logging
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mgsv_amd import synth  # noqa: E402
from mgsv_amd.config import cfg_native, cfg_plumbing  # noqa: E402
from mgsv_amd.model import Uni_model  # noqa: E402


def test_dropin_forward_and_state_dict(golden_dir):
    fix = np.load(os.path.join(golden_dir, "forward_native_Q3_B4.npz"))
    cfg = cfg_native()
    for k, v in ast.literal_eval(str(fix["meta_cfg_overrides"])):
        setattr(cfg, k, v)
    args = cfg.to_args(local_rank=0)
    model = Uni_model(args, device=torch.device("cuda:0"), logger=logging.getLogger("t"))
    sd_np = synth.make_state_dict(cfg, seed=0)
    assert set(model.state_dict().keys()) == set(sd_np.keys())            # the reference's key layout
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    sd["vit_model.visual.proj"] = torch.zeros(3)                          # frozen-encoder keys of a real checkpoint are tolerated
    model.load_state_dict(sd)
    model.eval()
    inp = synth.make_inputs(cfg, int(fix["meta_B"]), int(fix["meta_T_v"]), int(fix["meta_T_a"]), seed=1)
    t = {k: torch.from_numpy(v) for k, v in inp.items() if isinstance(v, np.ndarray)}      # CPU tensors, like a DataLoader batch
    with torch.no_grad():
        om, lm, fm, mm, im = model(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"],
                                   v_duration=t["v_duration"], video_ids=inp["video_ids"], music_ids=inp["music_ids"], is_train=False)
    torch.cuda.synchronize()
    for k in ("pred_logits", "pred_spans", "proj_queries"):
        np.testing.assert_allclose(om[k].cpu().numpy(), fix[k], atol=1e-4, rtol=0, err_msg=k)
    assert len(om["aux_outputs"]) == cfg.detr_dec_layers - 1
    for i, aux in enumerate(om["aux_outputs"]):
        np.testing.assert_allclose(aux["pred_spans"].cpu().numpy(), fix[f"aux{i}_pred_spans"], atol=1e-4, rtol=0)
    for k in ("video_feats", "music_feats"):
        np.testing.assert_allclose(fm[k].cpu().numpy(), fix[k], atol=1e-4, rtol=0)
    np.testing.assert_allclose(float(lm["retrieval_loss"]), float(fix["retrieval_loss"]), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(float(lm["localization_loss"]), float(fix["localization_loss"]), rtol=2e-4, atol=5e-4)
    assert set(lm["localization_loss_dict"]) == {k[5:] for k in fix.files if k.startswith("loss_")}
    wd = model.criterion.weight_dict
    total = sum(float(lm["localization_loss_dict"][k]) * wd[k] for k in lm["localization_loss_dict"] if k in wd)
    np.testing.assert_allclose(total, float(fix["localization_loss"]), rtol=2e-4, atol=5e-4)
    assert model.criterion.foreground_label == 0
    # parameter groups as the reference's optimizer builds them; decoder_query_embed is in none
    n_groups = sum(p.numel() for g in (model.get_temporal_parameter(), model.get_matching_parameter(), model.get_detection_parameter()) for p in g)
    n_all = sum(p.numel() for p in model.parameters())
    assert n_all - n_groups == model.decoder_query_embed.weight.numel()
    # the X-Pool block called directly on "the whole split", as test-MaDe.py:392-395 does
    xa = model.video_guided_to_music_pooling_cross_transformer
    xa.cpu()
    pooled = xa(fm["video_feats"].cpu(), fm["segment_feats"].cpu(), mm["segment_masks"])
    xa.to(torch.device("cuda:0"))
    np.testing.assert_allclose(pooled.numpy()[:, :, ::5], fix["music_feats_pooled_sub"], atol=1e-4, rtol=0)
    # training mode (three moment queries): the losses come back on the autograd tape and backward fills every gradient
    model.train()
    om2, lm2, *_ = model(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], is_train=True)
    loss = lm2["retrieval_loss"] + lm2["localization_loss"]
    assert loss.requires_grad and bool(torch.isfinite(loss))
    loss.backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for n, p in model.named_parameters() if n != "decoder_query_embed.weight" or True)
    assert float(model.decoder_query_embed.weight.grad.abs().sum()) > 0 and om2["pred_spans"].shape == (int(fix["meta_B"]), 3, 2)


def test_training_loop_body_of_the_reference_driver_runs_unchanged():
    """reference train-MaDe.py:337-381 verbatim in spirit: forward(is_train=True), weighted loss, loss.backward(), three
    clip_grad_norm_ calls, Adam step, zero_grad -- on the drop-in module.  Gradients equal the trainer's (which are pinned to the
    reference's autograd in tests/test_trainer_gpu.py), the loss goes down, and eval() afterwards sees the updated weights."""
    import numpy as np
    import torch
    from mgsv_amd import synth
    from mgsv_amd.config import cfg_native
    from mgsv_amd.model import Uni_model
    cfg = cfg_native()
    args = cfg.to_args(local_rank=0)
    args.compute_dtype = "bf16"
    model = Uni_model(args, device=torch.device("cuda:0"))
    keys_before = list(model.state_dict().keys())
    inp = synth.make_inputs(cfg, 8, 20, 40, seed=3)
    t = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
    opt = torch.optim.Adam([{"params": model.get_temporal_parameter(), "lr": 3e-4},
                            {"params": model.get_matching_parameter(), "lr": 3e-4},
                            {"params": model.get_detection_parameter(), "lr": 3e-4}])
    model.train()
    losses = []
    for it in range(10):
        om, lm, fm, mm, im = model(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"],
                                   v_duration=t["v_duration"], video_ids=inp["video_ids"], music_ids=inp["music_ids"], is_train=True)
        loss = lm["retrieval_loss"] * 1.0 + lm["localization_loss"] * 1.0
        loss.backward()
        if it == 0:
            trn = model._trainer
            for n, p in model.named_parameters():
                assert p.grad is not None and torch.equal(p.grad, trn.grad[n]), n
            assert float(model.vit_proj.weight.grad.abs().max()) > 0
        torch.nn.utils.clip_grad_norm_(model.get_temporal_parameter(), 1.0)
        torch.nn.utils.clip_grad_norm_(model.get_matching_parameter(), 1.0)
        torch.nn.utils.clip_grad_norm_(model.get_detection_parameter(), 1.0)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
        assert "loss_label" in lm["localization_loss_dict"] and om["pred_spans"].shape == (8, 1, 2)
    assert np.isfinite(losses).all() and np.mean(losses[-3:]) < 0.9 * np.mean(losses[:3]), losses
    assert list(model.state_dict().keys()) == keys_before
    model.eval()
    with torch.no_grad():
        om, lm, *_ = model(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
    assert float(lm["retrieval_loss"] + lm["localization_loss"]) < losses[0]


@pytest.mark.parametrize("name", ["xa_music_video_single", "regression_center_CA", "agg_mlp", "cls_token", "shared_temporal_block", "pre_norm_Q3_CA"])
def test_dropin_option_variants(golden_dir, name):
    """The drop-in module on two option variants (second X-Pool tower; regression head + predict_center + CA fusion): key layout,
    optimizer groups and outputs as the reference's (tests/golden/variants.npz)."""
    from test_oracle_golden import variant_case, check_variant
    fix = np.load(os.path.join(golden_dir, "variants.npz"))
    cfg, sd_np, inp = variant_case(fix, name)
    model = Uni_model(cfg.to_args(local_rank=0), device=torch.device("cuda:0"), logger=logging.getLogger("t"))
    assert set(model.state_dict().keys()) == set(sd_np.keys())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    model.eval()
    t = {k: torch.from_numpy(v) for k, v in inp.items() if isinstance(v, np.ndarray)}
    with torch.no_grad():
        om, lm, fm, mm, im = model(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"],
                                   v_duration=t["v_duration"], video_ids=inp["video_ids"], music_ids=inp["music_ids"], is_train=False)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in om.items() if torch.is_tensor(v)}
    got.update(retrieval_loss=lm["retrieval_loss"], localization_loss=lm["localization_loss"], loss_dict=lm["localization_loss_dict"])
    check_variant(fix, name, got)
    grouped = {id(p) for g in (model.get_temporal_parameter(), model.get_matching_parameter(), model.get_detection_parameter()) for p in g}
    left = {n.split(".")[0] for n, p in model.named_parameters() if id(p) not in grouped}
    if "regression" in name:       # reference model_Uni.py:112-113: only reg_mlp (and the CA block) train; the DETR stack is in no group
        assert set(om) == {"pred_spans"} and left == {"decoder_query_embed", "detr_transformer"}
    else:
        assert left == {"decoder_query_embed"}


@pytest.mark.parametrize("overrides", [{"mml_localization": "regression"}, {"mml_localization": "regression", "predict_center": 1, "mml_fusion": "CA"},
                                       {"audio_short_cut": 1, "predict_center": 1, "moment_loss": 1},
                                       {"detr_pre_norm": True}, {"dim_input": 128, "SA_temporal_heads": 4, "detr_nheads": 4}])
def test_training_loop_body_on_option_variants(overrides):
    """The reference's loop body on the drop-in module for variants the scripts do not use: the regression localisation head (its
    own optimizer group: reg_mlp + the CA block), predict_center, the audio short-cut (the CLI's default), moment_loss."""
    import numpy as np
    import torch
    from mgsv_amd import synth
    from mgsv_amd.config import cfg_native
    from mgsv_amd.model import Uni_model
    cfg = cfg_native()
    for k, v in overrides.items():
        setattr(cfg, k, v)
    args = cfg.to_args(local_rank=0)
    args.compute_dtype = "bf16"
    model = Uni_model(args, device=torch.device("cuda:0"))
    inp = synth.make_inputs(cfg, 8, 20, 40, seed=3)
    t = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
    opt = torch.optim.Adam([{"params": model.get_temporal_parameter(), "lr": 3e-4},
                            {"params": model.get_matching_parameter(), "lr": 3e-4},
                            {"params": model.get_detection_parameter(), "lr": 3e-4}])
    model.train()
    losses = []
    for it in range(8):
        om, lm, fm, mm, im = model(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"],
                                   v_duration=t["v_duration"], video_ids=inp["video_ids"], music_ids=inp["music_ids"], is_train=True)
        loss = lm["retrieval_loss"] + lm["localization_loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.get_temporal_parameter(), 1.0)
        torch.nn.utils.clip_grad_norm_(model.get_matching_parameter(), 1.0)
        torch.nn.utils.clip_grad_norm_(model.get_detection_parameter(), 1.0)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
        assert om["pred_spans"].shape == (8, 1, 2) and "loss_span" in lm["localization_loss_dict"]
    assert np.isfinite(losses).all() and np.mean(losses[-2:]) < np.mean(losses[:2]), losses


def test_mlp_aggregator_running_statistics_reach_eval_and_checkpoints():
    """agg_module = "mlp" through the drop-in module: the BatchNorm running statistics a train step moves live in the module's own
    registered buffers (shared storage with the trainer), so eval() (state_dict -> MadeEngine) and a saved state_dict see them."""
    cfg = cfg_native()
    cfg.agg_module, cfg.video_transformer_depth, cfg.audio_transformer_depth = "mlp", 0, 0          # (the reference asserts depth 0 with the EmbeddingNet)
    cfg.max_v_frames, cfg.max_snippet_num = 20, 40                                                   # BatchNorm over the token positions: T is fixed
    args = cfg.to_args(local_rank=0)
    model = Uni_model(args, device=torch.device("cuda:0"))
    inp = synth.make_inputs(cfg, 4, 20, 40, seed=3)
    t = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
    batch = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
    bn_keys = [k for k in model.state_dict() if k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    assert len(bn_keys) == 12
    before = {k: model.state_dict()[k].detach().cpu().clone() for k in bn_keys}
    model.eval()
    with torch.no_grad():
        e0 = model(*batch, v_duration=t["v_duration"])[2]["video_feats"].clone()
    model.train()
    om, lm, *_ = model(*batch, v_duration=t["v_duration"], is_train=True)
    (lm["retrieval_loss"] + lm["localization_loss"]).backward()        # no optimizer step: only the statistics move
    torch.cuda.synchronize()
    after = {k: model.state_dict()[k].detach().cpu().clone() for k in bn_keys}
    trn = model._trainer
    for k in bn_keys:
        assert torch.equal(after[k], trn.buffers[k].cpu()), k           # the module's buffers ARE the trainer's
        assert not torch.equal(after[k], before[k]), k
    assert all(int(after[k]) == 1 for k in bn_keys if k.endswith("num_batches_tracked"))
    model.eval()
    with torch.no_grad():
        e1 = model(*batch, v_duration=t["v_duration"])[2]["video_feats"].clone()
    assert float((e1 - e0).abs().max()) > 1e-4                          # eval normalises with the moved statistics
    # a fresh module loaded from the state_dict reproduces that eval output
    m2 = Uni_model(args, device=torch.device("cuda:0"))
    m2.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    m2.eval()
    with torch.no_grad():
        e2 = m2(*batch, v_duration=t["v_duration"])[2]["video_feats"]
    assert float((e2 - e1).abs().max()) <= 1e-5
