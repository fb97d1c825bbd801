"""CPU: the entry points accept exactly the reference's command-line flags with the same defaults and choices
(tests/golden/cli_flags.json was recorded from the reference's own argparse parsers), and derive the same fields."""
import json
import os

import numpy as np
import pytest

from mgsv_amd import driver


@pytest.mark.parametrize("script,for_test", [("train-MaDe.py", False), ("test-MaDe.py", True)])
def test_cli_flags_match_reference(golden_dir, script, for_test):
    ref = json.load(open(os.path.join(golden_dir, "cli_flags.json")))[script]
    p = driver.build_parser(for_test)
    ours = {}
    for act in p._actions:
        for o in act.option_strings:
            ours[o.lstrip("-")] = act
    extra = {n for n, *_ in driver.EXTRA}
    assert set(ref) <= set(ours), sorted(set(ref) - set(ours))
    assert set(ours) - set(ref) == extra
    for name, spec in ref.items():
        act = ours[name]
        if spec["flag"]:
            assert act.default is False and act.nargs == 0, name
            continue
        assert bool(act.required) == spec["required"], name
        if not spec["required"]:
            assert str(act.default) == str(spec["default"]), (name, act.default, spec["default"])
        assert (list(act.choices) if act.choices else None) == spec["choices"], name


def test_derived_fields_and_checks():
    a = driver.parse_option(["--name", "x", "--stride", "2.5", "--dim_input", "512", "--agg_module", "mlp", "--audio_short_cut", "0"])
    assert a.max_snippet_num == 96 and a.hidden_dim == 512 and a.detr_hidden_dim == 512
    assert a.video_transformer_depth == 0 and a.audio_transformer_depth == 0
    assert a.music_frozen_feature_path.endswith("ast_feature2p5") and a.frame_frozen_feature_path.endswith("vit_feature1")
    assert a.train_data == "kuai50k_uni"
    with pytest.raises(ValueError):
        driver.parse_option(["--name", "x", "--vmr_loss", "dual"])                      # XA fusion needs a single-tower loss
    with pytest.raises(ValueError):
        driver.parse_option(["--name", "x", "--num_moment_queries", "2"])               # needs decoder_SA
    # schedules (reference utils/scheduler.py)
    a = driver.parse_option(["--name", "x", "--audio_short_cut", "0"])
    assert driver.lr_factor(a, 0, 10, 100) == 0.0 and abs(driver.lr_factor(a, 5, 10, 100) - 0.5) < 1e-12
    assert abs(driver.lr_factor(a, 55, 10, 100) - 0.5) < 1e-12 and driver.lr_factor(a, 100, 10, 100) < 1e-12


def test_flags_that_select_unbuilt_code_fail_loudly():
    """A legal reference flag the HIP path has no code for must raise when the configuration is built, never be dropped silently."""
    import pytest
    from mgsv_amd.config import MadeConfig, cfg_native
    # span_loss_type=ce: the reference's own matcher cannot run it (music_detr/matcher.py:83-86 views the [B, Q, 2] spans as
    # [B * Q, 2, snippet_num] and indexes with float targets: RuntimeError / IndexError at the first iteration)
    args = cfg_native().to_args(local_rank=0)
    args.span_loss_type = "ce"
    with pytest.raises(NotImplementedError, match="span_loss_type=ce"):
        MadeConfig.from_args(args)
    # detr_pre_norm is carried through (round 5: built and parity-tested)
    args = cfg_native().to_args(local_rank=0)
    args.detr_pre_norm = True
    assert MadeConfig.from_args(args).detr_pre_norm is True and MadeConfig.from_args(args).to_args().detr_pre_norm is True
    # the reference itself refuses the learned position embedding (music_detr/position_encoding.py:98-105: ValueError "not supported learned")
    args = cfg_native().to_args(local_rank=0)
    args.position_embedding = "learned"
    with pytest.raises(ValueError, match="not supported learned"):
        MadeConfig.from_args(args)
    # encoder-variant flags are carried into the configuration (they used to be ignored)
    args = cfg_native().to_args(local_rank=0)
    args.with_cls_token, args.transformer_is_share, args.agg_module = 1, 1, "transf"
    c = MadeConfig.from_args(args)
    assert c.with_cls_token == 1 and c.transformer_is_share == 1 and c.agg_module == "transf"


def test_fresh_weights_follow_the_reference_initialisers():
    """args.seed seeds the initialisation; identity X-Pool projections, zero attention biases, LayerNorm 1 / 0, xavier DETR matrices
    (reference modules/transformer.py:148-154, music_detr/transformer.py:46-49)."""
    from mgsv_amd.config import cfg_native
    from mgsv_amd.model.init import reference_init
    c = cfg_native()
    a, b, a2 = reference_init(c, 1), reference_init(c, 2), reference_init(c, 1)
    assert all(np.array_equal(a[k], a2[k]) for k in a) and not np.array_equal(a["vit_proj.weight"], b["vit_proj.weight"])
    xa = "video_guided_to_music_pooling_cross_transformer."
    assert np.array_equal(a[xa + "cross_attn.k_proj.weight"], np.eye(c.D, dtype=np.float32)) and not a[xa + "linear_proj.bias"].any()
    assert np.all(a["detr_transformer.decoder.norm.weight"] == 1) and not a["detr_transformer.encoder.layers.0.self_attn.in_proj_bias"].any()
    w = a["detr_transformer.encoder.layers.0.linear1.weight"]
    bound = np.sqrt(6.0 / (w.shape[0] + w.shape[1]))
    assert np.abs(w).max() <= bound + 1e-6 and np.abs(w).max() > 0.95 * bound
    assert abs(float(a["logit_scale"]) - np.log(1 / c.temperature_init_value)) < 1e-6


def test_a_mistyped_checkpoint_path_fails_loudly(tmp_path):
    """--resume_path / --load_uni_model_path naming something that is not a file raises (the reference's torch.load does) instead of
    training or evaluating freshly initialised weights."""
    import logging
    a = driver.parse_option(["--name", "x", "--audio_short_cut", "0", "--resume_path", str(tmp_path / "no_such_checkpoint.bin")])
    with pytest.raises(FileNotFoundError):
        driver.build_model(a, "cpu", logging.getLogger("t"))
    a = driver.parse_option(["--name", "x", "--audio_short_cut", "0", "--load_uni_model_path", str(tmp_path)])      # a directory
    with pytest.raises(FileNotFoundError):
        driver.build_model(a, "cpu", logging.getLogger("t"))


def test_bench_and_entry_scripts_parse_their_flags_without_a_gpu():
    """`python bench.py --help` (and the two entry-point shims) must at least import and build their parsers here: a bench.py that
    does not start is only seen at round end otherwise."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "--workload" in out.stdout and "--gpus" in out.stdout, out.stderr[-2000:]
