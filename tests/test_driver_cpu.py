"""CPU: the entry points accept exactly the reference's command-line flags with the same defaults and choices
(tests/golden/cli_flags.json was recorded from the reference's own argparse parsers), and derive the same fields."""
import json
import os

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
