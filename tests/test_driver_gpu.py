"""GPU: the two entry points end to end on a tiny synthetic split (same CSV columns as dataset/MGSV-EC/*.csv, features
replaced by seeded random tensors): training lowers the loss, the evaluation prints the reference's metric set."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COMMON = ["--mml_fusion", "concat", "--detr_enc_layers", "2", "--audio_short_cut", "0", "--max_v_frames", "20", "--max_m_duration", "100",
          "--synthetic_features", "1", "--num_workers", "0", "--batch_size_val", "16", "--save_model", "0", "--tb_writer", "0"]


def _csv(path, n, seed):
    rng = np.random.default_rng(seed)
    cols = "video_id,music_id,video_start,video_end,music_start,music_end,music_total_duration,video_segment_duration,music_segment_duration," \
           "music_path,video_total_duration,video_width,video_height,video_total_frames,video_frame_rate,video_category"
    with open(path, "w") as f:
        f.write(cols + "\n")
        for i in range(n):
            dur = rng.uniform(40, 100)
            vd = rng.uniform(8, 19)
            ms = rng.uniform(0, dur - vd - 1)
            f.write(f"{100000 + i},m{int(rng.integers(0, max(2, n // 2)))},0.0,{vd:.3f},{ms:.3f},{ms + vd:.3f},{dur:.3f},{vd:.3f},{vd:.3f},/x.mp3,{vd:.2f},"
                    f"720,1280,300,30,Cat\n")


def test_train_and_test_entry_points(tmp_path):
    from mgsv_amd import driver
    tr, va = str(tmp_path / "train.csv"), str(tmp_path / "val.csv")
    _csv(tr, 48, 1); _csv(va, 32, 2)
    res = driver.main_train(["--name", "t", "--do_train", "--do_eval", "--epochs", "3", "--batch_size_train", "16", "--train_csv", tr, "--val_csv", va,
                             "--output_dir", str(tmp_path / "logs"), "--matching_lr", "3e-4", "--detection_lr", "3e-4", "--warmup_rate", "0.1"] + COMMON)
    assert sorted(res) == [1, 2, 3]
    assert np.isfinite([r["train_loss"] for r in res.values()]).all()
    assert res[3]["train_loss"] < res[1]["train_loss"]
    out = driver.main_test(["--name", "t", "--test_csv", va, "--output_dir", str(tmp_path / "logs")] + COMMON)
    assert set(out["ret"]) >= {"R1", "R5", "R10", "R100", "MedianR", "MeanR", "MRR"} and 0 <= out["ret"]["R1"] <= 100
    assert set(out["loc"]) == {"mIoU", "IoU@0.3", "IoU@0.5", "IoU@0.7"} and len(out["com"]) == 12
    # two batches in flight (the default) and one at a time score the split identically
    one = driver.main_test(["--name", "t", "--test_csv", va, "--output_dir", str(tmp_path / "logs"), "--eval_in_flight", "1"] + COMMON)
    assert one["ret"] == out["ret"] and one["loc"] == out["loc"] and one["com"] == out["com"]
    # the unfused path of the reference's loop body (torch Adam + clip_grad_norm_) also runs
    res2 = driver.main_train(["--name", "t2", "--do_train", "--epochs", "1", "--batch_size_train", "16", "--train_csv", tr, "--val_csv", va,
                              "--output_dir", str(tmp_path / "logs"), "--fused_step", "0"] + COMMON)
    assert np.isfinite(res2[1]["train_loss"])
    # the CLI's default --audio_short_cut 1 trains too (the reference's scripts pass 0)
    sc = [("1" if i > 0 and COMMON[i - 1] == "--audio_short_cut" else a) for i, a in enumerate(COMMON)]
    res3 = driver.main_train(["--name", "t3", "--do_train", "--do_eval", "--epochs", "1", "--batch_size_train", "16", "--train_csv", tr, "--val_csv", va,
                              "--output_dir", str(tmp_path / "logs")] + sc)
    assert np.isfinite(res3[1]["train_loss"])
    # the regression localisation variant: train + evaluate (IoU from the single regressed span)
    res4 = driver.main_train(["--name", "t4", "--do_train", "--do_eval", "--epochs", "1", "--batch_size_train", "16", "--train_csv", tr, "--val_csv", va,
                              "--output_dir", str(tmp_path / "logs"), "--mml_localization", "regression"] + COMMON)
    assert np.isfinite(res4[1]["train_loss"]) and 0 <= res4[1]["mIoU"] <= 1


def test_fused_training_updates_the_eval_weights_and_checkpoints_round_trip(tmp_path):
    """(1) With the fused optimizer step (the default) the masters change in place through raw pointers; the evaluation after every epoch
    must see the new weights (round 1 scored epoch-1 weights forever).  (2) The four best-* checkpoints carry the reference's file names
    and keys, load strictly, and reproduce the validation metrics of the epoch that wrote them (reference utils/util_train.py:21-60,
    train-MaDe.py:709-727, test-MaDe.py:486-514)."""
    import glob

    import torch

    from mgsv_amd import driver
    tr, va = str(tmp_path / "train.csv"), str(tmp_path / "val.csv")
    _csv(tr, 48, 3); _csv(va, 32, 4)
    common = [a if a != "0" or COMMON[i - 1] != "--save_model" else "1" for i, a in enumerate(COMMON)]
    res = driver.main_train(["--name", "ck", "--do_train", "--do_eval", "--epochs", "3", "--batch_size_train", "16", "--train_csv", tr, "--val_csv", va,
                             "--output_dir", str(tmp_path / "logs"), "--matching_lr", "1e-3", "--detection_lr", "1e-3", "--warmup_rate", "0.0",
                             "--scheduler", "constant", "--seed", "7"] + common)
    losses = [res[e]["val_loss"] for e in (1, 2, 3)]
    assert len({round(v, 6) for v in losses}) == 3, f"validation loss did not move between epochs: {losses}"
    logdir = glob.glob(str(tmp_path / "logs" / "*" / "*+ck"))
    assert len(logdir) == 1
    files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(logdir[0], "pytorch_model.bin.*")))
    assert files == ["pytorch_model.bin.best_iou", "pytorch_model.bin.best_r1", "pytorch_model.bin.best_r1iou05", "pytorch_model.bin.best_r1iou07"] or \
        set(files) >= {"pytorch_model.bin.best_iou", "pytorch_model.bin.best_r1", "pytorch_model.bin.best_r1iou07"}
    ck = torch.load(os.path.join(logdir[0], "pytorch_model.bin.best_r1"), map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "loss", "model_state_dict", "optimizer_state_dict"}
    ep = ck["epoch"]
    out = driver.main_test(["--name", "ck", "--test_csv", va, "--output_dir", str(tmp_path / "logs2"), "--load_uni_model_path", logdir[0], "--test_best", "1",
                            "--seed", "99"] + COMMON)
    got = out["pytorch_model.bin.best_r1"]
    assert got["epoch"] == ep
    np.testing.assert_allclose(got["loss"], res[ep]["val_loss"], rtol=1e-5)
    assert got["ret"]["R1"] == res[ep]["R1"] and abs(got["loc"]["mIoU"] - res[ep]["mIoU"]) < 1e-6
    # a single checkpoint file
    one = driver.main_test(["--name", "ck", "--test_csv", va, "--output_dir", str(tmp_path / "logs2"), "--load_uni_model_path",
                            os.path.join(logdir[0], "pytorch_model.bin.best_r1")] + COMMON)
    assert one["ret"] == got["ret"] and one["loc"] == got["loc"]
    # strict loading: a checkpoint with a missing key is refused
    bad = dict(ck); bad["model_state_dict"] = {k: v for k, v in ck["model_state_dict"].items() if k != "class_embed.weight"}
    torch.save(bad, str(tmp_path / "pytorch_model.bin.bad"))
    with pytest.raises(RuntimeError):
        driver.main_test(["--name", "ck", "--test_csv", va, "--output_dir", str(tmp_path / "logs2"), "--load_uni_model_path", str(tmp_path / "pytorch_model.bin.bad")] + COMMON)
