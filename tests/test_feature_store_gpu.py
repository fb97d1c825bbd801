"""GPU: the packed feature store in the data path of the drivers.

The same tiny split is scored three ways -- from per-id `.pt` files laid out like the reference's feature directories
(reference dataloaders/dataloader_MGSV_EC_feature.py:57-67), from an f32 packed store (bit-exact bytes, so identical metrics) and
from a bf16 packed store -- and a PackedBatcher batch is copied asynchronously from pinned memory and pushed through the HIP
forward next to the same batch uploaded the plain way."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--mml_fusion", "concat", "--detr_enc_layers", "2", "--audio_short_cut", "0", "--max_v_frames", "20", "--max_m_duration", "100",
        "--num_workers", "0", "--batch_size_val", "16", "--save_model", "0", "--tb_writer", "0", "--compute_dtype", "f32"]


def _split(tmp_path, n, seed):
    """CSV + per-id feature / mask files under <frozen>/vit_feature1 and <frozen>/ast_feature2p5 (stride 2.5)."""
    rng = np.random.default_rng(seed)
    frozen = tmp_path / "frozen"
    cols = "video_id,music_id,video_start,video_end,music_start,music_end,music_total_duration,video_segment_duration,music_segment_duration," \
           "music_path,video_total_duration,video_width,video_height,video_total_frames,video_frame_rate,video_category"
    g = torch.Generator().manual_seed(seed)
    for kind, sub in (("vit", "vit_feature1"), ("ast", "ast_feature2p5")):
        os.makedirs(frozen / sub / f"{kind}_feature"); os.makedirs(frozen / sub / f"{kind}_mask")
    csv = str(tmp_path / "split.csv")
    musics = {}
    with open(csv, "w") as f:
        f.write(cols + "\n")
        for i in range(n):
            vid, mid = str(100000 + i), f"m{int(rng.integers(0, n // 2))}"
            dur = musics.setdefault(mid, float(rng.uniform(40, 100)))
            vd = float(rng.uniform(8, 19))
            ms = float(rng.uniform(0, dur - vd - 1))
            f.write(f"{vid},{mid},0.0,{vd:.3f},{ms:.3f},{ms + vd:.3f},{dur:.3f},{vd:.3f},{vd:.3f},/x.mp3,{vd:.2f},720,1280,300,30,Cat\n")
            nv = max(1, min(20, int(round(vd))))
            torch.save(torch.randn(20, 512, generator=g), frozen / "vit_feature1" / "vit_feature" / f"{vid}.pt")
            torch.save((torch.arange(20) < nv).float(), frozen / "vit_feature1" / "vit_mask" / f"{vid}.pt")
        for mid, dur in musics.items():
            na = max(1, min(40, int(round(dur / 2.5))))
            torch.save(torch.randn(40, 768, generator=g), frozen / "ast_feature2p5" / "ast_feature" / f"{mid}.pt")
            torch.save((torch.arange(40) < na).float(), frozen / "ast_feature2p5" / "ast_mask" / f"{mid}.pt")
    return csv, str(frozen)


def _pack(frozen, csv, dtype):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pack_features.py"), "--frozen_feature_path", frozen, "--csv", csv,
                        "--dtype", dtype], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.isfile(os.path.join(frozen, "vit_feature1", "vit.made")) and os.path.isfile(os.path.join(frozen, "ast_feature2p5", "ast.made"))


def test_packed_store_scores_the_split_like_the_per_file_layout(tmp_path):
    from mgsv_amd import driver
    csv, frozen = _split(tmp_path, 32, 5)
    common = ["--name", "fs", "--test_csv", csv, "--output_dir", str(tmp_path / "logs"), "--frozen_feature_path", frozen] + ARGS
    files = driver.main_test(common)
    _pack(frozen, csv, "f32")
    packed = driver.main_test(common)
    assert packed["ret"] == files["ret"] and packed["loc"] == files["loc"] and packed["com"] == files["com"]
    _pack(frozen, csv, "bf16")                                # overwrites the stores: inputs rounded to bf16 once
    half = driver.main_test(common)
    assert abs(half["loc"]["mIoU"] - files["loc"]["mIoU"]) < 2e-2 and abs(half["ret"]["MeanR"] - files["ret"]["MeanR"]) <= 2.0


def test_batcher_feeds_the_forward_from_pinned_memory(tmp_path):
    from mgsv_amd import feature_store as fs, synth
    from mgsv_amd.config import MadeConfig
    from mgsv_amd.engine import MadeEngine
    csv, frozen = _split(tmp_path, 16, 6)
    _pack(frozen, csv, "f32")
    pv = fs.PackedFeatures(os.path.join(frozen, "vit_feature1", "vit.made"))
    pa = fs.PackedFeatures(os.path.join(frozen, "ast_feature2p5", "ast.made"))
    import pandas as pd
    df = pd.read_csv(csv)
    vids, mids = [str(v) for v in df["video_id"][:8]], [str(m) for m in df["music_id"][:8]]
    bt = fs.PackedBatcher(pv, pa, batch_size=8, device="cuda:0")
    assert bt.host[0]["ff"].is_pinned() and bt.host[1]["sf"].is_pinned()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                             # the loader's stream: the copy overlaps whatever the compute stream runs
        ff, fm, sf, sm = bt.load(vids, mids)
    torch.cuda.current_stream().wait_stream(side)
    # the same batch assembled from the per-id files the way the reference's dataset does
    ref = [torch.stack([pv.get(v)[0] for v in vids]), torch.stack([pv.get(v)[1] for v in vids]),
           torch.stack([pa.get(m)[0] for m in mids]), torch.stack([pa.get(m)[1] for m in mids])]
    for got, want in zip((ff, fm, sf, sm), ref):
        assert got.is_cuda and torch.equal(got.cpu(), want)
    cfg = MadeConfig(max_v_frames=20, max_snippet_num=40, detr_enc_layers=2, mml_fusion="concat", audio_short_cut=0)
    eng = MadeEngine(cfg, synth.make_state_dict(cfg, seed=0), device="cuda:0", dtype="f32")
    tgt = torch.tensor([[[0.5, 0.2]]] * 8, device="cuda:0")
    keys = ("pred_logits", "pred_spans", "video_feats", "music_feats")
    a = {k: v.clone() for k, v in eng.forward(ff, sf, fm, sm, tgt).items() if k in keys}
    b = eng.forward(ref[0].cuda(), ref[2].cuda(), ref[1].cuda(), ref[3].cuda(), tgt)
    for k in keys:
        assert torch.isfinite(a[k]).all() and torch.equal(a[k], b[k]), k
