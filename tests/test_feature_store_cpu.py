"""CPU: the packed feature store reproduces what the reference's per-file layout holds."""
import os

import numpy as np
import pytest
import torch

from mgsv_amd import feature_store as fs


def _make(root, kind, ids, T, D, seed):
    os.makedirs(os.path.join(root, f"{kind}_feature")); os.makedirs(os.path.join(root, f"{kind}_mask"))
    g = torch.Generator().manual_seed(seed)
    ref = {}
    for i in ids:
        n = int(torch.randint(1, T + 1, (1,), generator=g))
        mask = (torch.arange(T) < n).float()
        feats = torch.randn(T, D, generator=g)                       # the files hold unmasked features; the dataset zero-fills
        torch.save(feats, os.path.join(root, f"{kind}_feature", f"{i}.pt")); torch.save(mask, os.path.join(root, f"{kind}_mask", f"{i}.pt"))
        ref[i] = (feats.masked_fill(mask.unsqueeze(-1) == 0, 0), mask)
    return ref


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_pack_and_read_back(tmp_path, dtype):
    ids = ["108587485547", "7+5x4mjjeiubm49fg", "3", "113722188340", "a" * 30]
    ref = _make(str(tmp_path / "vit"), "vit", ids, 12, 64, 1)
    path = fs.pack(str(tmp_path / "vit"), "vit", ids + ids[:2], str(tmp_path / f"vit_{dtype}.made"), dtype=dtype)
    pf = fs.PackedFeatures(path)
    assert len(pf) == len(ids) and pf.T == 12 and pf.D == 64 and list(pf.ids) == sorted(ids)
    for i in ids:
        f, m = pf.get(i)
        assert torch.equal(m, ref[i][1])
        if dtype == "f32":
            assert torch.equal(f, ref[i][0])
        else:
            assert torch.equal(f, ref[i][0].to(torch.bfloat16).float())
    with pytest.raises(KeyError):
        pf.get("missing")
    out_f = torch.empty(3, 12, 64, dtype=pf.torch_dtype); out_m = torch.empty(3, 12)
    pf.gather([ids[2], ids[0], ids[2]], out_f, out_m)
    assert torch.equal(out_f[1].float(), pf.get(ids[0])[0]) and torch.equal(out_m[0], ref[ids[2]][1]) and torch.equal(out_f[0], out_f[2])


def test_batcher_matches_per_file_dataset_items(tmp_path):
    vids, mids = ["v1", "v2", "v3"], ["m1", "m2"]
    rv = _make(str(tmp_path / "vit"), "vit", vids, 10, 32, 2)
    ra = _make(str(tmp_path / "ast"), "ast", mids, 16, 48, 3)
    pv = fs.PackedFeatures(fs.pack(str(tmp_path / "vit"), "vit", vids, str(tmp_path / "v.made"), "f32"))
    pa = fs.PackedFeatures(fs.pack(str(tmp_path / "ast"), "ast", mids, str(tmp_path / "a.made"), "f32"))
    bt = fs.PackedBatcher(pv, pa, batch_size=4, device=None, pin=False)
    ff, fm, sf, sm = bt.load(["v3", "v1", "v3"], ["m2", "m2", "m1"])
    assert ff.shape == (3, 10, 32) and sf.shape == (3, 16, 48)
    assert torch.equal(ff[0], rv["v3"][0]) and torch.equal(fm[1], rv["v1"][1]) and torch.equal(sf[2], ra["m1"][0]) and torch.equal(sm[0], ra["m2"][1])
    ff2, *_ = bt.load(["v1"], ["m1"])                                  # the second buffer: the first batch is still intact
    assert torch.equal(ff[0], rv["v3"][0]) and torch.equal(ff2[0], rv["v1"][0])
