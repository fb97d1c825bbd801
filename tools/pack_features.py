#!/usr/bin/env python3
"""Pack the per-id feature files of a split into memory-mappable stores (mgsv_amd/feature_store.py).

    python tools/pack_features.py --frozen_feature_path features/Kuai_feature --stride 2.5 --csv dataset/MGSV-EC/train_data.csv \
        dataset/MGSV-EC/val_data.csv dataset/MGSV-EC/test_data.csv [--dtype bf16]

writes <frozen_feature_path>/vit_feature1/vit.made and <frozen_feature_path>/ast_feature2p5/ast.made; the drivers pick them up."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pandas as pd  # noqa: E402

from mgsv_amd import feature_store as fs  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--frozen_feature_path", default="features/Kuai_feature")
p.add_argument("--stride", type=float, default=2.5)
p.add_argument("--csv", nargs="+", required=True)
p.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
a = p.parse_args()
music_dir = {2.5: "ast_feature2p5", 5.0: "ast_feature5", 7.5: "ast_feature7p5", 10.0: "ast_feature10"}[a.stride]
df = pd.concat([pd.read_csv(c) for c in a.csv])
for kind, root, col in (("vit", os.path.join(a.frozen_feature_path, "vit_feature1"), "video_id"),
                        ("ast", os.path.join(a.frozen_feature_path, music_dir), "music_id")):
    out = fs.pack(root, kind, df[col].astype(str).unique(), os.path.join(root, f"{kind}.made"), a.dtype)
    print("wrote", out, os.path.getsize(out) >> 20, "MiB")
