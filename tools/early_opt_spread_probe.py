"""How far two runs of six training steps differ (f32 atomic weight-gradient sums: arrival order), per schedule -- the measurement behind the
fixed bounds of tests/test_trainer_gpu.py::test_train_step_with_the_early_optimizer_part_tracks_the_one_piece_step (GPU box)."""
import os, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_trainer_gpu import _setup
from mgsv_amd.trainer import MadeTrainer

cfg, sd, inp = _setup(8, 20, 40)
t = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
runs = {"1": [], "0": []}
for rep in range(8):
    for early in ("1", "0"):
        os.environ["MADE_EARLY_OPT"] = early
        trn = MadeTrainer(cfg, sd, dtype="bf16")
        losses = []
        for it in range(6):
            o = trn.train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=it, lrs=(3e-4, 3e-4, 3e-4))
            losses.append(float(o["retrieval_loss"]) + float(o["localization_loss"]))
        runs[early].append(losses)
a, b = np.asarray(runs["1"]), np.asarray(runs["0"])
allr = np.concatenate([a, b])
ref = np.median(allr, axis=0)
print("median loss per step:", np.round(ref, 4).tolist())
for name, x in (("early", a), ("one piece", b), ("all", allr)):
    dev = np.abs(x - ref) / np.maximum(np.abs(ref), 1.0)
    print(f"{name:10s} max relative deviation from the median, per step: {np.round(dev.max(0), 5).tolist()}")
print("max pairwise relative difference per step:", np.round(((allr[:, None] - allr[None]).__abs__() / np.maximum(np.abs(ref), 1.0)).max((0, 1)), 5).tolist())
