"""Eager step vs graph replays of the bench configuration, key by key (a difference = a race between the engine's two streams)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.engine import MadeEngine
cfg = cfg_headline(); B, Tv, Ta = 64, 30, 512
sd = synth.make_state_dict(cfg, seed=0); inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
dev = torch.device("cuda")
t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
eng = MadeEngine(cfg, sd, device=dev, dtype="bf16")
if len(sys.argv) > 1 and sys.argv[1] == "unfused": eng.force_unfused_decoder = True
step = lambda: eng.forward(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
keys = ["video_feats", "music_feats", "memory", "sims_single", "sims_dual", "retrieval_loss", "hs", "pred_logits", "pred_spans", "criterion_losses", "localization_loss"]
o = step(); torch.cuda.synchronize()
eager = {k: o[k].clone() for k in keys}
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    og = step()
for rep in range(6):
    g.replay(); torch.cuda.synchronize()
    bad = [(k, float((og[k].float() - eager[k].float()).abs().max())) for k in keys if not torch.equal(og[k], eager[k])]
    print("replay", rep, "differs:" if bad else "identical", bad, flush=True)
o2 = step(); torch.cuda.synchronize()
print("second eager identical:", all(torch.equal(o2[k], eager[k]) for k in keys))
