"""eval forward determinism: the captured graph of MadeEngine.forward replayed; every replay's outputs against the first eager run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.engine import MadeEngine
dev = torch.device("cuda", 0)
cfg = cfg_headline()
B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
sd = synth.make_state_dict(cfg, seed=0)
inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
eng = MadeEngine(cfg, sd, device=dev, dtype="bf16")
def step():
    return eng.forward(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
o = step(); torch.cuda.synchronize()
keys = [k for k in ("hs", "pred_spans", "pred_logits", "sims_single", "sims_dual") if k in o and isinstance(o[k], torch.Tensor)]
ref = {k: o[k].clone() for k in keys}
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    og = step()
bad = 0
N = int(os.environ.get("N", "200"))
for it in range(N):
    g.replay(); torch.cuda.synchronize()
    for k in keys:
        if not torch.equal(ref[k], og[k]):
            bad += 1
            if bad <= 3:
                d = (ref[k].float() - og[k].float()).abs()
                print(f"replay {it}: {k}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e}", flush=True)
            break
print(f"{bad} of {N} graph replays differ from the eager forward")
