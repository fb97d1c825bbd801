"""Debug helper: compare train-mode intermediates of the HIP path with the oracle (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_native
from mgsv_amd.trainer import MadeTrainer
from oracle import made_oracle as O

cfg = cfg_native()
sd = synth.make_state_dict(cfg, seed=0); inp = synth.make_inputs(cfg, 3, 20, 40, seed=1)
trn = MadeTrainer(cfg, sd, dtype="f32")
t = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
o = trn.forward_train(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=1234)
torch.cuda.synchronize()
P = O.to_torch_params(sd)
with torch.no_grad():
    r = O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"], inp["spans_target"],
                  v_duration=inp["v_duration"], drop=O.Drop(1234, p_detr=cfg.detr_dropout))
fm = torch.cat([t["frame_masks"], t["segment_masks"]], 1).cpu()
def d(a, b, m=None):
    a = a.float().cpu(); b = b.float()
    e = (a - b).abs()
    if m is not None: e = e * m
    return float(e.max())
print("video", d(o["video_feats"], r["video_feats"]), "music", d(o["music_feats"], r["music_feats"]))
print("seg", d(o["segment_feats"], r["segment_feats"]))
print("sims_single", d(o["sims_single"], r["sims_single"]), "sims_dual", d(o["sims_dual"], r["sims_dual"]))
print("memory", d(o["memory"], r["memory"], fm[:, :, None]))
for l in range(cfg.detr_dec_layers):
    print("hs", l, d(o["hs"][l], r["hs"][l]))
print("ret", float(o["retrieval_loss"]), float(r["retrieval_loss"]), "loc", float(o["localization_loss"]), float(r["localization_loss"]))
