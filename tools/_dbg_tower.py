import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import synth
from mgsv_amd.config import cfg_native
from mgsv_amd.trainer import MadeTrainer
from oracle import made_oracle as O
cfg = cfg_native(); cfg.moment_query_type = "xpool"
sd = synth.make_state_dict(cfg, seed=0); inp = synth.make_inputs(cfg, 3, 20, 40, seed=1)
trn = MadeTrainer(cfg, sd, dtype="f32"); trn.training_dropout = False
dev = trn.device
t = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
o = trn.forward_train(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=1234, v_duration=t.get("v_duration"))
torch.cuda.synchronize()
P = O.to_torch_params(sd)
r = O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"], inp["spans_target"], v_duration=inp["v_duration"])
tw = trn._train_buffers(3, 20, 40)
pooled = r["music_feats_pooled"]
print("pooled", float((tw["xpooled"].cpu().view(3, 3, 256) - pooled).abs().max()))
print("query", float((tw["xpool_q"].cpu() - pooled.mean(1)).abs().max()))
print("tgt0", float((tw["d.0.tgt"].cpu().float().view(3, 256) - pooled.mean(1)).abs().max()))
print("hs", float((o["hs"].cpu().float() - r["hs"]).abs().max()), "loc", float(o["localization_loss"]), float(r["localization_loss"]))
print("mem", float(((o["memory"].cpu().float() - r["memory"]) ).abs().max()))
