"""forward determinism probe: the same training forward (same seed) repeated; counts the repetitions whose outputs differ from the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.trainer import MadeTrainer
dev = torch.device("cuda", 0)
cfg = cfg_headline()
B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
sd = synth.make_state_dict(cfg, seed=0)
inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
trn = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
batch = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
keys = ("hs", "retrieval_loss", "localization_loss", "memory")
ref = None
bad = {k: 0 for k in keys}
N = int(os.environ.get("N", "40"))
for it in range(N):
    o = trn.forward_train(*batch, seed=7)
    if os.environ.get("WITH_BWD", "0") == "1":
        trn.backward(None, None)
    torch.cuda.synchronize()
    cur = {k: o[k].clone() for k in keys}
    if ref is None:
        ref = cur
        continue
    for k in keys:
        if not torch.equal(ref[k], cur[k]):
            bad[k] += 1
            if bad[k] == 1:
                d = (ref[k].float() - cur[k].float()).abs()
                print(f"iteration {it}: {k} differs: {int((d > 0).sum())} elements, max {float(d.max()):.3e}", flush=True)
                if k == "hs":
                    idx = (d > 0).nonzero()
                    print("   layers:", sorted(set(idx[:, 0].tolist())), "samples:", sorted(set(idx[:, 1].tolist()))[:20])
print("differing repetitions of", N - 1, ":", bad)
