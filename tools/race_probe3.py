"""first captured step (launch tape / hipGraph) against the eager step from the same state, repeated with fresh trainers: which forward
outputs, decoder buffers and (atomics-free) gradient stacks of the decoder's backward chain differ, and in which rows.
usage: python tools/race_probe3.py [tape] [graph]   (N = repetitions per mode)"""
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
t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
b = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
keys = ("memory", "hs", "retrieval_loss", "localization_loss")
eager = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
oe = eager.train_step(*b, seed=7, lrs=(1e-4, 1e-4, 1e-4))
torch.cuda.synchronize()
ref = {k: oe[k].clone() for k in keys}
tw = eager._train_buffers(B, Tv, Ta)
names = [k for k in tw if isinstance(tw[k], torch.Tensor) and k.startswith("d.")]
refb = {k: tw[k].clone() for k in names}
gnames = [k for k in tw["dstack"] if k.startswith("g_") or k == "dt1q"]          # the backward chain's per-layer gradients (no atomics)
refg = {k: tw["dstack"][k].clone() for k in gnames}
for mode in sys.argv[1:] or ("graph", "tape"):
    for rep in range(int(os.environ.get("N", "5"))):
        tr_ = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
        g = tr_.capture_train_step(*b, mode=mode)
        og = g.step(*b, seed=7, lrs=(1e-4, 1e-4, 1e-4))
        torch.cuda.synchronize()
        msg = []
        for k in keys:
            if not torch.equal(ref[k], og[k]):
                d = (ref[k].float() - og[k].float()).abs()
                msg.append(f"{k}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}")
        tw2 = tr_._train_buffers(B, Tv, Ta)
        first = None
        for k in names:
            if k in tw2 and tw2[k].shape == refb[k].shape and not torch.equal(tw2[k], refb[k]):
                d = (tw2[k].float() - refb[k].float()).abs()
                msg.append(f"   buffer {k}: {int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.3e}; rows {sorted(set((d > 0).nonzero()[:, 0].tolist()))[:12]}")
        for k in gnames:
            a_, b_ = tw2["dstack"][k], refg[k]
            if not torch.equal(a_, b_):
                d = (a_.float() - b_.float()).abs()
                lay = sorted(set((d > 0).nonzero()[:, 0].tolist()))
                msg.append(f"   gradient stack {k}: {int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.3e}; layers {lay}")
        print(mode, rep, "identical" if not msg else "\n  ".join(["DIFFERENT"] + msg[:14]), flush=True)
        del g, tr_
