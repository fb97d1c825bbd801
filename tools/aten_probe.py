"""Which ATen ops still run inside one eval forward / one training step (torch.profiler, CPU-side op names with call stacks)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.engine import MadeEngine
from mgsv_amd.trainer import MadeTrainer
from torch.profiler import profile, ProfilerActivity
cfg = cfg_headline(); B, Tv, Ta = 64, 30, 512
sd = synth.make_state_dict(cfg, seed=0); inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
dev = torch.device("cuda")
t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
which = sys.argv[1] if len(sys.argv) > 1 else "eval"
if which == "eval":
    eng = MadeEngine(cfg, sd, device=dev, dtype="bf16")
    step = lambda: eng.forward(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
else:
    trn = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
    it = [0]
    def step():
        it[0] += 1
        return trn.train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=it[0], lrs=(1e-4,) * 3)
step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
from collections import Counter
c = Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name not in ("aten::empty", "aten::view", "aten::slice", "aten::select", "aten::as_strided", "aten::empty_strided",
                                                         "aten::reshape", "aten::permute", "aten::transpose", "aten::t", "aten::expand", "aten::unsqueeze",
                                                         "aten::_unsafe_view", "aten::empty_like", "aten::detach", "aten::alias", "aten::squeeze", "aten::contiguous",
                                                         "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::resolve_conj", "aten::resolve_neg"):
        st = [s for s in (e.stack or []) if "mgsv_amd" in s]
        c[(e.name, st[0].split("/")[-1] if st else "?")] += 1
for (n, s), k in sorted(c.items(), key=lambda kv: -kv[1]):
    print(f"{k:4d} {n:28s} {s}")
