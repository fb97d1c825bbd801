"""Which forms of made_linear one training step of the headline shape uses (ops.LINEAR_LOG): kernel chosen, M, N, K, epilogue flags, count."""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from mgsv_amd import ops, synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.trainer import MadeTrainer
cfg = cfg_headline()
B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
dev = torch.device("cuda", 0)
trn = MadeTrainer(cfg, synth.make_state_dict(cfg, seed=0), device=dev, dtype=os.environ.get("DTYPE", "bf16"))
inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
batch = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
trn.train_step(*batch, seed=1, lrs=(1e-4, 1e-4, 1e-4), max_grad_norm=1.0)
ops.LINEAR_LOG = []
trn.train_step(*batch, seed=2, lrs=(1e-4, 1e-4, 1e-4), max_grad_norm=1.0)
torch.cuda.synchronize()
c = collections.Counter(ops.LINEAR_LOG)
for k, n in sorted(c.items(), key=lambda kv: -(kv[0][1] * kv[0][2] * kv[0][3] * kv[0][4] * kv[1])):
    if k[1] * k[4] >= int(os.environ.get('MIN_ROWS', 4096)):
        print(f"{n:3d} x {k[0]:34s} M={k[1]:6d} N={k[2]:5d} K={k[3]:5d} z={k[4]:3d}  {k[5]}")
