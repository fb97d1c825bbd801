"""Per-launch breakdown of one forward step (HIP events around every made_linear / made_attention launch)
plus total step time; other kernels are the remainder.   python tools/step_profile.py [--dtype bf16]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops, synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.engine import MadeEngine

ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--batch", type=int, default=64); a = ap.parse_args()
cfg = cfg_headline(); dev = torch.device("cuda")
eng = MadeEngine(cfg, synth.make_state_dict(cfg, 0), device=dev, dtype=a.dtype)
inp = synth.make_inputs(cfg, a.batch, seed=1)
t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
step = lambda: eng.forward(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
for _ in range(3): step()
torch.cuda.synchronize()
with ops.KernelTimer() as kt:
    step()
torch.cuda.synchronize()
tot = {}
for kind, s, e, fl, nb, desc in kt.records:
    ms = s.elapsed_time(e)
    key = (kind, desc[0] + ' skip' if isinstance(desc, tuple) else desc)
    d = tot.setdefault(key, [0, 0.0, fl])
    d[0] += 1; d[1] += ms
for (kind, desc), (n, ms, fl) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms*1e3:9.1f} us  x{n:2d}  {fl*n/ms/1e9 if ms>0 else 0:8.1f} TF  {kind:18s} {desc}")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): step()
e.record(); torch.cuda.synchronize()
print("eager step ms:", s.elapsed_time(e) / 10, " timed kernels ms:", sum(v[1] for v in tot.values()))
