"""Experiment: how fast is the training step when replayed as a hipGraph (fixed dropout seed / Adam step -> NOT a valid training
loop, only a measurement of how much of the eager step time is launch overhead)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.trainer import MadeTrainer

cfg = cfg_headline()
B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
trn = MadeTrainer(cfg, synth.make_state_dict(cfg, seed=0), dtype="bf16")
inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
t = {k: torch.from_numpy(v).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
step = lambda: trn.train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=5)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("eager  ms/step", (time.perf_counter() - t0) / 20 * 1e3)
t0 = time.perf_counter()
for _ in range(20): step()
print("eager  CPU issue ms/step (no sync)", (time.perf_counter() - t0) / 20 * 1e3)
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): step()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print("graph  ms/step", (time.perf_counter() - t0) / 20 * 1e3)
