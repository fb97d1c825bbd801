"""Upper bound of a half-batch pipeline of the training step: ONE taped step of B = 64 against TWO (FOUR) independent taped
steps of B = 32 (16) replayed concurrently on streams of their own -- every phase of one half may overlap every phase of the other, which is
more than a pipeline inside one step could arrange (the optimizer tail and the in-batch retrieval branch would stay whole-batch there).
If two halves in flight are not clearly faster than the whole batch, no placement of half-batch decoder chains can be."""
import os, statistics, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.trainer import MadeTrainer
cfg = cfg_headline()
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(cfg, seed=0)
def batch(B, seed):
    inp = synth.make_inputs(cfg, B, cfg.max_v_frames, cfg.max_snippet_num, seed=seed)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
    return (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
def build(B, n):
    lanes = []
    for i in range(n):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            trn = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
            b = batch(B, 1 + i)
            g = trn.capture_train_step(*b, mode="tape")
        torch.cuda.synchronize()
        lanes.append((st, trn, b, g))
    return lanes
def run(lanes, steps=30):
    def once(k):
        for st, trn, b, g in lanes:
            with torch.cuda.stream(st):
                g.step(*b, seed=100 + k, lrs=(1e-4, 1e-4, 1e-4))
    for k in range(8): once(k)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for k in range(steps): once(k)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps * 1e3)
    return statistics.median(ts), min(ts)
res = {}
for B, n in ((64, 1), (32, 2), (16, 4), (32, 1), (64, 2)):
    lanes = build(B, n)
    res[(B, n)] = run(lanes)
    print(f"{n} x B={B:3d} in flight: {res[(B, n)][0]:.3f} ms per round (min {res[(B, n)][1]:.3f}) = {n * B / res[(B, n)][0] * 1e3:.0f} pairs/s", flush=True)
    del lanes
    torch.cuda.empty_cache()
