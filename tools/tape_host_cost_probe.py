"""Host cost of one replay of the training step's launch tape: time of tape.replay() itself (no synchronisation; the device idle when it starts)
beside the device time of the step, for B = 64 / 32 / 8.  If the host loop takes as long as the device, the step is issue-bound somewhere."""
import os, statistics, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.trainer import MadeTrainer
cfg = cfg_headline(); dev = torch.device("cuda", 0); sd = synth.make_state_dict(cfg, seed=0)
for B in (64, 32, 8):
    inp = synth.make_inputs(cfg, B, cfg.max_v_frames, cfg.max_snippet_num, seed=1)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
    b = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
    trn = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
    g = trn.capture_train_step(*b, mode="tape")
    nk, nw, no = g.tape.counts()
    for k in range(5): g.step(*b, seed=k, lrs=(1e-4,) * 3)
    torch.cuda.synchronize()
    host, total = [], []
    for k in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); g.tape.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    # back-to-back (the bench's regime)
    t0 = time.perf_counter()
    for k in range(30): g.tape.replay()
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) / 30 * 1e3
    print(f"B={B:2d}: tape = {nk} kernels + {nw} waits + {no} other ops | host loop {statistics.median(host):.3f} ms (min {min(host):.3f}) | one replay from an idle device "
          f"{statistics.median(total):.3f} ms | back-to-back {b2b:.3f} ms per step", flush=True)
    del g, trn
    torch.cuda.empty_cache()
