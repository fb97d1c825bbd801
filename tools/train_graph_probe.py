"""Training step: eager launches against the captured iteration (MadeTrainer.capture_train_step), alternating, 3 rounds of 40 steps."""
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
batch = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
it = [0]
def eager():
    it[0] += 1
    trn.train_step(*batch, seed=it[0])
for _ in range(100): eager()
torch.cuda.synchronize()
g = trn.capture_train_step(*batch)
def graph():
    it[0] += 1
    g.step(*batch, seed=it[0])
def timeit(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0, c0 = time.perf_counter(), time.thread_time()
    for _ in range(n): f()
    wall_issue, busy = time.perf_counter() - t0, time.thread_time() - c0       # busy: CPU time of THIS thread (a blocked launch does not count)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, wall_issue / n * 1e3, busy / n * 1e3
for r in range(3):
    e, ec, eb = timeit(eager)
    gr, gc, gb = timeit(graph)
    print(f"round {r}: eager {e:.3f} ms/step (issue loop {ec:.3f} ms wall, {eb:.3f} ms of CPU), graph {gr:.3f} ms/step (issue loop {gc:.3f} ms wall, {gb:.3f} ms of CPU)")
