"""A/B of an environment knob on the eager training step inside ONE process (alternating, so box-to-box and warm-up differences cancel):
   python tools/ab_probe.py MADE_DEC_DW_SIDE 0 1"""
import sys, os, time
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.trainer import MadeTrainer
key, vals = sys.argv[1], sys.argv[2:]
cfg = cfg_headline()
B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
trn = MadeTrainer(cfg, synth.make_state_dict(cfg, seed=0), dtype="bf16")
inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
t = {k: torch.from_numpy(v).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
batch = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
it = [0]
def step():
    it[0] += 1
    trn.train_step(*batch, seed=it[0])
def timeit(n=50):
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(100): step()
res = {v: [] for v in vals}
for r in range(4):
    for v in vals:
        os.environ[key] = v
        res[v].append(timeit())
for v in vals:
    print(f"{key}={v}: " + " ".join(f"{x:.3f}" for x in res[v]) + f"  | mean {np.mean(res[v]):.3f} ms/step")
