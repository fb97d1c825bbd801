"""The recorded training step's operations around a launch, in replay order: kind (0 kernel, 1 stream-waits-stream, 2 memset, 3 copy, 4 event record,
5 event wait, 6 host callback), stream, grid.  usage: python tools/tape_ops_probe.py   (prints the operations behind the last launch of grid (1, 64, 4):
the first decoder layer's memory-space attention backward -- where the step's timeline shows the main stream idle for 60 us)"""
import os, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
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
g = trn.capture_train_step(*batch, max_grad_norm=1.0, mode="tape")
ops = g.tape.ops()
streams = {}
for k, fn, st, grid in ops:
    streams.setdefault(st, len(streams))
last = max(i for i, (k, fn, st, grid) in enumerate(ops) if k == 0 and grid == (1, 64, 4))
names = {0: "kernel", 1: "wait-stream", 2: "memset", 3: "copy", 4: "event-record", 5: "event-wait", 6: "callback"}
print(f"{len(ops)} operations, {sum(1 for o in ops if o[0] == 0)} kernels; streams {len(streams)}")
for i in range(max(0, last - 3), min(len(ops), last + 24)):
    k, fn, st, grid = ops[i]
    print(f"  {i:4d}  {names[k]:13s} s{streams[st]}  {grid if k == 0 else ''}")
import collections
print("non-kernel operations of the whole step:", dict(collections.Counter(names[o[0]] for o in ops if o[0] != 0)))

# every cross-stream operation of the step with the launches around it (grid of the last launch in front of it and the first behind it, per stream)
print("cross-stream operations (index, kind, stream; last launch issued on that stream in front of it):")
last = {}
for i, (k, fn, st, grid) in enumerate(ops):
    if k == 0:
        last[st] = (i, grid)
    elif k in (1, 4, 5):
        print(f"  {i:4d} {names[k]:13s} s{streams[st]}  after {last.get(st)}")
