"""One phase of the training step on its own: the step is recorded on the launch tape (program order), the decoder's forward and backward
layers are found by their memory-space attention launches, and those ranges alone are replayed in a loop -- the same kernels, arguments
and buffers as inside the step, without the rest of the step around them.  usage: python tools/phase_replay.py"""
import os, sys, time
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
os.environ["MADE_TAPE_INTERLEAVE"] = "0"
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
tb = tuple(g.inputs[k] for k in ("frame_feats", "segment_feats", "frame_masks", "segment_masks", "spans_target"))
for i in range(30):
    g.step(*tb, seed=i + 1, lrs=(1e-4, 1e-4, 1e-4))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    g.step(*tb, seed=i + 100, lrs=(1e-4, 1e-4, 1e-4))
torch.cuda.synchronize()
print(f"whole step (program order): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
ops = g.tape.ops()
main = max(set(o[2] for o in ops), key=lambda s: sum(1 for o in ops if o[2] == s))
wide = [i for i, o in enumerate(ops) if o[0] == 0 and o[3] == (1, 64, 4) and o[2] == main]
fns = []
for i in wide:
    if ops[i][1] not in fns: fns.append(ops[i][1])
by = {f: [i for i in wide if ops[i][1] == f] for f in fns}
print("memory-space attention launches on the main stream:", {hex(f): v for f, v in by.items()})


def time_range(first, count, reps=50):
    for _ in range(5): g.tape.replay_range(first, count)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    h0 = time.perf_counter()
    for _ in range(reps): g.tape.replay_range(first, count)
    h1 = time.perf_counter()
    e.record(); torch.cuda.synchronize()
    h2 = time.perf_counter()
    print(f"      [host: issuing took {(h1 - h0) / reps * 1e6:.1f} us per pass = {(h1 - h0) / reps / count * 1e6:.2f} us per op; the GPU needed {(h2 - h1) * 1e6 / reps:.1f} us per pass more]")
    return s.elapsed_time(e) * 1e3 / reps


def describe(first, count):
    nk = sum(1 for o in ops[first:first + count] if o[0] == 0)
    nm = sum(1 for o in ops[first:first + count] if o[0] == 0 and o[2] == main)
    return f"{count} ops, {nk} kernels ({nm} on the main stream)"


def time_range_queued(first, count, reps=20):
    """the same with the whole batch queued behind a long-running kernel, so the host is out of the picture"""
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    for _ in range(2): g.tape.replay_range(first, count)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(12): big.zero_()                      # ~ 2 ms of fills
    s.record()
    for _ in range(reps): g.tape.replay_range(first, count)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


for f, idx in by.items():
    if len(idx) < 6: continue
    idx = idx[-6:]
    a, b = idx[1], idx[5]
    us = time_range(a, b - a)
    print(f"function {hex(f)}: layers 1..4 of 6 = ops [{a}, {b}): {describe(a, b - a)}: {us:.1f} us = {us / 4:.1f} us per layer")
    us = time_range_queued(a, b - a)
    print(f"    queued behind 2 ms of fills: {us:.1f} us = {us / 4:.1f} us per layer")
    a, b = idx[2], idx[3]
    us = time_range(a, b - a)
    print(f"    one layer = ops [{a}, {b}): {describe(a, b - a)}: {us:.1f} us")
    # the same range, main-stream kernels only
    sel = [i for i in range(idx[1], idx[5]) if ops[i][0] == 0 and ops[i][2] == main]
    for _ in range(5):
        for i in sel: g.tape.replay_range(i, 1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        for i in sel: g.tape.replay_range(i, 1)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 50
    print(f"    main-stream kernels of layers 1..4 only ({len(sel)} launches): {us:.1f} us = {us / 4:.1f} us per layer, {us / len(sel):.2f} us per launch")
