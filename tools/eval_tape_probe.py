"""The eval forward (BASELINE configs[1]) replayed from a hipGraph against the library's launch tape: ms per batch of 64, one batch in flight."""
import os, sys, time
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import synth, tape as _tape
from mgsv_amd.config import cfg_headline
from mgsv_amd.engine import MadeEngine
dev = torch.device("cuda", 0)
cfg = cfg_headline()
B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
sd = synth.make_state_dict(cfg, seed=0)
inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
eng = MadeEngine(cfg, sd, device=dev, dtype=sys.argv[1] if len(sys.argv) > 1 else "bf16")
t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
step = lambda: eng.forward(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
out = step(); torch.cuda.synchronize()


def timed(fn, n=200):
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.75:
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    og = step()
print(f"hipGraph replay: {timed(g.replay):.4f} ms per batch")
try:
    with _tape.LaunchTape.record() as tp:
        ot = step()
    torch.cuda.synchronize()
    k, w, o = tp.counts()
    print(f"launch tape: {k} kernels, {o} other operations; {timed(tp.replay):.4f} ms per batch")
    tp.interleave(2)
    print(f"launch tape, streams fed round-robin: {timed(tp.replay):.4f} ms per batch")
    print("losses equal:", float(og['localization_loss']), float(ot['localization_loss']), float(og['retrieval_loss']), float(ot['retrieval_loss']))
except Exception as ex:
    print("tape refused:", type(ex).__name__, str(ex)[:300])
print(f"eager: {timed(step, 50):.4f} ms per batch")
