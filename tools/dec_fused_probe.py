"""Training step with the fused per-sample decoder kernels against the unfused launches (MADE_DEC_FUSED=0), headline config."""
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
def step():
    it[0] += 1
    return trn.train_step(*batch, seed=it[0])
def timeit(n=40):
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    cpu = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, cpu / n * 1e3
for _ in range(60): step()
for r in range(2):
    for mode in ("1", "0"):
        os.environ["MADE_DEC_FUSED"] = mode
        print(f"round {r} fused={mode} (used: {trn._dec_fused(B, Tv + Ta)}): %.3f ms/step, host issue %.3f ms" % timeit(), flush=True)
os.environ["MADE_DEC_FUSED"] = "1"
o1 = trn.forward_train(*batch, seed=7); l1 = (float(o1["retrieval_loss"]), float(o1["localization_loss"])); hs1 = o1["hs"].float().clone()
os.environ["MADE_DEC_FUSED"] = "0"
o0 = trn.forward_train(*batch, seed=7); l0 = (float(o0["retrieval_loss"]), float(o0["localization_loss"])); hs0 = o0["hs"].float()
print("losses fused", l1, "unfused", l0, "max |hs diff|", float((hs1 - hs0).abs().max()), "max |hs|", float(hs0.abs().max()))
