"""Training step with the decoder's 64-row stages chained (made_chain) against launching them one by one (MADE_CHAIN=0)."""
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
        os.environ["MADE_CHAIN"] = mode
        print(f"round {r} chain={mode}: %.3f ms/step, host issue %.3f ms" % timeit(), flush=True)
res = {}
for mode in ("1", "0"):
    os.environ["MADE_CHAIN"] = mode
    o = trn.forward_train(*batch, seed=7)
    trn.backward()
    torch.cuda.synchronize()
    res[mode] = (float(o["retrieval_loss"]), float(o["localization_loss"]), o["hs"].float().clone(), trn.flat_grad.clone())
print("losses chained", res["1"][:2], "separate", res["0"][:2], "hs identical:", torch.equal(res["1"][2], res["0"][2]),
      "grad cos", float(torch.nn.functional.cosine_similarity(res["1"][3], res["0"][3], dim=0)))
