"""Where the D = 512 / S = 512 retrieval pass spends its time, per chunk size of the tracks: wall time of a pass and the HIP-event totals of the
timed kernel kinds (made_linear variants, made_xpool_attention; LayerNorm and made_xpool_tail are the remainder).  usage: python tools/retr512_breakdown.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgsv_amd import ops, synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.engine import MadeEngine
cfg = cfg_headline(); dev = torch.device("cuda")
eng = MadeEngine(cfg, synth.make_state_dict(cfg, seed=0), device=dev, dtype="bf16")
n_v, n_m, S, D = 8192, 512, 512, cfg.D
g = torch.Generator(device=dev).manual_seed(5)
v = torch.nn.functional.normalize(torch.randn(n_v, D, device=dev, generator=g), dim=-1)
seg = torch.randn(n_m, S, D, device=dev, generator=g)
lens = torch.randint(12, S + 1, (n_m,), device=dev, generator=g)
mask = (torch.arange(S, device=dev)[None] < lens[:, None]).float()
seg = (seg * mask[:, :, None]).to(eng.tc)
mu = torch.nn.functional.normalize(torch.randn(n_m, D, device=dev, generator=g), dim=-1)
for cm in (None, 12, 24, 48, 96, 192):
    step = lambda: eng.retrieval_sim_matrix(v, seg, mask, mu, chunk_m=cm)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    with ops.KernelTimer() as kt:
        step()
    sm = kt.summary()
    print(f"chunk_m={cm}: {ms:.2f} ms per pass; " + "; ".join(f"{k} x{d['launches']} {d['ms']:.2f} ms" for k, d in sorted(sm.items(), key=lambda kv: -kv[1]['ms'])))
