"""Every weight-gradient product (made_gemm_tn / made_gemm_tn_grouped) of one eager training step at the headline shape: M, N, K, batch, reduction
splits, and the bytes its splits add to the gradient with atomics (the atomic units sustain 1.2 TB/s: docs/EXPERIMENTS.md 3f-4)."""
import os, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import synth, ops_train as tr, _lib
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
trn.train_step(*batch, seed=1)
torch.cuda.synchronize()
log = []
o_check = tr.check
lib = _lib.lib()
o_tn, o_g = lib.made_gemm_tn, lib.made_gemm_tn_grouped


class Spy:
    def __init__(self, fn, kind): self.fn, self.kind = fn, kind
    def __call__(self, ref, stream):
        a = ref._obj
        if self.kind == "tn":
            nz = a.batch1 * a.batch2
            rows = "gather" if a.row_index else ("mask" if a.row_mask else "")
            log.append(("gemm_tn", a.M, a.N, a.K, nz, a.split_m, rows, torch.cuda.current_stream().cuda_stream))
        else:
            for i in range(a.n_problems):
                log.append((f"grouped{a.tile_size or 128}[{i}/{a.n_problems}]", a.M, a.p[i].N, a.p[i].K, 1, a.split_m, "gather" if a.row_index else "", torch.cuda.current_stream().cuda_stream))
        return self.fn(ref, stream)


class LibSpy:
    def __getattr__(self, name):
        if name == "made_gemm_tn": return Spy(o_tn, "tn")
        if name == "made_gemm_tn_grouped": return Spy(o_g, "g")
        return getattr(lib, name)


tr.lib = lambda: LibSpy()
trn.train_step(*batch, seed=2)
torch.cuda.synchronize()
main = torch.cuda.current_stream().cuda_stream
tot = 0.0
print(f"{'call':22s} {'M':>6s} {'N':>5s} {'K':>5s} {'z':>3s} {'split':>5s} rows    stream   atomic MB (128 x 128 tiles x splits)")
for kind, M, N, K, nz, sp, rows, st in log:
    tiles = ((N + 127) // 128) * ((K + 127) // 128) * nz
    mb = tiles * max(sp, 1) * 65536 / 1e6 if not kind.startswith("grouped256") else float("nan")
    if mb == mb: tot += mb
    print(f"{kind:22s} {M:6d} {N:5d} {K:5d} {nz:3d} {sp:5d} {rows:7s} {'main' if st == main else 'side':6s} {mb:8.1f}")
print(f"total (without the 256 x 256-tile launches): {tot:.0f} MB = {tot / 1.2:.0f} us of the atomic units at 1.2 TB/s")
