"""Phase timeline of the fused decoder kernels (workgroup 0, s_memtime stamps at the phase boundaries), headline config."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.trainer import MadeTrainer
os.environ["MADE_DEC_FUSED"] = "1"
cfg = cfg_headline()
B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
trn = MadeTrainer(cfg, synth.make_state_dict(cfg, seed=0), dtype="bf16")
inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
t = {k: torch.from_numpy(v).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
batch = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
for i in range(5): trn.train_step(*batch, seed=i)
trn._dec_stamps = {"made_dec_train_fwd": torch.zeros(512, dtype=torch.int64, device="cuda"), "made_dec_train_bwd": torch.zeros(512, dtype=torch.int64, device="cuda")}
trn.train_step(*batch, seed=9)
torch.cuda.synchronize()
for fn, names in (("made_dec_train_fwd", ["sa_v", "sa_out", "ln1", "ca_q", "qfold", "scores", "softmax", "pooled", "vproj", "ca_out", "ln2", "ff1", "ff2", "ln3+norm"]),):
    st = trn._dec_stamps[fn].cpu().numpy()
    n = len(names)
    if st[1] == 0: continue
    per = np.zeros(n)
    nl = cfg.detr_dec_layers
    for l in range(nl):
        seg = st[l * n:(l + 1) * n + 1]
        per += np.diff(seg)
    tot = st[nl * n] - st[0]
    print(fn, "total ticks", tot, "(100 MHz ticks -> us: /100)")
    for k, v in zip(names, per / nl):
        print(f"  {k:10s} {v:9.0f} ticks/layer  {100 * v * nl / tot:5.1f} %")
