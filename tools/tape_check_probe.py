"""Which framework (ATen) operators touch device tensors inside the recorded training step of a configuration?  (mgsv_amd/tape.py: the
recorder's watcher; TrainStepGraph(mode='tape') refuses a step that has any.)  usage: python tools/tape_check_probe.py [headline|native]"""
import collections, os, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
os.environ["MADE_TAPE_CHECK"] = "0"
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
c = collections.Counter(g.tape.foreign_ops)
print("framework operators on device tensors inside the recorded headline step:", dict(c) if c else "none")
