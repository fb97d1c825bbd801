"""What frequency do the CUs run at inside the training step?  A one-wave probe kernel (tools/probes/clock_probe.hip: shader-clock cycles
per 100 MHz tick) is launched in front of every wide-attention call of the decoder (forward and backward) of the eager step, and in an
otherwise idle stream, and the frequencies are printed.  usage: python tools/clock_probe.py"""
import ctypes as C, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
so = "/tmp/clock_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so,
                       os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "clock_probe.hip")])
lib = C.CDLL(so)
lib.clock_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
from mgsv_amd import ops, ops_train as tr, synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.trainer import MadeTrainer
dev = torch.device("cuda", 0)
out = torch.zeros(3 * 256, dtype=torch.int64, device=dev)
slot = [0]
TICKS = int(os.environ.get("PROBE_TICKS", "300"))           # 3 us
def probe():
    if slot[0] < 256:
        lib.clock_probe(out.data_ptr(), slot[0], TICKS, torch.cuda.current_stream().cuda_stream)
        slot[0] += 1
def report(tag):
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(-1, 3)[:slot[0]]
    f = o[:, 0] / (o[:, 1] / 100.0)                              # cycles per us = MHz
    t = (o[:, 2] - o[0, 2]) / 100.0
    print(tag, " ".join(f"{x:.0f}@{tt:.0f}us" for x, tt in zip(f, t)), flush=True)
    slot[0] = 0
# idle GPU
for _ in range(8):
    probe(); torch.cuda.synchronize(); time.sleep(0.01)
report("idle, one probe at a time (MHz):")
for _ in range(16):
    probe()
report("idle, back to back:")
cfg = cfg_headline()
B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
sd = synth.make_state_dict(cfg, seed=0)
inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
trn = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
batch = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
it = [0]
def step():
    it[0] += 1
    return trn.train_step(*batch, seed=it[0], lrs=(1e-4, 1e-4, 1e-4), max_grad_norm=1.0, dist=None)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    step()
torch.cuda.synchronize()
aw, awb = ops.attention_wide, tr.attention_wide_bwd
def aw_p(*a, **k):
    probe(); return aw(*a, **k)
def awb_p(*a, **k):
    probe(); return awb(*a, **k)
ops.attention_wide, tr.attention_wide_bwd = aw_p, awb_p
import mgsv_amd.trainer as T
for _ in range(20):
    step()
torch.cuda.synchronize(); slot[0] = 0
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
print(f"step with probes: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
report("in the step (3 steps; per step: X-Pool, 6 decoder layers forward, X-Pool bwd?, 6 backward):")
# the decoder-like chain alone, in a loop: same probe between idle-ish launches
x = torch.randn(64, 512, device=dev).bfloat16(); w = torch.randn(512, 512, device=dev).bfloat16() * 0.05
for _ in range(200):
    x2 = ops.linear(x, w, None)
torch.cuda.synchronize()
for i in range(12):
    for _ in range(10):
        x2 = ops.linear(x, w, None)
    probe()
report("inside a chain of 64-row linears only:")
