"""first captured step (launch tape / hipGraph) against the eager step from the same state, repeated with fresh trainers: which forward
outputs, decoder buffers and (atomics-free) gradient stacks of the decoder's backward chain differ, and in which rows.
usage: python tools/race_probe3.py [tape] [graph]   (N = repetitions per mode)"""
import os, sys
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
t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
b = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
keys = ("memory", "hs", "retrieval_loss", "localization_loss")
eager = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
oe = eager.train_step(*b, seed=7, lrs=(1e-4, 1e-4, 1e-4))
torch.cuda.synchronize()
ref = {k: oe[k].clone() for k in keys}
tw = eager._train_buffers(B, Tv, Ta)
names = [k for k in tw if isinstance(tw[k], torch.Tensor) and k.startswith("d.")]
refb = {k: tw[k].clone() for k in names}
gnames = [k for k in tw["dstack"] if k.startswith("g_") or k == "dt1q"]          # the backward chain's per-layer gradients (no atomics)
refg = {k: tw["dstack"][k].clone() for k in gnames}
refin = {k: tw[k].clone() for k in ("dgN", "dhs") if k in tw}                        # what the chain starts from
refc = tw["dchain"].clone() if "dchain" in tw else None                             # [layer, hand-off, B*Q, D]: the chain's hand-off rows
# slot of tw["dchain"][layer] -> what the backward writes there (trainer.py: g1a, g1b, g2a, g2b, g2c, dt_out)
HANDOFF = ("FFN-1 dX + residual", "query dX + residual", "norm-3 stage dx", "norm-2 stage dx", "norm-1 stage dx", "layer output (d tgt)")
ORDER = (2, 0, 3, 1, 4, 5)                                                           # slots in the order the backward of a layer writes them
# SINGLE=1: the captured trainers run everything on ONE stream (their second stream is the current one): does the deviation need a
# second stream's kernels beside the chain?
if os.environ.get("SINGLE", "0") == "1":
    _orig_side = MadeTrainer._side_stream
# UNCACHED=1: the backward chain's hand-off buffers (dchain, dgN, dhs, the g_* stacks) in UNCACHED device memory
# (hipExtMallocWithFlags(hipDeviceMallocUncached)): does the deviation need a cache between a chain kernel and the next one?
_uncached_keep = []
def uncached_like(t):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    ptr = C.c_void_p()
    nbytes = t.numel() * t.element_size()
    rc = hip.hipExtMallocWithFlags(C.byref(ptr), C.c_size_t(nbytes), C.c_uint(3))
    assert rc == 0 and ptr.value, f"hipExtMallocWithFlags failed: {rc}"
    class H:
        pass
    h = H()
    typestr = {torch.bfloat16: "<u2", torch.float32: "<f4", torch.int32: "<i4"}[t.dtype]
    h.__cuda_array_interface__ = {"shape": tuple(t.shape), "typestr": typestr, "data": (ptr.value, False), "version": 2}
    u = torch.as_tensor(h, device=dev)
    if t.dtype == torch.bfloat16:
        u = u.view(torch.bfloat16)
    u.zero_()
    _uncached_keep.append(h)
    return u
def make_uncached(tw_):
    for k in ("dchain", "dgN", "dhs"):
        if k in tw_:
            tw_[k] = uncached_like(tw_[k])
    for k in list(tw_["dstack"].keys()):
        if k.startswith("g_") or k == "dt1q":
            tw_["dstack"][k] = uncached_like(tw_["dstack"][k])
for mode in sys.argv[1:] or ("graph", "tape"):
    for rep in range(int(os.environ.get("N", "5"))):
        tr_ = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
        if os.environ.get("SINGLE", "0") == "1":
            tr_._side_stream = lambda: torch.cuda.current_stream()
        if os.environ.get("RETMAIN", "0") == "1":
            # two streams as always, but the retrieval branch's backward (the only second-stream work beside the first decoder layers of
            # the backward chain) is issued on the MAIN stream in front of the chain
            main_ = torch.cuda.current_stream()
            orig_ = tr_._retrieval_bwd
            def on_main(*a_, _o=orig_, **k_):
                cur_ = torch.cuda.current_stream()
                main_.wait_stream(cur_)
                with torch.cuda.stream(main_):
                    _o(*a_, **k_)
                cur_.wait_stream(main_)
            tr_._retrieval_bwd = on_main
        if os.environ.get("UNCACHED", "0") == "1":
            make_uncached(tr_._train_buffers(B, Tv, Ta))
        g = tr_.capture_train_step(*b, mode=mode)
        # LR0=1: the captured step updates nothing (learning rates 0): whatever a later trainer finds in recycled memory -- e.g. the
        # bf16 weight copies of the trainer before it -- then equals what it writes there itself
        og = g.step(*b, seed=7, lrs=(0.0, 0.0, 0.0) if os.environ.get("LR0", "0") == "1" else (1e-4, 1e-4, 1e-4))
        torch.cuda.synchronize()
        msg = []
        for k in keys:
            if not torch.equal(ref[k], og[k]):
                d = (ref[k].float() - og[k].float()).abs()
                msg.append(f"{k}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}")
        tw2 = tr_._train_buffers(B, Tv, Ta)
        first = None
        for k in names:
            if k in tw2 and tw2[k].shape == refb[k].shape and not torch.equal(tw2[k], refb[k]):
                d = (tw2[k].float() - refb[k].float()).abs()
                msg.append(f"   buffer {k}: {int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.3e}; rows {sorted(set((d > 0).nonzero()[:, 0].tolist()))[:12]}")
        for k in gnames:
            a_, b_ = tw2["dstack"][k], refg[k]
            if not torch.equal(a_, b_):
                d = (a_.float() - b_.float()).abs()
                lay = sorted(set((d > 0).nonzero()[:, 0].tolist()))
                msg.append(f"   gradient stack {k}: {int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.3e}; layers {lay}")
        if refc is not None and "dchain" in tw2 and not torch.equal(tw2["dchain"], refc):
            dd = (tw2["dchain"].float() - refc.float()).abs()
            nd_ = dd.shape[0]
            first = None
            for l in range(nd_ - 1, -1, -1):
                for slot in ORDER:
                    if float(dd[l, slot].max()) > 0:
                        rows = sorted(set((dd[l, slot] > 0).nonzero()[:, 0].tolist()))
                        cols = (dd[l, slot] > 0).nonzero()[:, 1]
                        first = f"layer {l}, {HANDOFF[slot]}: {int((dd[l, slot] > 0).sum())} elements, rows {rows[:6]}, columns {int(cols.min())}..{int(cols.max())}, max {float(dd[l, slot].max()):.2e}"
                        break
                if first: break
            msg.insert(0, "   first hand-off of the backward chain that differs: " + str(first))
        # the highest layer in which anything differs: every per-layer product of the chain in the order the layer's backward writes it,
        # with the rows and the 16-column tiles that differ (a stage's workgroup owns 16 rows x 16 columns of its product)
        top = None
        for l in range(refg["g_z"].shape[0] - 1, -1, -1):
            if any(not torch.equal(tw2["dstack"][k][l], refg[k][l]) for k in gnames):
                top = l
                break
        if top is not None:
            def where(a_, b_):
                a2, b2 = a_.reshape(-1, a_.shape[-1]).float(), b_.reshape(-1, b_.shape[-1]).float()
                nz = (a2 != b2).nonzero()
                if nz.numel() == 0:
                    return "same"
                rows = sorted(set(nz[:, 0].tolist())); tiles = sorted(set((nz[:, 1] // 16).tolist()))
                return f"{nz.shape[0]} elements, rows {rows[:8]}, 16-column tiles {tiles[:40]}, max {float((a2 - b2).abs().max()):.2e}"
            det = [f"   layer {top} in the order of its backward:"]
            seq = [("g_ffn", None), ("g_z", None), (None, 2), (None, 0), ("g_ca", None), ("g_attc", None), (None, 3), ("g_q", None), ("g_qc", None),
                   ("dt1q", None), (None, 1), ("g_sa", None), (None, 4), (None, 5)]
            for k, slot in seq:
                if k is not None and k in refg:
                    det.append(f"      {k}: " + where(tw2["dstack"][k][top], refg[k][top]))
                elif slot is not None and refc is not None:
                    det.append(f"      hand-off '{HANDOFF[slot]}': " + where(tw2["dchain"][top, slot], refc[top, slot]))
            if refc is not None and top + 1 < refc.shape[0]:
                det.append(f"      (layer {top + 1}'s output, this layer's `add`: " + where(tw2["dchain"][top + 1, 5], refc[top + 1, 5]) + ")")
            for k in refin:
                det.append(f"      ({k}, all layers: " + where(tw2[k], refin[k]) + ")")
            msg[1:1] = det
        print(mode, rep, "identical" if not msg else "\n  ".join(["DIFFERENT"] + msg[:40]), flush=True)
        del g, tr_
