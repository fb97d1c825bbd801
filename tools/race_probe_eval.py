"""eval forward determinism: captured graphs of MadeEngine.forward replayed, one engine alone and two engines (two batches) in flight on
two streams as bench.py runs them; every replay's outputs against the engine's first eager run; on a bad replay the first workspace
buffer that differs, its damaged 1-KB blocks and where else those bytes exist.  Clean with the default kernels (0 of 6000 replays);
before the LDS-DMA kernels' raw barriers got their lgkmcnt(0), MADE_LINEAR_TILE=2128 (the ring kernel) showed 0.5-0.8 % replays with
garbage rows when two engines are in flight."""
import os, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.engine import MadeEngine
dev = torch.device("cuda", 0)
cfg = cfg_headline()
B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
sd = synth.make_state_dict(cfg, seed=0)
NL = 2
engs = [MadeEngine(cfg, sd, device=dev, dtype="bf16") for _ in range(NL)]
ts = []
for l in range(NL):
    inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1 + 1000 * l)
    ts.append({k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)})
def step(l):
    t = ts[l]
    return engs[l].forward(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
keys = None
refs, outs, graphs = [], [], []
for l in range(NL):
    o = step(l); torch.cuda.synchronize()
    keys = [k for k in ("hs", "pred_spans", "pred_logits", "sims_single", "sims_dual") if k in o and isinstance(o[k], torch.Tensor)]
    refs.append({k: o[k].clone() for k in keys})
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): step(l)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs.append(step(l))
    graphs.append(g)
N = int(os.environ.get("N", "200"))
streams = [torch.cuda.Stream() for _ in range(NL)]
for lanes in ((0,), (1,), (0, 1), (1, 0)):
    bad = 0
    for it in range(N):
        for l in lanes:
            with torch.cuda.stream(streams[l]):
                graphs[l].replay()
        torch.cuda.synchronize()
        for l in lanes:
            for k in keys:
                if not torch.equal(refs[l][k], outs[l][k]):
                    bad += 1
                    if bad <= 3:
                        d = (refs[l][k].float() - outs[l][k].float()).abs()
                        print(f"   {lanes} in flight, replay {it}, engine {l}: {k}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e}", flush=True)
                    break
    print(f"engines {lanes} in flight: {bad} of {N * len(lanes)} graph replays differ from the eager forward", flush=True)
# which workspace buffers differ after a bad replay (two engines in flight)?
wss = [list(e._ws.values())[0] for e in engs]
def snap(l):
    out = {}
    for k, v in wss[l].items():
        if isinstance(v, torch.Tensor): out[k] = v.clone()
    return out
for l in range(NL):
    graphs[l].replay()
torch.cuda.synchronize()
good = [snap(l) for l in range(NL)]
shown = 0
for it in range(3000):
    for l in (0, 1):
        with torch.cuda.stream(streams[l]):
            graphs[l].replay()
    torch.cuda.synchronize()
    for l in (0, 1):
        if not torch.equal(refs[l]["hs"], outs[l]["hs"]):
            diff = []
            for k, v in wss[l].items():
                if isinstance(v, torch.Tensor) and k in good[l] and v.shape == good[l][k].shape and not torch.equal(v, good[l][k]):
                    d = (v.float() - good[l][k].float()).abs()
                    rows = sorted(set((d.reshape(d.shape[0], -1) > 0).any(1).nonzero().reshape(-1).tolist()))[:6] if d.dim() >= 2 else []
                    diff.append(f"{k}{tuple(v.shape)}: {int((d > 0).sum()) + int(torch.isnan(d).sum())} elements, first-dim rows {rows}")
            print(f"bad replay {it}, engine {l}: differing workspace buffers:\n   " + "\n   ".join(diff[:6]), flush=True)
            first = None
            for k, v in wss[l].items():
                if isinstance(v, torch.Tensor) and k in good[l] and v.shape == good[l][k].shape and k not in ("dws", "part_o", "part_ml") and not torch.equal(v, good[l][k]):
                    first = k; break
            if first is not None:
                v, gd = wss[l][first].contiguous(), good[l][first].contiguous()
                bv, bg = v.view(-1).view(torch.uint8), gd.view(-1).view(torch.uint8)
                n1k = bv.numel() // 1024
                blk = (bv[:n1k * 1024].view(n1k, 1024) != bg[:n1k * 1024].view(n1k, 1024)).any(1).nonzero().reshape(-1).tolist()
                print(f"   first differing buffer {first}{tuple(v.shape)} {v.dtype}: damaged 1-KB blocks {blk[:16]} ({len(blk)})")
                raw = bv[blk[0] * 1024:(blk[0] + 1) * 1024]
                as32 = raw.view(torch.float32); as16 = raw.view(torch.bfloat16)
                print("   block read as f32:", [round(x, 4) for x in as32[:8].tolist()], "finite", bool(torch.isfinite(as32).all()), "max", float(as32.abs().max()),
                      "| as bf16:", [round(x, 3) for x in as16[:8].float().tolist()], "| expected (own dtype):", gd.view(-1)[blk[0] * 1024 // gd.element_size():][:6].float().tolist())
                hits = []
                for e_ in range(NL):
                    for k, w in wss[e_].items():
                        if not isinstance(w, torch.Tensor) or w.numel() * w.element_size() < 1024: continue
                        bb = w.contiguous().view(-1).view(torch.uint8)
                        n_ = bb.numel() // 1024
                        m_ = (bb[:n_ * 1024].view(n_, 1024) == raw[None]).all(1)
                        if bool(m_.any()) and not (e_ == l and k == first):
                            hits.append(f"engine {e_} {k}{tuple(w.shape)} {w.dtype} block {m_.nonzero().reshape(-1).tolist()[:4]}")
                    for k, w in good[e_].items():
                        bb = w.contiguous().view(-1).view(torch.uint8)
                        if bb.numel() < 1024: continue
                        n_ = bb.numel() // 1024
                        m_ = (bb[:n_ * 1024].view(n_, 1024) == raw[None]).all(1)
                        if bool(m_.any()):
                            hits.append(f"[good copy] engine {e_} {k}{tuple(w.shape)} block {m_.nonzero().reshape(-1).tolist()[:4]}")
                print("   the same 1 KB found in:", hits[:10])
            shown += 1
    if shown >= 2: break
