"""Which element of the fused matcher cost differs from (a) torch-CPU costs (the fixture's), (b) a numpy emulation of the kernel's formula."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from mgsv_amd import ops
from oracle import made_oracle as O
f32 = np.float32
fix = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "matcher.npz"))
def emu(lg, sp, tg, fg):
    Q, G = lg.shape[0], tg.shape[0]
    C = np.zeros((Q, G), f32); Pm = np.zeros(Q, f32)
    for q in range(Q):
        l0, l1 = lg[q]; mx = max(l0, l1)
        e0 = f32(np.exp(np.float64(f32(l0 - mx)))); e1 = f32(np.exp(np.float64(f32(l1 - mx))))
        p = f32((e0 if fg == 0 else e1) * f32(f32(1) / f32(e0 + e1))); Pm[q] = p
        pc, pw = sp[q]
        for j in range(G):
            tc, tw = tg[j]
            cs = f32(abs(f32(pc - tc)) + abs(f32(pw - tw)))
            hw = f32(f32(.5) * pw); ps, pe = f32(pc - hw), f32(pc + hw); hw2 = f32(f32(.5) * tw); ts, te = f32(tc - hw2), f32(tc + hw2)
            a1 = f32(pe - ps); a2 = f32(te - ts); inter = max(f32(min(pe, te) - max(ps, ts)), f32(0)); uni = f32(f32(a1 + a2) - inter); iou = f32(inter / uni)
            enc = max(f32(max(pe, te) - min(ps, ts)), f32(0)); g = f32(iou - f32(f32(enc - uni) / enc))
            C[q, j] = f32(f32(f32(f32(10) * cs) + f32(f32(1) * (-g))) + f32(f32(4) * (-p)))
    return C, Pm
for n, b in ((10, 1), (26, 1)):
    lg, sp, tg = fix[f"c{n}_logits"], fix[f"c{n}_spans"], fix[f"c{n}_targets"]; fg = int(fix[f"c{n}_fg"])
    keep = tg[b, :, 1] != 0
    _, _, _, _, cost = ops.hungarian_match(torch.from_numpy(lg).cuda(), torch.from_numpy(sp).cuda(), torch.from_numpy(tg).cuda(), fg)
    torch.cuda.synchronize()
    dev = cost[b].cpu().numpy()[:, :int(keep.sum())]
    ref = O.matcher_cost(torch.from_numpy(lg[b]), torch.from_numpy(sp[b]), torch.from_numpy(tg[b][keep]), fg).numpy()
    em, pm = emu(lg[b], sp[b], tg[b][keep], fg)
    pt = torch.from_numpy(lg[b]).softmax(-1).numpy()[:, fg]
    print(f"case {n} sample {b}: Q={lg.shape[1]} G={int(keep.sum())} | device != torch-CPU at {np.argwhere(dev != ref).tolist()} | device != emulation at {np.argwhere(dev != em).tolist()}"
          f" | emulation != torch-CPU at {np.argwhere(em != ref).tolist()} | p: emulation != torch at {np.argwhere(pm != pt).ravel().tolist()}")
    for q, j in np.argwhere(dev != em)[:6]:
        print(f"   [{q},{j}] device {dev[q, j]!r} emu {em[q, j]!r} torch {ref[q, j]!r} logits {lg[b, q].tolist()} p_emu {pm[q]!r} p_torch {pt[q]!r}")
