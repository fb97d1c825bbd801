"""Single-pass attention backward (made_attention_bwd, bf16, head dim 64) against the two-kernel form and f32 torch autograd, then the timing of
both forms at the step's shapes (GPU box):  python tools/attn_bwd_check.py [check|time|all]
The split form is selected per process with MADE_ATTN_BWD=split, so the script runs itself twice for the comparison."""
import os, sys, subprocess, json
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def make(B, H, hd, Lq, Lk, mask_kind, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    D = H * hd
    qkv = (torch.randn(B, max(Lq, Lk), 3 * D, generator=g) * 1.0).cuda().bfloat16()
    dO = torch.randn(B, Lq, D, generator=g).cuda().bfloat16()
    if mask_kind == "dense":
        km = torch.ones(B, Lk)
    elif mask_kind == "prefix":
        lens = torch.randint(max(1, Lk // 40), Lk + 1, (B,), generator=g)
        km = (torch.arange(Lk)[None] < lens[:, None]).float()
    else:                                  # the fused sequence of the DETR encoder: frames [0, lv) of 30, segments [30, 30 + la)
        lv, la = torch.randint(5, 31, (B,), generator=g), torch.randint(12, Lk - 30 + 1, (B,), generator=g)
        pos = torch.arange(Lk)[None]
        km = ((pos < lv[:, None]) | ((pos >= 30) & (pos < 30 + la[:, None]))).float()
    return qkv, dO, km.cuda()


def run(B, H, hd, Lq, Lk, mask_kind, p, use_bits, seed=0):
    from mgsv_amd import ops, ops_train as tr
    qkv, dO, km = make(B, H, hd, Lq, Lk, mask_kind, seed)
    D = H * hd
    q, k, v = qkv[:, :Lq, :D], qkv[:, :Lk, D:2 * D], qkv[:, :Lk, 2 * D:]
    qs = km if Lq == Lk else None
    O = torch.zeros(B, Lq, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, Lq, device="cuda")
    bits = torch.zeros(ops.attention_bits_shape(B, H, Lq, Lk), device="cuda", dtype=torch.int32) if (use_bits and p > 0) else None
    order = ops.batch_order(km) if Lq == Lk else None
    ops.attention(q, k, v, O, H, key_mask=km, q_skip_mask=qs, lse=lse, drop=(7, 3, p), keep_bits=bits, order=order)
    dqkv = torch.full((B, max(Lq, Lk), 3 * D), float("nan"), device="cuda", dtype=torch.bfloat16)
    delta = torch.zeros(B * H * Lq + 48 * B * H, device="cuda")          # (the stamps build writes 16 words per pair behind the deltas)
    f = lambda: tr.attention_bwd(q, k, v, O, dO, dqkv[:, :Lq, :D], dqkv[:, :Lk, D:2 * D], dqkv[:, :Lk, 2 * D:], lse, delta, H,
                                 key_mask=km, q_skip_mask=qs, drop=(7, 3, p), keep_bits=bits, order=order)
    f()
    torch.cuda.synchronize()
    return dqkv, f, (q, k, v, O, dO, km, qs, lse, delta)


CASES = [(2, 8, 64, 150, 150, "prefix", 0.0), (3, 8, 64, 542, 542, "fused", 0.1), (3, 8, 64, 542, 542, "fused", 0.0), (2, 4, 64, 70, 200, "prefix", 0.1),
         (2, 2, 64, 1, 1, "dense", 0.5), (4, 8, 64, 512, 512, "prefix", 0.8), (2, 8, 64, 30, 30, "prefix", 0.1), (1, 8, 64, 1024, 1024, "dense", 0.1),
         (2, 8, 64, 1054, 1054, "fused", 0.1), (64, 8, 64, 542, 542, "fused", 0.1)]


def check():
    out = {}
    for case in CASES:
        for bits in (True, False):
            if bits is False and case[0] == 64: continue
            dqkv, _, _ = run(*case, use_bits=bits)
            out[repr(case) + ("bits" if bits else "draw")] = dqkv.cpu()
    return out


def reference(case):
    """f32 torch autograd of the same op with the build's dropout mask"""
    from mgsv_amd import dropout
    B, H, hd, Lq, Lk, mk, p = case
    qkv, dO, km = make(B, H, hd, Lq, Lk, mk)
    D = H * hd
    q, k, v = [t.float().clone().requires_grad_(True) for t in (qkv[:, :Lq, :D], qkv[:, :Lk, D:2 * D], qkv[:, :Lk, 2 * D:])]
    qh, kh, vh = [t.view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v)]
    s = qh @ kh.transpose(-1, -2) * hd ** -0.5
    s = s.masked_fill((km == 0)[:, None, None, :], float("-inf"))
    a = torch.softmax(s, -1)
    if p > 0:
        import numpy as np
        keep = torch.from_numpy(dropout.keep_mask(7, 3, p, B * H * Lq * Lk).reshape(B, H, Lq, Lk)).cuda().float()
        a = a * keep / (1 - p)
    o = (a @ vh).transpose(1, 2).reshape(B, Lq, D)
    vq = (km if Lq == Lk else torch.ones(B, Lq, device="cuda"))[:, :, None]
    (o * dO.float() * vq).sum().backward()
    return q.grad, k.grad, v.grad


def timing():
    for case in [(64, 8, 64, 542, 542, "fused", 0.1), (64, 8, 64, 512, 512, "prefix", 0.1), (64, 8, 64, 542, 542, "dense", 0.1), (64, 8, 64, 542, 542, "fused", 0.0),
                 (64, 8, 64, 30, 30, "prefix", 0.1), (64, 8, 64, 1054, 1054, "fused", 0.1)]:
        _, f, _ = run(*case, use_bits=True)
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        print(f"{os.environ.get('MADE_ATTN_BWD', 'fused'):6s} {case}: {s.elapsed_time(e) / 20 * 1e3:8.1f} us", flush=True)


def stamps():
    """MADE_LIB_PATH=tools/_ab/fstamps.so: cycles per section of a workgroup (sum over the workgroup's life), averaged over the pairs"""
    names = ["scalars+flags", "tile lists", "zero fill", "key block prologue (K/V load, image, barrier)", "steps", "last dQ + dK/dV store + barrier",
             "dQ flush", "-"]
    for case in [(64, 8, 64, 542, 542, "fused", 0.1), (64, 8, 64, 542, 542, "dense", 0.1), (64, 8, 64, 512, 512, "prefix", 0.1)]:
        B, H, hd, Lq, Lk, mk, p = case
        _, f, t = run(*case, use_bits=True)
        f(); torch.cuda.synchronize()
        st = t[-1][B * H * Lq:B * H * Lq + 16 * B * H].view(torch.int32).view(B * H, 16).cpu().double()
        tot = st[:, 8]
        print(case, f"workgroup life: mean {tot.mean():.0f} max {tot.max():.0f} cycles; steps (nqt * ceil(nkt / 8)) mean {(st[:, 9] * ((st[:, 10] + 7) // 8)).mean():.1f}")
        for i in range(7):
            print(f"    {names[i]:55s} {st[:, i].mean():9.0f} cycles  {100 * st[:, i].sum() / tot.sum():5.1f} %")
        nsteps = (st[:, 9] * ((st[:, 10] + 7) // 8))
        print(f"    cycles per step: {st[:, 4].sum() / nsteps.sum():.0f}; per key block prologue {st[:, 3].sum() / (st[:, 11] * ((st[:, 10] + 7) // 8)).sum():.0f}, epilogue {st[:, 5].sum() / (st[:, 11] * ((st[:, 10] + 7) // 8)).sum():.0f}")
        pw = t[-1][B * H * Lq + 16 * B * H:].view(torch.int32).view(B * H, 8, 4).cpu().double()
        for wv in (0, 3, 4, 7):
            x = pw[:, wv].sum(0) / nsteps.sum()
            print(f"    wave {wv}: per step: staging {x[0]:.0f}  M phase {x[1]:.0f}  V phase {x[2]:.0f}  barrier wait {x[3]:.0f}")
        start = st[:, 12] - st[:, 12].min()
        end = start * 256 + tot
        print(f"    first start .. last end: {end.max():.0f} cycles; sum of lives / 256 CUs: {tot.sum() / 256:.0f}")


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode == "stamps":
        stamps(); sys.exit(0)
    if mode == "dump":
        torch.save(check(), sys.argv[2]); sys.exit(0)
    if mode in ("check", "all"):
        tmp = "/tmp/attn_bwd_split.pt"
        subprocess.run([sys.executable, __file__, "dump", tmp], env=dict(os.environ, MADE_ATTN_BWD="split"), check=True)
        split = torch.load(tmp)
        mine = check()
        bad = 0
        for case in CASES:
            B, H, hd, Lq, Lk, mk, p = case
            D = H * hd
            ref = reference(case) if B <= 4 else None
            for tag in ("bits", "draw"):
                key = repr(case) + tag
                if key not in mine: continue
                a_, b_ = mine[key].float(), split[key].float()
                fin = all(bool(torch.isfinite(a_[:, rs_, cs_]).all()) for rs_, cs_ in ((slice(0, Lq), slice(0, D)), (slice(0, Lk), slice(D, 3 * D))))
                parts = {"dq": (slice(0, Lq), slice(0, D)), "dk": (slice(0, Lk), slice(D, 2 * D)), "dv": (slice(0, Lk), slice(2 * D, 3 * D))}
                msg = []
                for n, (rs, cs) in parts.items():
                    x, y = a_[:, rs, cs], b_[:, rs, cs]
                    d = float((x - y).abs().max()); sc = float(y.abs().max())
                    m = f"{n}: |fused-split| {d:.3e} / {sc:.3e}"
                    if ref is not None:
                        rr = ref[["dq", "dk", "dv"].index(n)].cpu()
                        ef, es = float((x - rr).abs().max()), float((y - rr).abs().max())
                        m += f"  err vs f32: fused {ef:.3e} split {es:.3e}"
                        if ef > 1.5 * es + 1e-3 * float(rr.abs().max()): bad += 1; m += "  <-- WORSE"
                    elif d > 0.03 * sc: bad += 1; m += "  <-- FAR"
                    msg.append(m)
                if tag == "draw" and not torch.equal(mine[key].view(torch.int16), mine[repr(case) + "bits"].view(torch.int16)):
                    msg.append("draw != bits"); bad += 1
                print(case, tag, "finite" if fin else "NOT FINITE", *msg, sep="\n    ", flush=True)
                if not fin: bad += 1
        print("BAD" if bad else "OK", bad)
    if mode in ("time", "all"):
        timing()
        subprocess.run([sys.executable, __file__, "time"], env=dict(os.environ, MADE_ATTN_BWD="split"), check=False) if os.environ.get("MADE_ATTN_BWD") != "split" else None
