import os, sys, torch, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B, H, D = 64, 8, 512
for L in (32, 128, 542):
    for ns in (1, 4):
        q = torch.randn(B, 1, H * D, device=dev).to(dt); mem = torch.randn(B, L, D, device=dev).to(dt); mp = torch.randn(B, L, D, device=dev).to(dt)
        o = torch.empty(B, 1, H * D, device=dev, dtype=dt); mask = torch.ones(B, L, device=dev)
        q4 = q.view(B, 1, H, D).permute(0, 2, 1, 3); o4 = o.view(B, 1, H, D).permute(0, 2, 1, 3)
        po = torch.empty(B * ns * H * D, device=dev); pml = torch.empty(B * ns * H * 2, device=dev)
        t = timeit(lambda: ops.attention_wide(q4, mp, mem, o4, scale=0.125, key_mask=mask, n_split=ns, part_o=po, part_ml=pml))
        t2 = timeit(lambda: ops.attention_wide(q4, mp, mem, o4, scale=0.125, key_mask=None, n_split=ns, part_o=po, part_ml=pml))
        print(f"wide decoder L={L} n_split={ns}: {t:7.1f} us (masked) {t2:7.1f} us (no mask)")
