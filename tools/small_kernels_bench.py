"""The single-workgroup / small kernels at the head and tail of a step, for rocprofv3 --kernel-trace (tools/small_kernels_trace.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
dev = torch.device("cuda")
B, T, D = 64, 512, 512
lens = torch.randint(40, T + 1, (B,), device=dev)
mask = (torch.arange(T, device=dev)[None] < lens[:, None]).float()
vmask = (torch.arange(30, device=dev)[None] < torch.randint(5, 31, (B,), device=dev)[:, None]).float()
x = torch.randn(B, T, D, device=dev).bfloat16()
dim_t = torch.tensor([10000.0 ** (2 * (i // 2) / D) for i in range(D)], device=dev)
fmask = torch.cat([vmask, mask], 1).contiguous()
for _ in range(6):
    ops.sine_pe(fmask, dim_t, out_dtype=torch.bfloat16)
    ops.row_index(mask)
    ops.row_index(vmask)
    ops.batch_order(mask)
    ops.masked_mean(x, mask)
torch.cuda.synchronize()
