"""Where does made_dec_stage_bwd's result move beside a register-staged Linear (the SLP build, tools/_ab/dec_packed.so)?  Every differing launch's
(tensor, row, column, solo bits, observed bits); then a histogram.  MADE_LIB_PATH=tools/_ab/dec_packed.so python tools/dec_corun_diag.py [N]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.dec_corun_probe as P
from mgsv_amd import _lib
dev = P.dev; bf = P.bf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
I = P.stage_inputs()
M, D, N_ = 64, 512, 1024
Z = lambda: torch.zeros(D, device=dev)
ref = [torch.empty(M, N_, device=dev, dtype=bf), torch.empty(M, D, device=dev, dtype=bf), torch.empty(M, D, device=dev, dtype=bf)]
P.run_stage(I, *ref, [Z(), Z()]); torch.cuda.synchronize()
co = [v for k, v in P.corunners().items() if "register-staged" in k][0]
if os.environ.get("POISON"):                                  # POISON=<hex pattern>: the co-runner only leaves that pattern in the SIMDs' registers (tools/probes/vgpr_poison.hip)
    import ctypes
    pz = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", "vgpr_poison.so"))
    sink = torch.zeros(4, device=dev, dtype=torch.int32)
    pat = int(os.environ["POISON"], 16)
    co = lambda: pz.vgpr_poison(ctypes.c_uint(pat), 2048, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(sink.data_ptr()))
    print(f"co-runner: VGPR poison {pat:#010x}")
ring = [[torch.empty_like(r) for r in ref] for _ in range(8)]
pg = [Z(), Z()]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
hist = collections.Counter(); vals = collections.defaultdict(collections.Counter); nbad = 0
print(f"lib={os.path.basename(_lib.LIB_PATH)}")
for i in range(N):
    with torch.cuda.stream(s2):
        co()
        if i % 4 == 0: co()
    with torch.cuda.stream(s1):
        o = ring[i % 8]
        P.run_stage(I, *o, pg)
    s1.synchronize()
    for j, name in enumerate(("out", "dx_out", "a_out")):
        d = (o[j].view(torch.int16) != ref[j].view(torch.int16))
        if bool(d.any()):
            nbad += 1
            for r, c in d.nonzero().tolist()[:8]:
                hist[(name, r, c)] += 1
                vals[(name, r, c)][(int(ref[j].view(torch.int16)[r, c]) & 0xFFFF, int(o[j].view(torch.int16)[r, c]) & 0xFFFF)] += 1
torch.cuda.synchronize()
print(f"{N} launches, {nbad} differing tensors; positions (tensor, row, col): count, {{(solo bits, observed bits): count}}")
for k, v in hist.most_common(20):
    print(f"  {k}: {v}  " + ", ".join(f"({a:04x} -> {b:04x}) x{n}" for (a, b), n in vals[k].most_common(4)))
