"""CPU (hipcc cross-compiles): properties of the generated ISA that the source relies on but cannot express.

* `xs_dma16` (csrc/xpool_attn.hip) writes M0 from inline assembly without a clobber -- the compiler refuses M0 in a clobber list.  That is correct only while
  the compiler itself never reads or writes M0 anywhere in the two made_xpool_sims kernels: every mention of m0 in their ISA must be our own `s_mov_b32 m0, sN`.
* `matcher.hip` restates the reference's cost one rounding per operation: no FMA outside the IEEE division expansions (round 5 found contraction there).
* the made_xpool_sims kernels run two workgroups per CU on counted vector-memory waits: no scratch (a spill reload's wait would also cover the LDS-DMA pieces).
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mgsv_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-Wno-unused-function", "-Wno-pass-failed", "-w"]


def _isa(src, extra=()):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = subprocess.run([HIPCC, *FLAGS, *extra, "-o", "-", os.path.join(CSRC, src)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def _kernels(text, pattern):
    """{symbol: [instruction lines]} of the kernels whose symbol matches"""
    res, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1) if re.search(pattern, m.group(1)) else None
            if cur:
                res[cur] = []
            continue
        if cur is not None:
            s = ln.strip()
            if s.startswith("s_endpgm"):
                cur = None
            elif s and not s.startswith((";", ".")):
                res[cur].append(s)
    return res


def test_xpool_sims_kernels_leave_m0_to_the_inline_assembly_and_do_not_spill():
    text = _isa("xpool_attn.hip")
    ks = _kernels(text, r"xpool_sims(32|64)_kernel")
    assert len(ks) >= 3, list(ks)                               # 32-video (+ its stamp build), 64-video (+ stamp build)
    for name, ins in ks.items():
        m0 = [i for i in ins if re.search(r"\bm0\b", i)]
        assert m0, name                                         # the LDS-DMA pieces are there
        foreign = [i for i in m0 if not re.match(r"s_mov_b32 m0, s\d+", i)]
        assert not foreign, (name, foreign[:5])
        dma = sum(1 for i in ins if i.startswith("global_load_lds_dwordx4"))
        assert dma > 0 and len(m0) == dma, (name, len(m0), dma)   # one M0 write per piece, nothing else
    # no scratch in the shipped builds (the stamp builds may spill a register or two: measurement aids)
    meta = re.findall(r"\.name:\s+(\S*xpool_sims(?:32|64)_kernel\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text)
    shipped = {n: int(b) for n, b in meta if "ELb0E" in n}
    assert len(shipped) == 2, meta
    assert all(b == 0 for b in shipped.values()), shipped


def test_matcher_cost_has_no_fma_contraction():
    text = _isa("matcher.hip", extra=["-ffp-contract=off"])      # (the flag csrc/Makefile compiles this file with)
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert re.search(r"build/matcher\.o:\s*CXXFLAGS\s*\+=\s*-ffp-contract=off", mk)
    ks = _kernels(text, r"hungarian_kernel")
    assert len(ks) == 1
    ins = next(iter(ks.values()))
    # FMAs are legal only inside the IEEE f32 division expansions (v_div_scale ... v_div_fmas ... v_div_fixup) and in the f64 exponential
    fma32 = [k for k, i in enumerate(ins) if re.match(r"v_(fma|fmac|mad)_f32", i)]
    assert fma32, "expected the division expansions"
    for k in fma32:
        window = ins[max(0, k - 12):k + 12]
        assert any(w.startswith(("v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_rcp_f32")) for w in window), ins[max(0, k - 3):k + 3]
    n_div = sum(1 for i in ins if i.startswith("v_div_fixup_f32"))
    assert n_div == 3 and len(fma32) <= 7 * n_div, (n_div, len(fma32))   # the two GIoU quotients and 1 / (e0 + e1)
