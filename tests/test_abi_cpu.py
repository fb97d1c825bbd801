"""CPU: the C-ABI library loads and exports every symbol include/made_hip.h declares; argument
validation works without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from mgsv_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "made_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(made_[a-z0-9_]+)\s*\(", text))
    inline = set(re.findall(r"static inline \w+ (made_[a-z0-9_]+)\s*\(", text))     # header-only helpers (dropout RNG)
    return sorted(names - inline)


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    names = _declared_symbols()
    assert len(names) >= 14
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in made_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    assert _lib.lib().made_abi_version() == _lib.ABI_VERSION == 8


def _all_structs():
    return sorted((n, t) for n, t in vars(_lib).items() if isinstance(t, type) and issubclass(t, C.Structure) and t is not C.Structure
                  and n.startswith("Made"))


def test_struct_layout_matches_header():
    """EVERY ctypes mirror in mgsv_amd/_lib.py against the C compiler's layout of include/made_hip.h: size and the offset of every
    field (padding fields included), so a drifting struct fails here and not as a corrupted kernel argument."""
    import subprocess, tempfile
    structs = _all_structs()
    assert len(structs) >= 15 and {"MadeDecStageArgs", "MadeGemmTNGroup", "MadeGemmTNProblem", "MadeAdamDeviceState", "MadeXpoolFusedArgs"} <= {n for n, _ in structs}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "made_hip.h"', "int main(void){"]
    want = []
    for n, t in structs:
        lines.append(f'printf("%zu\\n", sizeof({n}));')
        want.append(C.sizeof(t))
        for f in t._fields_:
            lines.append(f'printf("%zu\\n", offsetof({n}, {f[0]}));')
            want.append(getattr(t, f[0]).offset)
    lines.append("return 0; }")
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "s.c")
        open(p, "w").write("\n".join(lines))
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), p, "-o", exe])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    labels = [x for n, t in structs for x in [f"sizeof({n})"] + [f"{n}.{f[0]}" for f in t._fields_]]
    bad = [(l, g, w) for l, g, w in zip(labels, got, want) if g != w]
    assert not bad and len(got) == len(want), bad[:10]


def test_dropout_rng_header_matches_numpy_restatement():
    """made_rng_mix in include/made_hip.h (compiled here with gcc) == mgsv_amd.dropout.rng_mix."""
    import subprocess, tempfile, textwrap
    import numpy as np
    from mgsv_amd import dropout as dr
    src = textwrap.dedent('''
        #include <stdio.h>
        #include "made_hip.h"
        int main(void){
            unsigned long long seeds[2] = {1234ULL, 0xDEADBEEF12345678ULL};
            for (int s = 0; s < 2; ++s) for (unsigned site = 5; site < 4000000000u; site += 1999999999u)
                for (unsigned long long i = 0; i < 6; ++i) {
                    unsigned long long idx = i * 0x40000001ULL + i;
                    printf("%u\\n", made_rng_mix(seeds[s], site, idx));
                }
            return 0; }
    ''')
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "r.c")
        open(p, "w").write(src)
        exe = os.path.join(d, "r")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), p, "-o", exe])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    want = []
    for seed in (1234, 0xDEADBEEF12345678):
        for site in range(5, 4000000000, 1999999999):
            idx = np.array([i * 0x40000001 + i for i in range(6)], dtype=np.uint64)
            want += [int(x) for x in dr.rng_mix(seed, site, idx)]
    assert got == want


def test_argument_validation_without_gpu():
    l = _lib.lib()
    assert l.made_linear(None, None) == -1
    assert b"null args" in l.made_last_error()
    a = _lib.MadeAttnArgs()
    assert l.made_attention(C.byref(a), None) == -1
    with pytest.raises(_lib.MadeError):
        _lib.check(-1, "x")


def test_product_path_refuses_cpu_tensors():
    import torch
    from mgsv_amd import ops
    with pytest.raises(_lib.MadeError):
        ops.layernorm(torch.zeros(4, 256), torch.ones(256), torch.zeros(256))


def test_weight_gradient_split_model():
    """ops_train._tn_splits (reduction splits of the 128 x 128-tile weight-gradient kernels by cost: products against f32 atomic adds): never more
    splits than slabs, never more than 64 slabs' row indices per workgroup, short reductions few splits, long ones the 16 measured best."""
    from mgsv_amd.ops_train import _tn_splits
    for tiles in (1, 4, 16, 24, 32, 80, 128, 256):
        for nslab in (1, 2, 4, 6, 30, 64, 141, 512, 542, 1000, 4096):
            for gathered in (False, True):
                sp = _tn_splits(tiles, nslab, gathered, 64)
                assert 1 <= sp and (sp <= nslab or nslab > 64 * sp - 1 or (nslab + sp - 1) // sp <= 64), (tiles, nslab, sp)
                assert (nslab + sp - 1) // sp <= 64, (tiles, nslab, sp)
    assert _tn_splits(128, 30, True, 64) <= 3            # the video tower's grouped launch (1 920 rows): 14 splits until round 6
    assert _tn_splits(16, 30, True, 64) <= 8
    assert _tn_splits(16, 64, False, 64) <= 8            # the 4 096 pair rows: 32 splits until round 6
    assert _tn_splits(32, 512, True, 64) == 16           # the audio tower's 32 768 rows: as measured best in rounds 2-3
