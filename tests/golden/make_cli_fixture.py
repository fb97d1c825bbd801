"""Record the reference's command-line interface as data  --  build container only.

Loads the reference's two driver scripts from /root/reference (nothing is copied), intercepts `ArgumentParser.parse_args` and
stores, per flag, its default, choices and whether it is a store_true switch / required -> tests/golden/cli_flags.json.

    python tests/golden/make_cli_fixture.py
"""
import argparse
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True                              # (before anything of the reference is imported: no __pycache__ in its tree)

from oracle import ref_import  # noqa: E402

ref_import.import_reference()                               # (puts the reference root on sys.path)
if "torch.utils.tensorboard" not in sys.modules:            # the drivers import SummaryWriter at module level; not installed here
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb

import torch.distributed as _dist  # noqa: E402

_dist.init_process_group = lambda *a, **k: None               # the drivers initialise the process group at import time
_dist.get_world_size = lambda *a, **k: 1
_dist.get_rank = lambda *a, **k: 0

out = {}
for script in ("train-MaDe.py", "test-MaDe.py"):
    captured = {}
    orig = argparse.ArgumentParser.parse_args

    def fake(self, *a, **k):
        for act in self._actions:
            for o in act.option_strings:
                captured[o.lstrip("-")] = {"default": act.default, "choices": list(act.choices) if act.choices else None,
                                           "flag": isinstance(act, argparse._StoreTrueAction), "required": bool(act.required)}
        raise SystemExit(0)

    argparse.ArgumentParser.parse_args = fake
    try:
        spec = importlib.util.spec_from_file_location("ref_" + script[:4], os.path.join(ref_import.REFERENCE_ROOT, script))
        mod = importlib.util.module_from_spec(spec)
        try:
            spec.loader.exec_module(mod)
            mod.parse_option()
        except SystemExit:
            pass
    finally:
        argparse.ArgumentParser.parse_args = orig
    out[script] = captured
json.dump(out, open(os.path.join(HERE, "cli_flags.json"), "w"), indent=1, sort_keys=True, default=str)
print({k: len(v) for k, v in out.items()})
