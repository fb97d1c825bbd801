import os
import sys

import pytest

# the suite exercises the library's measurement variants (MADE_LINEAR_TILE, MADE_XPOOL_SIMS_PQ, MADE_DEC_STAGE ...): they are honoured only
# under this switch (mgsv_amd/_lib.py variant_env, csrc/common.h made_variant_env)
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
