"""Free functions the reference's drivers import (reference modules/metrics.py, modules/loss.py) over libmade_hip.so."""
from . import loss, metrics  # noqa: F401
