"""Dropout bookkeeping of the training path.

The reference draws its dropout masks from torch's global generator (reference model/model_Base.py:69-75,
modules/transformer.py:145,177, music_detr/transformer.py:153-162,229-241).  The HIP path cannot (and need not)
reproduce that stream; instead every mask bit is a pure function of (seed, site, element index) so that

  * the forward and the backward kernels regenerate the same mask without storing it,
  * the mask does not depend on tiling, stream order or the number of GPUs,
  * the CPU oracle can rebuild the identical mask and autograd through it (tests/).

keep(seed, site, idx)  <=>  (mix(seed, site, idx) >> 8) >= floor(p * 2^24)       kept values are scaled by 1/(1-p)

mix is two rounds of the murmur3 32-bit finaliser (`made_rng_mix` in include/made_hip.h; restated in numpy below for
host-side checks).  `site` names the dropout module, `idx` is the element's flat index in the site's LOGICAL layout:

  activation sites : idx = row * ncols + col,  row = batch-major token row (b * T + t), col = feature
  attention sites  : idx = ((b * H + h) * Lq + q) * Lk + k
  X-Pool site      : idx = (m * Nv + n) * D + d          (music-major rows, as the reference's [Nm, Nv, D] output)
"""
from __future__ import annotations

import zlib

import numpy as np

P_TEMPORAL = 0.8      # reference model/model_Uni.py:41 (both temporal transformers)
P_XPOOL = 0.3         # reference modules/transformer.py:133


def site_id(name: str) -> int:
    """Stable 32-bit id of a dropout site, e.g. "enc.0.attn", "audio.0.ffn_act", "xa.linear_out"."""
    return zlib.crc32(name.encode()) & 0xFFFFFFFF


def threshold(p: float) -> int:
    """floor(float32(p) * 2^24), exactly (made_drop_threshold in include/made_hip.h)."""
    return int(np.floor(float(np.float32(p)) * (1 << 24)))


def _fmix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h


def rng_mix(seed: int, site: int, idx: np.ndarray) -> np.ndarray:
    """numpy restatement of made_rng_mix (include/made_hip.h): uint32 per element."""
    idx = np.asarray(idx, dtype=np.uint64)
    lo, hi = idx & np.uint64(0xFFFFFFFF), idx >> np.uint64(32)
    s_lo, s_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    k = (s_lo ^ ((np.uint64(site) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF))
         ^ ((s_hi * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)) ^ ((hi * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)))
    k = _fmix32(k)
    return _fmix32(lo ^ k).astype(np.uint32)


def keep_mask(seed: int, site: int, p: float, n: int, offset: int = 0) -> np.ndarray:
    """bool[n]: keep flags of elements offset .. offset+n-1 of a site."""
    h = rng_mix(seed, site, np.arange(offset, offset + n, dtype=np.uint64))
    return (h >> np.uint32(8)) >= np.uint32(threshold(p))
