"""Sharded all-pairs video x music retrieval scoring (reference test-MaDe.py:386-413), one process per GPU.

The reference concatenates the embeddings of the whole split on one host and scores all pairs on the CPU.
Every (video, music) pair is independent given the two embedding sets, so here each rank keeps the VIDEO rows
it encoded, the MUSIC side (per-segment embeddings, masks, pooled music vectors: 393 MB f32 at 4k x 96 x 256) is
exchanged once with an all-gather (RCCL over xGMI when the backend is "nccl"), and each rank scores its own
[N_v/W, N_m] row block with the local kernels.  Rows are complete on their owner, so ranking / top-k stay local;
`gather_rows=True` additionally collects the full matrix on every rank the way the reference returns it.

`score_fn(video, seg_all, mask_all, music_all) -> [n_v_local, N_m]` is the compute backend: the HIP engine in
production (`MadeEngine.retrieval_sim_matrix`); tests inject the CPU oracle to check the sharding logic under gloo.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def _all_gather_ragged(t: Tensor, group=None) -> Tuple[Tensor, List[int]]:
    """All-gather along dim 0 when ranks hold different row counts: pad to the largest shard, gather, trim."""
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], device=t.device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    if t.shape[0] < mx:
        pad = torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        t = torch.cat([t, pad], dim=0)
    out = torch.empty((world * mx,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    if all(c == mx for c in counts):
        return out, counts
    return torch.cat([out[r * mx:r * mx + c] for r, c in enumerate(counts)], dim=0), counts


def shard_rows(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row partition [lo, hi) of n rows over `world` ranks (first ranks take the remainder)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedRetrieval:
    def __init__(self, score_fn: Callable[[Tensor, Tensor, Tensor, Tensor], Tensor], group=None):
        self.score_fn = score_fn
        self.group = group

    def gather_music_side(self, seg_local: Tensor, mask_local: Tensor, music_local: Tensor):
        """One exchange step: every rank ends up with all tracks, in rank order."""
        if not (dist.is_available() and dist.is_initialized()):
            return seg_local, mask_local, music_local
        seg_all, _ = _all_gather_ragged(seg_local, self.group)
        mask_all, _ = _all_gather_ragged(mask_local, self.group)
        music_all, _ = _all_gather_ragged(music_local, self.group)
        return seg_all, mask_all, music_all

    def sim_rows(self, video_local: Tensor, seg_local: Tensor, mask_local: Tensor, music_local: Tensor) -> Tensor:
        """This rank's complete rows of the similarity matrix: [n_v_local, N_m]."""
        seg_all, mask_all, music_all = self.gather_music_side(seg_local, mask_local, music_local)
        return self.score_fn(video_local, seg_all, mask_all, music_all)

    def sim_matrix(self, video_local: Tensor, seg_local: Tensor, mask_local: Tensor, music_local: Tensor,
                   gather_rows: bool = True) -> Tensor:
        rows = self.sim_rows(video_local, seg_local, mask_local, music_local)
        if not gather_rows or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return rows
        full, _ = _all_gather_ragged(rows, self.group)
        return full
