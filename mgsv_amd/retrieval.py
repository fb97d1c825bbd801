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


def _all_gather_ragged(t: Tensor, group=None, counts: Optional[List[int]] = None) -> Tuple[Tensor, List[int]]:
    """All-gather along dim 0 when ranks hold different row counts: pad to the largest shard, gather, trim.  `counts` (rows per rank,
    e.g. from shard_rows) saves the count exchange and its host synchronisation."""
    world = dist.get_world_size(group)
    if counts is None:
        n = torch.tensor([t.shape[0]], device=t.device, dtype=torch.int64)
        cl = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(cl, n, group=group)
        counts = [int(c.item()) for c in cl]
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
    """`pack_dtype`: dtype the per-segment embeddings travel in (bf16 when the scoring kernel consumes bf16: half the bytes of the one
    large tensor).  The music side travels as ONE packed buffer per track -- [S*D segment embeddings | S mask floats | D pooled
    vector] -- in ONE all-gather; with `counts` (tracks per rank, known from the split's partition: shard_rows) nothing is read back
    to the host.  Unequal shards are scored block by block straight out of the gathered buffer (no compaction copy)."""

    def __init__(self, score_fn: Callable[[Tensor, Tensor, Tensor, Tensor], Tensor], group=None, pack_dtype: Optional[torch.dtype] = None):
        self.score_fn = score_fn
        self.group = group
        self.pack_dtype = pack_dtype

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    @staticmethod
    def _layout(S: int, D: int, esz: int):
        a = S * D * esz                       # segment embeddings
        b = a + S * 4                         # + mask (f32)
        c = b + D * 4                         # + pooled music vector (f32)
        return a, b, (c + 15) // 16 * 16      # 16-byte aligned records (S*D*esz is a multiple of 4 for every esz we send)

    def gather_music_side(self, seg_local: Tensor, mask_local: Tensor, music_local: Tensor, counts: Optional[List[int]] = None):
        """One exchange step.  Returns a list of per-source-rank blocks [(seg [c, S, D], mask [c, S], music [c, D]), ...] in rank order:
        views into the gathered buffer (one block covering everything when all shards have the same size)."""
        if not self._active():
            return [(seg_local, mask_local, music_local)]
        world = dist.get_world_size(self.group)
        n, S, D = seg_local.shape
        sdt = self.pack_dtype or seg_local.dtype
        esz = torch.empty((), dtype=sdt).element_size()
        a, b, rec = self._layout(S, D, esz)
        if counts is None:
            cnt = torch.tensor([n], device=seg_local.device, dtype=torch.int64)
            cl = [torch.zeros_like(cnt) for _ in range(world)]
            dist.all_gather(cl, cnt, group=self.group)
            counts = [int(x.item()) for x in cl]             # (pass `counts` to avoid this synchronisation)
        assert len(counts) == world and counts[dist.get_rank(self.group)] == n
        mx = max(counts)
        if seg_local.is_cuda and sdt in (torch.float32, torch.bfloat16) and seg_local.dtype in (torch.float32, torch.bfloat16) and D % 8 == 0:
            # one launch of the library packs (and converts) the three tensors of every track into its record and zero-fills the padding
            from . import ops
            send = torch.empty(mx, rec, device=seg_local.device, dtype=torch.uint8)
            ops.pack_music_records(seg_local.contiguous(), mask_local, music_local, send, sdt)
        else:                                                               # (host tensors: the two-rank gloo tests of the exchange logic)
            send = torch.zeros(mx, rec, device=seg_local.device, dtype=torch.uint8)
            send[:n, :a].view(sdt).view(n, S, D).copy_(seg_local)          # (converts to the travel dtype)
            send[:n, a:b].view(torch.float32).view(n, S).copy_(mask_local)
            send[:n, b:b + D * 4].view(torch.float32).view(n, D).copy_(music_local)
        recv = torch.empty(world * mx, rec, device=seg_local.device, dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, send, group=self.group)

        def views(lo, cnt_):
            blk = recv[lo:lo + cnt_]
            seg = blk[:, :a].view(sdt).view(cnt_, S, D)                    # row stride = rec bytes: strided, unit inner stride
            mask = blk[:, a:b].view(torch.float32).view(cnt_, S)
            music = blk[:, b:b + D * 4].view(torch.float32).view(cnt_, D)
            return seg, mask, music

        if all(c == mx for c in counts):
            return [views(0, world * mx)]
        return [views(r * mx, c) for r, c in enumerate(counts) if c > 0]

    def sim_rows(self, video_local: Tensor, seg_local: Tensor, mask_local: Tensor, music_local: Tensor,
                 counts: Optional[List[int]] = None) -> Tensor:
        """This rank's complete rows of the similarity matrix: [n_v_local, N_m]."""
        blocks = self.gather_music_side(seg_local, mask_local, music_local, counts)
        if len(blocks) == 1:
            return self.score_fn(video_local, *blocks[0])
        n_m = sum(b[0].shape[0] for b in blocks)
        out = None
        off = 0
        for seg, mask, music in blocks:
            part = self.score_fn(video_local, seg, mask, music)
            if out is None:
                out = torch.empty(video_local.shape[0], n_m, device=part.device, dtype=part.dtype)
            out[:, off:off + seg.shape[0]] = part
            off += seg.shape[0]
        return out

    def sim_matrix(self, video_local: Tensor, seg_local: Tensor, mask_local: Tensor, music_local: Tensor,
                   gather_rows: bool = True, counts: Optional[List[int]] = None, video_counts: Optional[List[int]] = None) -> Tensor:
        rows = self.sim_rows(video_local, seg_local, mask_local, music_local, counts)
        if not gather_rows or not self._active():
            return rows
        full, _ = _all_gather_ragged(rows, self.group, video_counts)
        return full
