"""Packed feature store for the pre-extracted CLIP-ViT frame / AST segment features.

The reference keeps every clip as two tiny files, `<root>/<kind>_feature/<id>.pt` and `<root>/<kind>_mask/<id>.pt`
(reference train-MaDe.py:162-169, dataloaders/dataloader_MGSV_EC_feature.py:57-67) and `torch.load`s four of them per sample
with 32 worker processes.  `pack()` turns one such directory pair into ONE memory-mappable file:

    header  (64 bytes)  magic "MADEFS01", n, T, D, dtype code (0 = f32, 1 = bf16), offsets of the sections
    ids     n fixed-width byte strings, sorted (binary search; the original files stay readable, nothing is deleted)
    masks   [n, T] uint8
    data    [n, T, D] f32 or bf16 (padded rows zero, as the dataset emits them after its masked_fill)

`PackedFeatures` memory-maps it; `gather(ids, out_feats, out_mask)` assembles a batch straight into (pinned) host buffers with
one memcpy per sample and no Python-side tensor construction, so the H2D copy can run with `non_blocking=True`.  bf16 storage
halves the bytes read per batch (the model's GEMMs consume bf16 anyway; f32 storage is bit-exact with the `.pt` files).
"""
from __future__ import annotations

import os
import struct
from typing import Iterable, List, Sequence, Tuple

import numpy as np
import torch

MAGIC = b"MADEFS01"
HEADER = struct.Struct("<8sQQQQQQQ")           # magic, n, T, D, dtype, id_width, off_masks, off_data  (64 bytes)


def pack(root: str, kind: str, ids: Iterable[str], out_path: str, dtype: str = "bf16") -> str:
    """root/<kind>_feature/<id>.pt + root/<kind>_mask/<id>.pt  ->  out_path.  kind: "vit" or "ast"."""
    ids = sorted({str(i) for i in ids})
    assert ids, "nothing to pack"
    first = torch.load(os.path.join(root, f"{kind}_feature", f"{ids[0]}.pt"), map_location="cpu")
    T, D = first.shape
    width = max(len(i.encode()) for i in ids)
    code = {"f32": 0, "bf16": 1}[dtype]
    esz = 4 if code == 0 else 2
    off_ids = HEADER.size
    off_masks = off_ids + len(ids) * width
    off_data = (off_masks + len(ids) * T + 63) // 64 * 64
    with open(out_path, "wb") as f:
        f.write(HEADER.pack(MAGIC, len(ids), T, D, code, width, off_masks, off_data))
        for i in ids:
            f.write(i.encode().ljust(width, b"\0"))
        f.truncate(off_data + len(ids) * T * D * esz)
    mm = np.memmap(out_path, mode="r+", dtype=np.uint8)
    masks = mm[off_masks:off_masks + len(ids) * T].reshape(len(ids), T)
    data = mm[off_data:].view(np.float32 if code == 0 else np.uint16).reshape(len(ids), T, D)
    for k, i in enumerate(ids):
        feats = torch.load(os.path.join(root, f"{kind}_feature", f"{i}.pt"), map_location="cpu").float()
        mask = torch.load(os.path.join(root, f"{kind}_mask", f"{i}.pt"), map_location="cpu").float()
        assert tuple(feats.shape) == (T, D) and tuple(mask.shape) == (T,), (i, feats.shape, mask.shape)
        feats = feats.masked_fill(mask.unsqueeze(-1) == 0, 0)
        masks[k] = (mask != 0).numpy().astype(np.uint8)
        data[k] = feats.numpy() if code == 0 else feats.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    mm.flush()
    del mm
    return out_path


class PackedFeatures:
    def __init__(self, path: str):
        self.path = path
        self.mm = np.memmap(path, mode="r", dtype=np.uint8)
        magic, self.n, self.T, self.D, self.code, width, off_masks, off_data = HEADER.unpack(bytes(self.mm[:HEADER.size]))
        if magic != MAGIC:
            raise ValueError(f"{path}: not a packed feature file")
        raw = self.mm[HEADER.size:HEADER.size + self.n * width].reshape(self.n, width)
        self.ids = np.array([bytes(r).rstrip(b"\0").decode() for r in raw])
        self.masks = self.mm[off_masks:off_masks + self.n * self.T].reshape(self.n, self.T)
        self.data = self.mm[off_data:].view(np.float32 if self.code == 0 else np.uint16).reshape(self.n, self.T, self.D)
        self.torch_dtype = torch.float32 if self.code == 0 else torch.bfloat16

    def __len__(self):
        return int(self.n)

    def index(self, ident: str) -> int:
        k = int(np.searchsorted(self.ids, str(ident)))
        if k >= self.n or self.ids[k] != str(ident):
            raise KeyError(ident)
        return k

    def get(self, ident: str) -> Tuple[torch.Tensor, torch.Tensor]:
        """(feats [T, D] float32, mask [T] float32) -- what the reference's dataset item holds for this id."""
        k = self.index(ident)
        raw = torch.from_numpy(np.ascontiguousarray(self.data[k]))
        feats = raw.float() if self.code == 0 else raw.view(torch.int16).view(torch.bfloat16).float()
        return feats, torch.from_numpy(self.masks[k].astype(np.float32))

    def gather(self, ids: Sequence[str], out_feats: torch.Tensor, out_mask: torch.Tensor) -> None:
        """batch assembly into preallocated (ideally pinned) host tensors: out_feats [B, T, D] in the store's dtype
        (`torch_dtype`), out_mask [B, T] float32."""
        assert out_feats.dtype == self.torch_dtype and tuple(out_feats.shape[1:]) == (self.T, self.D)
        dst = out_feats.view(torch.int16).numpy().view(np.uint16) if self.code == 1 else out_feats.numpy()
        msk = out_mask.numpy()
        for b, ident in enumerate(ids):
            k = self.index(ident)
            dst[b] = self.data[k]
            msk[b] = self.masks[k]


class PackedBatcher:
    """Batches of (frame_feats, frame_mask, segment_feats, segment_mask) for lists of (video_id, music_id), double-buffered in
    pinned memory and copied to the GPU asynchronously on a side stream: the loader of the training loop without worker
    processes or per-sample torch.load."""

    def __init__(self, vit: PackedFeatures, ast: PackedFeatures, batch_size: int, device=None, pin: bool = True):
        self.vit, self.ast, self.B = vit, ast, batch_size
        self.device = torch.device(device) if device is not None else None
        pin = pin and torch.cuda.is_available()
        mk = lambda *s, dt: torch.empty(*s, dtype=dt, pin_memory=pin)      # noqa: E731
        self.host = [dict(ff=mk(batch_size, vit.T, vit.D, dt=vit.torch_dtype), fm=mk(batch_size, vit.T, dt=torch.float32),
                          sf=mk(batch_size, ast.T, ast.D, dt=ast.torch_dtype), sm=mk(batch_size, ast.T, dt=torch.float32)) for _ in range(2)]
        self.turn = 0

    def load(self, video_ids: Sequence[str], music_ids: Sequence[str]):
        n = len(video_ids)
        h = self.host[self.turn]
        self.turn ^= 1
        self.vit.gather(video_ids, h["ff"][:n], h["fm"][:n])
        self.ast.gather(music_ids, h["sf"][:n], h["sm"][:n])
        out = {k: v[:n] for k, v in h.items()}
        if self.device is not None:
            out = {k: v.to(self.device, non_blocking=True) for k, v in out.items()}
        return out["ff"], out["fm"], out["sf"], out["sm"]
