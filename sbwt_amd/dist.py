"""Multi-GPU plumbing: one process per GPU, reads sharded, read-only index replicated.

The only collective on the path is the one-time broadcast of the device image at load time
(RCCL over xGMI on GPUs; the same code runs over gloo on CPU tensors in the tests).  There are no
per-step collectives: every rank searches its own contiguous range of reads.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def contiguous_shard(read_off: np.ndarray, rank: int, world: int) -> Tuple[int, int]:
    """Reads [lo, hi) of rank `rank`: contiguous ranges (output order is preserved by concatenating
    ranks in order) balanced by total bases rather than by read count."""
    n = len(read_off) - 1
    if world <= 1 or n == 0:
        return 0, n
    total = int(read_off[-1] - read_off[0])
    targets = read_off[0] + (np.arange(world + 1, dtype=np.float64) * total / world)
    cuts = np.searchsorted(read_off, targets, side="left").astype(np.int64)
    cuts[0], cuts[-1] = 0, n
    cuts = np.maximum.accumulate(np.clip(cuts, 0, n))
    return int(cuts[rank]), int(cuts[rank + 1])


def broadcast_blob(header: Optional[bytes], blob: Optional[torch.Tensor], device: torch.device, src: int = 0
                   ) -> Tuple[bytes, torch.Tensor]:
    """Replicates (header bytes, device-image tensor) from rank `src` to every rank.

    `blob` is a uint8 tensor on `device` on the source rank (a view of the index's device image);
    the other ranks receive a freshly allocated tensor.  One broadcast of 16 header bytes worth of
    metadata + one broadcast of the image; nothing else is ever exchanged."""
    rank = dist.get_rank()
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        meta[0], meta[1] = len(header), blob.numel()
    dist.broadcast(meta, src=src)
    hlen, blen = int(meta[0].item()), int(meta[1].item())
    hdr = torch.zeros(hlen, dtype=torch.uint8, device=device)
    if rank == src:
        hdr.copy_(torch.frombuffer(bytearray(header), dtype=torch.uint8))
        out = blob
    else:
        out = torch.empty(blen, dtype=torch.uint8, device=device)
    dist.broadcast(hdr, src=src)
    dist.broadcast(out, src=src)
    return bytes(hdr.cpu().numpy().tobytes()), out


def max_over_ranks(value: float, device: torch.device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device: torch.device):
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def gather_floats(value: float, device: torch.device):
    """value of every rank, in rank order (on every rank)."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
