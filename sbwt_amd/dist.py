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


def blob_preflight(header: Optional[bytes], blob: Optional[torch.Tensor], device: torch.device, group=None, src: int = 0,
                   fail: bool = False) -> Tuple[Optional[bytes], Optional[torch.Tensor], int]:
    """What every rank must get right BEFORE the data collective is entered, agreed over `group` (the gloo control group
    beside an RCCL data group): the header and the image's length travel as CPU tensors, every other rank allocates its
    receiving tensor, and all ranks count the failures (`fail`: a test hook's, an allocation's).  Returns (header, tensor to
    broadcast into, failures): with failures > 0 NO rank enters the data collective -- a rank that fails on its own can
    therefore never leave the others blocked inside RCCL."""
    rank = dist.get_rank()
    backend = dist.get_backend(group) if group is not None else dist.get_backend()
    cdev = device if backend == "nccl" else torch.device("cpu")
    meta = torch.zeros(2, dtype=torch.int64, device=cdev)
    if rank == src:
        meta[0], meta[1] = len(header), blob.numel()
    dist.broadcast(meta, src=src, group=group)
    hlen, blen = int(meta[0].item()), int(meta[1].item())
    hdr = torch.zeros(hlen, dtype=torch.uint8, device=cdev)
    if rank == src:
        hdr.copy_(torch.frombuffer(bytearray(header), dtype=torch.uint8))
    dist.broadcast(hdr, src=src, group=group)
    out, failed = blob, fail
    if rank != src and not failed:
        try:
            out = torch.empty(blen, dtype=torch.uint8, device=device)
        except Exception:                                                  # noqa: BLE001 (out of memory on this rank only)
            out, failed = None, True
    n_failed = count_failures(failed, group)
    return bytes(hdr.cpu().numpy().tobytes()), out, n_failed


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


def count_failures(failed: bool, group=None) -> int:
    """How many ranks report a failure (all-reduce of one flag over `group`: the gloo control group beside an RCCL data
    group, or the default group).  Every rank gets the same answer, so every rank takes the same fall-back."""
    t = torch.tensor([1 if failed else 0], dtype=torch.int64)
    backend = dist.get_backend(group) if group is not None else dist.get_backend()
    if backend == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def broadcast_bits_cpu(bits, k: int, group, bits_class):
    """The fall-back replication: the five bit vectors of the index (0.7 bytes per column) as ONE uint64 CPU tensor from
    rank 0 over `group` (gloo), for ranks that then derive their own device image.  `bits` is the builder's output on rank 0
    (None elsewhere); returns a `bits_class` (capi.BuiltBits) on every rank."""
    rank = dist.get_rank()
    backend = dist.get_backend(group) if group is not None else dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    meta = torch.zeros(3, dtype=torch.int64, device=dev)
    if rank == 0:
        meta[0], meta[1], meta[2] = bits.n_nodes, bits.n_kmers, 1 if bits.ssup is not None else 0
    dist.broadcast(meta, src=0, group=group)
    n_nodes, n_kmers, has_ssup = int(meta[0]), int(meta[1]), int(meta[2])
    nw = (n_nodes + 63) // 64
    rows = torch.empty((4 + has_ssup) * nw, dtype=torch.int64, device=dev)
    if rank == 0:
        parts = list(bits.cols) + ([bits.ssup] if has_ssup else [])
        rows.copy_(torch.from_numpy(np.concatenate([np.asarray(p)[:nw] for p in parts]).view(np.int64)))
    dist.broadcast(rows, src=0, group=group)
    if rank == 0:
        return bits
    h = rows.cpu().numpy().view(np.uint64)
    return bits_class([h[c * nw:(c + 1) * nw] for c in range(4)], h[4 * nw:5 * nw] if has_ssup else None, n_nodes, n_kmers, k)
