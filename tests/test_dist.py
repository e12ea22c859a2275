"""CPU, world_size 2 over gloo: the multi-GPU plumbing of bench.py (index image broadcast, read
sharding, max-over-ranks timing) -- the same code path runs over RCCL on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from sbwt_amd import dist as sdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    header, blob = None, None
    if rank == 0:
        header = bytes(range(200)) + b"SBWTGPU1"
        blob = torch.arange(100_003, dtype=torch.int64).to(torch.uint8)
    h, b = sdist.broadcast_blob(header, blob, dev, src=0)
    ok = h == bytes(range(200)) + b"SBWTGPU1" and torch.equal(b, torch.arange(100_003, dtype=torch.int64).to(torch.uint8))
    # ragged reads: contiguous shards balanced by bases cover everything exactly once, in order
    lens = np.random.default_rng(5).integers(0, 400, size=1001)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    lo, hi = sdist.contiguous_shard(off, rank, world)
    mine = int(off[hi] - off[lo])
    tot = sdist.sum_over_ranks([mine, hi - lo], dev)
    mx = sdist.max_over_ranks(1.0 + rank, dev)
    ok = ok and sdist.gather_floats(10.0 + rank, dev) == [10.0, 11.0]          # per-rank values in rank order
    q.put((rank, ok, lo, hi, mine, tot, mx, int(off[-1])))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ok0, lo0, hi0, m0, tot0, mx0, total), (r1, ok1, lo1, hi1, m1, tot1, mx1, _) = res
    assert ok0 and ok1
    assert lo0 == 0 and hi0 == lo1 and hi1 == 1001
    assert tot0 == tot1 == [float(total), 1001.0]
    assert mx0 == mx1 == 2.0
    assert abs(m0 - m1) < 0.05 * total          # balanced by bases


def test_contiguous_shard_edge_cases():
    sys.path.insert(0, ROOT)
    from sbwt_amd.dist import contiguous_shard
    off = np.array([0, 10, 20, 30, 40], dtype=np.int64)
    assert [contiguous_shard(off, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 4)]
    assert contiguous_shard(off, 0, 1) == (0, 4)
    off = np.array([0], dtype=np.int64)
    assert contiguous_shard(off, 1, 2) == (0, 0)
    off = np.array([0, 0, 0, 100], dtype=np.int64)       # empty reads in front
    parts = [contiguous_shard(off, r, 2) for r in range(2)]
    assert parts[0][0] == 0 and parts[0][1] == parts[1][0] and parts[1][1] == 3


class _Bits:
    def __init__(self, cols, ssup, n_nodes, n_kmers, k):
        self.cols, self.ssup, self.n_nodes, self.n_kmers, self.k = cols, ssup, n_nodes, n_kmers, k


def _fallback_worker(rank, world, port, q):
    # the control plane of bench.py's replication fall-back: a second (gloo) group beside the data
    # group, agreement on "did any rank fail", the five bit vectors as one CPU tensor
    sys.path.insert(0, ROOT)
    from sbwt_amd import dist as sdist
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=60))
    none_failed = sdist.count_failures(False, ctl)
    one_failed = sdist.count_failures(rank == world - 1, ctl)
    # the pre-flight of the image broadcast: header, length and the receiving tensor are agreed over the control
    # group, and a rank that fails on its own is counted BEFORE anyone enters the data collective
    hdr0 = bytes(range(100)) + b"SBWTGPU3"
    blob0 = torch.arange(5003, dtype=torch.int64).to(torch.uint8)
    for fail_last in (False, True):
        h, out, nf = sdist.blob_preflight(hdr0 if rank == 0 else None, blob0 if rank == 0 else None, torch.device("cpu"), ctl,
                                          src=0, fail=fail_last and rank == world - 1)
        assert h == hdr0 and nf == (1 if fail_last else 0)
        if not fail_last:
            assert out.numel() == 5003 and out.dtype == torch.uint8
            dist.broadcast(out, src=0)                 # the data collective, entered by every rank or by none
            assert torch.equal(out, blob0)
    n_nodes = 1000 * 64 + 17
    nw = (n_nodes + 63) // 64
    rng = np.random.default_rng(11)
    cols = [rng.integers(0, 2**63, size=nw, dtype=np.uint64) for _ in range(4)]
    ssup = rng.integers(0, 2**63, size=nw, dtype=np.uint64)
    for with_ssup in (True, False):
        src = _Bits(cols, ssup if with_ssup else None, n_nodes, 12345, 31) if rank == 0 else None
        got = sdist.broadcast_bits_cpu(src, 31, ctl, _Bits)
        ok = (got.n_nodes, got.n_kmers, got.k) == (n_nodes, 12345, 31) and all(np.array_equal(a, b) for a, b in zip(got.cols, cols))
        ok = ok and ((got.ssup is None) if not with_ssup else np.array_equal(got.ssup, ssup))
        q.put((rank, with_ssup, ok, none_failed, one_failed))
    dist.barrier()
    dist.destroy_process_group()


def test_replication_fallback_control_plane_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(res) == 4
    for rank, with_ssup, ok, none_failed, one_failed in res:
        assert ok, (rank, with_ssup)
        assert none_failed == 0 and one_failed == 1
