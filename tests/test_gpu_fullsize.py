"""GPU, BASELINE.json configs 2, 3 and 5 at FULL size (config 3: 65 x 5 Mbp pan-genome-like index, 142 M columns, 100 M reads;
config 5: k = 63 without streaming support, 10 M reads; below).  Config 2 at full size (10 M x 150 bp reads, 12.8 M-column index): size-independent
properties through the device-buffer entry points -- the path-order kernel, the certificate kernel on the
blocks and the reference-order kernel agree bit for bit (three independent code paths), streaming == per-k-mer search (tests/test_large.hh:104-115),
hits are valid columns, misses are exactly the k-mers touching a mismatch or nothing else can explain,
the run is deterministic, and a sample equals the oracle."""
import os
import sys

import numpy as np
import pytest

from oracle import OracleIndex
from sbwt_amd import capi, hostlib, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config2_full_size_properties(gpu):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    k, n_reads, L = 30, 10_000_000, 150
    m = L - k + 1
    genomes = synth.coli3_like(5_000_000)
    bits = hostlib.build_bits([g.tobytes() for g in genomes], k, False, True, n_threads=bench.effective_cores())
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                            bits.n_kmers, 8)
    dev = torch.device("cuda:0")
    d_bases = bench.gpu_reads(genomes, n_reads, 4242, dev)
    d_roff = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * L
    d_ooff = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * m
    wsb = capi.search_workspace_bytes(d_bases.numel())
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run(streaming, variant):
        out = torch.full((n_reads * m,), -9, dtype=torch.int64, device=dev)
        capi.set_tuning("search_variant", variant)
        try:
            idx.streaming_search_dev(d_bases.data_ptr(), d_bases.numel(), d_roff.data_ptr(), n_reads, out.data_ptr(),
                                     d_ooff.data_ptr(), d_ws.data_ptr(), wsb, st, streaming)
            torch.cuda.synchronize()
            assert idx.workspace_status(d_ws.data_ptr(), st) == 0
        finally:
            capi.set_tuning("search_variant", -1)
        return out

    a = run(True, 5)                      # the product path: the fused route (k_search_fused, sbwt_search_fused.hip)
    assert idx.workspace_stats(d_ws.data_ptr(), st)[4] > 0
    assert torch.equal(a, run(False, 5))  # the fused route under SBWT::search (internal streaming; upper-case input)
    assert torch.equal(a, run(True, 4))   # the general path kernel over all reads (two passes: k_encode + k_search_cert<PATH>)
    assert torch.equal(a, run(True, 1))   # certificates on the blocks only
    assert torch.equal(a, run(True, 0))   # the reference's order of searches
    assert torch.equal(a, run(False, 4))  # per-k-mer search loop (internal streaming) == streaming (upper-case input)
    assert torch.equal(a, run(False, 1))  # per-k-mer search loop == streaming (upper-case input)
    assert torch.equal(a, run(True, 1))   # deterministic
    assert int(a.min()) == -1 and int(a.max()) < bits.n_nodes
    hit = (a >= 0).double().mean().item()
    assert 0.70 < hit < 0.78              # 0.99^30 = 0.74 of the k-mers avoid every substituted base
    # every hit is a k-mer column, never a dummy: rank structure consistency via a second lookup path
    sample = 5000
    h_bases = d_bases[: sample * L].cpu().numpy()
    orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                                bits.n_kmers, 8)
    want, _ = orc.batch_search(h_bases, np.arange(sample + 1, dtype=np.int64) * L,
                               np.arange(sample + 1, dtype=np.int64) * m, 4)
    assert np.array_equal(a[: sample * m].cpu().numpy(), want)
    # checksum of checksums, stable across runs and kernels
    w = torch.arange(1, m + 1, device=dev, dtype=torch.int64)
    per_read = (a.view(n_reads, m) * w).sum(dim=1)
    assert int((per_read * torch.arange(1, n_reads + 1, device=dev)).sum().item()) == \
        int(((run(True, 0).view(n_reads, m) * w).sum(dim=1) * torch.arange(1, n_reads + 1, device=dev)).sum().item())


class _Batch:
    """Reads of one length resident in HBM + one result buffer that every route writes in turn (config 3's results are
    96 GB: routes are compared through per-read checksums and an exact copy of the first reads, not through two buffers)."""

    def __init__(self, idx, genomes, n_reads, L, k, seed):
        import torch
        import bench
        self.torch, self.idx, self.n_reads, self.L, self.m = torch, idx, n_reads, L, L - k + 1
        self.dev = torch.device("cuda:0")
        self.d_bases = bench.gpu_reads(genomes, n_reads, seed, self.dev)
        self.d_roff = torch.arange(n_reads + 1, dtype=torch.int64, device=self.dev) * L
        self.d_ooff = torch.arange(n_reads + 1, dtype=torch.int64, device=self.dev) * self.m
        self.wsb = capi.search_workspace_bytes(self.d_bases.numel())
        self.d_ws = torch.empty(self.wsb, dtype=torch.uint8, device=self.dev)
        self.out = torch.empty(n_reads * self.m, dtype=torch.int64, device=self.dev)
        self.st = torch.cuda.current_stream().cuda_stream

    def run(self, streaming, variant, n_reads=None):
        torch = self.torch
        n = self.n_reads if n_reads is None else n_reads
        self.out[: n * self.m].fill_(-9)
        capi.set_tuning("search_variant", variant)
        try:
            self.idx.streaming_search_dev(self.d_bases.data_ptr(), n * self.L, self.d_roff.data_ptr(), n, self.out.data_ptr(),
                                          self.d_ooff.data_ptr(), self.d_ws.data_ptr(), self.wsb, self.st, streaming)
            torch.cuda.synchronize()
            assert self.idx.workspace_status(self.d_ws.data_ptr(), self.st) == 0
        finally:
            capi.set_tuning("search_variant", -1)
        return self.out[: n * self.m]

    def checksums(self, n_reads=None):
        """per read: sum of (position + 1) * result, and the number of hits -- in chunks (the temporaries of one pass
        over 1.2e10 results would not fit beside them)"""
        torch = self.torch
        n = self.n_reads if n_reads is None else n_reads
        w = torch.arange(1, self.m + 1, device=self.dev, dtype=torch.int64)
        cs = torch.empty(n, dtype=torch.int64, device=self.dev)
        hits = 0
        lo_v, hi_v = 0, -1
        step = 2_000_000
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            v = self.out[lo * self.m: hi * self.m].view(hi - lo, self.m)
            cs[lo:hi] = (v * w).sum(dim=1)
            hits += int((v >= 0).sum().item())
            lo_v, hi_v = min(lo_v, int(v.min().item())), max(hi_v, int(v.max().item()))
        return cs, hits, lo_v, hi_v


def test_config3_full_size_pangenome(gpu):
    """BASELINE config 3 at its full size: 1 + 64 genomes x 5 Mbp at 2 % divergence (synth.pan_like), k = 31, streaming
    support, 100 M x 150 bp reads = 1.2e10 k-mers.  The fused route, the two-pass path kernel and the certificates on the
    blocks give the same results (per-read checksums over all reads, bit for bit on the first 10 M reads, where the
    reference-order kernel joins in), streaming == per-k-mer search (tests/test_large.hh:104-115), a 5 000-read sample
    equals the oracle, the hit rate is what 1 % substitutions leave."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    k, n_reads, L = 31, 100_000_000, 150
    genomes = synth.pan_like(64, 5_000_000)
    bits = capi.build_bits_gpu([g.tobytes() for g in genomes], k, False, True)
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                            bits.n_kmers, 8)
    assert idx.n_nodes > 100_000_000 and idx.image_level == 0 and idx.default_search_variant == 5
    B = _Batch(idx, genomes, n_reads, L, k, 4243)
    m, head = B.m, 10_000_000
    a = B.run(True, 5)                                   # the product path
    stats = idx.workspace_stats(B.d_ws.data_ptr(), B.st)
    assert stats[4] > 0.5 * n_reads * m, "the path-order kernel answered too few k-mers along path runs"
    cs5, hits, vmin, vmax = B.checksums()
    assert vmin == -1 and vmax < bits.n_nodes
    assert 0.68 < hits / (n_reads * m) < 0.80            # 0.99^31 = 0.73 of the k-mers avoid every substituted base
    a_head = a[: head * m].clone()
    # a 5 000-read sample against the oracle
    sample = 5000
    orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                                bits.n_kmers, 8)
    want, _ = orc.batch_search(B.d_bases[: sample * L].cpu().numpy(), np.arange(sample + 1, dtype=np.int64) * L,
                               np.arange(sample + 1, dtype=np.int64) * m, bench.effective_cores())
    assert np.array_equal(a_head[: sample * m].cpu().numpy(), want)
    del orc
    for streaming, variant in ((True, 4), (True, 1), (False, 5)):
        b = B.run(streaming, variant)
        cs, h2, _, _ = B.checksums()
        assert h2 == hits, (streaming, variant)
        assert torch.equal(cs, cs5), (streaming, variant)
        assert torch.equal(b[: head * m], a_head), (streaming, variant)
    # the reference's own order of searches, on the first 10 M reads
    assert torch.equal(B.run(True, 0, head), a_head)


def test_config5_full_size_k63_no_streaming_support(gpu):
    """BASELINE config 5 at its full size: coli3-like genomes, k = 63, index WITHOUT streaming support, 10 M x 150 bp
    reads, SBWT::search of every k-mer (run_queries_not_streaming, sbwt_search.cpp:67-91).  All four routes bit for bit,
    a sample against the oracle, streaming_search refused (SBWT.hh:546-547)."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    k, n_reads, L = 63, 10_000_000, 150
    genomes = synth.coli3_like(5_000_000)
    bits = capi.build_bits_gpu([g.tobytes() for g in genomes], k, False, False)
    assert bits.ssup is None
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], None, bits.n_nodes, k, bits.n_kmers, 8)
    assert not idx.has_streaming_support and idx.image_level == 0
    B = _Batch(idx, genomes, n_reads, L, k, 4245)
    m = B.m
    with pytest.raises(capi.SbwtGpuError):
        B.run(True, 5)
    a = B.run(False, 5).clone()
    assert idx.workspace_stats(B.d_ws.data_ptr(), B.st)[4] > 0
    for variant in (4, 1, 0):
        assert torch.equal(B.run(False, variant), a), variant
    assert int(a.min()) == -1 and int(a.max()) < bits.n_nodes
    hit = (a >= 0).double().mean().item()
    assert 0.48 < hit < 0.58                             # 0.99^63 = 0.53
    sample = 3000
    orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], None, bits.n_nodes, k, bits.n_kmers, 8)
    want, _ = orc.batch_search(B.d_bases[: sample * L].cpu().numpy(), np.arange(sample + 1, dtype=np.int64) * L,
                               np.arange(sample + 1, dtype=np.int64) * m, bench.effective_cores())
    assert np.array_equal(a[: sample * m].cpu().numpy(), want)
