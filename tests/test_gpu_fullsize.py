"""GPU, BASELINE.json config 2 at FULL size (10 M x 150 bp reads, 12.8 M-column index): size-independent
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
