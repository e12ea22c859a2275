"""GPU, BASELINE.json config 3 (pan-genome-like, k=31, plain-matrix with streaming support) at a size the oracle can
sample: a star pan-genome of 1 + 16 genomes (synth.pan_like, 2 % divergence) and 2 M x 150 bp reads with 1 %
substitutions.  The three search kernels (path order, certificates on the blocks, reference order) must agree bit
for bit on the full batch, with transitions running on along their quoted steps or not, a 5 000-read sample must
equal the oracle, and the per-k-mer search loop must equal streaming_search (tests/test_large.hh:104-115)."""
import os
import sys

import numpy as np
import pytest

from oracle import OracleIndex
from sbwt_amd import capi, hostlib, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config3_pangenome_k31(gpu):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    k, n_reads, L = 31, 2_000_000, 150
    m = L - k + 1
    genomes = synth.pan_like(16, 2_000_000)
    bits = hostlib.build_bits([g.tobytes() for g in genomes], k, False, True, n_threads=bench.effective_cores())
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                            bits.n_kmers, 8)
    assert idx.n_nodes > 10_000_000 and idx.has_streaming_support
    dev = torch.device("cuda:0")

    d_bases = bench.gpu_reads(genomes, n_reads, 31337, dev)
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

    a = run(True, 5)                          # the product path: the fused route
    stats = idx.workspace_stats(d_ws.data_ptr(), st)
    assert stats[4] > 0, "the path-order kernel did not run"      # k-mers answered along path runs
    assert torch.equal(a, run(True, 4))       # the general path kernel over all reads (two passes)
    assert torch.equal(a, run(True, 1))       # certificates on the blocks only
    assert torch.equal(a, run(True, 0))       # the reference's order of searches
    assert torch.equal(a, run(False, 5))      # per-k-mer search loop == streaming (upper-case input)
    assert int(a.min()) == -1 and int(a.max()) < bits.n_nodes
    hit = (a >= 0).double().mean().item()
    assert 0.68 < hit < 0.80                  # 0.99^31 = 0.73 of the k-mers avoid every substituted base
    sample = 5000
    h_bases = d_bases[: sample * L].cpu().numpy()
    orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                                bits.n_kmers, 8)
    want, _ = orc.batch_search(h_bases, np.arange(sample + 1, dtype=np.int64) * L,
                               np.arange(sample + 1, dtype=np.int64) * m, 4)
    assert np.array_equal(a[: sample * m].cpu().numpy(), want)

    # "path_lookahead": the paths of the default index follow the core of the pan-genome (they look 8 steps ahead and
    # behind for branch points when they choose a successor); an index whose paths choose blindly answers with the same
    # bits, but its reads leave their paths several times as often
    run(True, 4)
    trans_core = idx.workspace_stats(d_ws.data_ptr(), st)[0]
    capi.set_tuning("path_lookahead", 0)
    try:
        idx0 = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                                 bits.n_kmers, 8)
    finally:
        capi.set_tuning("path_lookahead", 8)
    idx_core, idx = idx, idx0
    assert torch.equal(a, run(True, 4))
    trans_blind = idx.workspace_stats(d_ws.data_ptr(), st)[0]
    assert torch.equal(a, run(True, 5))
    idx = idx_core
    assert trans_blind > 2 * trans_core, (trans_blind, trans_core)
