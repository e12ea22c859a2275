"""Parity corners the earlier suites left out ("weak" 1): NUL bytes inside reads, prefix tables of the index file
with p in 9..12, batches without any read / without any k-mer through every entry point, k = 1 and 2, result ranges with
gaps between them.  Everything against the oracle, through the C ABI."""
import numpy as np
import pytest

from oracle import OracleIndex, print_vector
from sbwt_amd import capi, synth

pytestmark = pytest.mark.gpu


def gpu_index_from_oracle(orc: OracleIndex, precalc_from_file: bool = False) -> capi.Index:
    cols = orc.columns()
    return capi.Index.create(cols[0], cols[1], cols[2], cols[3], orc.ssup_words(), orc.n_nodes, orc.k,
                             orc.n_kmers, orc.precalc_k, orc.precalc() if precalc_from_file else None)


def oracle_batch(orc: OracleIndex, bases, off, streaming=True):
    out = []
    for r in range(len(off) - 1):
        s = bases[off[r]:off[r + 1]].tobytes()
        out.append(orc.streaming_search(s) if streaming else orc.search_all(s))
    return np.concatenate(out) if out else np.zeros(0, np.int64)


@pytest.fixture(scope="module")
def genome_case():
    k = 30
    genomes = [synth.random_genome(150_000, 11)]
    genomes.append(synth.mutate(genomes[0], 0.05, 12))
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 8)
    return genomes, orc


def _dev_call(idx, bases, off, ooff, n_out_alloc, streaming, i32=False, fill=-7):
    """One sbwtgpu_*_dev(_i32) call on device copies of the arguments; returns the whole result array (guards included)."""
    import torch
    dev = torch.device("cuda:0")
    d_b = torch.from_numpy(np.ascontiguousarray(bases)).to(dev) if len(bases) else torch.zeros(1, dtype=torch.uint8, device=dev)
    d_ro = torch.from_numpy(np.ascontiguousarray(off, dtype=np.int64)).to(dev)
    d_oo = torch.from_numpy(np.ascontiguousarray(ooff, dtype=np.int64)).to(dev)
    d_out = torch.full((max(n_out_alloc, 1),), fill, dtype=torch.int32 if i32 else torch.int64, device=dev)
    wsb = capi.search_workspace_bytes(len(bases))
    d_ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    fn = idx.streaming_search_dev_i32 if i32 else idx.streaming_search_dev
    fn(d_b.data_ptr(), len(bases), d_ro.data_ptr(), len(off) - 1, d_out.data_ptr(), d_oo.data_ptr(), d_ws.data_ptr(), wsb, st,
       streaming)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()[:n_out_alloc]


@pytest.mark.parametrize("variant", [5, 4, 1, 0])
def test_nul_bytes_inside_reads(gpu, genome_case, variant):
    # SBWT.hh:544: streaming_search takes (pointer, length) and does not depend on NUL termination; byte 0x00 is just another
    # byte that is not ACGT (-1 for every k-mer that holds it, a full search behind it: SBWT.hh:557-559)
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    bases, off = synth.sample_reads(genomes, 3000, 150, 0.005, 77)
    rng = np.random.default_rng(7)
    bases = bases.copy()
    where = rng.choice(len(bases), size=500, replace=False)
    bases[where] = 0
    bases[off[5]:off[6]] = 0                               # a read of NULs only
    bases[off[9]] = 0                                      # first / last byte of a read
    bases[off[11] - 1] = 0
    capi.set_tuning("search_variant", variant)
    try:
        for streaming in (True, False):
            want = oracle_batch(orc, bases, off, streaming)
            got, _ = (idx.streaming_search if streaming else idx.search)(bases, off)
            assert np.array_equal(got, want), streaming
            ooff = capi.out_offsets(off, orc.k)
            assert np.array_equal(_dev_call(idx, bases, off, ooff, int(ooff[-1]), streaming), want)
        txt, _ = idx.search_text(bases, off, True)
        assert txt == b"".join(print_vector(orc.streaming_search(bases[off[r]:off[r + 1]].tobytes())) for r in range(len(off) - 1))
    finally:
        capi.set_tuning("search_variant", -1)


@pytest.mark.parametrize("p", [9, 10, 12])
def test_file_prefix_tables_with_p_9_to_12(gpu, p):
    # SBWT.hh:616-645 (Q12: p <= 15 is what the reference's 32-bit shift defines): the table that comes with the index file,
    # uploaded as it is, and the one the device computes, both equal the oracle's do_kmer_prefix_precalc; searches through an
    # image whose file table is deeper / shallower than the device's own
    k = 31
    genomes = [synth.random_genome(40_000, 31)]
    genomes.append(synth.mutate(genomes[0], 0.03, 32))
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, p)
    want_tab = orc.precalc()
    assert want_tab.shape == (4 ** p, 2)
    bases, off = synth.sample_reads(genomes, 600, 120, 0.01, 5 + p)
    bases = synth.inject(bases, 20, ord("N"), 3)
    want_s, want_n = oracle_batch(orc, bases, off, True), oracle_batch(orc, bases, off, False)
    for from_file in (True, False):
        idx = gpu_index_from_oracle(orc, precalc_from_file=from_file)
        assert idx.precalc_k == p and idx.device_precalc_k >= p
        assert np.array_equal(idx.get_precalc(), want_tab), from_file
        for variant in (-1, 1, 0):
            capi.set_tuning("search_variant", variant)
            try:
                assert np.array_equal(idx.streaming_search(bases, off)[0], want_s), (from_file, variant)
                assert np.array_equal(idx.search(bases, off)[0], want_n), (from_file, variant)
            finally:
                capi.set_tuning("search_variant", -1)
        del idx


def test_file_table_shallower_than_the_device_table(gpu):
    # an index file with p = 5 whose image gets a deeper dense table of its own: get_precalc still returns the FILE's table
    # (SBWT::get_precalc), searches use the device's
    k = 31
    genomes = [synth.random_genome(1_500_000, 41)]
    bits = capi.build_bits_gpu([g.tobytes() for g in genomes], k, False, True)
    orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k, bits.n_kmers, 5)
    idx = gpu_index_from_oracle(orc, precalc_from_file=True)
    assert idx.precalc_k == 5 and idx.device_precalc_k > 5, idx.device_precalc_k
    assert np.array_equal(idx.get_precalc(), orc.precalc())
    bases, off = synth.sample_reads(genomes, 800, 150, 0.01, 3)
    for variant in (-1, 1, 0):
        capi.set_tuning("search_variant", variant)
        try:
            assert np.array_equal(idx.streaming_search(bases, off)[0], oracle_batch(orc, bases, off, True)), variant
        finally:
            capi.set_tuning("search_variant", -1)


@pytest.mark.parametrize("k,streaming_support", [(30, True), (63, False)])
def test_batches_without_reads_or_without_kmers(gpu, k, streaming_support):
    # n_reads == 0, and batches whose reads are all shorter than k (Q9: an empty result per read, an empty line in the text),
    # through every _batch / _dev / _i32 / _text entry point; nothing is written anywhere
    genomes = [synth.random_genome(30_000, 51)]
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, streaming_support, False, 4)
    idx = gpu_index_from_oracle(orc)
    g0 = genomes[0].tobytes()
    cases = [([], "no reads"), ([b""], "one empty read"), ([b"", b"", b""], "empty reads"),
             ([g0[:k - 1], b"A", g0[5:5 + k - 1], b"", b"ACGTN"[: min(5, k - 1)]], "all shorter than k")]
    for reads, what in cases:
        bases, off = capi.concat_reads(reads)
        n = len(reads)
        assert len(off) == n + 1
        ooff = capi.out_offsets(off, k)
        assert int(ooff[-1]) == 0
        for streaming in ((True, False) if streaming_support else (False,)):
            got, oo = (idx.streaming_search if streaming else idx.search)(bases, off)
            assert len(got) == 0 and np.array_equal(oo, ooff), what
            got32, _ = idx.search_i32(bases, off, streaming)
            assert len(got32) == 0, what
            txt, nk = idx.search_text(bases, off, streaming)
            assert txt == b"\n" * n and nk == 0, (what, txt)
            for i32 in (False, True):
                guard = _dev_call(idx, bases, off, ooff, 64, streaming, i32=i32, fill=-7)
                assert (guard == -7).all(), (what, streaming, i32)
        if not streaming_support and n > 0:
            # SBWT.hh:546-547: streaming search without the support is an error, also for a batch that holds no k-mer
            with pytest.raises(capi.SbwtGpuError):
                idx.streaming_search(bases, off)


@pytest.mark.parametrize("k", [1, 2])
@pytest.mark.parametrize("precalc", [0, 1])
def test_k_1_and_2(gpu, k, precalc):
    # the smallest k-mers: every table is shallower than usual (precalc <= k), the path order's windows are one or two chars
    if precalc > k:
        pytest.skip("precalc > k")
    seqs = [b"ACGTTGCAACGGT", b"TTTTACG", b"G"]
    for ssup in (True, False):
        orc = OracleIndex.build(seqs, k, ssup, False, precalc)
        idx = gpu_index_from_oracle(orc)
        assert idx.n_nodes == orc.n_nodes and idx.C == orc.C
        reads = [b"ACGTTGCAACGGT", b"A", b"C", b"G", b"T", b"N", b"", b"AC", b"CA", b"NN", b"ANC", b"acgt", b"ACGTNACGT" * 20,
                 bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(k).integers(0, 4, size=400)])]
        bases, off = capi.concat_reads(reads)
        for variant in (-1, 1, 0):
            capi.set_tuning("search_variant", variant)
            try:
                if ssup:
                    assert np.array_equal(idx.streaming_search(bases, off)[0], oracle_batch(orc, bases, off, True)), (ssup, variant)
                assert np.array_equal(idx.search(bases, off)[0], oracle_batch(orc, bases, off, False)), (ssup, variant)
            finally:
                capi.set_tuning("search_variant", -1)
        # rank at every position, every symbol
        pos = np.repeat(np.arange(orc.n_nodes + 1), 5)
        sym = np.tile(np.frombuffer(b"ACGTN", dtype=np.uint8), orc.n_nodes + 1)
        assert np.array_equal(idx.rank(pos, sym), np.array([orc.rank(int(a), bytes([int(c)])) for a, c in zip(pos, sym)]))


@pytest.mark.parametrize("variant", [5, 4, 1, 0])
@pytest.mark.parametrize("i32", [False, True])
def test_result_ranges_with_gaps_between_them(gpu, genome_case, variant, i32):
    # the _dev calls take any out_off: result ranges with gaps between them (a caller that pads every read's results to a
    # line, or interleaves two batches) -- the slots between the ranges keep their guard values.  Reads of one length (the
    # fused kernel's stride arithmetic must notice that out_off is not one stride) and ragged ones.
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    k = orc.k
    rng = np.random.default_rng(3)
    batches = [synth.sample_reads(genomes, 3000, 150, 0.01, 61), synth.ragged_reads(genomes, 3000, 20, 200, 0.01, 62)]
    b3, o3 = synth.sample_reads(genomes, 2000, 150, 0.01, 63)
    batches.append((synth.inject(b3, 100, ord("N"), 1), o3))
    capi.set_tuning("search_variant", variant)
    capi.set_tuning("poison_results", 0)       # (the suite's poison fills out[out_off[0] .. out_off[n]), gaps included)
    try:
        for bi, (bases, off) in enumerate(batches):
            m = np.maximum(np.diff(off) - k + 1, 0)
            for gaps in ("random", "pad16", "one stride with a gap"):
                if gaps == "random":
                    gap = rng.integers(0, 9, size=len(m))
                elif gaps == "pad16":
                    gap = (-m) % 16
                else:
                    gap = np.full(len(m), 7)
                starts = np.concatenate([[3], 3 + np.cumsum(m + gap)])[:-1]
                ooff = np.concatenate([starts, [starts[-1] + m[-1]]]).astype(np.int64)
                total = int(starts[-1] + m[-1] + gap[-1]) + 5
                for streaming in (True, False):
                    want = oracle_batch(orc, bases, off, streaming)
                    got = _dev_call(idx, bases, off, ooff, total, streaming, i32=i32, fill=-7).astype(np.int64)
                    mask = np.zeros(total, dtype=bool)
                    flat = np.empty(int(m.sum()), dtype=np.int64)
                    at = 0
                    for r in range(len(m)):
                        mask[starts[r]:starts[r] + m[r]] = True
                        flat[at:at + m[r]] = got[starts[r]:starts[r] + m[r]]
                        at += m[r]
                    assert np.array_equal(flat, want), (bi, gaps, streaming)
                    assert (got[~mask] == -7).all(), (bi, gaps, streaming)
    finally:
        capi.set_tuning("search_variant", -1)
        capi.set_tuning("poison_results", 1)
