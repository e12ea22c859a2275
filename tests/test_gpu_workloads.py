"""GPU parity on the harder synthetic workloads of DESIGN.md section 7 (one test per generator in sbwt_amd/synth.py):
repeated content in the genome, reads with indels, ragged read lengths, a reverse-complement index queried from both
strands -- every kernel variant against the oracle."""
import numpy as np
import pytest

from oracle import OracleIndex
from sbwt_amd import capi, synth

pytestmark = pytest.mark.gpu


def check(genomes, k, bases, off, revcomp=False):
    seqs = [g.tobytes() for g in genomes]
    bits = capi.build_bits_gpu(seqs, k, revcomp, True)
    orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                                bits.n_kmers, 8)
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                            bits.n_kmers, 8)
    want = [orc.streaming_search(bases[off[r]:off[r + 1]].tobytes()) for r in range(len(off) - 1)]
    want = np.concatenate(want) if want else np.zeros(0, np.int64)
    for variant in (5, 4, 1, 0):
        capi.set_tuning("search_variant", variant)
        try:
            got, _ = idx.streaming_search(bases, off)
            got2, _ = idx.search(bases, off)
        finally:
            capi.set_tuning("search_variant", -1)
        assert np.array_equal(got, want), variant
        assert np.array_equal(got2, want), variant      # upper-case input: search == streaming_search
    return want


def test_repeated_content_genome(gpu):
    g0 = synth.repeat_genome(300_000, 5, 0.08)
    genomes = [g0, synth.mutate(g0, 0.05, 2)]
    bases, off = synth.sample_reads(genomes, 3000, 150, 0.01, 46)
    want = check(genomes, 30, bases, off)
    assert 0.6 < (want >= 0).mean() < 0.85


def test_reads_with_indels(gpu):
    genomes = [synth.random_genome(200_000, 1), None]
    genomes[1] = synth.mutate(genomes[0], 0.05, 2)
    bases, off = synth.indel_reads(genomes, 3000, 150, 0.01, 0.004, 44)
    assert np.diff(off).min() < 150 < np.diff(off).max()
    check(genomes, 30, bases, off)


def test_ragged_read_lengths(gpu):
    genomes = [synth.random_genome(200_000, 1), None]
    genomes[1] = synth.mutate(genomes[0], 0.05, 2)
    bases, off = synth.ragged_reads(genomes, 3000, 20, 250, 0.01, 43)      # some reads shorter than k
    assert (np.diff(off) < 31).any()
    check(genomes, 31, bases, off)


def test_revcomp_index_reads_from_both_strands(gpu):
    """How the tool is normally used (and how the reference's own KAT is built, tests/test_CLI.hh:43): the index holds
    every k-mer and its reverse complement, the reads come from either strand.  Palindromic k-mers merge the two strands'
    paths; the hit rate must be the same as on a forward-only index queried with forward reads."""
    genomes = [synth.random_genome(200_000, 11), None]
    genomes[1] = synth.mutate(genomes[0], 0.05, 12)
    bases, off = synth.both_strand_reads(genomes, 3000, 150, 0.01, 47)
    for k in (30, 31):                       # even k: palindromic k-mers exist; odd k: none
        want = check(genomes, k, bases, off, revcomp=True)
        assert 0.65 < (want >= 0).mean() < 0.82
    # a read and its reverse complement hit the same number of k-mers, mirrored
    fwd, foff = synth.sample_reads(genomes, 500, 150, 0.01, 48)
    rc = np.concatenate([synth.revcomp(fwd[foff[r]:foff[r + 1]]) for r in range(500)])
    bits = capi.build_bits_gpu([g.tobytes() for g in genomes], 31, True, True)
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, 31,
                            bits.n_kmers, 8)
    a, _ = idx.streaming_search(fwd, foff)
    b, _ = idx.streaming_search(rc, foff)
    m = 150 - 31 + 1
    assert np.array_equal((a.reshape(500, m) >= 0), (b.reshape(500, m) >= 0)[:, ::-1])


@pytest.mark.parametrize("k", [30, 63])
def test_substitutions_that_are_another_strains_base(gpu, k):
    """A sequencing error that happens to be ANOTHER strain's base: every short window around it is in the index, so no
    certificate probe proves anything, and only each k-mer's own search says it is absent -- the reference's loop after
    a miss (SBWT.hh:557-559).  The planner's blind mode (and, for k > 31, the 31-base windows) must not change a bit."""
    g0 = synth.random_genome(150_000, 21)
    genomes = [g0, synth.mutate(g0, 0.05, 22)]
    bases, off = synth.variant_reads(genomes, 2500, 150, 23)
    bases = synth.mutate(bases, 0.005, 24)                   # and ordinary errors beside it
    want = check(genomes, k, bases, off)
    assert 0.3 < (want >= 0).mean() < 0.9


def test_reads_without_a_single_kmer_in_shared_stretches(gpu):
    """k = 63, two substitutions 60 bases apart: every k-mer of the read is absent, so there is no k-mer to anchor on; in a
    stretch two strains share every 31-base prefix sits in TWO columns.  The fused kernel aligns such a read through a seed
    of two columns (sbwt_search_fused.hip, CF_SEED2) -- the slowest reads of BASELINE config 5 before it did."""
    g0 = synth.random_genome(150_000, 5)
    genomes = [g0, synth.mutate(g0, 0.02, 6)]                       # long shared stretches between the two strains
    rng = np.random.Generator(np.random.PCG64(77))
    n, L = 3000, 150
    start = rng.integers(0, len(g0) - L, size=n)
    src = rng.integers(0, 2, size=n)
    bases = np.concatenate([genomes[s][a:a + L] for s, a in zip(src, start)]).copy()
    for r in range(n):
        for at in (int(rng.integers(40, 62)), int(rng.integers(100, 122))):
            c = bases[r * L + at]
            bases[r * L + at] = b"ACGT"[(b"ACGT".index(bytes([c])) + 1 + int(rng.integers(0, 3))) % 4]
    off = np.arange(n + 1, dtype=np.int64) * L
    want = check(genomes, 63, bases, off)
    assert (want >= 0).mean() < 0.05


def test_batch_sizes_around_the_pool_and_workgroup_boundaries(gpu):
    """Batches of 1 .. 70 000 reads (a wave's pool is 64 tickets, a workgroup 256 lanes; the launcher picks the number of
    workgroups by the batch's bases, `sbwt_launch_search_fused`) and the same batch on 1, 2 and 1280 workgroups ("debug" >> 8):
    the fused route against the oracle, int64 results."""
    genomes = [synth.random_genome(200_000, 21)]
    genomes.append(synth.mutate(genomes[0], 0.04, 22))
    k = 30
    bits = capi.build_bits_gpu([g.tobytes() for g in genomes], k, False, True)
    orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                                bits.n_kmers, 8)
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                            bits.n_kmers, 8)
    bases_all, off_all = synth.sample_reads(genomes, 70_000, 150, 0.015, 47)
    bases_all = synth.inject(bases_all, 20, ord("N"), 5)
    want_all = np.concatenate([orc.streaming_search(bases_all[off_all[r]:off_all[r + 1]].tobytes()) for r in range(6000)])
    m = 150 - k + 1
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4097, 6000):
        got, _ = idx.streaming_search(bases_all[:off_all[n]], off_all[:n + 1])
        assert np.array_equal(got, want_all[:n * m]), n
    for wgs in (1, 2, 1280):
        capi.set_tuning("debug", wgs << 8)
        try:
            got, _ = idx.streaming_search(bases_all[:off_all[6000]], off_all[:6001])
        finally:
            capi.set_tuning("debug", 0)
        assert np.array_equal(got, want_all), wgs
    # the whole batch: every route agrees (the oracle has vouched for the first 6000 reads)
    ref = None
    for variant in (5, 4):
        capi.set_tuning("search_variant", variant)
        try:
            got, _ = idx.streaming_search(bases_all, off_all)
        finally:
            capi.set_tuning("search_variant", -1)
        assert np.array_equal(got[:6000 * m], want_all), variant
        if ref is None:
            ref = got
        assert np.array_equal(got, ref), variant
