"""GPU parity tests proper: the HIP path, called through the C ABI (include/sbwtgpu.h), must be
bit-identical to the oracle on the reference's known-answer vectors, on seeded synthetic inputs,
and on the edge cases the reference tests (empty/short/ragged reads, N and lower-case characters,
no streaming support, pos == n_nodes)."""
import json
import os
import random

import numpy as np
import pytest

from bruteforce import BruteSBWT, kmer_set
from oracle import OracleIndex, print_vector
from sbwt_amd import capi, synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "ref_kats.json")))


def b(s):
    return s.encode()


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


def test_cli_kat_exact_output(gpu):
    kat = KATS["cli_end_to_end"]
    orc = OracleIndex.build([b(s) for s in kat["seqs"]], kat["k"], True, True, kat["precalc"])
    idx = gpu_index_from_oracle(orc)
    assert idx.n_nodes == 87 and idx.C == [1, 25, 43, 59] and idx.precalc_k == 4
    res = idx.streaming_search_reads([b(q) for q in kat["queries"]])
    assert b"".join(print_vector(r) for r in res) == kat["expected_output"].encode()
    res = idx.search_reads([b(q) for q in kat["queries"]])
    assert b"".join(print_vector(r) for r in res) == kat["expected_output"].encode()
    # the device-computed prefix table equals do_kmer_prefix_precalc of the oracle
    assert np.array_equal(idx.get_precalc(), orc.precalc())


@pytest.mark.parametrize("case", KATS["small_cases"]["cases"], ids=lambda c: c["name"])
@pytest.mark.parametrize("precalc", [0, 2])
def test_small_cases_exhaustive(gpu, case, precalc):
    k, seqs = case["k"], case["seqs"]
    orc = OracleIndex.build([b(s) for s in seqs], k, True, False, min(precalc, k))
    idx = gpu_index_from_oracle(orc)
    brute = BruteSBWT(seqs, k)
    kmers = ["".join("ACGT"[(m >> (2 * i)) & 3] for i in range(k)) for m in range(4 ** k)] + ["N" * k]
    got = idx.search_reads([b(x) for x in kmers])
    truth = kmer_set(seqs, k)
    for x, g in zip(kmers, got):
        assert len(g) == 1
        assert g[0] == (brute.rank_of[x] if x in truth else -1)


def test_serialization_strings_streaming_and_N(gpu):
    kat = KATS["serialization_strings"]
    k = kat["k"]
    orc = OracleIndex.build([b(s) for s in kat["seqs"]], k, True, False, kat["precalc"])
    idx = gpu_index_from_oracle(orc, precalc_from_file=True)
    rnd = random.Random(5)
    inputs = kat["seqs"] + ["".join(rnd.choice("ACGT") for _ in range(100)), "N" * 100]
    got = idx.streaming_search_reads([b(s) for s in inputs])
    for s, g in zip(inputs, got):
        assert np.array_equal(g, orc.streaming_search(b(s)))
    assert list(got[-1]) == [-1] * (100 - k + 1)


@pytest.fixture(scope="module")
def genome_case():
    k = 30
    genomes = [synth.random_genome(200_000, 1)]
    genomes.append(synth.mutate(genomes[0], 0.05, 2))
    genomes.append(synth.mutate(genomes[0], 0.05, 3))
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 8)
    return genomes, orc


def test_rank_batch_random_and_edges(gpu, genome_case):
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    rng = np.random.default_rng(3)
    n = orc.n_nodes
    pos = np.concatenate([rng.integers(0, n + 1, size=5000),
                          np.array([0, 1, 63, 64, 65, 127, 128, n - 1, n, (n // 64) * 64, max((n // 64) * 64 - 1, 0)])])
    sym = rng.choice(np.frombuffer(b"ACGTNacgt$", dtype=np.uint8), size=len(pos))
    got = idx.rank(pos, sym)
    want = np.array([orc.rank(int(p), bytes([int(c)])) for p, c in zip(pos, sym)], dtype=np.int64)
    assert np.array_equal(got, want)


def test_streaming_parity_synthetic(gpu, genome_case):
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    bases, off = synth.sample_reads(genomes, 4000, 150, 0.01, 42)
    bases = synth.inject(bases, 60, ord("N"), 7)
    bases = synth.inject(bases, 60, ord("c"), 8)         # lower-case: Q1/Q2 path dependence
    bases = synth.inject(bases, 20, ord("n"), 9)
    got, oo = idx.streaming_search(bases, off)
    want = oracle_batch(orc, bases, off, True)
    assert np.array_equal(got, want)
    assert (got >= 0).mean() > 0.4                       # the case really exercises hits and misses
    got2, _ = idx.search(bases, off)
    assert np.array_equal(got2, oracle_batch(orc, bases, off, False))


def test_all_miss_reads(gpu, genome_case):
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    bases, off = synth.random_reads(1000, 100, 99)
    got, _ = idx.streaming_search(bases, off)
    assert np.array_equal(got, oracle_batch(orc, bases, off, True))


def test_ragged_empty_and_short_reads(gpu, genome_case):
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    g = genomes[0].tobytes()
    rnd = random.Random(11)
    reads = [b"", b"A", g[:29], g[:30], g[100:131], g[5000:5400], b"", g[777:777 + 64], b"N" * 40,
             g[10:10 + 33].lower(), g[-30:], g[-31:]]
    for _ in range(200):
        L = rnd.choice([0, 1, 29, 30, 31, 32, 33, 63, 64, 65, 95, 96, 97, 150, 257])
        s = rnd.randrange(0, len(g) - 300)
        reads.append(g[s:s + L])
    got = idx.streaming_search_reads(reads)
    for r, gg in zip(reads, got):
        assert np.array_equal(gg, orc.streaming_search(r)), r
    got = idx.search_reads(reads)
    for r, gg in zip(reads, got):
        assert np.array_equal(gg, orc.search_all(r)), r


def test_no_streaming_support_k63(gpu):
    # config 5 of BASELINE.json: k=63 without suffix_group_starts -> non-streaming path
    k = 63
    genomes = [synth.random_genome(100_000, 5)]
    genomes.append(synth.mutate(genomes[0], 0.03, 6))
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, False, False, 8)
    idx = gpu_index_from_oracle(orc)
    assert not idx.has_streaming_support
    bases, off = synth.sample_reads(genomes, 1500, 150, 0.005, 42)
    bases = synth.inject(bases, 10, ord("N"), 7)
    with pytest.raises(capi.SbwtGpuError) as ei:
        idx.streaming_search(bases, off)
    assert ei.value.code == capi.ERR_NO_STREAMING and "streaming search support not built" in ei.value.msg
    got, _ = idx.search(bases, off)
    assert np.array_equal(got, oracle_batch(orc, bases, off, False))
    assert (got >= 0).mean() > 0.3


def test_update_interval_and_forward(gpu, genome_case):
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    g = genomes[1].tobytes()
    rnd = random.Random(21)
    strs = [g[s:s + rnd.randrange(0, 40)] for s in (rnd.randrange(0, len(g) - 50) for _ in range(500))]
    strs += [b"ACGTN", b"acgt", b""]
    bases, off = capi.concat_reads(strs)
    first = np.zeros(len(strs), dtype=np.int64)
    second = np.full(len(strs), orc.n_nodes - 1, dtype=np.int64)
    first[5], second[5] = -1, -1
    f, s = idx.update_interval(bases, off, first, second)
    for i, st in enumerate(strs):
        assert (f[i], s[i]) == orc.update_interval(st, int(first[i]), int(second[i]))
    # forward (SBWT.hh:368-381) from random found columns
    bases2, off2 = synth.sample_reads(genomes, 300, 31, 0.0, 5)
    cols, _ = idx.search(bases2, off2)
    nodes = np.repeat(cols[cols >= 0][:300], 5)
    sym = np.tile(np.frombuffer(b"ACGTN", dtype=np.uint8), len(nodes) // 5)
    got = idx.forward(nodes, sym)
    want = np.array([orc.forward(int(v), bytes([int(c)])) for v, c in zip(nodes, sym)], dtype=np.int64)
    assert np.array_equal(got, want)


def test_precalc_table_and_limits(gpu, genome_case):
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)                       # table computed on the device
    assert np.array_equal(idx.get_precalc(), orc.precalc())
    cols = orc.columns()
    with pytest.raises(capi.SbwtGpuError) as ei:
        capi.Index.create(cols[0], cols[1], cols[2], cols[3], None, orc.n_nodes, 30, 0, 21)
    assert ei.value.code == capi.ERR_PRECALC_TOO_LONG
    with pytest.raises(capi.SbwtGpuError) as ei:
        capi.Index.create(cols[0], cols[1], cols[2], cols[3], None, orc.n_nodes, 5, 0, 6)
    assert ei.value.code == capi.ERR_PRECALC_GT_K


def test_streaming_equals_search_at_scale_properties(gpu, genome_case):
    # size-independent properties at a larger batch: streaming == per-k-mer search (test_large.hh:104-115),
    # every hit really is a valid column, and a checksum of checksums is stable across two runs
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    bases, off = synth.sample_reads(genomes, 200_000, 150, 0.01, 4242)
    a, oo = idx.streaming_search(bases, off)
    bb, _ = idx.search(bases, off)
    assert np.array_equal(a, bb)
    assert a.min() >= -1 and a.max() < orc.n_nodes
    a2, _ = idx.streaming_search(bases, off)
    assert int(np.bitwise_xor.reduce(a * np.arange(1, len(a) + 1))) == int(np.bitwise_xor.reduce(a2 * np.arange(1, len(a2) + 1)))
    # sample 2000 reads against the oracle
    want = oracle_batch(orc, bases[: 2000 * 150], off[:2001], True)
    assert np.array_equal(a[: len(want)], want)


@pytest.mark.parametrize("variant,probe", [(0, -1), (1, -1), (1, 0), (1, 9), (1, 11), (1, 12), (1, 13), (1, 20), (1, 29),
                                           (4, -1), (4, 0), (4, 9), (4, 12), (4, 13), (4, 29),
                                           (5, -1), (5, 0), (5, 9), (5, 12), (5, 13), (5, 29)])
def test_results_do_not_depend_on_search_variant_or_probe_length(gpu, genome_case, variant, probe):
    # k_search (reference order), k_search_cert (absent-substring certificates) and its path-order form
    # must give the same bits for every probe length, including reads with N / lower case and all-miss reads
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    capi.set_tuning("search_variant", variant)
    capi.set_tuning("probe_len", probe)
    try:
        bases, off = synth.sample_reads(genomes, 3000, 150, 0.02, 77)
        bases = synth.inject(bases, 80, ord("N"), 1)
        bases = synth.inject(bases, 80, ord("g"), 2)
        rb, ro = synth.random_reads(300, 150, 5)
        bases = np.concatenate([bases, rb])
        off = np.concatenate([off, ro[1:] + off[-1]])
        got, _ = idx.streaming_search(bases, off)
        assert np.array_equal(got, oracle_batch(orc, bases, off, True))
        got, _ = idx.search(bases, off)
        assert np.array_equal(got, oracle_batch(orc, bases, off, False))
    finally:
        capi.set_tuning("search_variant", -1)
        capi.set_tuning("probe_len", -1)


@pytest.mark.parametrize("sparse,path,safe", [(0, 1, 1), (20, 1, 1), (31, 0, 1), (0, 0, 1), (16, 1, 1), (31, 1, 0)])
def test_results_do_not_depend_on_acceleration_structures(gpu, genome_case, sparse, path, safe):
    # the sparse prefix table (any depth), the path order and its substitution-safe bits are derived data:
    # with or without them, same bits
    genomes, orc = genome_case
    capi.set_tuning("sparse_depth", sparse)
    capi.set_tuning("path_order", path)
    capi.set_tuning("path_safe", safe)
    try:
        idx = gpu_index_from_oracle(orc)
    finally:
        capi.set_tuning("sparse_depth", 31)
        capi.set_tuning("path_order", 1)
        capi.set_tuning("path_safe", 1)
    bases, off = synth.sample_reads(genomes, 3000, 150, 0.02, 78)
    bases = synth.inject(bases, 80, ord("N"), 1)
    bases = synth.inject(bases, 80, ord("t"), 2)
    rb, ro = synth.random_reads(200, 150, 6)
    bases = np.concatenate([bases, rb])
    off = np.concatenate([off, ro[1:] + off[-1]])
    got, _ = idx.streaming_search(bases, off)
    assert np.array_equal(got, oracle_batch(orc, bases, off, True))
    got, _ = idx.search(bases, off)
    assert np.array_equal(got, oracle_batch(orc, bases, off, False))


def test_wide_kernel_instantiation_matches(gpu, genome_case):
    # the 64-bit-position instantiation (used when n_nodes >= 2^31) forced onto a small index
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    bases, off = synth.sample_reads(genomes, 3000, 150, 0.02, 91)
    bases = synth.inject(bases, 40, ord("N"), 3)
    capi.set_tuning("debug", 16)
    try:
        got, _ = idx.streaming_search(bases, off)
        got2, _ = idx.search(bases, off)
    finally:
        capi.set_tuning("debug", 0)
    assert np.array_equal(got, oracle_batch(orc, bases, off, True))
    assert np.array_equal(got2, oracle_batch(orc, bases, off, False))


def test_fused_kernel_both_instantiations(gpu, genome_case):
    # batches of one read length run through k_search_fused<.., UNI = true>; "debug" 128 sends them through the general
    # instantiation (the one ragged batches and pieces use): same bits, both equal to the oracle; int32 results as well
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    for length in (150, 100, 33, 160):
        bases, off = synth.sample_reads(genomes, 4000, length, 0.02, 50 + length)
        bases = synth.inject(bases, 25, ord("N"), 9)
        want_s, want_f = oracle_batch(orc, bases, off, True), oracle_batch(orc, bases, off, False)
        for dbg in (0, 128):
            capi.set_tuning("debug", dbg)
            try:
                got_s, _ = idx.streaming_search(bases, off)
                got_f, _ = idx.search(bases, off)
            finally:
                capi.set_tuning("debug", 0)
            assert np.array_equal(got_s, want_s), (length, dbg)
            assert np.array_equal(got_f, want_f), (length, dbg)


def test_rank_beyond_2_pow_31_columns(gpu):
    # mega-block path: more than 2^31 columns (random bit matrix; rank() is defined for any bits)
    n = (1 << 31) + 1_000_003
    nw = (n + 63) // 64
    rng = np.random.default_rng(17)
    cols = [rng.integers(0, 1 << 63, size=nw, dtype=np.int64).astype(np.uint64) for _ in range(4)]
    idx = capi.Index.create(cols[0], cols[1], cols[2], cols[3], None, n, 31, 0, 0)
    with pytest.raises(capi.SbwtGpuError, match="only rank"):      # random bits are not an SBWT
        idx.search(np.frombuffer(b"A" * 40, dtype=np.uint8), np.array([0, 40]))
    pos = np.concatenate([rng.integers(0, n + 1, size=4000),
                          np.array([0, 1, (1 << 31) - 1, 1 << 31, (1 << 31) + 1, (1 << 31) + 64, n - 1, n])])
    sym = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=len(pos))
    got = idx.rank(pos, sym)
    pc = [np.concatenate([[0], np.cumsum(np.bitwise_count(c).astype(np.int64))]) for c in cols]
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    tail_mask = (np.uint64(1) << np.uint64(n & 63)) - np.uint64(1) if n & 63 else None
    for p_, s_, g in zip(pos, sym, got):
        c = code[int(s_)]
        w, b = int(p_) >> 6, int(p_) & 63
        want = int(pc[c][w])
        if b:
            want += bin(int(cols[c][w]) & ((1 << b) - 1)).count("1")
        assert g == want, (p_, s_)
    idx.close()


@pytest.mark.parametrize("L,k", [(1_200_000_000, 31), (2_250_000_000, 31), (2_250_000_000, 32)])
def test_search_on_indexes_of_more_than_2_pow_30_and_2_pow_31_columns(gpu, L, k):
    """The reference is int64 throughout (SBWT.hh:36-45).  One random sequence of L bases, k = 31, columns built on the GPU:
    1.2 x 10^9 columns -- the builders of the derived structures start more than 2^32 threads there (four per item), which one
    dispatch silently truncates (round 5: sliced launches; before, such an index lost most of its sparse table); 2.25 x 10^9
    columns -- beyond 2^31 the full image holds 32-bit UNSIGNED columns and positions and the fused kernel runs its BIG
    instantiation (round 5; before, such an index got the blocks-only kernel: 23 G k-mers/s).  Both must get the FULL image
    (level 0).  streaming_search and search of reads from all over the sequence (substitutions, N, lower case) on every route
    against each other and, on a sample, against the oracle; int32 results refused beyond 2^31 columns.
    Round 6: k = 32 at 2.25 x 10^9 columns -- 31 < k <= 63 gets the full image there too (the second-level table without flags in
    bit 31: position + 1, a lookup goes on past a full bucket), read by the fused kernel's WIDE + BIG instantiation."""
    import torch
    if torch.cuda.mem_get_info()[1] < (250 << 30):
        pytest.skip("needs a GPU with 288 GB: the image of 2.25e9 columns is 130 GB, its builders' scratch as much again")
    import gc
    gc.collect()
    torch.cuda.empty_cache()        # (what earlier tests of this process left in torch's caching allocator: the full-size suites hold 100 GB)
    genome = synth.random_genome(L, 7)
    bits = capi.build_bits_gpu([genome.tobytes()], k, False, True)
    big = bits.n_nodes >= (1 << 31)
    assert big == (L > 2_000_000_000)
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k, bits.n_kmers, 8)
    want_level = int(os.environ.get("SBWTGPU_IMAGE_LEVEL", "0"))     # (the knob sweep of tools/final_session.sh forces levels 1 and 2)
    assert idx.image_level == 0 if want_level == 0 else idx.image_level >= want_level, (idx.image_level, want_level)
    if want_level == 0 and "SBWTGPU_SEARCH_VARIANT" not in os.environ:
        assert idx.default_search_variant == 5, idx.default_search_variant
    bases, off = synth.sample_reads([genome], 3000, 150, 0.01, 5)
    bases = synth.inject(bases, 40, ord("N"), 6)
    bases = synth.inject(bases, 20, ord("c"), 8)
    got, _ = idx.streaming_search(bases, off)
    got2, _ = idx.search(bases, off)
    for variant in (1, 0):                                           # the blocks-only kernel, the reference-order kernel
        capi.set_tuning("search_variant", variant)
        try:
            assert np.array_equal(idx.streaming_search(bases, off)[0], got), variant
            assert np.array_equal(idx.search(bases, off)[0], got2), variant
        finally:
            capi.set_tuning("search_variant", -1)
    if big:
        with pytest.raises(capi.SbwtGpuError, match="2\\^31"):
            idx.search_i32(bases, off)
    else:
        assert np.array_equal(idx.search_i32(bases, off)[0].astype(np.int64), got)
    # the device-resident entry point (the fused kernel itself, no host pipeline in between)
    assert np.array_equal(_search_dev(idx, bases, off, k, True), got)
    # reads of other lengths: pieces of the fused kernel (161 .. 422 bases), zones of the general kernel beyond, reads shorter than
    # k; the formatted text of the CLI (values of ten digits) -- against the reference-order kernel
    b2, o2 = synth.ragged_reads([genome], 600, 20, 700, 0.01, 11)
    b2 = synth.inject(b2, 30, ord("N"), 12)
    mixed, _ = idx.streaming_search(b2, o2)
    text, _ = idx.search_text(b2, o2, True)
    capi.set_tuning("search_variant", 0)
    try:
        ref_mixed, oo2 = idx.streaming_search(b2, o2)
    finally:
        capi.set_tuning("search_variant", -1)
    assert np.array_equal(mixed, ref_mixed)
    assert np.array_equal(_search_dev(idx, b2, o2, k, True), ref_mixed)
    assert text == b"".join(print_vector(ref_mixed[oo2[r]:oo2[r + 1]]) for r in range(len(o2) - 1))
    idx.close()
    del genome
    orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k, bits.n_kmers, 8)
    sample = 700
    want = oracle_batch(orc, bases[:off[sample]], off[:sample + 1], True)
    assert np.array_equal(got[:len(want)], want)
    assert np.array_equal(got2[:len(want)], oracle_batch(orc, bases[:off[sample]], off[:sample + 1], False))
    assert 0.6 < (got >= 0).mean() < 0.85
    assert (got >= (1 << 31)).any() == big and (got >= (1 << 30)).any()


def test_device_side_print_vector(gpu, genome_case):
    # print_vector of src/CLI/sbwt_search.cpp:21-43 on the device, incl. the pipelined
    # host path with more than one chunk, empty reads, and the 0 -> empty-token quirk
    import ctypes as C
    import torch
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    g = genomes[0].tobytes()
    reads = [b"", g[:29], g[100:400], b"N" * 50, g[5000:5031], b""] + [g[i * 97:i * 97 + 150] for i in range(3000)]
    bases, off = capi.concat_reads(reads)
    bases = synth.inject(bases, 200, ord("N"), 4)
    want = b"".join(print_vector(orc.streaming_search(bases[off[r]:off[r + 1]].tobytes())) for r in range(len(reads)))
    text, nq = idx.search_text(bases, off, True)
    assert text == want and nq == int(np.maximum(np.diff(off) - 29, 0).sum())
    text2, _ = idx.search_text(bases, off, False)
    assert text2 == want
    # many chunks: > 4 Mi reads per chunk is the limit, so use short reads in bulk
    big_b, big_o = synth.sample_reads(genomes, 600_000, 64, 0.01, 3)      # 38 M bases -> 2 chunks
    t3, nq3 = idx.search_text(big_b, big_o, True)
    got, oo = idx.streaming_search(big_b, big_o)
    lines = t3.split(b"\n")
    assert len(lines) == 600_001 and lines[-1] == b"" and nq3 == len(got)
    for r in (0, 1, 299_999, 300_000, 599_999):
        assert lines[r] + b"\n" == print_vector(got[oo[r]:oo[r + 1]])
    # whole text of two chunk-sized stretches (one per pipeline chunk), every line's token count elsewhere
    for lo, hi in ((0, 40_000), (560_000, 600_000)):
        assert b"\n".join(lines[lo:hi]) + b"\n" == b"".join(print_vector(got[oo[r]:oo[r + 1]]) for r in range(lo, hi))
    assert all(ln.count(b" ") == 35 for ln in lines[40_000:560_000:97])
    # raw formatter on arbitrary values: 0 prints as an empty token, large values keep every digit
    vals = np.array([0, -1, 7, 10, 99, 100, 12345678901234, -1, 0, 0, 9223372036854775807, 1], dtype=np.int64)
    ooff = np.array([0, 3, 3, 11, 12], dtype=np.int64)
    dev = torch.device("cuda:0")
    d_v, d_o = torch.from_numpy(vals).to(dev), torch.from_numpy(ooff).to(dev)
    L = capi.lib()
    cap = 21 * len(vals) + 64
    d_t = torch.zeros(cap, dtype=torch.uint8, device=dev)
    d_l = torch.zeros(5, dtype=torch.int64, device=dev)
    scr = L.sbwtgpu_format_scratch_bytes(4)
    d_s = torch.zeros(scr, dtype=torch.uint8, device=dev)
    big = capi.Index.create(*[np.zeros(1, np.uint64)] * 4, None, 1, 2)     # n_nodes = 1: bound would be tiny ...
    rc = L.sbwtgpu_format_results_dev(big.handle, d_v.data_ptr(), d_o.data_ptr(), 4, len(vals), d_t.data_ptr(), cap,
                                      d_l.data_ptr(), d_s.data_ptr(), scr, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    n = int(d_l[4].item())
    assert bytes(d_t[:n].cpu().numpy()) == b"".join(print_vector(vals[ooff[r]:ooff[r + 1]]) for r in range(4))
    assert list(d_l.cpu().numpy()) == [0, 7, 8, 8 + len(print_vector(vals[3:11])), n]


@pytest.mark.parametrize("derive", [1, 0])
def test_internal_streaming_for_indexes_without_ssup(gpu, derive):
    # An index created without suffix_group_starts gets the marks derived on the device
    # (mark_suffix_groups, suffix_group_optimization.cpp:66-121) and the search loop streams internally
    # with raw-char validation; results must equal SBWT::search of every k-mer, lower case included.
    capi.set_tuning("derive_ssup", derive)
    try:
        for k, glen in ((63, 60_000), (30, 60_000), (6, 0)):
            if k == 6:
                kat = KATS["cli_end_to_end"]
                seqs = [b(s) for s in kat["seqs"]]
                orc = OracleIndex.build(seqs, 6, False, True, 4)
                reads = [b(q) for q in kat["queries"]] + [b"ACTAGTGTAGCTACAAA", b"ACTAGtGTAGCTACAAA", b"NNNNNNNNN"]
                bases, off = capi.concat_reads(reads)
            else:
                genomes = [synth.random_genome(glen, 5)]
                genomes.append(synth.mutate(genomes[0], 0.03, 6))
                orc = OracleIndex.build([g.tobytes() for g in genomes], k, False, False, 8)
                bases, off = synth.sample_reads(genomes, 1500, 200, 0.005, 42)
                bases = synth.inject(bases, 30, ord("N"), 7)
                bases = synth.inject(bases, 30, ord("a"), 8)      # lower case must give -1 (raw validation)
            idx = gpu_index_from_oracle(orc)
            assert not idx.has_streaming_support
            with pytest.raises(capi.SbwtGpuError):
                idx.streaming_search(bases, off)                  # still "not built", like the reference
            got, _ = idx.search(bases, off)
            assert np.array_equal(got, oracle_batch(orc, bases, off, False)), (k, derive)
    finally:
        capi.set_tuning("derive_ssup", 1)


@pytest.mark.parametrize("k", [65, 100, 255])
def test_long_kmers_up_to_255(gpu, k):
    # The reference supports k up to 255 (-DMAX_KMER_LENGTH, CMakeLists.txt:70-78); search itself is
    # k-agnostic.  Columns come from the definition-level brute force (any k), truth from a set.
    rnd = random.Random(k)
    g = "".join(rnd.choice("ACGT") for _ in range(k + 300))
    g2 = g[:150] + ("A" if g[150] != "A" else "C") + g[151:]
    seqs = [g, g2]
    brute = BruteSBWT(seqs, k)
    cols, ssup = brute.columns()
    n = len(brute.nodes)
    from bruteforce import int_to_words
    idx = capi.Index.create(*[int_to_words(c, n) for c in cols], int_to_words(ssup, n), n, k, len(brute.kmers), 8)
    reads = [g.encode(), g2.encode(), (g[:k + 40] + "N" + g[k + 41:]).encode(), g[5:k + 4].encode(), g[7:k + 7].encode(),
             ("".join(rnd.choice("ACGT") for _ in range(k + 20))).encode()]
    got = idx.streaming_search_reads(reads)
    got2 = idx.search_reads(reads)
    for r, a, a2 in zip(reads, got, got2):
        want = np.array(brute.search_all(r.decode()), dtype=np.int64)
        assert np.array_equal(a, want) and np.array_equal(a2, want)
    assert (got[0] >= 0).all() and (got[1] >= 0).all() and len(got[3]) == 0 and len(got[4]) == 1


@pytest.mark.parametrize("k,revcomp", [(30, True), (31, False), (12, True)])
def test_reference_query_file(gpu, k, revcomp):
    # the reference's own example_data/queries.fastq reads (tests/test_large.hh:104-115), indexed against
    # themselves: overlapping real reads -> big suffix groups, branching, walk-backs across blocks
    from test_oracle_golden import load_reference_queries
    from sbwt_amd import hostlib
    reads = load_reference_queries()
    bits = hostlib.build_bits(reads[:3000], k, revcomp, True, n_threads=4)
    orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                                bits.n_kmers, 8)
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k,
                            bits.n_kmers, 8)
    bases, off = capi.concat_reads(reads)
    got, oo = idx.streaming_search(bases, off)
    want = oracle_batch(orc, bases, off, True)
    assert np.array_equal(got, want)
    assert (got >= 0).mean() > 0.5
    got2, _ = idx.search(bases, off)
    assert np.array_equal(got2, want)                      # streaming == per-k-mer search (upper-case input)
    text, _ = idx.search_text(bases, off, True)
    assert text == b"".join(print_vector(want[oo[r]:oo[r + 1]]) for r in range(len(reads)))
    # the same index without streaming support: internal streaming must agree too
    idx2 = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], None, bits.n_nodes, k,
                             bits.n_kmers, 8)
    got3, _ = idx2.search(bases, off)
    assert np.array_equal(got3, want)


def test_long_reads_are_split_exactly(gpu, genome_case):
    # reads far longer than a lane should walk alone are cut into overlapping pieces by the host entry
    # points; the cut points avoid windows with lower-case bases, so results stay bit-identical
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    g0 = genomes[0].tobytes()
    long1 = bytearray(g0[1000:151000])                      # 150 kbp
    rnd = random.Random(5)
    for _ in range(300):
        long1[rnd.randrange(len(long1))] = ord(rnd.choice("ACGTNacgtn"))
    lower_run = bytearray(g0[20000:60000])
    lower_run[4090:4200] = bytes(lower_run[4090:4200]).lower()        # lower case exactly around a cut point
    lower_run[8190:8200] = b"acgtacgtac"
    all_lower = g0[500:20500].lower()
    reads = [bytes(long1), g0[:100], bytes(lower_run), all_lower, g0[300:300 + 2 * 2048 + 29], g0[300:300 + 2 * 2048 + 30],
             g0[7:7 + 3 * 2048 + 29 + 5], b"", genomes[1].tobytes()[:90000]]
    bases, off = capi.concat_reads(reads)
    want = oracle_batch(orc, bases, off, True)
    got, oo = idx.streaming_search(bases, off)
    assert np.array_equal(got, want)
    got2, _ = idx.search(bases, off)
    assert np.array_equal(got2, oracle_batch(orc, bases, off, False))
    text, nq = idx.search_text(bases, off, True)
    assert text == b"".join(print_vector(want[oo[r]:oo[r + 1]]) for r in range(len(reads))) and nq == len(want)


@pytest.mark.parametrize("k", [3, 4, 8, 16, 31])
def test_periodic_sequences_cycles_in_the_path_order(gpu, k):
    # Tandem repeats make cycles in the graph of streaming steps: the path order must cut every one of them
    # (k_path_cut) and still answer exactly.  Homopolymers (a column that follows itself), short periods,
    # a period longer than k, and the same with random flanks and branching variants.
    rng = random.Random(1000 + k)
    unit = "".join(rng.choice("ACGT") for _ in range(k + 5))
    flank = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    seqs = ["A" * 120, "AC" * 80, "ACG" * 60, "ACGT" * 50, "AACCGGTT" * 30, unit * 14,
            flank(60) + "GATTACA" * 25 + flank(60), flank(40) + unit * 6 + flank(40),
            flank(30) + ("ACGT" * 20) + "T" + ("ACGT" * 20) + flank(30)]
    orc = OracleIndex.build([b(s) for s in seqs], k, True, False, min(k, 2))
    idx = gpu_index_from_oracle(orc)
    genomes = [np.frombuffer(b(s), dtype=np.uint8) for s in seqs]
    bases, off = synth.sample_reads(genomes, 600, 100, 0.03, 5 + k)
    bases = synth.inject(bases, 30, ord("N"), 2)
    whole = np.concatenate(genomes)
    woff = np.concatenate([[0], np.cumsum([len(g) for g in genomes])]).astype(np.int64)
    bases = np.concatenate([bases, whole])
    off = np.concatenate([off, woff[1:] + off[-1]])
    for variant in (5, 4, 1):
        capi.set_tuning("search_variant", variant)
        try:
            got, _ = idx.streaming_search(bases, off)
            got2, _ = idx.search(bases, off)
        finally:
            capi.set_tuning("search_variant", -1)
        assert np.array_equal(got, oracle_batch(orc, bases, off, True))
        assert np.array_equal(got2, oracle_batch(orc, bases, off, False))


def test_exported_image_adopted_as_a_replica_gives_the_same_bits(gpu, genome_case):
    # what multi-GPU replication does, on one device: header + image copied into a
    # caller-owned buffer and adopted; the derived structures (sparse table, filter, path order) travel inside
    import torch
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    hdr = idx.export_header()
    buf = torch.empty(idx.blob_bytes, dtype=torch.uint8, device="cuda:0")
    idx.copy_blob(buf.data_ptr(), idx.blob_bytes, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rep = capi.Index.adopt(hdr, buf.data_ptr(), idx.blob_bytes, 0, keepalive=buf)
    del idx
    bases, off = synth.sample_reads(genomes, 2000, 150, 0.02, 314)
    bases = synth.inject(bases, 40, ord("N"), 4)
    got, _ = rep.streaming_search(bases, off)
    assert np.array_equal(got, oracle_batch(orc, bases, off, True))


@pytest.mark.parametrize("k", [32, 33, 47, 62, 63])
def test_second_level_sparse_table_for_k_up_to_63(gpu, k):
    # 31 < k <= 63: the walk for a whole k-mer is the 31-base sparse lookup + one second-level lookup keyed by
    # (the prefix's interval, the remaining k-31 bases); hits, misses in either level, N inside either window
    genomes = [synth.random_genome(60_000, 21)]
    genomes.append(synth.mutate(genomes[0], 0.03, 22))
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 4)
    idx = gpu_index_from_oracle(orc)
    bases, off = synth.sample_reads(genomes, 1500, 150, 0.015, 23 + k)
    bases = synth.inject(bases, 60, ord("N"), 3)
    bases = synth.inject(bases, 30, ord("a"), 4)
    rb, ro = synth.random_reads(100, 150, 9)
    bases = np.concatenate([bases, rb])
    off = np.concatenate([off, ro[1:] + off[-1]])
    got, _ = idx.streaming_search(bases, off)
    assert np.array_equal(got, oracle_batch(orc, bases, off, True))
    got, _ = idx.search(bases, off)
    assert np.array_equal(got, oracle_batch(orc, bases, off, False))


@pytest.mark.parametrize("k", [20, 31, 32, 40, 63])
def test_the_layout_of_2_pow_31_columns_on_a_small_index(gpu, k):
    """The image layout of 2^31 .. 2^32 columns (32-bit unsigned columns and positions, no flag in bit 31 of any of them) forced
    onto a small index ("big_path" 2): the fused kernel's BIG instantiations -- k <= 31, and since round 6 31 < k <= 63 (the
    second-level table's entries hold position + 1, no overflow flag; depth-31 entries carry no position) -- against the
    oracle, on every route, reads of 150 bases, pieces (250 bases), ragged and long reads, N and lower case."""
    genomes = [synth.random_genome(80_000, 31)]
    genomes.append(synth.mutate(genomes[0], 0.03, 32))
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 4)
    capi.set_tuning("big_path", 2)
    try:
        idx = gpu_index_from_oracle(orc)
    finally:
        capi.set_tuning("big_path", 1)
    plain = gpu_index_from_oracle(orc)
    if int(os.environ.get("SBWTGPU_IMAGE_LEVEL", "0")) == 0:          # (the knob sweep of tools/final_session.sh forces levels 1 and 2)
        assert idx.image_level == 0 and idx.default_search_variant == 5
        import struct
        # (SbwtBlobHeader::big_layout is the int32 at byte 160 of the exported header)
        assert struct.unpack_from("<i", idx.export_header(), 160)[0] == 1 and struct.unpack_from("<i", plain.export_header(), 160)[0] == 0
    for (nr, L, seed) in ((1500, 150, 5), (600, 250, 6)):
        bases, off = synth.sample_reads(genomes, nr, L, 0.015, seed + k)
        bases = synth.inject(bases, 40, ord("N"), 3)
        bases = synth.inject(bases, 20, ord("a"), 4)
        rb, ro = synth.random_reads(60, L, 9)
        bases = np.concatenate([bases, rb])
        off = np.concatenate([off, ro[1:] + off[-1]])
        want = oracle_batch(orc, bases, off, True)
        want2 = oracle_batch(orc, bases, off, False)
        for variant in (-1, 5, 4, 1, 0):
            capi.set_tuning("search_variant", variant)
            try:
                assert np.array_equal(idx.streaming_search(bases, off)[0], want), (k, L, variant)
                assert np.array_equal(idx.search(bases, off)[0], want2), (k, L, variant)
            finally:
                capi.set_tuning("search_variant", -1)
        assert np.array_equal(_search_dev(idx, bases, off, k, True), want)
        assert np.array_equal(plain.streaming_search(bases, off)[0], want)
    b2, o2 = synth.ragged_reads(genomes, 500, 20, 700, 0.01, 11)
    b2 = synth.inject(b2, 30, ord("N"), 12)
    assert np.array_equal(idx.streaming_search(b2, o2)[0], oracle_batch(orc, b2, o2, True))
    assert np.array_equal(_search_dev(idx, b2, o2, k, True), oracle_batch(orc, b2, o2, True))


def test_arbitrary_bytes_in_reads(gpu, genome_case):
    # every byte value 1..255 somewhere in the reads: only upper-case ACGT is valid for a full search, ACGT after
    # toupper for a streaming step (the encode kernel classifies four bases per 32-bit operation)
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    bases, off = synth.sample_reads(genomes, 3000, 150, 0.005, 99)
    rng = np.random.default_rng(17)
    where = rng.choice(len(bases), size=6000, replace=False)
    bases = bases.copy()
    bases[where] = rng.integers(1, 256, size=len(where), dtype=np.uint8)
    bases[where[:256]] = np.arange(256, dtype=np.uint8).clip(1)      # each value at least once
    got, _ = idx.streaming_search(bases, off)
    assert np.array_equal(got, oracle_batch(orc, bases, off, True))
    got, _ = idx.search(bases, off)
    assert np.array_equal(got, oracle_batch(orc, bases, off, False))


def test_many_short_reads_of_mixed_lengths(gpu, genome_case):
    # reads of k-2 .. k+12 bases (0 .. 13 k-mers each): every alignment of a read's results to the 64-byte lines of
    # `out`, reads without results between reads with results, runs that end after one or two results
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    k = orc.k
    rng = np.random.default_rng(5)
    cat = np.concatenate(genomes)
    n = 60_000
    lens = rng.integers(k - 2, k + 13, size=n)
    starts = rng.integers(0, len(cat) - 64, size=n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    bases = np.empty(int(off[-1]), dtype=np.uint8)
    for r in range(n):
        bases[off[r]:off[r + 1]] = cat[starts[r]:starts[r] + lens[r]]
    flip = rng.choice(len(bases), size=len(bases) // 60, replace=False)
    bases[flip] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=len(flip))]
    got, oo = idx.streaming_search(bases, off)
    assert np.array_equal(got, oracle_batch(orc, bases, off, True))
    got2, _ = idx.search(bases, off)
    assert np.array_equal(got2, oracle_batch(orc, bases, off, False))


@pytest.mark.parametrize("variant", [4])
def test_sorted_reads_give_the_same_bits(gpu, genome_case, variant):
    # "sort_reads": the path-order kernels take the reads in the order of their first k-mer's path position (a radix
    # sort before the search); every result still lands at its own place.  Fixed-length and ragged batches, reads
    # without any anchor (random, N-only, shorter than k), lower case.
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    bases, off = synth.sample_reads(genomes, 5000, 150, 0.02, 123)
    bases = synth.inject(bases, 120, ord("N"), 1)
    bases = synth.inject(bases, 60, ord("a"), 2)
    rb, ro = synth.random_reads(400, 150, 9)
    bases = np.concatenate([bases, rb, np.full(300, ord("N"), dtype=np.uint8)])
    off = np.concatenate([off, ro[1:] + off[-1], [off[-1] + ro[-1] + 150, off[-1] + ro[-1] + 300]])
    rg_b, rg_o = synth.ragged_reads(genomes, 3000, 0, 260, 0.01, 77)
    capi.set_tuning("search_variant", variant)
    try:
        for b_, o_ in ((bases, off), (rg_b, rg_o)):
            want = oracle_batch(orc, b_, o_, True)
            for sort in (1, 0):
                capi.set_tuning("sort_reads", sort)
                got, _ = idx.streaming_search(b_, o_)
                assert np.array_equal(got, want), (variant, sort)
                got, _ = idx.search(b_, o_)
                assert np.array_equal(got, oracle_batch(orc, b_, o_, False)), (variant, sort)
    finally:
        capi.set_tuning("search_variant", -1)
        capi.set_tuning("sort_reads", -1)


@pytest.mark.parametrize("variant", [4, 5])
def test_reads_that_hop_between_strains(gpu, genome_case, variant):
    # reads that hop between the strains every few bases leave their path all the time: transition entries (hashed on
    # (path position, char), probed linearly), runs that end inside the 32 steps an entry quotes, path ends
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    rng = np.random.default_rng(4)
    n, L = 3000, 150
    start = rng.integers(0, len(genomes[0]) - L, size=n)
    which = rng.integers(0, 3, size=(n, L // 10))
    bases = np.empty(n * L, dtype=np.uint8)
    for r in range(n):
        for seg in range(L // 10):
            bases[r * L + seg * 10:r * L + seg * 10 + 10] = genomes[which[r, seg]][start[r] + seg * 10:start[r] + seg * 10 + 10]
    bases = synth.mutate(bases, 0.005, 9)
    off = np.arange(n + 1, dtype=np.int64) * L
    want = oracle_batch(orc, bases, off, True)
    capi.set_tuning("search_variant", variant)
    try:
        got, _ = idx.streaming_search(bases, off)
        assert np.array_equal(got, want), variant
        bases2 = synth.inject(synth.inject(bases, 200, ord("N"), 3), 200, ord("t"), 4)      # ... with bases the fused kernel hands on
        got, _ = idx.streaming_search(bases2, off)
        assert np.array_equal(got, oracle_batch(orc, bases2, off, True)), variant
    finally:
        capi.set_tuning("search_variant", -1)


@pytest.mark.parametrize("variant", [4, 5])
def test_stitched_chains_are_derived_data(gpu, variant):
    # strains that share long stretches: the path order joins tails to heads through COPIES of the shared stretch
    # ("path_stitch", on by default; "path_stitch_min" = shortest copy).  With copies, without, with few: the same bits,
    # for reads that follow one strain and for reads that hop between strains inside the shared stretches.
    k = 30
    genomes = synth.coli3_like(150_000)
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 8)
    rng = np.random.default_rng(12)
    bases, off = synth.sample_reads(genomes, 3000, 150, 0.01, 5)
    n, L = 1500, 150
    start = rng.integers(0, min(len(g) for g in genomes) - L, size=n)
    which = rng.integers(0, 3, size=(n, 3))
    hop = np.empty(n * L, dtype=np.uint8)
    for r in range(n):
        for seg in range(3):
            hop[r * L + seg * 50:r * L + seg * 50 + 50] = genomes[which[r, seg]][start[r] + seg * 50:start[r] + seg * 50 + 50]
    bases = np.concatenate([bases, hop])
    off = np.concatenate([off, off[-1] + np.arange(1, n + 1, dtype=np.int64) * L])
    bases = synth.inject(bases, 60, ord("N"), 3)
    want_s = oracle_batch(orc, bases, off, True)
    want_f = oracle_batch(orc, bases, off, False)
    chains = {}
    capi.set_tuning("search_variant", variant)
    try:
        for stitch, mn in ((1, 1), (0, 1), (1, 12)):
            capi.set_tuning("path_stitch", stitch)
            capi.set_tuning("path_stitch_min", mn)
            idx = gpu_index_from_oracle(orc)
            chains[(stitch, mn)] = idx.n_paths
            got, _ = idx.streaming_search(bases, off)
            assert np.array_equal(got, want_s), (stitch, mn)
            got, _ = idx.search(bases, off)
            assert np.array_equal(got, want_f), (stitch, mn)
    finally:
        capi.set_tuning("search_variant", -1)
        capi.set_tuning("path_stitch", 1)
        capi.set_tuning("path_stitch_min", 1)
    if chains[(0, 1)] > 0:                                  # (an image without a path order, SBWTGPU_IMAGE_LEVEL >= 1: nothing to join)
        assert chains[(1, 1)] < chains[(0, 1)], chains      # the copies did join paths


@pytest.mark.parametrize("k", [24, 31, 40, 63])
def test_long_reads_through_the_fused_kernels_ticket_table(gpu, k):
    """Round 6: a batch whose reads are mostly longer than three pieces of 160 bases used to be the general
    kernel's altogether.  Now such a batch is cut into tickets of <= 160 bases listed in a table (k_fused_tickets) and the fused
    kernel answers it -- for 31 < k <= 63 with its aligned compare.  Reads of 300 .. 6000 bases and a whole genome as ONE read,
    with substitutions, N and lower case (those reads are handed on: their zones are searched by the kernel behind, the others'
    zones stay empty), short reads and reads shorter than k in between; against the oracle, against the route without the table,
    streaming_search and search, host and device entry points, int32 results."""
    genomes = [synth.random_genome(150_000, 41)]
    genomes.append(synth.mutate(genomes[0], 0.03, 42))
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 4)
    idx = gpu_index_from_oracle(orc)
    bases, off = synth.ragged_reads(genomes, 700, 300, 6000, 0.01, 13)
    short_b, short_o = synth.ragged_reads(genomes, 60, 5, 200, 0.01, 14)
    whole = synth.mutate(genomes[1], 0.005, 15)
    bases = np.concatenate([bases, short_b, whole])
    off = np.concatenate([off, short_o[1:] + off[-1], [off[-1] + short_o[-1] + len(whole)]])
    bases = synth.inject(bases, 25, ord("N"), 3)
    bases = synth.inject(bases, 10, ord("g"), 4)
    want = oracle_batch(orc, bases, off, True)
    want2 = oracle_batch(orc, bases, off, False)
    for table in (1, 0):
        capi.set_tuning("fused_table", table)
        try:
            assert np.array_equal(idx.streaming_search(bases, off)[0], want), (k, table)
            assert np.array_equal(idx.search(bases, off)[0], want2), (k, table)
            assert np.array_equal(_search_dev(idx, bases, off, k, True), want), (k, table)
            assert np.array_equal(idx.search_i32(bases, off)[0].astype(np.int64), want), (k, table)
        finally:
            capi.set_tuning("fused_table", 1)
    # clean long reads only: nothing is handed on, every zone of the kernel behind stays empty
    cb, co = synth.ragged_reads(genomes, 300, 500, 3000, 0.02, 21)
    assert np.array_equal(_search_dev(idx, cb, co, k, True), oracle_batch(orc, cb, co, True))


@pytest.mark.parametrize("k", [20, 31])
def test_sorted_instantiation_and_the_hint_that_picks_it(gpu, k):
    """Round 6: `k_search_fused<..., SORT>` (lanes sorted by state: searcher waves and path-follower waves, reads handed over
    through LDS slots and rings).  "fused_sort" | 4096 runs it whatever the workspace says: reads that follow their paths, reads
    of unrelated sequence (the followers have nothing to do for the whole launch), 8 % substitutions, ragged lengths, pieces, tiny
    batches (fewer reads than one wave; one read) -- against the oracle and the unsorted kernel.  As shipped the instantiation is
    picked by the hint the call before left in the workspace's header: a second call on the same workspace with reads that
    follow their paths runs sorted, one with unrelated reads unsorted again -- with the same results either way."""
    import torch
    genomes = [synth.random_genome(120_000, 51)]
    genomes.append(synth.mutate(genomes[0], 0.04, 52))
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 4)
    idx = gpu_index_from_oracle(orc)
    batches = []
    batches.append(synth.sample_reads(genomes, 3000, 150, 0.01, 1))
    batches.append(synth.sample_reads(genomes, 3000, 150, 0.08, 2))
    batches.append(synth.random_reads(2000, 150, 3))
    batches.append(synth.ragged_reads(genomes, 2500, 10, 160, 0.02, 4))
    batches.append(synth.sample_reads(genomes, 1200, 300, 0.01, 5))
    batches.append(synth.sample_reads(genomes, 40, 150, 0.01, 6))
    batches.append(synth.sample_reads(genomes, 1, 150, 0.0, 7))
    nb, no = synth.sample_reads(genomes, 2000, 150, 0.01, 8)
    batches.append((synth.inject(nb, 50, ord("N"), 9), no))
    try:
        for bi, (bases, off) in enumerate(batches):
            want = oracle_batch(orc, bases, off, True)
            for fs in (7728, 4097, 6192, 0):
                capi.set_tuning("fused_sort", fs)
                assert np.array_equal(_search_dev(idx, bases, off, k, True), want), (k, bi, fs)
                assert np.array_equal(idx.streaming_search(bases, off)[0], want), (k, bi, fs)
            capi.set_tuning("fused_sort", 7728)
            assert np.array_equal(idx.search_i32(bases, off)[0].astype(np.int64), want), (k, bi)
    finally:
        capi.set_tuning("fused_sort", 3632)
    if int(os.environ.get("SBWTGPU_IMAGE_LEVEL", "0")) != 0 or idx.image_level != 0:
        return                      # (no path order: the fused route, and with it the hint, is not in play)
    # the hint: one workspace, a sequence of calls
    dev = torch.device("cuda:0")
    a_b, a_o = synth.sample_reads(genomes, 3000, 150, 0.002, 11)      # (few substitutions: > 12 k-mers along paths per search started)
    r_b, r_o = batches[2]
    nmax = max(len(a_b), len(r_b))
    wsb = capi.search_workspace_bytes(nmax)
    d_ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)

    def call(bases, off):
        oo = capi.out_offsets(off, k)
        d_b = torch.from_numpy(np.ascontiguousarray(bases)).to(dev)
        d_ro, d_oo = torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(oo).to(dev)
        d_out = torch.full((int(oo[-1]),), -7, dtype=torch.int64, device=dev)
        idx.streaming_search_dev(d_b.data_ptr(), d_b.numel(), d_ro.data_ptr(), len(off) - 1, d_out.data_ptr(), d_oo.data_ptr(),
                                 d_ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream, True)
        torch.cuda.synchronize()
        return d_out.cpu().numpy(), int(d_ws[312:316].view(torch.int32).item()) & 0xFFFFFFFF      # (SbwtWorkHeader::hint: the header's last word)
    w_a, w_r = oracle_batch(orc, a_b, a_o, True), oracle_batch(orc, r_b, r_o, True)
    got, hint = call(a_b, a_o)                       # a fresh workspace: the unsorted kernel; its reads follow their paths
    assert np.array_equal(got, w_a) and hint == 0x5B377A01, hex(hint)
    got, hint = call(a_b, a_o)                       # ... a second call in a row like that (still unsorted): the next one runs sorted
    assert np.array_equal(got, w_a) and hint == 0x5B377A02, hex(hint)
    got, hint = call(a_b, a_o)                       # sorted; the count stays at its cap
    assert np.array_equal(got, w_a) and hint == 0x5B377A02, hex(hint)
    got, hint = call(r_b, r_o)                       # unrelated reads, sorted once; the count starts again
    assert np.array_equal(got, w_r) and hint == 0x5B377A00, hex(hint)
    got, hint = call(r_b, r_o)
    assert np.array_equal(got, w_r) and hint == 0x5B377A00, hex(hint)
    for _ in range(3):                               # batches of alternating kinds: never two in a row that fit, never sorted
        got, hint = call(a_b, a_o)
        assert np.array_equal(got, w_a) and hint == 0x5B377A01, hex(hint)
        got, hint = call(r_b, r_o)
        assert np.array_equal(got, w_r) and hint == 0x5B377A00, hex(hint)


def _search_dev(idx, bases, off, k, streaming):
    import torch
    dev = torch.device("cuda:0")
    lens = np.diff(off)
    oo = np.concatenate([[0], np.cumsum(np.maximum(lens - k + 1, 0))]).astype(np.int64)
    d_b = torch.from_numpy(np.ascontiguousarray(bases)).to(dev)
    d_ro = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_oo = torch.from_numpy(oo).to(dev)
    d_out = torch.full((int(oo[-1]),), -7, dtype=torch.int64, device=dev)
    wsb = capi.search_workspace_bytes(d_b.numel())
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    idx.streaming_search_dev(d_b.data_ptr(), d_b.numel(), d_ro.data_ptr(), len(off) - 1, d_out.data_ptr(), d_oo.data_ptr(),
                             d_ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream, streaming)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


@pytest.mark.parametrize("variant", [5, 4, 1])
def test_long_reads_are_cut_on_the_device_too(gpu, genome_case, variant):
    # the device-pointer entry points cannot cut reads on the host: the check kernel lists the zones of reads longer than
    # 2 * SBWT_PIECE (1024) k-mers, k_piece_bounds moves each cut to a k-mer without lower-case bases, lanes take the pieces
    # as tickets of their own.  Lower case around the cut points, a read that is all lower case, lengths around the
    # threshold, short reads in between; then a batch of equally long reads (the path kernel's offset arithmetic).
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    k = orc.k
    g0 = genomes[0].tobytes()
    long1 = bytearray(g0[1000:151000])
    rnd = random.Random(6)
    for _ in range(300):
        long1[rnd.randrange(len(long1))] = ord(rnd.choice("ACGTNacgtn"))
    lower_run = bytearray(g0[20000:60000])
    lower_run[1000:1100] = bytes(lower_run[1000:1100]).lower()          # lower case exactly around cut points
    lower_run[2040:2060] = bytes(lower_run[2040:2060]).lower()
    lower_run[3070:3075] = b"acgta"
    lower_run[5000:9000] = bytes(lower_run[5000:9000]).lower()          # ... and across several zones
    all_lower = g0[500:20500].lower()
    reads = [bytes(long1), g0[:100], bytes(lower_run), all_lower, g0[300:300 + 2048 + k - 1], g0[300:300 + 2048 + k],
             g0[7:7 + 3 * 1024 + k + 4], b"", b"ACGT", genomes[1].tobytes()[:90000], g0[40:190]]
    bases, off = capi.concat_reads(reads)
    capi.set_tuning("search_variant", variant)
    try:
        for streaming in (True, False):
            want = oracle_batch(orc, bases, off, streaming)
            assert np.array_equal(_search_dev(idx, bases, off, k, streaming), want), streaming
            capi.set_tuning("split_long", 0)               # one lane per read: the same bits
            try:
                assert np.array_equal(_search_dev(idx, bases, off, k, streaming), want), streaming
            finally:
                capi.set_tuning("split_long", 1)
        uni = [genomes[2].tobytes()[s:s + 5000] for s in range(0, 100000, 4000)]
        uni[3] = uni[3][:2000] + uni[3][2000:2100].lower() + uni[3][2100:]
        bases, off = capi.concat_reads(uni)
        assert np.array_equal(_search_dev(idx, bases, off, k, True), oracle_batch(orc, bases, off, True))
    finally:
        capi.set_tuning("search_variant", -1)


def test_fused_kernel_takes_batches_of_mixed_lengths(gpu, genome_case):
    # "fused_ragged" (on by default): a batch whose reads differ in length still goes through the fused kernel -- it fetches
    # the offsets of its 64 tickets with every refill --, which hands on what it cannot take: reads of more than 160 bases
    # (one of them a 120 kbp read that the general kernel behind it answers in pieces), reads with N or lower case.  Reads
    # shorter than k and empty reads answer nothing.  Same bits as the two-pass route and as the oracle.
    genomes, orc = genome_case
    idx = gpu_index_from_oracle(orc)
    k = orc.k
    rng = np.random.default_rng(21)
    cat = np.concatenate(genomes)
    n = 6000
    lens = rng.integers(0, 161, size=n)
    lens[rng.integers(0, n, size=60)] = rng.integers(161, 400, size=60)       # too long for the fused kernel
    lens[17] = 120_000
    lens[n - 1] = 150
    st = (rng.random(n) * (len(cat) - lens)).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    bases = np.empty(int(off[-1]), dtype=np.uint8)
    for r in range(n):
        bases[off[r]:off[r + 1]] = cat[st[r]:st[r] + lens[r]]
    bases = synth.mutate(bases, 0.01, 3)
    bases = synth.inject(bases, 100, ord("N"), 4)
    bases = synth.inject(bases, 100, ord("c"), 5)
    for streaming in (True, False):
        want = oracle_batch(orc, bases, off, streaming)
        for ragged in (1, 0):
            capi.set_tuning("fused_ragged", ragged)
            try:
                got = _search_dev(idx, bases, off, k, streaming)
            finally:
                capi.set_tuning("fused_ragged", 1)
            assert np.array_equal(got, want), (streaming, ragged)
    # mostly long reads: the check kernel's sample sends the batch to the general route (same bits)
    lens2 = rng.integers(150, 400, size=3000)
    st2 = (rng.random(3000) * (len(cat) - lens2)).astype(np.int64)
    off2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int64)
    bases2 = np.concatenate([cat[a:a + l] for a, l in zip(st2, lens2)])
    assert np.array_equal(_search_dev(idx, bases2, off2, k, True), oracle_batch(orc, bases2, off2, True))


@pytest.mark.parametrize("k", [30, 63])
def test_fused_kernel_takes_reads_of_161_to_400_bases_as_pieces(gpu, k):
    # Round 4, "fused_pieces" = 2 or 3 (the default since round 5; before, for k <= 31 the default was 1: no faster
    # than the two-pass route -- NOTES.md): a read of more than
    # 160 bases is taken by the fused kernel as up to three pieces of 160 bases that overlap by k-1 (a ticket is (read, piece);
    # SBWT.hh:556-579 has no length limit).  Reads of ONE length 161 .. 3 * (161 - k) + k - 1
    # (offset arithmetic), reads of mixed lengths with two or three tickets each, N / lower case inside one piece (the read
    # is handed on to the general kernel, its clean pieces are still answered here), reads beyond the limit (handed on
    # whole), reads shorter than k.  Same bits as the oracle, and the fused kernel did the work (its run counter).
    import torch
    genomes = [synth.random_genome(120_000, 5)]
    genomes.append(synth.mutate(genomes[0], 0.05, 6))
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 8)
    idx = gpu_index_from_oracle(orc)
    st = torch.cuda.current_stream().cuda_stream
    cat = np.concatenate(genomes)
    rng = np.random.default_rng(77)
    kpp = 161 - k
    limit3 = 3 * kpp + k - 1

    def run(bases, off):
        for streaming in (True, False):
            want = oracle_batch(orc, bases, off, streaming)
            for variant, pieces in ((5, 3), (5, 2), (5, 1), (4, 1)):
                capi.set_tuning("search_variant", variant)
                capi.set_tuning("fused_pieces", pieces)
                try:
                    got = _search_dev(idx, bases, off, k, streaming)
                finally:
                    capi.set_tuning("search_variant", -1)
                    capi.set_tuning("fused_pieces", -1)
                assert np.array_equal(got, want), (streaming, variant, pieces)

    # (1) reads of one length
    for L in (161, 250, 2 * kpp + k - 1, 2 * kpp + k, min(320, limit3), limit3):
        bases, off = synth.sample_reads(genomes, 700, L, 0.01, 100 + L)
        bases = synth.inject(bases, 20, ord("N"), 9)
        bases = synth.inject(bases, 20, ord("g"), 10)
        run(bases, off)
    # ... and the fused kernel really takes them: 4 000 clean reads of 250 bases, all k-mers answered along path runs
    bases, off = synth.sample_reads(genomes, 4000, 250, 0.0, 5)
    d_b = torch.from_numpy(bases).cuda()
    oo = capi.out_offsets(off, k)
    d_ro, d_oo = torch.from_numpy(off).cuda(), torch.from_numpy(oo).cuda()
    d_out = torch.empty(int(oo[-1]), dtype=torch.int64, device="cuda")
    wsb = capi.search_workspace_bytes(len(bases))
    d_ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    capi.set_tuning("fused_pieces", 2)
    try:
        idx.streaming_search_dev(d_b.data_ptr(), len(bases), d_ro.data_ptr(), 4000, d_out.data_ptr(), d_oo.data_ptr(), d_ws.data_ptr(),
                                 wsb, st, True)
        torch.cuda.synchronize()
    finally:
        capi.set_tuning("fused_pieces", -1)
    stats = idx.workspace_stats(d_ws.data_ptr(), st)
    assert int((d_out >= 0).sum()) == int(oo[-1])
    if idx.image_level == 0:                             # (SBWTGPU_IMAGE_LEVEL > 0 in the knob sweep: no path order, no fused kernel)
        assert stats[4] > 0.9 * int(oo[-1])
        hdr = d_ws[:256].cpu().numpy().view(np.uint64)
        assert int(hdr[13]) == 0, "reads of 250 bases were handed on to the general kernel"  # SbwtWorkHeader.n_deferred (byte 104)
    # (2) mixed lengths: mostly 100 .. 320, some too long even for three pieces, some shorter than k, some empty
    n = 5000
    lens = rng.integers(100, 321, size=n)
    lens[rng.integers(0, n, size=40)] = rng.integers(limit3 + 1, limit3 + 300, size=40)
    lens[rng.integers(0, n, size=40)] = rng.integers(0, k, size=40)
    sta = (rng.random(n) * (len(cat) - lens)).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    bases = np.concatenate([cat[a:a + l] for a, l in zip(sta, lens)])
    bases = synth.mutate(bases, 0.01, 3)
    bases = synth.inject(bases, 120, ord("N"), 4)
    bases = synth.inject(bases, 120, ord("t"), 5)
    run(bases, off)


@pytest.mark.parametrize("case", ["periodic3", "periodic8", "periodic31", "genomes30", "pan31", "one_path20"])
def test_path_order_by_splitters_equals_the_doubling_over_every_column(gpu, case):
    """Round 6: the list ranking behind the path order (head of its path and distance from it, per column) is done by splitters
    (sbwt_derived.hip k_rank_*: heads + one column in 64 walk to the next splitter, the splitters are ranked by doubling, the
    stretches are walked again) instead of doubling over every column.  Same path order, bit for bit: col[], pos[] and the path
    groups of images built with SBWTGPU_PATH_RANK = 0 (doubling), 2 (splitters whatever their share) and unset (splitters
    unless more than one column in eight is one) -- on graphs with cycles of every size (with and without a splitter on them),
    on many short paths and on one long path."""
    import struct
    rng = random.Random(77)
    if case.startswith("periodic"):
        k = int(case[8:])
        unit = "".join(rng.choice("ACGT") for _ in range(k + 5))
        flank = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
        seqs = ["A" * 120, "AC" * 80, "ACG" * 60, "ACGT" * 50, "AACCGGTT" * 30, unit * 14, flank(60) + "GATTACA" * 25 + flank(60),
                flank(40) + unit * 6 + flank(40), ("".join(rng.choice("ACGT") for _ in range(700))) * 3]      # (a cycle of 700 columns)
        orc = OracleIndex.build([b(s) for s in seqs], k, True, False, min(k, 2))
    elif case == "genomes30":
        genomes = [synth.random_genome(120_000, 5)]
        genomes.append(synth.mutate(genomes[0], 0.03, 6))
        orc = OracleIndex.build([g.tobytes() for g in genomes], 30, True, False, 6)
    elif case == "pan31":
        genomes = synth.pan_like(12, 40_000)
        orc = OracleIndex.build([g.tobytes() for g in genomes], 31, True, False, 6)
    else:
        orc = OracleIndex.build([synth.random_genome(300_000, 9).tobytes()], 20, False, False, 6)
    old = os.environ.get("SBWTGPU_PATH_RANK")
    images = {}
    try:
        for mode in ("0", "2", None, "gave_up"):
            os.environ.pop("SBWTGPU_PATH_RANK_LIMIT", None)
            if mode is None:
                os.environ.pop("SBWTGPU_PATH_RANK", None)
            elif mode == "gave_up":                     # walks of at most 8 steps: some walk gives up, the doubling takes over
                os.environ["SBWTGPU_PATH_RANK"] = "2"
                os.environ["SBWTGPU_PATH_RANK_LIMIT"] = "8"
            else:
                os.environ["SBWTGPU_PATH_RANK"] = mode
            idx = gpu_index_from_oracle(orc)
            hdr = idx.export_header()
            has_path, = struct.unpack_from("<i", hdr, 164)
            off_col, off_pos, off_pq = struct.unpack_from("<3q", hdr, 168)
            n_pos, = struct.unpack_from("<q", hdr, 248)
            if not has_path:
                pytest.skip("an image without a path order (SBWTGPU_IMAGE_LEVEL >= 1)")
            blob = idx.blob_tensor()
            images[mode] = (idx.n_paths, n_pos, blob[off_pos:off_pos + 4 * idx.n_nodes].cpu().numpy().copy(),
                            blob[off_col:off_col + 4 * n_pos].cpu().numpy().copy(),
                            blob[off_pq:off_pq + 16 * (n_pos // 32 + 1)].cpu().numpy().copy())
            idx.close()
    finally:
        os.environ.pop("SBWTGPU_PATH_RANK_LIMIT", None)
        if old is None:
            os.environ.pop("SBWTGPU_PATH_RANK", None)
        else:
            os.environ["SBWTGPU_PATH_RANK"] = old
    ref = images["0"]
    assert ref[0] > 0
    for mode in ("2", None, "gave_up"):
        got = images[mode]
        assert got[0] == ref[0] and got[1] == ref[1], (mode, got[:2], ref[:2])
        for a, b_ in zip(got[2:], ref[2:]):
            assert np.array_equal(a, b_), mode
