"""Pins the oracle (oracle/sbwt_oracle.c) against every known-answer vector the reference's own
tests hold for the search path (tests/golden/ref_kats.json) and against a definition-level pure
Python brute force.  CPU only."""
import json
import os
import random

import numpy as np
import pytest

from bruteforce import BruteSBWT, int_to_words, kmer_set
from oracle import OracleIndex, print_vector

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "ref_kats.json")))


def b(s):
    return s.encode()


def test_cli_end_to_end_exact_output():
    # tests/test_CLI.hh:90 -- the strongest KAT: construction order, C array, precalc, streaming, format
    kat = KATS["cli_end_to_end"]
    idx = OracleIndex.build([b(s) for s in kat["seqs"]], kat["k"], True, kat["add_reverse_complements"],
                            kat["precalc"])
    assert idx.n_nodes == 87 and idx.n_kmers == 73 and idx.C == [1, 25, 43, 59]
    got = b"".join(print_vector(idx.streaming_search(b(q))) for q in kat["queries"])
    assert got == kat["expected_output"].encode()
    # non-streaming loop gives the same lines (sbwt_search.cpp:67-91)
    got2 = b"".join(print_vector(idx.search_all(b(q))) for q in kat["queries"])
    assert got2 == kat["expected_output"].encode()


def test_redundant_dummies_nine_columns():
    kat = KATS["redundant_dummies"]
    idx = OracleIndex.build([b(s) for s in kat["seqs"]], kat["k"], False)
    assert idx.n_nodes == kat["n_subsets"]


def test_partial_search():
    kat = KATS["partial_search"]
    idx = OracleIndex.build([b(s) for s in kat["seqs"]], kat["k"], False)
    (l, r), n = idx.partial_search(b(kat["query"]))
    assert n == kat["matched_len"]
    brute = BruteSBWT(kat["seqs"], kat["k"])
    for i, label in enumerate(brute.nodes):     # tests/test_small.hh:117-124
        assert (l <= i <= r) == label.endswith(kat["interval_suffix"])


def all_kmers(k):
    for mask in range(4 ** k):
        yield "".join("ACGT"[(mask >> (2 * i)) & 3] for i in range(k))


def check_bits_against_bruteforce(idx: OracleIndex, brute: BruteSBWT):
    assert idx.n_nodes == len(brute.nodes)
    cols, ssup = brute.columns()
    for got, want in zip(idx.columns(), cols):
        assert np.array_equal(got, int_to_words(want, idx.n_nodes))
    if idx.has_streaming_support:
        assert np.array_equal(idx.ssup_words(), int_to_words(ssup, idx.n_nodes))


def random_cases():
    rnd = random.Random(247829347)           # tests/setup_tests.hh:120 seed
    lots = ["".join(rnd.choice("ACGT") for _ in range(6)) for _ in range(20)]   # lots_of_dummies
    return [{"name": "lots_of_dummies", "k": 6, "seqs": lots},
            {"name": "random_k5", "k": 5, "seqs": ["".join(rnd.choice("ACGT") for _ in range(60)) for _ in range(3)]}]


@pytest.mark.parametrize("case", KATS["small_cases"]["cases"] + random_cases(), ids=lambda c: c["name"])
@pytest.mark.parametrize("precalc", [0, 2])
def test_small_cases_exhaustive(case, precalc):
    # run_small_testcase + check_all_queries (tests/test_small.hh:24-43,65-99)
    k, seqs = case["k"], case["seqs"]
    idx = OracleIndex.build([b(s) for s in seqs], k, True, False, min(precalc, k))
    brute = BruteSBWT(seqs, k)
    check_bits_against_bruteforce(idx, brute)
    truth = kmer_set(seqs, k)
    for kmer in all_kmers(k):
        col = idx.search(b(kmer))
        if kmer in truth:
            assert col >= 0 and col == brute.rank_of[kmer]
        else:
            assert col == -1
    assert idx.search(b("N" * k)) == -1
    # the recomputed suffix-group marks equal the stored ones (tests/test_large.hh:89-92)
    assert np.array_equal(idx.mark_suffix_groups(), idx.ssup_words())


def test_serialization_strings_streaming_and_N():
    # tests/test_small.hh:324-428: input with NN, precalc 2, streaming found <=> in truth set, 100xN -> -1
    kat = KATS["serialization_strings"]
    k = kat["k"]
    idx = OracleIndex.build([b(s) for s in kat["seqs"]], k, True, False, kat["precalc"])
    truth = {w for w in kmer_set(kat["seqs"], k)}
    for kmer in all_kmers(k):
        assert (idx.search(b(kmer)) >= 0) == (kmer in truth)
    rnd = random.Random(5)
    inputs = kat["seqs"] + ["".join(rnd.choice("ACGT") for _ in range(100))]
    for s in inputs:
        res = idx.streaming_search(b(s))
        assert len(res) == len(s) - k + 1
        for i, x in enumerate(res):
            assert (x >= 0) == (s[i:i + k] in truth)
    assert list(idx.streaming_search(b"N" * 100)) == [-1] * (100 - k + 1)


def test_api_example_consistency():
    kat = KATS["api_example"]
    idx = OracleIndex.build([b(s) for s in kat["seqs"]], kat["k"], True, False, kat["precalc"])
    brute = BruteSBWT(kat["seqs"], kat["k"])
    assert idx.search(b(kat["search"])) == brute.search(kat["search"])
    assert list(idx.streaming_search(b(kat["streaming"]))) == brute.search_all(kat["streaming"])


def test_streaming_equals_search_and_forward_on_random_genome():
    # tests/test_large.hh:104-115 (streaming == per-k-mer search) and :126-170 (forward consistency)
    from sbwt_amd import synth
    k = 30
    genomes = [synth.random_genome(20000, 1)]
    genomes.append(synth.mutate(genomes[0], 0.05, 2))
    idx = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 8)
    bases, off = synth.sample_reads(genomes, 300, 150, 0.01, 42)
    bases = synth.inject(bases, 20, ord("N"), 7)
    for r in range(300):
        s = bases[off[r]:off[r + 1]].tobytes()
        assert np.array_equal(idx.streaming_search(s), idx.search_all(s))
    all_kmers_set = set()
    for g in genomes:
        gb = g.tobytes()
        for i in range(len(gb) - k + 1):
            all_kmers_set.add(gb[i:i + k])
    truth = sorted(all_kmers_set)[::37]
    rnd = random.Random(12514)
    for kmer in truth[:400]:
        col = idx.search(kmer)
        assert col >= 0
        for c in b"ACGT":
            nxt = kmer[1:] + bytes([c])
            want = idx.search(nxt) if nxt in all_kmers_set else -1     # tests/test_large.hh:142-149
            assert idx.forward(col, bytes([c])) == want
    for _ in range(2000):                          # random absent k-mers -> -1 (tests/test_large.hh:157-168)
        kmer = bytes(rnd.choice(b"ACGT") for _ in range(k))
        if kmer not in all_kmers_set:
            assert idx.search(kmer) == -1
    assert np.array_equal(idx.mark_suffix_groups(), idx.ssup_words())


def test_quirks_lowercase_and_short_reads():
    # Q1/Q2/Q3/Q9 of SURVEY 8a
    seqs = ["ACGTACGTTGCAGTCAGTCCATG"]
    idx = OracleIndex.build([b(s) for s in seqs], 5, True, False, 2)
    up = idx.streaming_search(b"ACGTACGTTGCAGTC")
    assert (up >= 0).all()
    # a lower-case char inside the first k-mer makes the full search fail (raw char validated) ...
    low_first = idx.streaming_search(b"ACgTACGTTGCAGTC")
    assert low_first[0] == -1 and low_first[1] == -1 and low_first[2] == -1 and low_first[3] >= 0
    # ... but a lower-case char consumed by a streaming step is accepted (toupper'd)
    low_stream = idx.streaming_search(b"ACGTACgTTGCAGTC")
    assert np.array_equal(low_stream, up)
    assert len(idx.streaming_search(b"ACGT")) == 0       # shorter than k -> empty
    assert len(idx.search_all(b"ACGT")) == 0


def test_print_vector_format():
    assert print_vector(np.array([-1, 74, 0, 5], dtype=np.int64)) == b"-1 74  5 \n"   # 0 prints as empty token
    assert print_vector(np.array([], dtype=np.int64)) == b"\n"


def test_precalc_limits():
    idx = OracleIndex.build([b"ACGTACGTTGCAGTCAGTCCATG"], 5, True)
    assert idx.do_precalc(21) == -1       # SBWT.hh:619-621
    assert idx.do_precalc(6) == -2        # SBWT.hh:623-624
    assert idx.do_precalc(5) == 0
    for kmer in all_kmers(5):
        pass


def load_reference_queries():
    import gzip
    data = gzip.open(os.path.join(HERE, "golden", "queries_seqs.txt.gz"), "rb").read().split(b"\n")
    return [s for s in data if s]


def test_reference_query_file_streaming_equals_search():
    # TEST_LARGE.streaming_queries (tests/test_large.hh:104-115) on the reference's own example_data/queries.fastq
    # reads (5000 x 100 bp, 21 N).  coli3.fna is not in the checkout, so the index is built from the first
    # 3000 reads (+ reverse complements): real, heavily overlapping reads give multi-member suffix groups
    # and branching that random genomes do not.
    reads = load_reference_queries()
    assert len(reads) == 5000 and all(len(r) == 100 for r in reads) and sum(r.count(b"N") for r in reads) == 21
    idx = OracleIndex.build(reads[:3000], 30, True, True, 8)
    assert np.array_equal(idx.mark_suffix_groups(), idx.ssup_words())
    hits = 0
    for r in reads[::7]:
        res = idx.streaming_search(r)
        assert np.array_equal(res, idx.search_all(r))
        hits += int((res >= 0).sum())
    assert hits > 10000


def test_oracle_cli_loop_reproduces_the_reference_cli_output(tmp_path):
    """orc_search_file (the CPU end-to-end figure of bench.py) = run_file + print_vector, src/CLI/sbwt_search.cpp:21-105:
    on the reference's own CLI known-answer test (tests/test_CLI.hh:21-22,43,49,90) it writes the exact expected text,
    from FASTQ and from multi-line FASTA, lower-case input upper-cased on read."""
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kats.json")))["cli_end_to_end"]
    orc = OracleIndex.build([s.encode() for s in kat["seqs"]], kat["k"], True, True, kat["precalc"])
    fq = tmp_path / "q.fastq"
    fq.write_bytes(b"".join(b"@r\n%s\n+\n%s\n" % (q.encode(), b"I" * len(q)) for q in kat["queries"]))
    wall, qsecs, nr, nk = orc.search_file(str(fq), str(tmp_path / "o1.txt"))
    assert (tmp_path / "o1.txt").read_bytes() == kat["expected_output"].encode()
    assert nr == len(kat["queries"]) and nk == sum(max(0, len(q) - kat["k"] + 1) for q in kat["queries"]) and 0 <= qsecs <= wall
    fa = tmp_path / "q.fna"
    fa.write_bytes(b"".join(b">r\n%s\n%s\n" % (q[:7].lower().encode(), q[7:].encode()) for q in kat["queries"]))
    orc.search_file(str(fa), str(tmp_path / "o2.txt"))
    assert (tmp_path / "o2.txt").read_bytes() == kat["expected_output"].encode()
