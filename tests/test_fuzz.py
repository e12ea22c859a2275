"""CPU: randomized differential tests (hypothesis) of the host builder against the oracle's literal
NodeBOSSInMemoryConstructor restatement and the definition-level brute force, and of the index file
round trip.  Mirrors the reference's strategy of checking two constructors against each other
(tests/test_small.hh:65-99) on inputs the hand-written cases do not reach."""
import os
import tempfile

import numpy as np
from hypothesis import given, settings, strategies as st

from bruteforce import BruteSBWT, int_to_words
from oracle import OracleIndex
from sbwt_amd import hostlib

dna = st.text(alphabet="ACGT", min_size=0, max_size=40)
dirty = st.text(alphabet="ACGTNacgt", min_size=0, max_size=30)


@settings(max_examples=120, deadline=None)
@given(seqs=st.lists(st.one_of(dna, dirty), min_size=1, max_size=6), k=st.integers(2, 9), rc=st.booleans())
def test_builder_oracle_bruteforce_agree(seqs, k, rc):
    bs = [s.encode() for s in seqs]
    b = hostlib.build_bits(bs, k, rc, True)
    o = OracleIndex.build(bs, k, True, rc, 0)
    assert b.n_nodes == o.n_nodes and b.n_kmers == o.n_kmers
    assert all(np.array_equal(x, y) for x, y in zip(b.cols, o.columns()))
    assert np.array_equal(b.ssup, o.ssup_words())
    br = BruteSBWT(seqs, k, add_revcomp=rc)
    cols, ssup = br.columns()
    assert b.n_nodes == len(br.nodes)
    for x, y in zip(b.cols, cols):
        assert np.array_equal(x, int_to_words(y, b.n_nodes))
    assert np.array_equal(b.ssup, int_to_words(ssup, b.n_nodes))
    assert np.array_equal(o.mark_suffix_groups(), o.ssup_words())
    # oracle queries against set membership
    for s in seqs:
        if len(s) >= k:
            want = br.search_all(s)
            assert list(o.search_all(s.encode())) == want


@settings(max_examples=40, deadline=None)
@given(seqs=st.lists(dna, min_size=1, max_size=4), k=st.integers(2, 7), p=st.integers(0, 3), ssup=st.booleans())
def test_index_file_roundtrip_random(seqs, k, p, ssup):
    p = min(p, k)
    o = OracleIndex.build([s.encode() for s in seqs], k, ssup, False, p)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.sbwt")
        hostlib.write_index_file(path, o.columns(), o.ssup_words(), o.C, o.precalc() if p else None, p, o.n_nodes,
                                 o.n_kmers, k)
        f = hostlib.read_index_file(path)
    assert (f.n_nodes, f.n_kmers, f.k, f.precalc_k, f.C) == (o.n_nodes, o.n_kmers, k, p, o.C)
    assert all(np.array_equal(x, y) for x, y in zip(f.cols, o.columns()))
    assert (f.ssup is None) == (not ssup) and (not ssup or np.array_equal(f.ssup, o.ssup_words()))
    assert (p == 0 and f.precalc is None) or np.array_equal(f.precalc, o.precalc())
