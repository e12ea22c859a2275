"""GPU: index construction on the device: sbwtgpu_build_plain_matrix must give the bits of the
reference's constructor -- compared with the oracle's literal restatement of NodeBOSSInMemoryConstructor.hh:98-213 on
small inputs (incl. the reference's own known answers) and with the host sort-based builder on larger ones -- and
index_create's device-side block construction must serve the same ranks as before."""
import json
import os

import numpy as np
import pytest

from oracle import OracleIndex
from sbwt_amd import capi, hostlib, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "ref_kats.json")))


def same_bits(a, b_cols, b_ssup, n_nodes):
    nw = (n_nodes + 63) // 64
    for x, y in zip(a.cols, b_cols):
        assert np.array_equal(x[:nw], np.asarray(y)[:nw])
    if b_ssup is not None:
        assert np.array_equal(a.ssup[:nw], np.asarray(b_ssup)[:nw])


def test_reference_known_answers(gpu):
    kat = KATS["cli_end_to_end"]                       # tests/test_CLI.hh:21-49: with reverse complements, k=6
    seqs = [s.encode() for s in kat["seqs"]]
    got = capi.build_bits_gpu(seqs, kat["k"], True, True)
    orc = OracleIndex.build(seqs, kat["k"], True, True, 0)
    assert got.n_nodes == 87 and got.n_kmers == 73 and orc.n_nodes == 87
    same_bits(got, orc.columns(), orc.ssup_words(), 87)
    for case in KATS["small_cases"]["cases"]:          # tests/test_small.hh (5 strings k=4 -> 9 columns, ...)
        seqs = [s.encode() for s in case["seqs"]]
        if case["k"] < 2:
            continue
        got = capi.build_bits_gpu(seqs, case["k"], False, True)
        orc = OracleIndex.build(seqs, case["k"], True, False, 0)
        assert got.n_nodes == orc.n_nodes, case["name"]
        same_bits(got, orc.columns(), orc.ssup_words(), orc.n_nodes)


@pytest.mark.parametrize("k", [2, 3, 7, 16, 21, 30, 31, 32, 33, 48, 63, 64])
@pytest.mark.parametrize("rc", [False, True])
def test_random_inputs_equal_oracle_constructor(gpu, k, rc):
    rng = np.random.default_rng(100 * k + rc)
    g0 = synth.random_genome(3000, 5 + k)
    seqs = [g0.tobytes(), synth.mutate(g0, 0.03, 9).tobytes()]
    # short sequences (many dummy nodes), sequences shorter than k, N and lower case inside, an empty one, a tandem repeat
    for _ in range(30):
        L = int(rng.integers(0, 3 * k + 5))
        seqs.append(synth.random_genome(L, int(rng.integers(1, 1 << 30))).tobytes())
    noisy = bytearray(synth.random_genome(400, 77).tobytes())
    noisy[50] = ord("N"); noisy[51] = ord("N"); noisy[200] = ord("a"); noisy[399] = ord("$")
    seqs += [bytes(noisy), b"", b"ACGT" * 40, b"A" * 70]
    got = capi.build_bits_gpu(seqs, k, rc, True)
    orc = OracleIndex.build(seqs, k, True, rc, 0)
    assert (got.n_nodes, got.n_kmers) == (orc.n_nodes, orc.n_kmers)
    same_bits(got, orc.columns(), orc.ssup_words(), orc.n_nodes)
    no_ssup = capi.build_bits_gpu(seqs, k, rc, False)
    assert no_ssup.ssup is None
    same_bits(no_ssup, orc.columns(), None, orc.n_nodes)


def test_empty_input_is_the_root_alone(gpu):
    got = capi.build_bits_gpu([b"ACG", b""], 5, False, True)
    assert got.n_nodes == 1 and got.n_kmers == 0 and int(got.ssup[0]) == 1
    assert all(int(c[0]) == 0 for c in got.cols)
    with pytest.raises(capi.SbwtGpuError):
        capi.build_bits_gpu([b"ACGT"], 65, False, True)     # 128-bit keys end at k = 64; the C++ host builder goes on


@pytest.mark.parametrize("k", [33, 63])
def test_long_kmers_equal_host_builder(gpu, k):
    # 32 < k <= 64: 128-bit keys (config 5's index is k = 63); the threaded host builder is the second opinion
    genomes = synth.coli3_like(300_000)
    seqs = [g.tobytes() for g in genomes] + [b"ACGT" * 30, synth.random_genome(k + 3, 4).tobytes()]
    for rc in (False, True):
        got = capi.build_bits_gpu(seqs, k, rc, True)
        host = hostlib.build_bits(seqs, k, rc, True, n_threads=8)
        assert (got.n_nodes, got.n_kmers) == (host.n_nodes, host.n_kmers)
        same_bits(got, host.cols, host.ssup, host.n_nodes)


def test_genome_scale_equals_host_builder_and_searches(gpu):
    k = 31
    genomes = synth.pan_like(6, 400_000)
    seqs = [g.tobytes() for g in genomes]
    got = capi.build_bits_gpu(seqs, k, True, True)
    host = hostlib.build_bits(seqs, k, True, True, n_threads=8)
    assert (got.n_nodes, got.n_kmers) == (host.n_nodes, host.n_kmers)
    same_bits(got, host.cols, host.ssup, host.n_nodes)
    idx = capi.Index.create(got.cols[0], got.cols[1], got.cols[2], got.cols[3], got.ssup, got.n_nodes, k, got.n_kmers, 8)
    orc = OracleIndex.from_bits(host.cols[0], host.cols[1], host.cols[2], host.cols[3], host.ssup, host.n_nodes, k,
                                host.n_kmers, 8)
    assert idx.C == orc.C                              # C array from the device-side counts (SBWT.hh:344-349)
    bases, off = synth.sample_reads(genomes, 1500, 120, 0.01, 5)
    out, _ = idx.streaming_search(bases, off)
    want = np.concatenate([orc.streaming_search(bases[off[r]:off[r + 1]].tobytes()) for r in range(1500)])
    assert np.array_equal(out, want)
    rng = np.random.default_rng(1)
    pos = rng.integers(0, got.n_nodes + 1, size=20000)
    sym = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=20000)
    assert np.array_equal(idx.rank(pos, sym), orc.batch_rank(pos, sym, 4)[0])
