"""CPU: host-side pieces either side of the hot path -- the sort-based builder (differentially
against the oracle's literal restatement of NodeBOSSInMemoryConstructor, like the reference tests
KMC vs in-memory construction, tests/test_small.hh:65-99), the index file format, the sequence
reader."""
import gzip
import json
import os
import random
import struct

import numpy as np
import pytest

from oracle import OracleIndex
from sbwt_amd import hostlib, synth

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "ref_kats.json")))


def same_bits(b, o):
    return (b.n_nodes == o.n_nodes and b.n_kmers == o.n_kmers
            and all(np.array_equal(x, y) for x, y in zip(b.cols, o.columns()))
            and ((b.ssup is None and not o.has_streaming_support) or np.array_equal(b.ssup, o.ssup_words())))


def test_builder_cli_kat_columns():
    kat = KATS["cli_end_to_end"]
    seqs = [s.encode() for s in kat["seqs"]]
    b = hostlib.build_bits(seqs, kat["k"], True, True)
    assert (b.n_nodes, b.n_kmers) == (87, 73)
    assert same_bits(b, OracleIndex.build(seqs, kat["k"], True, True, 0))


@pytest.mark.parametrize("case", KATS["small_cases"]["cases"], ids=lambda c: c["name"])
def test_builder_small_cases(case):
    seqs = [s.encode() for s in case["seqs"]]
    for ssup in (True, False):
        assert same_bits(hostlib.build_bits(seqs, case["k"], False, ssup), OracleIndex.build(seqs, case["k"], ssup))


@pytest.mark.parametrize("k", [2, 3, 7, 30, 31, 32, 33, 63, 64])
def test_builder_vs_oracle_random(k):
    rnd = random.Random(k)
    g0 = synth.random_genome(4000, k)
    seqs = [g0.tobytes(), synth.mutate(g0, 0.05, k + 1).tobytes(), b"ACGTNNACGTTTGAnnACC" * 7,
            bytes(rnd.choice(b"ACGT") for _ in range(k)), b"ACG"]
    for rc in (False, True):
        b = hostlib.build_bits(seqs, k, rc, True, n_threads=3)
        assert same_bits(b, OracleIndex.build(seqs, k, True, rc, 0)), (k, rc)


def test_builder_many_dummies_and_threads():
    rnd = random.Random(3)
    seqs = [bytes(rnd.choice(b"ACGT") for _ in range(12)) for _ in range(3000)]   # read-set like input
    b1 = hostlib.build_bits(seqs, 11, False, True, n_threads=1)
    b8 = hostlib.build_bits(seqs, 11, False, True, n_threads=8)
    assert same_bits(b1, OracleIndex.build(seqs, 11, True, False, 0))
    assert all(np.array_equal(x, y) for x, y in zip(b1.cols, b8.cols)) and np.array_equal(b1.ssup, b8.ssup)


def test_index_file_roundtrip_and_layout(tmp_path):
    kat = KATS["cli_end_to_end"]
    seqs = [s.encode() for s in kat["seqs"]]
    orc = OracleIndex.build(seqs, kat["k"], True, True, kat["precalc"])
    path = str(tmp_path / "kat.sbwt")
    hostlib.write_index_file(path, orc.columns(), orc.ssup_words(), orc.C, orc.precalc(), orc.precalc_k, orc.n_nodes,
                             orc.n_kmers, orc.k)
    f = hostlib.read_index_file(path)
    assert (f.n_nodes, f.n_kmers, f.k, f.precalc_k, f.C) == (87, 73, 6, 4, [1, 25, 43, 59])
    assert all(np.array_equal(x, y) for x, y in zip(f.cols, orc.columns()))
    assert np.array_equal(f.ssup, orc.ssup_words()) and np.array_equal(f.precalc, orc.precalc())
    # byte layout of SURVEY App. A
    raw = open(path, "rb").read()
    pos = 0

    def take(n):
        nonlocal pos
        b = raw[pos:pos + n]
        pos += n
        return b
    assert struct.unpack("<q", take(8))[0] == 12 and take(12) == b"plain-matrix"
    assert struct.unpack("<q", take(8))[0] == 4 and take(4) == b"v0.1"
    nw = (87 + 63) // 64
    for c in range(4):
        assert struct.unpack("<Q", take(8))[0] == 87
        assert np.array_equal(np.frombuffer(take(8 * nw), dtype=np.uint64), orc.columns()[c])
    for c in range(4):                                  # rank_support_v5: 2 * ((cap >> 11) + 1) words
        bits = struct.unpack("<Q", take(8))[0]
        assert bits == 64 * 2 * (((64 * nw) >> 11) + 1)
        words = np.frombuffer(take(bits // 8), dtype=np.uint64)
        col = orc.columns()[c]
        assert words[0] == 0
        ones6 = sum(bin(int(w)).count("1") for w in col[:6]) if nw >= 6 else None
        if nw == 2:                                     # fewer than 6 words: no packed field is reached
            assert words[1] == 0
    assert struct.unpack("<Q", take(8))[0] == 87        # suffix_group_starts
    take(8 * nw)
    assert struct.unpack("<q", take(8))[0] == 32 and struct.unpack("<4q", take(32)) == (1, 25, 43, 59)
    assert struct.unpack("<q", take(8))[0] == 16 * 4 ** 4
    take(16 * 4 ** 4)
    assert struct.unpack("<4q", take(32)) == (4, 87, 73, 6)
    assert pos == len(raw)


def test_index_file_v5_directory_values(tmp_path):
    # a bigger vector so that superblocks and packed 12-bit fields are exercised
    g = synth.random_genome(30_000, 9)
    b = hostlib.build_bits([g.tobytes()], 20, False, True)
    path = str(tmp_path / "big.sbwt")
    hostlib.write_index_file(path, b.cols, b.ssup, [1, 2, 3, 4], None, 0, b.n_nodes, b.n_kmers, 20)
    raw = open(path, "rb").read()
    nw = (b.n_nodes + 63) // 64
    pos = 8 + 12 + 8 + 4 + 4 * (8 + 8 * nw)
    for c in range(4):
        bits = struct.unpack_from("<Q", raw, pos)[0]
        words = np.frombuffer(raw, dtype=np.uint64, count=bits // 64, offset=pos + 8)
        pos += 8 + bits // 8
        pc = np.array([bin(int(w)).count("1") for w in b.cols[c]], dtype=np.int64)
        cum = np.concatenate([[0], np.cumsum(pc)])
        for sb in range(len(words) // 2):
            first = sb * 32
            if first <= nw:
                assert words[2 * sb] == cum[min(first, nw)]
            for blk in range(1, 6):
                w_end = first + 6 * blk
                if w_end <= nw and first + 32 <= nw:    # a complete superblock: every field is defined
                    field = (int(words[2 * sb + 1]) >> (60 - 12 * blk)) & 0x7FF
                    assert field == cum[w_end] - cum[first]
    f = hostlib.read_index_file(path)
    assert not hasattr(f, "x") and f.precalc is None and f.n_nodes == b.n_nodes


@pytest.mark.parametrize("n_bits", [1, 63, 64, 383, 384, 385, 2047, 2048, 2049, 4096 + 777, 100_000])
def test_written_rank_support_equals_oracle_directory_and_serves_ranks(tmp_path, n_bits):
    """f1 (upstream file format): no upstream-built .sbwt exists offline, so the sdsl members cannot be pinned to golden
    bytes; what CAN be checked here: the rank_support_v5 blob the writer emits is word for word the directory the oracle
    builds (oracle/sbwt_oracle.c, orc_bitvec_init), and a rank computed from the WRITTEN bytes alone with SURVEY App. A's
    formula (superblock word + 11-bit field + whole-word popcounts + masked popcount) equals the prefix popcount at
    every tested position -- so the only missing piece is an upstream file to compare with."""
    rng = np.random.default_rng(n_bits)
    nw = (n_bits + 63) // 64
    cols = [rng.integers(0, 2**64, size=nw, dtype=np.uint64) for _ in range(4)]
    for c in cols:                                            # bits past n_bits are zero in a bit_vector
        if n_bits & 63:
            c[-1] &= np.uint64((1 << (n_bits & 63)) - 1)
    path = str(tmp_path / "rs.sbwt")
    hostlib.write_index_file(path, cols, None, [1, 2, 3, 4], None, 0, n_bits, 0, 3)
    raw = open(path, "rb").read()
    orc = OracleIndex.from_bits(cols[0], cols[1], cols[2], cols[3], None, n_bits, 3, 0, 0)
    pos = 8 + 12 + 8 + 4 + 4 * (8 + 8 * nw)
    for c in range(4):
        bits = struct.unpack_from("<Q", raw, pos)[0]
        words = np.frombuffer(raw, dtype=np.uint64, count=bits // 64, offset=pos + 8)
        pos += 8 + bits // 8
        bv = orc._p.contents.col[c]
        assert bv.n_dir == len(words) == 2 * (((64 * nw) >> 11) + 1)
        assert np.array_equal(words, np.ctypeslib.as_array(bv.dir, shape=(bv.n_dir,)))
        # App. A: rank(idx) from the written directory and the written bit vector words only
        data = np.frombuffer(raw, dtype=np.uint64, count=nw, offset=8 + 12 + 8 + 4 + c * (8 + 8 * nw) + 8)
        pc = np.array([bin(int(w)).count("1") for w in data], dtype=np.int64)
        cum = np.concatenate([[0], np.cumsum(pc)])
        idxs = np.unique(np.concatenate([[0, n_bits, n_bits - 1, min(n_bits, 384), min(n_bits, 2048)],
                                         rng.integers(0, n_bits + 1, size=300)]))
        for idx in idxs:
            idx = int(idx)
            p0, p1 = int(words[(idx >> 11) * 2]), int(words[(idx >> 11) * 2 + 1])
            blk = (idx & 0x7FF) // 384
            r = p0 + ((p1 >> (60 - 12 * blk)) & 0x7FF)
            first = ((idx >> 11) << 5) + blk * 6
            for w in range(first, idx >> 6):
                r += int(pc[w])
            if idx & 63:
                r += bin(int(data[idx >> 6]) & ((1 << (idx & 63)) - 1)).count("1")
            want = int(cum[idx >> 6]) + (bin(int(data[idx >> 6]) & ((1 << (idx & 63)) - 1)).count("1") if idx & 63 else 0)
            assert r == want == orc.rank(idx, b"ACGT"[c:c + 1]), (n_bits, c, idx)


def test_index_file_without_streaming_support_and_errors(tmp_path):
    seqs = [b"CCCGTGATGGCTA", b"TAATGCTGTAGC"]
    orc = OracleIndex.build(seqs, 4, False, False, 2)
    path = str(tmp_path / "nossup.sbwt")
    hostlib.write_index_file(path, orc.columns(), None, orc.C, orc.precalc(), 2, orc.n_nodes, orc.n_kmers, 4)
    f = hostlib.read_index_file(path)
    assert f.ssup is None and f.precalc_k == 2
    bad = str(tmp_path / "bad.sbwt")
    raw = bytearray(open(path, "rb").read())
    raw[8 + 12 + 8] = ord("x")                          # version string "v0.1" -> "x0.1"
    open(bad, "wb").write(raw)
    with pytest.raises(RuntimeError, match="incompatible version"):
        hostlib.read_index_file(bad)
    with pytest.raises(RuntimeError, match="Error opening file"):
        hostlib.read_index_file(str(tmp_path / "missing.sbwt"))


def test_sequence_reader_formats(tmp_path):
    reads = [b"GGAGAACTAGTGTAGCTACAAAGAGAG", b"AGTGTGTAGCAAAATGTGCTGATGCTAGCAAAAAAAA", b"CTCTACACACTTC", b"acgtNNac"]
    fq = b"".join(b"@r%d desc\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))
    fa = b"".join(b">r%d\n%s\n%s\n" % (i, r[:10], r[10:]) for i, r in enumerate(reads))   # multi-line FASTA
    files = {"q.fq": fq, "q.fna": fa, "q.fastq": fq, "q.fa": fa}
    for name, data in list(files.items()):
        (tmp_path / name).write_bytes(data)
        with gzip.open(str(tmp_path / (name + ".gz")), "wb") as g:
            g.write(data)
        files[name + ".gz"] = data
    want = b"".join(r.upper() for r in reads)
    for name in files:
        bases, off = hostlib.read_sequences(str(tmp_path / name))
        assert bases.tobytes() == want, name
        assert list(np.diff(off)) == [len(r) for r in reads], name
    (tmp_path / "q.txt").write_bytes(fa)
    with pytest.raises(RuntimeError, match="Unknown file format"):
        hostlib.read_sequences(str(tmp_path / "q.txt"))


def test_example_queries_fastq_shape():
    # example_data/queries.fastq of the reference: 5000 reads x 100 bp with 21 N; the file is not
    # shipped here, so a synthetic stand-in with the same shape goes through the reader
    import tempfile
    bases, off = synth.sample_reads([synth.random_genome(100_000, 1)], 5000, 100, 0.0, 42)
    bases = synth.inject(bases, 21, ord("N"), 3)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "queries.fastq")
        with open(p, "wb") as f:
            for r in range(5000):
                f.write(b"@q%d\n" % r + bases[off[r]:off[r + 1]].tobytes() + b"\n+\n" + b"F" * 100 + b"\n")
        b2, o2 = hostlib.read_sequences(p)
    assert np.array_equal(b2, bases) and np.array_equal(o2, off)


@pytest.mark.parametrize("k", [65, 100, 128, 129, 200, 255])
def test_builder_long_kmers_vs_bruteforce(k):
    # k > 64 (the reference supports up to 255, CMakeLists.txt:70-78): multi-word keys in the builder,
    # checked against the definition-level brute force (the oracle's own builder stops at 64)
    from bruteforce import BruteSBWT, int_to_words
    rnd = random.Random(1000 + k)
    g = "".join(rnd.choice("ACGT") for _ in range(k + 120))
    seqs = [g, g[:60] + ("A" if g[60] != "A" else "G") + g[61:], g[10:k + 9], "ACGTN" * 10]
    for rc in (False, True):
        b = hostlib.build_bits([s.encode() for s in seqs], k, rc, True, n_threads=2)
        br = BruteSBWT(seqs, k, add_revcomp=rc)
        cols, ssup = br.columns()
        assert b.n_nodes == len(br.nodes) and b.n_kmers == len(br.kmers)
        for x, y in zip(b.cols, cols):
            assert np.array_equal(x, int_to_words(y, b.n_nodes))
        assert np.array_equal(b.ssup, int_to_words(ssup, b.n_nodes))



def test_parallel_gzip_writer_roundtrip(tmp_path):
    # Buffered_ofstream (the CLI's writer) compresses 1 MiB blocks on several threads into a multi-member
    # gzip file: it must read back as one stream with Python's gzip and with the C++ reader (zlib gzread)
    rng = np.random.default_rng(1)
    for n in (0, 1, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, 5 * (1 << 20) + 123):
        seq = synth.ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)].tobytes()
        data = (b">x\n" + seq + b"\n") if n else b""
        for threads in (1, 3):
            p = str(tmp_path / ("w%d_%d.fna.gz" % (n, threads)))
            hostlib.write_file(p, data, gzip_output=True, n_threads=threads)
            assert gzip.open(p, "rb").read() == data
            bases, off = hostlib.read_sequences(p)
            assert bases.tobytes() == seq and len(off) == (2 if n else 1)
    p = str(tmp_path / "plain.fna")
    hostlib.write_file(p, b">y\nACGT\n", gzip_output=False)
    assert open(p, "rb").read() == b">y\nACGT\n"


def test_writer_replaces_existing_files_without_truncating_on_open(tmp_path):
    """Buffered_ofstream opens without O_TRUNC (truncating -- even an empty file, as check_writable() leaves one -- makes ext4
    write the whole output back inside close(): 0.77 s of a 10 M-read search) and cuts the file to what it wrote when it
    closes: over a longer file, a shorter one, an empty one, none at all, plain and gzip, large pieces and small ones --
    the result must be exactly the new contents, on the same inode."""
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, size=3 * (1 << 20) + 17, dtype=np.uint8).tobytes()      # goes straight to the file
    small = b"-1 -1 7 \n" * 1000                                                          # gathered first
    for new in (big, small, b""):
        for old in (None, b"", b"x" * 10, b"y" * (5 << 20)):
            p = str(tmp_path / "out.txt")
            if os.path.exists(p):
                os.unlink(p)
            ino = None
            if old is not None:
                open(p, "wb").write(old)
                ino = os.stat(p).st_ino
            hostlib.write_file(p, new, gzip_output=False)
            assert open(p, "rb").read() == new, (len(new), None if old is None else len(old))
            if ino is not None:
                assert os.stat(p).st_ino == ino
            if old is not None and len(old) > (1 << 20):
                hostlib.write_file(p, new, gzip_output=True, n_threads=2)              # gzip over the longer plain file
                assert gzip.open(p, "rb").read() == new


def test_chunked_reader_equals_the_sequential_one(tmp_path):
    """read_file_chunked (the CLI's reader for plain regular files: pieces cut at record starts, parsed by several threads,
    put together in order) against the sequential Reader on files made to confuse the cut: quality lines that begin with
    '@' or '+', headers with '+' and '@' inside, CRLF line ends, lower case, N, no newline at the end, long and short
    records, multi-line FASTA -- and a record without bases, which ends the stream for both."""
    rng = random.Random(11)

    def fastq(n, crlf=False, empty_at=None, final_newline=True):
        nl = b"\r\n" if crlf else b"\n"
        out = []
        for i in range(n):
            ln = rng.choice([1, 5, 30, 31, 150, 151, 400]) if i != empty_at else 0
            seq = bytes(rng.choice(b"ACGTacgtN") for _ in range(ln))
            qual = bytes(rng.choice(b"@+I#5>") for _ in range(ln))
            if ln and rng.random() < 0.3:
                qual = rng.choice([b"@", b"+"]) + qual[1:]
            hdr = b"@r%d %s" % (i, rng.choice([b"", b"+x", b"@y", b"a b+c@d"]))
            out.append(hdr + nl + seq + nl + b"+" + rng.choice([b"", b"r%d" % i]) + nl + qual + nl)
        data = b"".join(out)
        return data if final_newline else data.rstrip(b"\r\n")

    def fasta(n, crlf=False):
        nl = b"\r\n" if crlf else b"\n"
        out = []
        for i in range(n):
            ln = rng.choice([1, 60, 61, 500, 5000])
            seq = bytes(rng.choice(b"ACGTacgtN") for _ in range(ln))
            width = rng.choice([60, 70, 100000])
            lines = [seq[j:j + width] for j in range(0, ln, width)]
            out.append(b">s%d @x +y" % i + nl + nl.join(lines) + nl)
        return b"".join(out)

    cases = [("a.fastq", fastq(3000)), ("b.fq", fastq(2000, crlf=True)), ("c.fastq", fastq(1500, final_newline=False)),
             ("d.fastq", fastq(2500, empty_at=1700)), ("e.fastq", fastq(1)), ("f.fastq", b""),
             ("g.fna", fasta(400)), ("h.fasta", fasta(300, crlf=True)), ("i.fa", fasta(1))]
    for name, data in cases:
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        want_b, want_off = hostlib.read_sequences(p)
        for chunk, threads in ((4096, 1), (5000, 3), (70_000, 4), (1 << 20, 2), (1 << 30, 2)):
            got = hostlib.read_sequences_chunked(p, chunk, threads)
            assert got is not None, name
            assert np.array_equal(got[1], want_off), (name, chunk, threads)
            assert np.array_equal(got[0], want_b), (name, chunk, threads)
    # (pieces of 4 KB .. 1 GB: hundreds of cuts per file, a few, none; then a larger file with pieces of 1 MiB)
    p = str(tmp_path / "big.fastq")
    open(p, "wb").write(fastq(40000))
    want_b, want_off = hostlib.read_sequences(p)
    got = hostlib.read_sequences_chunked(p, 1 << 20, 4)
    assert np.array_equal(got[1], want_off) and np.array_equal(got[0], want_b)
    # what cannot be cut is refused (the CLI then reads it sequentially): gzip files, gzip data under a plain name
    pz = str(tmp_path / "z.fastq.gz")
    gzip.open(pz, "wb").write(fastq(10))
    assert hostlib.read_sequences_chunked(pz, 1 << 20, 2) is None
    pz2 = str(tmp_path / "z2.fastq")
    open(pz2, "wb").write(open(pz, "rb").read())
    assert hostlib.read_sequences_chunked(pz2, 1 << 20, 2) is None


def test_reader_and_writer_on_pipes(tmp_path):
    """Non-seekable input and output: `-q <(zcat reads.fq.gz)`-style FIFOs feed the reader, and the writer
    appends to a pipe (`-o /dev/stdout | ...`), like the reference's ifstream / ofstream do.  The reader's gzip probe
    keeps the bytes it read instead of rewinding; the writer uses write(2), not pwrite."""
    import threading
    reads = [b"ACGTACGTAC", b"ttgaN", b"G" * 70000]
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))
    fifo = str(tmp_path / "in.fastq")
    os.mkfifo(fifo)

    def feed():
        with open(fifo, "wb") as f:
            f.write(fq)
    t = threading.Thread(target=feed)
    t.start()
    bases, off = hostlib.read_sequences(fifo)
    t.join()
    assert len(off) == len(reads) + 1
    assert bases.tobytes() == b"".join(r.upper() for r in reads)
    # one-byte and empty inputs through a FIFO (the probe read less than it asked for)
    for content, want in ((b">", 1), (b"", 0)):
        fa = str(tmp_path / ("in%d.fna" % len(content)))
        os.mkfifo(fa)
        t = threading.Thread(target=lambda: open(fa, "wb").write(content))
        t.start()
        _, off = hostlib.read_sequences(fa)
        t.join()
        assert len(off) == 1                                 # no reads either way (a header without bases ends the stream)
    # gzip data arriving on a pipe under a plain name: a loud error, not text parsing
    fz = str(tmp_path / "z.fastq")
    os.mkfifo(fz)
    t = threading.Thread(target=lambda: open(fz, "wb").write(gzip.compress(fq)))
    t.start()
    with pytest.raises(RuntimeError):
        hostlib.read_sequences(fz)
    t.join()
    # the writer on a FIFO, plain and gzip
    for gz in (False, True):
        out = str(tmp_path / ("out%d" % gz))
        os.mkfifo(out)
        got = []
        t = threading.Thread(target=lambda: got.append(open(out, "rb").read()))
        t.start()
        data = os.urandom(3_000_000) + b"tail"
        hostlib.write_file(out, data, gz, 2)
        t.join()
        assert (gzip.decompress(got[0]) if gz else got[0]) == data


def test_host_rank_directory_of_the_scalar_api():
    """The C++ mirror answers SubsetMatrixRank::rank of ONE position on the host from the rank_support_v5 directory it
    serialises (host/bitvector.hh).  Its arithmetic -- superblock count + 12-bit block field + whole words + masked
    word -- against a numpy prefix popcount: every position around word, 384-bit block and 2048-bit superblock boundaries,
    pos == n_bits, vectors whose length is and is not a multiple of 64 / 2048, all-ones and all-zeros."""
    rng = np.random.default_rng(21)
    for n_bits in (0, 1, 63, 64, 65, 383, 384, 385, 2047, 2048, 2049, 6 * 64 * 7 + 5, 4096, 100_003, 1 << 17):
        nw = (n_bits + 63) // 64
        for kind in ("random", "ones", "zeros"):
            if kind == "random":
                w = rng.integers(0, 1 << 63, size=nw, dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, size=nw).astype(np.uint64)
            elif kind == "ones":
                w = np.full(nw, np.uint64(0xFFFFFFFFFFFFFFFF))
            else:
                w = np.zeros(nw, dtype=np.uint64)
            if n_bits & 63 and nw:
                w[-1] &= np.uint64((1 << (n_bits & 63)) - 1)
            bits = np.unpackbits(w.view(np.uint8), bitorder="little")[:n_bits] if nw else np.zeros(0, np.uint8)
            prefix = np.concatenate([[0], np.cumsum(bits.astype(np.int64))])
            pos = np.unique(np.concatenate([np.arange(0, min(n_bits, 5000) + 1), rng.integers(0, n_bits + 1, size=3000),
                                            np.array([n_bits, max(n_bits - 1, 0), n_bits // 2])]))
            pos = pos[(pos >= 0) & (pos <= n_bits)]
            got = hostlib.rank_batch(w, n_bits, pos)
            assert np.array_equal(got, prefix[pos]), (n_bits, kind)
    with pytest.raises(RuntimeError):
        hostlib.rank_batch(np.zeros(1, np.uint64), 10, [11])
