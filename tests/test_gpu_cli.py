"""GPU: the `sbwt` command (C++ host mirror, sbwt_amd/csrc/host/sbwt_cli.cpp) end to end, restating
the reference's CLI.end_to_end_build_and_query (tests/test_CLI.hh:20-113): build k=6 index with
reverse complements and precalc 4 from two gz FASTA files listed in a .txt, query 3 reads in four
formats through list files, compare the output files byte for byte with the known answer, plain
and gzipped; plus the error conventions of src/CLI/sbwt.cpp:42-57."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import OracleIndex, print_vector
from sbwt_amd import hostlib, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SBWT = os.path.join(ROOT, "sbwt_amd", "bin", "sbwt")
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_kats.json")))


def run(*args, check=True, env=None):
    p = subprocess.run([SBWT] + list(args), capture_output=True, timeout=300, env=dict(os.environ, **env) if env else None)
    if check:
        assert p.returncode == 0, p.stderr.decode()
    return p


def write_fasta(path, seqs, gz=False):
    data = b"".join(b">s%d\n%s\n" % (i, s) for i, s in enumerate(seqs))
    (gzip.open(path, "wb") if gz else open(path, "wb")).write(data)


def write_fastq(path, seqs, gz=False):
    data = b"".join(b"@s%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(seqs))
    (gzip.open(path, "wb") if gz else open(path, "wb")).write(data)


def test_cli_end_to_end_build_and_query(gpu, tmp_path):
    kat = KATS["cli_end_to_end"]
    d = str(tmp_path)
    f1, f2 = d + "/a.fna.gz", d + "/b.fna.gz"
    write_fasta(f1, [s.encode() for s in kat["seqs"][:2]], gz=True)
    write_fasta(f2, [s.encode() for s in kat["seqs"][2:]], gz=True)
    open(d + "/in.txt", "w").write(f1 + "\n" + f2 + "\n")
    index = d + "/index.sbwt"
    run("build", "-i", d + "/in.txt", "-o", index, "-k", "6", "--add-reverse-complements", "--temp-dir", d,
        "--precalc-length", "4")
    f = hostlib.read_index_file(index)
    assert (f.n_nodes, f.n_kmers, f.k, f.precalc_k, f.C) == (87, 73, 6, 4, [1, 25, 43, 59])

    queries = [q.encode() for q in kat["queries"]]
    q = [d + "/q1.fq", d + "/q2.fna", d + "/q3.fq.gz", d + "/q4.fna.gz"]
    write_fastq(q[0], queries)
    write_fasta(q[1], queries)
    write_fastq(q[2], queries, gz=True)
    write_fasta(q[3], queries, gz=True)
    outs = [d + "/o%d.txt" % i for i in range(4)]
    open(d + "/qlist.txt", "w").write("\n".join(q) + "\n")
    open(d + "/olist.txt", "w").write("\n".join(outs) + "\n")
    p = run("search", "-o", d + "/olist.txt", "-i", index, "-q", d + "/qlist.txt")
    assert b"us/query: " in p.stderr and b"(excluding I/O etc)" in p.stderr and b"us/query end-to-end: " in p.stderr
    for o in outs:
        assert open(o, "rb").read() == kat["expected_output"].encode()
    open(d + "/olist_gz.txt", "w").write("\n".join(o + ".gz" for o in outs) + "\n")
    run("search", "-o", d + "/olist_gz.txt", "-i", index, "-q", d + "/qlist.txt", "--gzip-output")
    for o in outs:
        assert gzip.open(o + ".gz", "rb").read() == kat["expected_output"].encode()
    # single file form, -z short flag
    run("search", "-o", d + "/single.txt.gz", "-i", index, "-q", q[0], "-z")
    assert gzip.open(d + "/single.txt.gz", "rb").read() == kat["expected_output"].encode()


def test_cli_non_streaming_index_and_batches(gpu, tmp_path):
    d = str(tmp_path)
    genomes = [synth.random_genome(30_000, 4)]
    write_fasta(d + "/g.fna", [genomes[0].tobytes()])
    run("build", "-i", d + "/g.fna", "-o", d + "/ns.sbwt", "-k", "31", "--no-streaming-support")
    run("build", "-i", d + "/g.fna", "-o", d + "/st.sbwt", "-k", "31", "-p", "5", "-t", "4")
    bases, off = synth.sample_reads(genomes, 700, 120, 0.02, 4)
    bases = synth.inject(bases, 15, ord("N"), 1)
    reads = [bases[off[r]:off[r + 1]].tobytes() for r in range(700)] + [b"ACGT", b"A" * 31]
    write_fastq(d + "/r.fastq", reads)
    orc = OracleIndex.build([genomes[0].tobytes()], 31, True, False, 5)
    want = b"".join(print_vector(orc.streaming_search(r)) for r in reads)
    p = run("search", "-o", d + "/ns.out", "-i", d + "/ns.sbwt", "-q", d + "/r.fastq")
    assert b"Running non-streaming queries" in p.stderr
    assert open(d + "/ns.out", "rb").read() == want
    # tiny batches force many GPU round trips; the output must not depend on the batching
    p = run("search", "-o", d + "/st.out", "-i", d + "/st.sbwt", "-q", d + "/r.fastq", "--batch-bases", "1000")
    assert b"Running streaming queries" in p.stderr
    assert open(d + "/st.out", "rb").read() == want
    # sharded over three GPU contexts (the same device listed three times on a 1-GPU box): same bytes
    p = run("search", "-o", d + "/sh.out", "-i", d + "/st.sbwt", "-q", d + "/r.fastq", "--gpu-list", "0,0,0",
            "--batch-bases", "20000")
    assert b"replicated onto 3 GPU contexts" in p.stderr
    assert open(d + "/sh.out", "rb").read() == want
    p = run("search", "-o", d + "/hf.out", "-i", d + "/st.sbwt", "-q", d + "/r.fastq", "--host-format")
    assert open(d + "/hf.out", "rb").read() == want
    p = run("search", "-o", d + "/hf2.out", "-i", d + "/st.sbwt", "-q", d + "/r.fastq", "--host-format", "--gpu-list", "0,0")
    assert open(d + "/hf2.out", "rb").read() == want
    p = run("search", "-o", d + "/g1.out", "-i", d + "/ns.sbwt", "-q", d + "/r.fastq", "--gpus", "1")
    assert open(d + "/g1.out", "rb").read() == want
    p = run("search", "-o", d + "/g9.out", "-i", d + "/st.sbwt", "-q", d + "/r.fastq", "--gpus", "9", check=False)
    assert p.returncode == 1 and b"Runtime error" in p.stderr        # more GPUs than the box has
    # an output file that exists already -- longer, then a hard link's other name -- is replaced in place (the writer opens
    # without O_TRUNC and cuts the file when it closes it; host/seqio.hh): same bytes, same inode, plain and gzipped
    few = reads[:40]
    write_fastq(d + "/few.fastq", few)
    want_few = b"".join(print_vector(orc.streaming_search(r)) for r in few)
    ino = os.stat(d + "/st.out").st_ino
    os.link(d + "/st.out", d + "/st_link.out")
    run("search", "-o", d + "/st.out", "-i", d + "/st.sbwt", "-q", d + "/few.fastq")
    assert open(d + "/st.out", "rb").read() == want_few and os.stat(d + "/st.out").st_ino == ino
    assert open(d + "/st_link.out", "rb").read() == want_few
    run("search", "-o", d + "/st.out", "-i", d + "/st.sbwt", "-q", d + "/r.fastq", "-z")
    assert gzip.open(d + "/st.out", "rb").read() == want
    run("search", "-o", d + "/st.out", "-i", d + "/st.sbwt", "-q", d + "/few.fastq")
    assert open(d + "/st.out", "rb").read() == want_few


def test_cli_many_batches_keep_their_order(gpu, tmp_path):
    # two threads take the batches in turn and write their text in input order (host/sbwt_cli.cpp run_file): hundreds of
    # batches of uneven cost -- stretches of reads that are absent, reads with N, short reads -- against one batch, the
    # host-formatted output and the oracle
    d = str(tmp_path)
    genomes = [synth.random_genome(60_000, 8)]
    genomes.append(synth.mutate(genomes[0], 0.03, 9))
    write_fasta(d + "/g.fna", [g.tobytes() for g in genomes])
    run("build", "-i", d + "/g.fna", "-o", d + "/i.sbwt", "-k", "30", "-t", "4")
    bases, off = synth.sample_reads(genomes, 30_000, 150, 0.02, 10)
    bases = synth.inject(bases, 300, ord("N"), 2)
    rb, ro = synth.random_reads(3_000, 150, 11)
    reads = [bases[off[r]:off[r + 1]].tobytes() for r in range(30_000)]
    absent = [rb[ro[r]:ro[r + 1]].tobytes() for r in range(3_000)]
    mixed = reads[:10_000] + absent + [b"ACGT", b"A" * 29] + reads[10_000:]
    write_fastq(d + "/r.fastq", mixed)
    run("search", "-o", d + "/one.out", "-i", d + "/i.sbwt", "-q", d + "/r.fastq", "--batch-bases", "100000000")
    run("search", "-o", d + "/many.out", "-i", d + "/i.sbwt", "-q", d + "/r.fastq", "--batch-bases", "20000")
    run("search", "-o", d + "/host.out", "-i", d + "/i.sbwt", "-q", d + "/r.fastq", "--host-format", "--batch-bases", "300000")
    # ... and with the input parsed piece by piece by several threads (SBWT_CLI_PARSER_THREADS: off by default)
    run("search", "-o", d + "/chunked.out", "-i", d + "/i.sbwt", "-q", d + "/r.fastq", "--batch-bases", "20000",
        env={"SBWT_CLI_CHUNK_MIN_BYTES": "0", "SBWT_CLI_PARSER_THREADS": "3"})
    write_fasta(d + "/r.fna", mixed)
    run("search", "-o", d + "/chunked_fa.out", "-i", d + "/i.sbwt", "-q", d + "/r.fna", "--batch-bases", "50000",
        env={"SBWT_CLI_CHUNK_MIN_BYTES": "0", "SBWT_CLI_PARSER_THREADS": "2"})
    one = open(d + "/one.out", "rb").read()
    assert open(d + "/many.out", "rb").read() == one
    assert open(d + "/chunked.out", "rb").read() == one
    assert open(d + "/chunked_fa.out", "rb").read() == one
    assert open(d + "/host.out", "rb").read() == one
    orc = OracleIndex.build([g.tobytes() for g in genomes], 30, True, False, 8)
    lines = one.split(b"\n")
    assert len(lines) == len(mixed) + 1
    for r in list(range(0, 300)) + list(range(9_990, 10_010)) + list(range(13_000, 13_010)):
        assert lines[r] + b"\n" == print_vector(orc.streaming_search(mixed[r])), r


def test_cli_error_conventions(gpu, tmp_path):
    d = str(tmp_path)
    p = run("search", "-o", d + "/o.txt", "-i", d + "/missing.sbwt", "-q", d + "/q.fna", check=False)
    assert p.returncode == 1 and p.stderr.strip().endswith(b"Runtime error: Error opening file: " + (d + "/missing.sbwt").encode())
    p = run("frobnicate", check=False)
    assert p.returncode == 1 and b"Runtime error: Invalid command: frobnicate" in p.stderr
    p = run("search", check=False)
    assert p.returncode == 1 and b"--index-file" in p.stderr          # help text, exit(1)
    p = run(check=False)
    assert p.returncode == 0 and b"Available commands" in p.stderr
    # unknown variant string in the file
    import struct
    open(d + "/bad.sbwt", "wb").write(struct.pack("<q", 5) + b"bogus")
    write_fasta(d + "/q.fna", [b"ACGTACGT"])
    p = run("search", "-o", d + "/o.txt", "-i", d + "/bad.sbwt", "-q", d + "/q.fna", check=False)
    assert p.returncode == 1 and b"unrecognized variant" in p.stderr
    # list length mismatch (sbwt_search.cpp:111-115)
    seqs = [b"ACTAGTGTAGCTACAAA"]
    write_fasta(d + "/s.fna", seqs)
    run("build", "-i", d + "/s.fna", "-o", d + "/s.sbwt", "-k", "6", "-p", "2")
    open(d + "/ql.txt", "w").write(d + "/q.fna\n" + d + "/q.fna\n")
    open(d + "/ol.txt", "w").write(d + "/o1.txt\n")
    p = run("search", "-o", d + "/ol.txt", "-i", d + "/s.sbwt", "-q", d + "/ql.txt", check=False)
    assert p.returncode == 1 and b"Number of input and output files does not match (2 vs 1)" in p.stderr


def test_cpp_api_mirror(gpu, tmp_path):
    """Compiles tests/cpp/test_api.cpp against the host mirror headers (like api_examples/Makefile does
    against the reference) and runs it: the reference's API tests, restated, on the GPU path."""
    exe = str(tmp_path / "test_api")
    lib = os.path.join(ROOT, "sbwt_amd", "lib")
    cmd = ["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "sbwt_amd", "csrc", "host"),
           os.path.join(ROOT, "tests", "cpp", "test_api.cpp"), "-o", exe, "-L" + lib, "-lsbwtgpu", "-lz",
           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, timeout=300)
    p = subprocess.run([exe], capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert p.stdout.splitlines()[-1].startswith(b"OK "), p.stdout[-500:]
    print(p.stdout.decode())                 # includes the measured cost of the scalar API (us per call)


def test_cli_reads_from_a_fifo_and_writes_to_a_pipe(gpu, tmp_path):
    """`sbwt search -q <(zcat reads.fq.gz) -o /dev/stdout | ...`: the query arrives on a FIFO, the output leaves through a
    pipe (the reader must not rewind, the writer must not pwrite).  Same bytes as the reference's KAT, plain and
    with -z."""
    import threading
    kat = KATS["cli_end_to_end"]
    d = str(tmp_path)
    write_fasta(d + "/g.fna", [s.encode() for s in kat["seqs"]])
    index = d + "/index.sbwt"
    run("build", "-i", d + "/g.fna", "-o", index, "-k", "6", "--add-reverse-complements", "--precalc-length", "4")
    queries = [q.encode() for q in kat["queries"]]
    fq = b"".join(b"@s%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(queries))
    for zflag in ([], ["-z"]):
        fifo = d + "/in%d.fastq" % len(zflag)
        os.mkfifo(fifo)
        t = threading.Thread(target=lambda: open(fifo, "wb").write(fq))
        t.start()
        p = subprocess.Popen([SBWT, "search", "-i", index, "-q", fifo, "-o", "/dev/stdout"] + zflag, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE)
        out, err = p.communicate(timeout=300)
        t.join()
        assert p.returncode == 0, err.decode()
        # (the log lines go to stderr; stdout carries the result text only)
        text = gzip.decompress(out) if zflag else out
        assert text == kat["expected_output"].encode()


def test_upstream_index_check_script(gpu):
    """tools/check_upstream_index.py is the runnable closing step for file-format parity with upstream (no
    upstream-built index exists offline).  Here it runs on a file of this repository's own writer -- that exercises the script
    (load, C array, search text, prefix table, byte-identical rewrite, and a corrupted rank support that must fail), it pins
    nothing."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_upstream_index.py"), "--self-test"], capture_output=True,
                       timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-2000:] + p.stderr.decode()[-2000:]
    assert out.count("PASS") >= 9 and "rewrite  FAIL" in out and "rank supports" in out, out
