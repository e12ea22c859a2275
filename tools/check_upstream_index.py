#!/usr/bin/env python3
"""check_upstream_index.py -- the closing step for SURVEY 8f row 1 (file-format parity with upstream `sbwt`), runnable
by anyone who holds an upstream-built plain-matrix index.  Offline no such file exists (the reference cannot be built here:
sdsl-lite and SeqIO are absent), so the byte layout of sdsl::bit_vector / rank_support_v5 in host/bitvector.hh is restated
from upstream knowledge and stays "parity unpinned" until this script has passed once on a real upstream file.

  upstream side (any machine with algbio/SBWT built):
      sbwt build -i genomes.fna -o upstream.sbwt -k 31 --add-reverse-complements
      sbwt search -i upstream.sbwt -q queries.fastq -o upstream_out.txt
  here (GPU box):
      python tools/check_upstream_index.py upstream.sbwt queries.fastq upstream_out.txt

Checks, each printed as PASS / FAIL:
  1. load      the file parses with host/index_file.hh (framing strings, four bit vectors, four rank supports skipped,
               suffix_group_starts, C, prefix table, the four trailing integers; SBWT.hh:500-516, SubsetMatrixRank.hh:102-125)
  2. C array   C[0] = 1, C[i+1] = C[i] + ones of row i, recomputed from the loaded bits (SBWT.hh:344-349)
  3. search    the GPU's `sbwt search` text for the queries equals upstream's output byte for byte (sbwt_search.cpp:21-91)
  4. precalc   the prefix table the device computes from the bits equals the one in the file (SBWT.hh:616-645)
  5. rewrite   the file written back by host/index_file.hh is byte-identical to the upstream file (SBWT.hh:462-491,
               SubsetMatrixRank.hh:86-100: this is what pins the rank_support_v5 directory bytes)

Exit code 0 iff every check passes.  --self-test writes an index with this repository's own writer first (what
tests/test_gpu_cli.py runs: it exercises the script, it does not pin anything).
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def check(index_path: str, query_path: str, expected_out_path: str) -> int:
    from sbwt_amd import capi, hostlib
    failures = 0

    def report(name: str, ok: bool, detail: str = "") -> None:
        nonlocal failures
        print("%-8s %s%s" % (name, "PASS" if ok else "FAIL", (": " + detail) if detail else ""), flush=True)
        if not ok:
            failures += 1

    try:
        f = hostlib.read_index_file(index_path)
    except Exception as e:                                 # noqa: BLE001 -- the message is the finding
        report("load", False, str(e))
        return 1
    report("load", True, "n_nodes %d, n_kmers %d, k %d, precalc %d, streaming support %s" %
           (f.n_nodes, f.n_kmers, f.k, f.precalc_k, f.ssup is not None))

    ones = [int(np.unpackbits(c.view(np.uint8)).sum()) for c in f.cols]
    want_C = [1, 1 + ones[0], 1 + ones[0] + ones[1], 1 + ones[0] + ones[1] + ones[2]]
    report("C array", list(f.C) == want_C, "file %s, from the bits %s" % (list(f.C), want_C))

    idx = capi.Index.create(f.cols[0], f.cols[1], f.cols[2], f.cols[3], f.ssup, f.n_nodes, f.k, f.n_kmers, f.precalc_k,
                            f.precalc)
    bases, off = hostlib.read_sequences(query_path)
    text, n_kmers = idx.search_text(bases, off, streaming=f.ssup is not None)
    expected = open(expected_out_path, "rb").read()
    ok = text == expected
    detail = "%d reads, %d k-mers, %d bytes" % (len(off) - 1, n_kmers, len(text))
    if not ok:
        n = min(len(text), len(expected))
        first = next((i for i in range(n) if text[i] != expected[i]), n)
        line = text[:first].count(b"\n") + 1
        detail += "; first difference at byte %d (output line %d): ours %r, upstream %r" % (
            first, line, text[max(0, first - 20):first + 20], expected[max(0, first - 20):first + 20])
    report("search", ok, detail)

    if f.precalc_k:
        idx2 = capi.Index.create(f.cols[0], f.cols[1], f.cols[2], f.cols[3], f.ssup, f.n_nodes, f.k, f.n_kmers, f.precalc_k, None)
        report("precalc", np.array_equal(idx2.get_precalc(), f.precalc), "4^%d entries" % f.precalc_k)
    else:
        report("precalc", True, "the file holds no prefix table")

    with tempfile.TemporaryDirectory() as d:
        again = os.path.join(d, "rewritten.sbwt")
        hostlib.write_index_file(again, f.cols, f.ssup, f.C, f.precalc, f.precalc_k, f.n_nodes, f.n_kmers, f.k)
        a, b = open(index_path, "rb").read(), open(again, "rb").read()
        ok = a == b
        detail = "%d bytes" % len(a)
        if not ok:
            n = min(len(a), len(b))
            first = next((i for i in range(n) if a[i] != b[i]), n)
            nw = (f.n_nodes + 63) // 64
            bits_end = 8 + 12 + 8 + 4 + 4 * (8 + 8 * nw)
            where = "framing / bit vectors" if first < bits_end else "rank supports or later (offset %d past the bit vectors)" % (first - bits_end)
            detail = "sizes %d (upstream) vs %d (rewritten); first difference at byte %d, in the %s" % (len(a), len(b), first, where)
        report("rewrite", ok, detail)
    return 1 if failures else 0


def self_test() -> int:
    """Builds a small index with this repository's own builder + writer and its own expected output (the oracle is not
    involved: a script under tools/ ships with the product).  Exercises every branch of check(); pins nothing."""
    from sbwt_amd import capi, hostlib, synth
    with tempfile.TemporaryDirectory() as d:
        genomes = [synth.random_genome(20_000, 5)]
        genomes.append(synth.mutate(genomes[0], 0.03, 6))
        bits = hostlib.build_bits([g.tobytes() for g in genomes], 31, True, True)
        idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, 31, bits.n_kmers, 6)
        path = os.path.join(d, "own.sbwt")
        hostlib.write_index_file(path, bits.cols, bits.ssup, idx.C, idx.get_precalc(), 6, bits.n_nodes, bits.n_kmers, 31)
        bases, off = synth.sample_reads(genomes, 300, 120, 0.01, 7)
        q = os.path.join(d, "q.fastq")
        with open(q, "wb") as fh:
            for r in range(len(off) - 1):
                s = bases[off[r]:off[r + 1]].tobytes()
                fh.write(b"@r%d\n%s\n+\n%s\n" % (r, s, b"I" * len(s)))
        text, _ = idx.search_text(bases, off, True)
        out = os.path.join(d, "expected.txt")
        open(out, "wb").write(text)
        rc = check(path, q, out)
        # a corrupted copy must FAIL the rewrite check and name the region
        raw = bytearray(open(path, "rb").read())
        nw = (bits.n_nodes + 63) // 64
        raw[8 + 12 + 8 + 4 + 4 * (8 + 8 * nw) + 24] ^= 1            # one bit inside the first rank support
        bad = os.path.join(d, "bad.sbwt")
        open(bad, "wb").write(bytes(raw))
        rc_bad = check(bad, q, out)
        print("self-test: own file rc %d (want 0), corrupted rank support rc %d (want 1)" % (rc, rc_bad))
        return 0 if (rc == 0 and rc_bad == 1) else 1


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("index", nargs="?", help="upstream-built plain-matrix index (.sbwt)")
    ap.add_argument("queries", nargs="?", help="FASTA / FASTQ (.gz) query file given to upstream `sbwt search`")
    ap.add_argument("expected", nargs="?", help="upstream's uncompressed output for those queries")
    ap.add_argument("--self-test", action="store_true")
    a = ap.parse_args()
    if a.self_test:
        return self_test()
    if not (a.index and a.queries and a.expected):
        ap.error("index, queries and expected output are required (or --self-test)")
    return check(a.index, a.queries, a.expected)


if __name__ == "__main__":
    sys.exit(main())
