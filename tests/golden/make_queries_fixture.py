"""Writes tests/golden/queries_seqs.txt.gz: the 5000 sequence lines of the reference's own query file
example_data/queries.fastq (the data its TEST_LARGE.streaming_queries test reads, tests/test_large.hh:105).
Data only (headers and quality lines dropped); the index the reference pairs it with (coli3.fna) is not
in the checkout, so the tests index the reads themselves.
Run (in the container that has /root/reference):  python tests/golden/make_queries_fixture.py"""
import gzip
import os

src = "/root/reference/example_data/queries.fastq"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "queries_seqs.txt.gz")
lines = open(src, "rb").read().split(b"\n")
seqs = [lines[i] for i in range(1, len(lines), 4) if lines[i]]
assert len(seqs) == 5000 and all(len(s) == 100 for s in seqs) and sum(s.count(b"N") for s in seqs) == 21
with gzip.GzipFile(dst, "wb", compresslevel=9, mtime=0) as f:
    f.write(b"\n".join(seqs) + b"\n")
print("wrote", dst, os.path.getsize(dst), "bytes")
