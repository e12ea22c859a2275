"""End-to-end timing of the `sbwt` CLI and of the host-buffer entry points (PCIe inclusive)."""
import os, subprocess, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sbwt_amd import capi, hostlib, synth

n_reads = int(os.environ.get("NREADS", 2_000_000))
SBWT = os.path.join(ROOT, "sbwt_amd", "bin", "sbwt")
genomes = synth.coli3_like(5_000_000)
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
with open(d + "/g.fna", "wb") as f:
    for i, g in enumerate(genomes):
        f.write(b">g%d\n" % i + g.tobytes() + b"\n")
t = time.time(); subprocess.run([SBWT, "build", "-i", d + "/g.fna", "-o", d + "/i.sbwt", "-k", "30", "-t", "16"], check=True, capture_output=True); print("build s", time.time() - t, flush=True)
bases, off = synth.sample_reads(genomes, n_reads, 150, 0.01, 42)
t = time.time()
with open(d + "/r.fastq", "wb") as f:
    q = b"I" * 150
    rows = bases.reshape(n_reads, 150)
    chunk = []
    for r in range(n_reads):
        chunk.append(b"@r\n" + rows[r].tobytes() + b"\n+\n" + q + b"\n")
        if len(chunk) == 100000: f.write(b"".join(chunk)); chunk = []
    f.write(b"".join(chunk))
print("fastq written s", time.time() - t, os.path.getsize(d + "/r.fastq") / 1e6, "MB", flush=True)
n_kmers = n_reads * 121
for extra in ([], [], ["--batch-bases", "268435456"]):
    t = time.time()
    p = subprocess.run([SBWT, "search", "-i", d + "/i.sbwt", "-q", d + "/r.fastq", "-o", d + "/out.txt"] + extra, capture_output=True,
                       env=dict(os.environ, SBWT_CLI_TIMING="1"))
    print("\n".join(l for l in p.stderr.decode().splitlines() if l.startswith("timing")))
    dt = time.time() - t
    logs = [l for l in p.stderr.decode().splitlines() if "us/query" in l]
    print("CLI", extra, "wall s %.2f -> %.1f M k-mers/s end to end;" % (dt, n_kmers / dt / 1e6), logs, "out MB", os.path.getsize(d + "/out.txt") / 1e6, flush=True)
# host-buffer entry points
f = hostlib.read_index_file(d + "/i.sbwt")
idx = capi.Index.create(f.cols[0], f.cols[1], f.cols[2], f.cols[3], f.ssup, f.n_nodes, f.k, f.n_kmers, f.precalc_k, f.precalc)
for name, fn in (("streaming_search_batch (raw int64 out)", lambda: idx.streaming_search(bases, off)),
                 ("search_text_batch (GPU formatted, pipelined)", lambda: idx.search_text(bases, off, True))):
    fn()
    t = time.time(); fn(); dt = time.time() - t
    print("%s: %.3f s -> %.2f G k-mers/s PCIe inclusive" % (name, dt, n_kmers / dt / 1e9), flush=True)
