"""Debug helper: `sbwt search` with two kernel variants on the CLI test's data; prints differing lines."""
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from sbwt_amd import synth
BIN = ROOT + "/sbwt_amd/bin/sbwt"
d = tempfile.mkdtemp()
genomes = [synth.random_genome(30_000, 4)]
open(d + "/g.fna", "wb").write(b">g\n" + genomes[0].tobytes() + b"\n")
subprocess.run([BIN, "build", "-i", d + "/g.fna", "-o", d + "/ns.sbwt", "-k", os.environ.get("K", "31")] + ([] if os.environ.get("SSUP") else ["--no-streaming-support"]), check=True, capture_output=True)
bases, off = synth.sample_reads(genomes, 700, 120, 0.02, 4)
bases = synth.inject(bases, 15, ord("N"), 1)
reads = [bases[off[r]:off[r + 1]].tobytes() for r in range(int(os.environ.get("NR", 700)))]
if not os.environ.get("NOEXTRA"):
    reads += [b"ACGT", b"A" * 31]
with open(d + "/r.fastq", "wb") as f:
    for i, r in enumerate(reads):
        f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
outs = {}
for v in (1, 2):
    env = dict(os.environ, SBWTGPU_SEARCH_VARIANT=str(v))
    subprocess.run([BIN, "search", "-o", d + "/o%d" % v, "-i", d + "/ns.sbwt", "-q", d + "/r.fastq"], check=True, capture_output=True, env=env)
    outs[v] = open(d + "/o%d" % v, "rb").read().split(b"\n")
n = 0
for i, (a, b) in enumerate(zip(outs[1], outs[2])):
    if a != b:
        n += 1
        if n <= 5:
            ta, tb = a.split(b" "), b.split(b" ")
            ds = [j for j in range(min(len(ta), len(tb))) if ta[j] != tb[j]]
            print("read", i, "len", len(reads[i]), "tokens", len(ta), len(tb), "diff at", ds[:10], [(ta[j], tb[j]) for j in ds[:5]])
            print("   read:", reads[i])
print("differing lines", n)
