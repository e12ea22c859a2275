"""Debug helper: compare the search variants on a small coli3-like case through the host API and explain the first
differences (read, k-mer range, expected vs got).  Env: K, GLEN, NR, L, SUB, DIRTY (inject N / lower case), VARIANTS."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sbwt_amd import capi, hostlib, synth

k = int(os.environ.get("K", 30))
glen = int(os.environ.get("GLEN", 50_000))
nr = int(os.environ.get("NR", 4000))
L = int(os.environ.get("L", 150))
variants = json.loads(os.environ.get("VARIANTS", "[0,4,5]"))
genomes = synth.coli3_like(glen)
bits = hostlib.build_bits([g.tobytes() for g in genomes], k, False, True, n_threads=8)
idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k, bits.n_kmers, 8)
print("index n_nodes", idx.n_nodes, "image", idx.blob_bytes, "paths", idx.n_paths, "branching", idx.n_branch)
bases, off = synth.sample_reads(genomes, nr, L, float(os.environ.get("SUB", 0.01)), 42)
if int(os.environ.get("DIRTY", 0)):
    bases = synth.inject(bases, 60, ord("N"), 7)
    bases = synth.inject(bases, 60, ord("c"), 8)
if os.environ.get("ONLY"):
    r = int(os.environ["ONLY"])
    bases = bases[off[r]:off[r + 1]].copy()
    off = np.array([0, len(bases)], dtype=np.int64)
oo = capi.out_offsets(off, k)
capi.set_tuning("poison_results", 1)
res = {}
for v in variants:
    capi.set_tuning("search_variant", v)
    res[v] = idx.streaming_search(bases, off)[0]
    print("variant", v, "found", int((res[v] >= 0).sum()), "of", len(res[v]))
capi.set_tuning("search_variant", -1)
ref = res[variants[0]]
for v in variants[1:]:
    d = np.nonzero(res[v] != ref)[0]
    print("variant", v, "differences", len(d))
    shown = set()
    for x in d:
        r = int(np.searchsorted(oo, x, side="right") - 1)
        if r in shown or len(shown) >= 5:
            continue
        shown.add(r)
        lo, hi = int(oo[r]), int(oo[r + 1])
        bad = np.nonzero(res[v][lo:hi] != ref[lo:hi])[0]
        print("  read", r, "bad k-mers", bad.tolist()[:40])
        print("     read:", bases[off[r]:off[r + 1]].tobytes().decode("latin1"))
        print("     want", ref[lo:hi].tolist())
        print("     got ", res[v][lo:hi].tolist())
