import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
from sbwt_amd import capi, synth
from oracle import OracleIndex
k = int(os.environ.get("K", 32)); L = int(os.environ.get("L", 399)); n = int(os.environ.get("N", 1))
genomes = [synth.random_genome(50_000, 5)]
orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 8)
cols = orc.columns()
idx = capi.Index.create(cols[0], cols[1], cols[2], cols[3], orc.ssup_words(), orc.n_nodes, orc.k, orc.n_kmers, orc.precalc_k, None)
bases, off = synth.sample_reads(genomes, n, L, 0.0, 5)
got, _ = idx.streaming_search(bases, off)
want = np.concatenate([orc.streaming_search(bases[off[r]:off[r + 1]].tobytes()) for r in range(n)])
print("equal", np.array_equal(got, want), "wrong at", np.flatnonzero(got != want)[:20].tolist(), "of", len(got))
print("got", got[:8].tolist(), got[-8:].tolist())
print("want", want[:8].tolist(), want[-8:].tolist())
