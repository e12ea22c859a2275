"""k = 1 / 2 indexes: every route's results for a few reads, beside the oracle's (debugging aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import OracleIndex
from sbwt_amd import capi
seqs = [b"ACGTTGCAACGGT", b"TTTTACG", b"G"]
for k in (1, 2):
    orc = OracleIndex.build(seqs, k, True, False, 0)
    cols = orc.columns()
    idx = capi.Index.create(cols[0], cols[1], cols[2], cols[3], orc.ssup_words(), orc.n_nodes, k, orc.n_kmers, 0)
    print("k", k, "n_nodes", orc.n_nodes, "image level", idx.image_level, "default variant", idx.default_search_variant, "paths", idx.n_paths)
    reads = [b"ACGTTGCAACGGT", b"AC", b"ACGTNACGT"]
    bases, off = capi.concat_reads(reads)
    for streaming in (True, False):
        want = np.concatenate([orc.streaming_search(r) if streaming else orc.search_all(r) for r in reads])
        print(" streaming", streaming, "oracle ", list(want))
        for v in (-1, 5, 4, 2, 1, 0):
            capi.set_tuning("search_variant", v)
            got = (idx.streaming_search if streaming else idx.search)(bases, off)[0]
            capi.set_tuning("search_variant", -1)
            print("   variant %2d %s %s" % (v, "ok  " if np.array_equal(got, want) else "DIFF", list(got)))
