"""Definition-level brute force of the plain-matrix SBWT in pure Python (tiny inputs only).

Independent of oracle/sbwt_oracle.c: it builds the node set straight from the definition used by
the reference (NodeBOSSInMemoryConstructor.hh:98-154 describes the same set constructively) and
answers queries by dictionary lookup, so it can check both the oracle and the GPU path.
"""
from __future__ import annotations

from typing import Dict, List, Set, Tuple

RC = {"A": "T", "C": "G", "G": "C", "T": "A"}


def revcomp(s: str) -> str:
    return "".join(RC.get(c, c) for c in reversed(s))


def kmer_set(seqs: List[str], k: int) -> Set[str]:
    out = set()
    for s in seqs:
        for i in range(len(s) - k + 1):
            w = s[i:i + k]
            if all(c in "ACGT" for c in w):
                out.add(w)
    return out


def colex_key(label: str, k: int) -> Tuple:
    # Kmer::operator< (Kmer.hh:108-123): compare from the last char backwards; missing (left) chars
    # count as 'A'; on a tie the shorter string is smaller.
    rev = label[::-1] + "A" * (k - len(label))
    return (rev, len(label))


class BruteSBWT:
    def __init__(self, seqs: List[str], k: int, add_revcomp: bool = False):
        if add_revcomp:
            seqs = list(seqs) + [revcomp(s) for s in seqs]
        self.k = k
        self.kmers = kmer_set(seqs, k)
        suffixes = {x[1:] for x in self.kmers}
        nodes = set(self.kmers)
        nodes.add("")
        for z in self.kmers:
            if z[:-1] not in suffixes:                 # no predecessor -> all proper prefixes are dummies
                for j in range(k):
                    nodes.add(z[:j])
        self.nodes: List[str] = sorted(nodes, key=lambda s: colex_key(s, k))
        self.rank_of: Dict[str, int] = {s: i for i, s in enumerate(self.nodes)}
        n = len(self.nodes)
        # suffix group starts (NodeBOSSInMemoryConstructor.hh:174-185)
        def sfx(s):
            return s[1:] if len(s) == k else s
        self.ssup = [1 if i == 0 or sfx(self.nodes[i]) != sfx(self.nodes[i - 1]) else 0 for i in range(n)]
        # edges, only on suffix group starts (:113-137)
        self.edges = [set() for _ in range(n)]
        for i, s in enumerate(self.nodes):
            if not self.ssup[i]:
                continue
            for c in "ACGT":
                t = (s + c) if len(s) < k else (s[1:] + c)
                if t in self.rank_of and len(t) == min(len(s) + 1, k):
                    self.edges[i].add(c)

    def columns(self):
        """A,C,G,T,ssup as Python ints (bit i = column i)."""
        cols = []
        for c in "ACGT":
            v = 0
            for i, e in enumerate(self.edges):
                if c in e:
                    v |= 1 << i
            cols.append(v)
        s = 0
        for i, b in enumerate(self.ssup):
            if b:
                s |= 1 << i
        return cols, s

    def search(self, kmer: str) -> int:
        if any(c not in "ACGT" for c in kmer[: self.k]):
            return -1
        return self.rank_of.get(kmer[: self.k], -1) if kmer[: self.k] in self.kmers else -1

    def search_all(self, s: str) -> List[int]:
        return [self.search(s[i:i + self.k]) for i in range(len(s) - self.k + 1)]


def int_to_words(v: int, n_bits: int):
    import numpy as np
    nw = (n_bits + 63) // 64
    return np.array([(v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(nw)], dtype=np.uint64)
