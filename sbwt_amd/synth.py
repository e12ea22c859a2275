"""Seeded synthetic genomes and reads: coli3.fna and the pan-genome are not available
offline, so tests and bench.py generate stand-ins with fixed seeds (numpy PCG64 => identical bytes
on every box).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_genome(n: int, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]


def mutate(genome: np.ndarray, rate: float, seed: int) -> np.ndarray:
    """Per-base substitutions with probability `rate` (always to a different base)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = genome.copy()
    hit = np.nonzero(rng.random(len(g)) < rate)[0]
    code = np.zeros(256, dtype=np.uint8)
    code[ACGT] = np.arange(4, dtype=np.uint8)
    shift = rng.integers(1, 4, size=len(hit), dtype=np.uint8)
    g[hit] = ACGT[(code[g[hit]] + shift) & 3]
    return g


def coli3_like(genome_len: int = 5_000_000, divergence: float = 0.05) -> List[np.ndarray]:
    """G-small of SURVEY 8d: genome 0 random (seed 1), genomes 1,2 = 5 % substituted copies (seeds 2,3)."""
    g0 = random_genome(genome_len, 1)
    return [g0, mutate(g0, divergence, 2), mutate(g0, divergence, 3)]


def pan_like(n_derived: int = 64, genome_len: int = 5_000_000, divergence: float = 0.02) -> List[np.ndarray]:
    """G-pan of SURVEY 8d: genome 0 + n_derived copies at 2 % divergence (seeds 100..)."""
    g0 = random_genome(genome_len, 1)
    return [g0] + [mutate(g0, divergence, 100 + i) for i in range(n_derived)]


def sample_reads(genomes: List[np.ndarray], n_reads: int, read_len: int, sub_rate: float, seed: int
                 ) -> Tuple[np.ndarray, np.ndarray]:
    """Uniform (genome, offset) substrings with per-base substitutions; returns (bases, read_off)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.array([len(g) for g in genomes])
    cat = np.concatenate(genomes)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    which = rng.integers(0, len(genomes), size=n_reads)
    off = (rng.random(n_reads) * (lens[which] - read_len + 1)).astype(np.int64) + starts[which]
    bases = np.empty(n_reads * read_len, dtype=np.uint8)
    chunk = max(1, (1 << 24) // max(read_len, 1))
    ar = np.arange(read_len, dtype=np.int64)
    for lo in range(0, n_reads, chunk):
        hi = min(n_reads, lo + chunk)
        bases[lo * read_len: hi * read_len] = cat[(off[lo:hi, None] + ar[None, :]).ravel()]
    if sub_rate > 0:
        bases = mutate(bases, sub_rate, seed + 1)
    read_off = np.arange(n_reads + 1, dtype=np.int64) * read_len
    return bases, read_off


def random_reads(n_reads: int, read_len: int, seed: int) -> Tuple[np.ndarray, np.ndarray]:
    """All-miss stress set: uniform random reads."""
    return random_genome(n_reads * read_len, seed), np.arange(n_reads + 1, dtype=np.int64) * read_len


def inject(bases: np.ndarray, n: int, char: int, seed: int) -> np.ndarray:
    """Overwrite n random positions with `char` (e.g. ord('N'))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    b = bases.copy()
    if len(b) and n:
        b[rng.integers(0, len(b), size=n)] = char
    return b


# ---- harder workloads (robustness table) ----
def repeat_genome(n: int, seed: int, repeat_fraction: float = 0.05) -> np.ndarray:
    """A random genome of n bases in which about `repeat_fraction` of the positions are repeated content:
    insertion-sequence-like elements (10 families of 1-5 kbp, several copies each, 1 % diverged between copies),
    one rRNA-operon-like 5 kbp element in 7 exact copies, and low-complexity tracts (homopolymers of 15-40 bases and
    tandem repeats of 2-6 bp units, 30-120 bases long)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = random_genome(n, seed + 1000003)
    budget = int(n * repeat_fraction)
    used = 0

    def paste(seq):
        nonlocal used
        at = int(rng.integers(0, max(1, n - len(seq))))
        g[at:at + len(seq)] = seq[: n - at]
        used += len(seq)
    rrna = random_genome(5000, seed + 7)
    for _ in range(7):
        paste(rrna)
    families = [random_genome(int(rng.integers(1000, 5001)), seed + 100 + f) for f in range(10)]
    while used < 0.8 * budget:
        paste(mutate(families[int(rng.integers(0, 10))], 0.01, int(rng.integers(1, 1 << 30))))
    while used < budget:
        if rng.random() < 0.5:
            paste(np.full(int(rng.integers(15, 41)), ACGT[int(rng.integers(0, 4))], dtype=np.uint8))
        else:
            unit = ACGT[rng.integers(0, 4, size=int(rng.integers(2, 7)))]
            paste(np.tile(unit, int(rng.integers(30, 121)) // len(unit) + 1))
    return g


def ragged_reads(genomes: List[np.ndarray], n_reads: int, min_len: int, max_len: int, sub_rate: float, seed: int
                 ) -> Tuple[np.ndarray, np.ndarray]:
    """Reads of uniformly distributed lengths in [min_len, max_len] at uniform (genome, offset), substitutions as in
    sample_reads; returns (bases, read_off)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens_g = np.array([len(g) for g in genomes])
    cat = np.concatenate(genomes)
    starts = np.concatenate([[0], np.cumsum(lens_g)[:-1]])
    L = rng.integers(min_len, max_len + 1, size=n_reads)
    which = rng.integers(0, len(genomes), size=n_reads)
    off = (rng.random(n_reads) * (lens_g[which] - L + 1)).astype(np.int64) + starts[which]
    read_off = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(L, out=read_off[1:])
    idx = np.repeat(off - read_off[:-1], L) + np.arange(read_off[-1], dtype=np.int64)
    bases = cat[idx]
    if sub_rate > 0:
        bases = mutate(bases, sub_rate, seed + 1)
    return bases, read_off


def indel_reads(genomes: List[np.ndarray], n_reads: int, read_len: int, sub_rate: float, indel_rate: float, seed: int
                ) -> Tuple[np.ndarray, np.ndarray]:
    """Reads with substitutions AND insertions/deletions: every base is deleted with probability indel_rate/2 and
    followed by one inserted random base with probability indel_rate/2 (so the reads come out ragged, read_len +- a
    few bases); returns (bases, read_off)."""
    bases, off = sample_reads(genomes, n_reads, read_len, sub_rate, seed)
    rng = np.random.Generator(np.random.PCG64(seed + 17))
    u = rng.random(len(bases))
    count = np.ones(len(bases), dtype=np.int64)
    count[u < indel_rate / 2] = 0                                   # deletions
    ins = (u >= indel_rate / 2) & (u < indel_rate)
    count[ins] = 2                                                  # the base + one inserted base
    out = np.repeat(bases, count)
    pos = np.cumsum(count)                                          # end position (exclusive) of every source base in out
    out[pos[ins] - 1] = ACGT[rng.integers(0, 4, size=int(ins.sum()))]
    read_off = np.concatenate([[0], pos[off[1:] - 1]]).astype(np.int64)
    return out, read_off


_COMP = np.zeros(256, dtype=np.uint8)
_COMP[ACGT] = np.frombuffer(b"TGCA", dtype=np.uint8)


def revcomp(seq: np.ndarray) -> np.ndarray:
    """Reverse complement of an ACGT byte array (globals.hh:19-35)."""
    return _COMP[seq[::-1]]


def both_strand_reads(genomes: List[np.ndarray], n_reads: int, read_len: int, sub_rate: float, seed: int
                      ) -> Tuple[np.ndarray, np.ndarray]:
    """sample_reads, but every read comes from the forward or the reverse strand with probability 1/2 -- what a
    sequencer delivers, and what an index built with --add-reverse-complements (tests/test_CLI.hh:43) is for."""
    bases, off = sample_reads(genomes, n_reads, read_len, 0.0, seed)
    rng = np.random.Generator(np.random.PCG64(seed + 29))
    flip = rng.random(n_reads) < 0.5
    rows = bases.reshape(n_reads, read_len)
    rows[flip] = _COMP[rows[flip][:, ::-1]]
    bases = rows.reshape(-1)
    if sub_rate > 0:
        bases = mutate(bases, sub_rate, seed + 1)
    return bases, off


def variant_reads(genomes: List[np.ndarray], n_reads: int, read_len: int, seed: int) -> Tuple[np.ndarray, np.ndarray]:
    """Reads of genomes[0] in which ONE base is replaced by the base genomes[1] has at that position, at places where
    the two differ (genomes[1] = mutate(genomes[0], ...): same coordinates).  The substituted k-mers are usually absent --
    the other strain's next difference is within k -- but every SHORT window around the base is in the index (as the
    other strain's): the case where certificate probes prove nothing and only each k-mer's own search does."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a, b = genomes[0], genomes[1]
    diff = np.flatnonzero(a != b)
    diff = diff[(diff >= read_len) & (diff < len(a) - read_len)]
    at = rng.choice(diff, size=n_reads)
    start = at - rng.integers(0, read_len, size=n_reads)
    bases = a[(start[:, None] + np.arange(read_len)[None, :]).ravel()].copy()
    bases[np.arange(n_reads) * read_len + (at - start)] = b[at]
    return bases, np.arange(n_reads + 1, dtype=np.int64) * read_len
