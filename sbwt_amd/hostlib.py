"""ctypes binding of include/sbwthost.h (libsbwthost.so): the sort-based index builder, the
reference index file format and the FASTA/FASTQ reader.  GPU-free host code either side of the
hot path."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SBWT_HOST_LIB", os.path.join(_HERE, "lib", "libsbwthost.so"))   # override: sanitizer build

EXPORTED_SYMBOLS = [
    "sbwthost_last_error", "sbwthost_build", "sbwthost_bits_free", "sbwthost_bits_info", "sbwthost_bits_words",
    "sbwthost_file_write", "sbwthost_file_read", "sbwthost_file_free", "sbwthost_file_info",
    "sbwthost_file_words", "sbwthost_file_precalc", "sbwthost_read_sequences", "sbwthost_read_sequences_chunked", "sbwthost_free", "sbwthost_write_file",
    "sbwthost_rank_batch",
]

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m sbwt_amd.build`")
    L = C.CDLL(LIB_PATH)
    vp, i64, ci = C.c_void_p, C.c_int64, C.c_int
    L.sbwthost_last_error.restype = C.c_char_p
    L.sbwthost_build.argtypes = [C.POINTER(C.c_char_p), C.POINTER(i64), i64, i64, ci, ci, ci, C.POINTER(vp)]
    L.sbwthost_bits_free.argtypes = [vp]
    L.sbwthost_bits_free.restype = None
    L.sbwthost_bits_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(ci)]
    L.sbwthost_bits_words.argtypes = [vp, ci]
    L.sbwthost_bits_words.restype = C.POINTER(C.c_uint64)
    L.sbwthost_file_write.argtypes = [C.c_char_p, i64, vp, vp, vp, vp, vp, C.POINTER(i64), vp, i64, i64, i64]
    L.sbwthost_file_read.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.sbwthost_file_free.argtypes = [vp]
    L.sbwthost_file_free.restype = None
    L.sbwthost_file_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64),
                                     C.POINTER(i64), C.POINTER(ci)]
    L.sbwthost_file_words.argtypes = [vp, ci]
    L.sbwthost_file_words.restype = C.POINTER(C.c_uint64)
    L.sbwthost_file_precalc.argtypes = [vp]
    L.sbwthost_file_precalc.restype = C.POINTER(i64)
    L.sbwthost_read_sequences.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
    L.sbwthost_read_sequences_chunked.argtypes = [C.c_char_p, i64, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
    L.sbwthost_free.argtypes = [vp]
    L.sbwthost_free.restype = None
    L.sbwthost_write_file.argtypes = [C.c_char_p, C.c_char_p, i64, ci, ci]
    L.sbwthost_rank_batch.argtypes = [vp, i64, vp, i64, vp]
    _lib = L
    return L


def _err() -> str:
    return lib().sbwthost_last_error().decode(errors="replace")


class IndexBits:
    """A/C/G/T (+ suffix_group_starts) columns of one plain-matrix SBWT, as numpy uint64 words."""

    def __init__(self, cols: List[np.ndarray], ssup: Optional[np.ndarray], n_nodes: int, n_kmers: int, k: int):
        self.cols, self.ssup, self.n_nodes, self.n_kmers, self.k = cols, ssup, n_nodes, n_kmers, k


def build_bits(seqs: Sequence[bytes], k: int, add_revcomp: bool = False, streaming_support: bool = True,
               n_threads: int = 1) -> IndexBits:
    """Sort-based in-memory construction (index_builder.hh)."""
    L = lib()
    seqs = [bytes(s) for s in seqs]
    arr = (C.c_char_p * len(seqs))(*seqs)
    lens = (C.c_int64 * len(seqs))(*[len(s) for s in seqs])
    h = C.c_void_p()
    if L.sbwthost_build(arr, lens, len(seqs), k, int(add_revcomp), int(streaming_support), n_threads, C.byref(h)) != 0:
        raise RuntimeError(_err())
    try:
        n, nk, kk, hs = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        L.sbwthost_bits_info(h, C.byref(n), C.byref(nk), C.byref(kk), C.byref(hs))
        nw = (n.value + 63) // 64
        cols = [np.ctypeslib.as_array(L.sbwthost_bits_words(h, c), shape=(nw,)).copy() for c in range(4)]
        ssup = np.ctypeslib.as_array(L.sbwthost_bits_words(h, 4), shape=(nw,)).copy() if hs.value else None
        return IndexBits(cols, ssup, n.value, nk.value, kk.value)
    finally:
        L.sbwthost_bits_free(h)


class IndexFile:
    def __init__(self, cols, ssup, C_array, precalc, precalc_k, n_nodes, n_kmers, k):
        self.cols, self.ssup, self.C, self.precalc = cols, ssup, C_array, precalc
        self.precalc_k, self.n_nodes, self.n_kmers, self.k = precalc_k, n_nodes, n_kmers, k


def write_index_file(path: str, cols, ssup, C_array, precalc: Optional[np.ndarray], precalc_k: int, n_nodes: int,
                     n_kmers: int, k: int) -> None:
    cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in cols]
    s = np.ascontiguousarray(ssup, dtype=np.uint64) if ssup is not None else None
    pc = np.ascontiguousarray(precalc, dtype=np.int64) if precalc is not None and precalc_k else None
    Carr = (C.c_int64 * 4)(*[int(x) for x in C_array])
    rc = lib().sbwthost_file_write(path.encode(), n_nodes, cols[0].ctypes.data, cols[1].ctypes.data,
                                   cols[2].ctypes.data, cols[3].ctypes.data, s.ctypes.data if s is not None else None,
                                   Carr, pc.ctypes.data if pc is not None else None, precalc_k, n_kmers, k)
    if rc != 0:
        raise RuntimeError(_err())


def read_index_file(path: str) -> IndexFile:
    L = lib()
    h = C.c_void_p()
    if L.sbwthost_file_read(path.encode(), C.byref(h)) != 0:
        raise RuntimeError(_err())
    try:
        n, nk, k, p, hs = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        Carr = (C.c_int64 * 4)()
        L.sbwthost_file_info(h, C.byref(n), C.byref(nk), C.byref(k), C.byref(p), Carr, C.byref(hs))
        nw = (n.value + 63) // 64
        cols = [np.ctypeslib.as_array(L.sbwthost_file_words(h, c), shape=(nw,)).copy() for c in range(4)]
        ssup = np.ctypeslib.as_array(L.sbwthost_file_words(h, 4), shape=(nw,)).copy() if hs.value else None
        precalc = None
        if p.value:
            precalc = np.ctypeslib.as_array(L.sbwthost_file_precalc(h), shape=(4 ** p.value, 2)).copy()
        return IndexFile(cols, ssup, [Carr[i] for i in range(4)], precalc, p.value, n.value, nk.value, k.value)
    finally:
        L.sbwthost_file_free(h)


def read_sequences_chunked(path: str, chunk_bytes: int, n_threads: int):
    """The same reads through the CLI's chunked reader for plain regular files; None when the file cannot be cut."""
    L = lib()
    pb, po, n = C.c_void_p(), C.c_void_p(), C.c_int64()
    rc = L.sbwthost_read_sequences_chunked(path.encode(), chunk_bytes, n_threads, C.byref(pb), C.byref(po), C.byref(n))
    if rc == 1:
        return None
    if rc != 0:
        raise RuntimeError(_err())
    try:
        off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_int64)), shape=(n.value + 1,)).copy()
        total = int(off[-1])
        bases = (np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), shape=(total,)).copy()
                 if total else np.zeros(0, np.uint8))
        return bases, off
    finally:
        L.sbwthost_free(pb)
        L.sbwthost_free(po)


def read_sequences(path: str):
    """All reads of a FASTA/FASTQ(.gz) file: (bases uint8 array, read_off int64 array)."""
    L = lib()
    pb, po, n = C.c_void_p(), C.c_void_p(), C.c_int64()
    if L.sbwthost_read_sequences(path.encode(), C.byref(pb), C.byref(po), C.byref(n)) != 0:
        raise RuntimeError(_err())
    try:
        off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_int64)), shape=(n.value + 1,)).copy()
        total = int(off[-1])
        bases = (np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), shape=(total,)).copy()
                 if total else np.zeros(0, np.uint8))
        return bases, off
    finally:
        L.sbwthost_free(pb)
        L.sbwthost_free(po)


def write_file(path: str, data: bytes, gzip_output: bool = False, n_threads: int = 0) -> None:
    """Writes bytes through the CLI's buffered writer (parallel multi-member gzip when gzip_output)."""
    if lib().sbwthost_write_file(path.encode(), data, len(data), int(gzip_output), n_threads) != 0:
        raise RuntimeError(_err())


def rank_batch(bits: np.ndarray, n_bits: int, pos) -> np.ndarray:
    """Host rank directory of the scalar API (rank_support_v5 layout): ones in bits[0, pos) for every position."""
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    out = np.empty(len(pos), dtype=np.int64)
    if lib().sbwthost_rank_batch(bits.ctypes.data, n_bits, pos.ctypes.data, len(pos), out.ctypes.data) != 0:
        raise RuntimeError(_err())
    return out
