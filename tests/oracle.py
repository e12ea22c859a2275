"""ctypes wrapper of oracle/liboracle.so -- the CPU restatement of the reference path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py; never by anything under sbwt_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.environ.get("SBWT_ORACLE_LIB", os.path.join(ORACLE_DIR, "liboracle.so"))   # override: sanitizer build


class _BitVec(C.Structure):
    _fields_ = [("n_bits", C.c_int64), ("n_words", C.c_int64), ("words", C.POINTER(C.c_uint64)),
                ("n_dir", C.c_int64), ("dir", C.POINTER(C.c_uint64))]


class _Index(C.Structure):
    _fields_ = [("col", _BitVec * 4), ("ssup", _BitVec), ("C", C.c_int64 * 4),
                ("precalc", C.POINTER(C.c_int64)), ("precalc_k", C.c_int64),
                ("n_nodes", C.c_int64), ("n_kmers", C.c_int64), ("k", C.c_int64)]


_lib: Optional[C.CDLL] = None


def build_if_needed() -> None:
    src = os.path.join(ORACLE_DIR, "sbwt_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "sbwt_oracle.h")
    if (not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build_if_needed()
    L = C.CDLL(LIB_PATH)
    vp, i64 = C.c_void_p, C.c_int64
    P = C.POINTER(_Index)
    L.orc_index_build.restype = P
    L.orc_index_build.argtypes = [C.POINTER(C.c_char_p), i64, i64, C.c_int, C.c_int, i64]
    L.orc_index_from_bits.restype = P
    L.orc_index_from_bits.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64]
    L.orc_index_free.argtypes = [P]
    L.orc_index_free.restype = None
    L.orc_do_precalc.argtypes = [P, i64]
    L.orc_rank.restype = i64
    L.orc_rank.argtypes = [P, i64, C.c_char]
    L.orc_contains.argtypes = [P, i64, C.c_char]
    L.orc_update_interval.argtypes = [P, C.c_char_p, i64, C.POINTER(i64), C.POINTER(i64)]
    L.orc_update_interval.restype = None
    L.orc_search.restype = i64
    L.orc_search.argtypes = [P, C.c_char_p]
    L.orc_streaming_search.restype = i64
    L.orc_streaming_search.argtypes = [P, C.c_char_p, i64, vp]
    L.orc_search_all.restype = i64
    L.orc_search_all.argtypes = [P, C.c_char_p, i64, vp]
    L.orc_forward.restype = i64
    L.orc_forward.argtypes = [P, i64, C.c_char]
    L.orc_partial_search.restype = i64
    L.orc_partial_search.argtypes = [P, C.c_char_p, i64, C.POINTER(i64), C.POINTER(i64)]
    L.orc_mark_suffix_groups.argtypes = [P, vp]
    L.orc_mark_suffix_groups.restype = None
    L.orc_print_vector.restype = i64
    L.orc_print_vector.argtypes = [vp, i64, C.c_char_p]
    L.orc_search_file.restype = C.c_double
    L.orc_search_file.argtypes = [P, C.c_char_p, C.c_char_p, C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_double)]
    L.orc_batch_search.restype = C.c_double
    L.orc_batch_search.argtypes = [P, vp, vp, i64, vp, vp, C.c_int]
    L.orc_get_kmer.restype = None
    L.orc_get_kmer.argtypes = [P, i64, C.c_char_p]
    L.orc_select.restype = i64
    L.orc_select.argtypes = [P, i64, C.c_char]
    L.orc_batch_rank.restype = C.c_double
    L.orc_batch_rank.argtypes = [P, vp, vp, i64, vp, C.c_int]
    L.orc_count_work.restype = None
    L.orc_count_work.argtypes = [P, vp, vp, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    _lib = L
    return L


def _bv_words(bv: _BitVec) -> np.ndarray:
    nw = (bv.n_bits + 63) // 64
    if nw == 0:
        return np.zeros(0, dtype=np.uint64)
    return np.ctypeslib.as_array(bv.words, shape=(nw,)).copy()


class OracleIndex:
    """The oracle's restatement of sbwt::SBWT<SubsetMatrixRank<...>> (plain-matrix)."""

    def __init__(self, ptr):
        if not ptr:
            raise RuntimeError("oracle index construction failed")
        self._p = ptr
        s = ptr.contents
        self.n_nodes, self.n_kmers, self.k, self.precalc_k = s.n_nodes, s.n_kmers, s.k, s.precalc_k
        self.C = [s.C[i] for i in range(4)]
        self.has_streaming_support = s.ssup.n_bits > 0

    # -- construction --
    @classmethod
    def build(cls, seqs: Sequence[bytes], k: int, streaming_support: bool = True, add_revcomp: bool = False,
              precalc_k: int = 0) -> "OracleIndex":
        arr = (C.c_char_p * len(seqs))(*[bytes(s) for s in seqs])
        return cls(lib().orc_index_build(arr, len(seqs), k, int(streaming_support), int(add_revcomp), precalc_k))

    @classmethod
    def from_bits(cls, A, Cb, G, T, ssup, n_nodes: int, k: int, n_kmers: int, precalc_k: int) -> "OracleIndex":
        arrs = [np.ascontiguousarray(x, dtype=np.uint64) for x in (A, Cb, G, T)]
        s = np.ascontiguousarray(ssup, dtype=np.uint64) if ssup is not None else None
        return cls(lib().orc_index_from_bits(arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                                             arrs[3].ctypes.data, s.ctypes.data if s is not None else None,
                                             n_nodes, k, n_kmers, precalc_k))

    def __del__(self):
        try:
            if self._p:
                lib().orc_index_free(self._p)
                self._p = None
        except Exception:
            pass

    # -- raw members --
    def columns(self) -> List[np.ndarray]:
        return [_bv_words(self._p.contents.col[i]) for i in range(4)]

    def ssup_words(self) -> Optional[np.ndarray]:
        return _bv_words(self._p.contents.ssup) if self.has_streaming_support else None

    def precalc(self) -> np.ndarray:
        n = 4 ** self.precalc_k if self.precalc_k else 0
        if n == 0:
            return np.zeros((0, 2), dtype=np.int64)
        return np.ctypeslib.as_array(self._p.contents.precalc, shape=(n, 2)).copy()

    def do_precalc(self, p: int) -> int:
        rc = lib().orc_do_precalc(self._p, p)
        self.precalc_k = self._p.contents.precalc_k
        return rc

    # -- queries --
    def rank(self, pos: int, c: bytes) -> int:
        return lib().orc_rank(self._p, pos, c)

    def search(self, kmer: bytes) -> int:
        assert len(kmer) >= self.k
        return lib().orc_search(self._p, kmer)

    def streaming_search(self, s: bytes) -> np.ndarray:
        out = np.zeros(max(len(s) - self.k + 1, 0), dtype=np.int64)
        n = lib().orc_streaming_search(self._p, s, len(s), out.ctypes.data)
        if n < 0:
            raise RuntimeError("Error: streaming search support not built")
        return out[:n]

    def search_all(self, s: bytes) -> np.ndarray:
        out = np.zeros(max(len(s) - self.k + 1, 0), dtype=np.int64)
        n = lib().orc_search_all(self._p, s, len(s), out.ctypes.data)
        return out[:n]

    def update_interval(self, s: bytes, first: int, second: int):
        a, b = C.c_int64(first), C.c_int64(second)
        lib().orc_update_interval(self._p, s, len(s), C.byref(a), C.byref(b))
        return a.value, b.value

    def forward(self, node: int, c: bytes) -> int:
        return lib().orc_forward(self._p, node, c)

    def partial_search(self, s: bytes):
        a, b = C.c_int64(0), C.c_int64(0)
        n = lib().orc_partial_search(self._p, s, len(s), C.byref(a), C.byref(b))
        return (a.value, b.value), n

    def get_kmer(self, colex_rank: int) -> bytes:
        buf = C.create_string_buffer(int(self.k) + 1)
        lib().orc_get_kmer(self._p, colex_rank, buf)
        return buf.raw[: self.k]

    def select(self, j: int, c: bytes) -> int:
        return lib().orc_select(self._p, j, c)

    def mark_suffix_groups(self) -> np.ndarray:
        out = np.zeros((self.n_nodes + 63) // 64, dtype=np.uint64)
        lib().orc_mark_suffix_groups(self._p, out.ctypes.data)
        return out

    def search_file(self, query_path: str, out_path: str):
        """The reference CLI's single-threaded loop over one query file (sbwt_search.cpp:21-105): returns
        (wall seconds, query seconds, reads, k-mers)."""
        nr, nk, qs = C.c_int64(), C.c_int64(), C.c_double()
        wall = lib().orc_search_file(self._p, query_path.encode(), out_path.encode(), C.byref(nr), C.byref(nk), C.byref(qs))
        if wall < 0:
            raise RuntimeError("orc_search_file: I/O error")
        return wall, qs.value, nr.value, nk.value

    def batch_search(self, bases: np.ndarray, read_off: np.ndarray, out_off: np.ndarray, n_threads: int = 1,
                     out: Optional[np.ndarray] = None):
        """Returns (out, seconds): seconds = the largest per-thread sum of per-read query times, timed
        like sbwt_search.cpp:54-56 ("excluding I/O etc").  Pass a pre-faulted `out` to keep first-touch
        page faults out of the timing."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        out_off = np.ascontiguousarray(out_off, dtype=np.int64)
        if out is None:
            out = np.full(int(out_off[-1]), -3, dtype=np.int64)
        secs = lib().orc_batch_search(self._p, bases.ctypes.data, read_off.ctypes.data, len(read_off) - 1,
                                      out.ctypes.data, out_off.ctypes.data, n_threads)
        return out, secs

    def batch_rank(self, pos: np.ndarray, sym: np.ndarray, n_threads: int = 1):
        """(out, seconds): SubsetMatrixRank::rank for every (pos, sym) pair."""
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        out = np.full(len(pos), -3, dtype=np.int64)
        secs = lib().orc_batch_rank(self._p, pos.ctypes.data, sym.ctypes.data, len(pos), out.ctypes.data, n_threads)
        return out, secs

    def count_work(self, bases: np.ndarray, read_off: np.ndarray):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        lib().orc_count_work(self._p, bases.ctypes.data, read_off.ctypes.data, len(read_off) - 1,
                             C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value


def print_vector(v: np.ndarray) -> bytes:
    """print_vector of src/CLI/sbwt_search.cpp:21-43."""
    v = np.ascontiguousarray(v, dtype=np.int64)
    buf = C.create_string_buffer(22 * len(v) + 2)
    n = lib().orc_print_vector(v.ctypes.data, len(v), buf)
    return buf.raw[:n]
