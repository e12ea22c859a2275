"""ctypes binding of the C ABI in include/sbwtgpu.h (libsbwtgpu.so).

Plumbing only: tests and bench.py call the HIP path through this exactly as a C/C++ host
would.  There is no Python or CPU implementation of any query behind these calls -- if the
shared library is missing, importing/using this module fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SBWTGPU_LIB", os.path.join(_HERE, "lib", "libsbwtgpu.so"))   # override: A/B builds only

OK = 0
ERR_INVALID_ARG = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3
ERR_NO_STREAMING = -4
ERR_PRECALC_TOO_LONG = -5
ERR_PRECALC_GT_K = -6
ERR_NOT_SINGLETON = -7
ERR_OOM = -8
ERR_READ_TOO_LONG = -9

# every symbol include/sbwtgpu.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "sbwtgpu_version", "sbwtgpu_last_error", "sbwtgpu_device_count", "sbwtgpu_set_tuning",
    "sbwtgpu_index_create", "sbwtgpu_index_destroy", "sbwtgpu_index_get_info", "sbwtgpu_index_get_precalc",
    "sbwtgpu_index_export_header", "sbwtgpu_index_blob", "sbwtgpu_index_copy_blob", "sbwtgpu_index_adopt", "sbwtgpu_index_bcast",
    "sbwtgpu_rank_batch", "sbwtgpu_streaming_search_batch", "sbwtgpu_search_batch",
    "sbwtgpu_streaming_search_batch_i32", "sbwtgpu_search_batch_i32",
    "sbwtgpu_update_interval_batch", "sbwtgpu_forward_batch",
    "sbwtgpu_build_plain_matrix", "sbwtgpu_free_plain_matrix",
    "sbwtgpu_partial_search_batch", "sbwtgpu_get_kmer_batch", "sbwtgpu_select_batch",
    "sbwtgpu_search_workspace_bytes", "sbwtgpu_streaming_search_dev", "sbwtgpu_search_dev",
    "sbwtgpu_streaming_search_dev_i32", "sbwtgpu_search_dev_i32",
    "sbwtgpu_rank_dev", "sbwtgpu_encode_bases_dev", "sbwtgpu_search_encoded_dev",
    "sbwtgpu_workspace_status", "sbwtgpu_workspace_stats", "sbwtgpu_kernel_times",
    "sbwtgpu_format_text_bound", "sbwtgpu_format_scratch_bytes", "sbwtgpu_format_results_dev",
    "sbwtgpu_search_text_batch", "sbwtgpu_search_text_stream", "sbwtgpu_free_host", "sbwtgpu_release_cached_buffers",
]


class SbwtGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"sbwtgpu error {code}: {msg}")
        self.code = code
        self.msg = msg


class IndexDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int64),
        ("A_bits", C.c_void_p), ("C_bits", C.c_void_p), ("G_bits", C.c_void_p), ("T_bits", C.c_void_p),
        ("suffix_group_starts", C.c_void_p),
        ("k", C.c_int64), ("n_kmers", C.c_int64), ("precalc_k", C.c_int64),
        ("precalc", C.c_void_p),
    ]


class IndexInfo(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int64), ("n_kmers", C.c_int64), ("k", C.c_int64), ("precalc_k", C.c_int64),
        ("C", C.c_int64 * 4),
        ("has_streaming_support", C.c_int32), ("device", C.c_int32),
        ("device_precalc_k", C.c_int64), ("blob_bytes", C.c_int64), ("image_level", C.c_int64),
        ("n_paths", C.c_int64), ("n_branch", C.c_int64), ("default_search_variant", C.c_int64),
    ]


class PlainMatrixBitsC(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_kmers", C.c_int64), ("k", C.c_int64),
                ("A_bits", C.c_void_p), ("C_bits", C.c_void_p), ("G_bits", C.c_void_p), ("T_bits", C.c_void_p),
                ("suffix_group_starts", C.c_void_p)]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Loads libsbwtgpu.so (built by `python -m sbwt_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m sbwt_amd.build` "
                          "(there is no CPU fallback for the GPU search path)")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i64, ci = C.c_void_p, C.c_int64, C.c_int
    L.sbwtgpu_version.restype = C.c_char_p
    L.sbwtgpu_last_error.restype = C.c_char_p
    L.sbwtgpu_device_count.argtypes = [C.POINTER(ci)]
    L.sbwtgpu_set_tuning.argtypes = [C.c_char_p, i64]
    L.sbwtgpu_index_create.argtypes = [C.POINTER(IndexDesc), ci, C.POINTER(vp)]
    L.sbwtgpu_index_destroy.argtypes = [vp]
    L.sbwtgpu_index_destroy.restype = None
    L.sbwtgpu_index_get_info.argtypes = [vp, C.POINTER(IndexInfo)]
    L.sbwtgpu_index_get_precalc.argtypes = [vp, vp]
    L.sbwtgpu_index_export_header.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.sbwtgpu_index_blob.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.sbwtgpu_index_copy_blob.argtypes = [vp, vp, i64, vp]
    L.sbwtgpu_index_adopt.argtypes = [vp, i64, vp, i64, ci, C.POINTER(vp)]
    L.sbwtgpu_index_bcast.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(vp)]
    L.sbwtgpu_rank_batch.argtypes = [vp, vp, vp, i64, vp]
    L.sbwtgpu_streaming_search_batch.argtypes = [vp, vp, vp, i64, vp, vp]
    L.sbwtgpu_search_batch.argtypes = [vp, vp, vp, i64, vp, vp]
    try:                                    # (absent from older builds loaded through SBWTGPU_LIB for A/B runs)
        L.sbwtgpu_streaming_search_batch_i32.argtypes = [vp, vp, vp, i64, vp, vp]
        L.sbwtgpu_search_batch_i32.argtypes = [vp, vp, vp, i64, vp, vp]
    except AttributeError:
        if "SBWTGPU_LIB" not in os.environ:
            raise
    L.sbwtgpu_update_interval_batch.argtypes = [vp, vp, vp, i64, vp, vp]
    L.sbwtgpu_forward_batch.argtypes = [vp, vp, vp, i64, vp]
    L.sbwtgpu_build_plain_matrix.argtypes = [C.POINTER(C.c_char_p), vp, i64, i64, ci, ci, ci, C.POINTER(PlainMatrixBitsC)]
    L.sbwtgpu_free_plain_matrix.argtypes = [C.POINTER(PlainMatrixBitsC)]
    L.sbwtgpu_free_plain_matrix.restype = None
    L.sbwtgpu_partial_search_batch.argtypes = [vp, vp, vp, i64, vp, vp, vp]
    L.sbwtgpu_get_kmer_batch.argtypes = [vp, vp, i64, vp]
    L.sbwtgpu_select_batch.argtypes = [vp, vp, vp, i64, vp]
    L.sbwtgpu_search_workspace_bytes.argtypes = [i64]
    L.sbwtgpu_search_workspace_bytes.restype = i64
    L.sbwtgpu_streaming_search_dev.argtypes = [vp, vp, i64, vp, i64, vp, vp, vp, i64, vp]
    L.sbwtgpu_search_dev.argtypes = [vp, vp, i64, vp, i64, vp, vp, vp, i64, vp]
    try:
        L.sbwtgpu_streaming_search_dev_i32.argtypes = [vp, vp, i64, vp, i64, vp, vp, vp, i64, vp]
        L.sbwtgpu_search_dev_i32.argtypes = [vp, vp, i64, vp, i64, vp, vp, vp, i64, vp]
    except AttributeError:                      # an older A/B build named by SBWTGPU_LIB
        if not os.environ.get("SBWTGPU_LIB"):
            raise
    L.sbwtgpu_rank_dev.argtypes = [vp, vp, vp, i64, vp, vp]
    L.sbwtgpu_workspace_status.argtypes = [vp, vp, C.POINTER(ci)]
    L.sbwtgpu_encode_bases_dev.argtypes = [vp, vp, i64, vp, i64, vp]
    L.sbwtgpu_search_encoded_dev.argtypes = [vp, i64, vp, i64, vp, vp, vp, i64, ci, vp]
    L.sbwtgpu_workspace_stats.argtypes = [vp, vp, C.POINTER(i64)]
    L.sbwtgpu_kernel_times.argtypes = [vp, i64, C.POINTER(i64)]
    L.sbwtgpu_format_text_bound.argtypes = [vp, i64, i64]
    L.sbwtgpu_format_text_bound.restype = i64
    L.sbwtgpu_format_scratch_bytes.argtypes = [i64]
    L.sbwtgpu_format_scratch_bytes.restype = i64
    L.sbwtgpu_format_results_dev.argtypes = [vp, vp, vp, i64, i64, vp, i64, vp, vp, i64, vp]
    L.sbwtgpu_search_text_batch.argtypes = [vp, vp, vp, i64, ci, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]
    L.sbwtgpu_free_host.argtypes = [vp]
    L.sbwtgpu_free_host.restype = None
    L.sbwtgpu_release_cached_buffers.restype = None
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != OK:
        raise SbwtGpuError(rc, lib().sbwtgpu_last_error().decode(errors="replace"))


def set_tuning(key: str, value: int) -> None:
    _check(lib().sbwtgpu_set_tuning(key.encode(), value))


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().sbwtgpu_device_count(C.byref(n))
    return n.value if rc == OK else 0


def _words(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def concat_reads(reads: Sequence[bytes]):
    """bases / read_off arrays for a list of byte strings."""
    lens = np.fromiter((len(r) for r in reads), dtype=np.int64, count=len(reads))
    off = np.zeros(len(reads) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if len(reads) else np.zeros(0, np.uint8)
    return bases, off


def out_offsets(read_off: np.ndarray, k: int) -> np.ndarray:
    lens = np.diff(read_off)
    m = np.maximum(lens - k + 1, 0)
    off = np.zeros(len(read_off), dtype=np.int64)
    np.cumsum(m, out=off[1:])
    return off


class BuiltBits:
    """What the builders return: the four rows + suffix_group_starts as uint64 word arrays (numpy copies)."""

    def __init__(self, cols, ssup, n_nodes, n_kmers, k):
        self.cols, self.ssup, self.n_nodes, self.n_kmers, self.k = cols, ssup, n_nodes, n_kmers, k


def build_bits_gpu(seqs: Sequence[bytes], k: int, add_revcomp: bool = False, streaming_support: bool = True,
                   device: int = 0) -> BuiltBits:
    """sbwtgpu_build_plain_matrix: the plain-matrix SBWT columns of `seqs`, built on the GPU (2 <= k <= 64)."""
    arr = (C.c_char_p * len(seqs))(*[bytes(s) for s in seqs])
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    out = PlainMatrixBitsC()
    _check(lib().sbwtgpu_build_plain_matrix(arr, lens.ctypes.data, len(seqs), k, int(add_revcomp), int(streaming_support),
                                            device, C.byref(out)))
    try:
        nw = (out.n_nodes + 63) // 64

        def words(p):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(nw,)).copy()
        cols = [words(out.A_bits), words(out.C_bits), words(out.G_bits), words(out.T_bits)]
        ssup = words(out.suffix_group_starts) if out.suffix_group_starts else None
        return BuiltBits(cols, ssup, out.n_nodes, out.n_kmers, out.k)
    finally:
        lib().sbwtgpu_free_plain_matrix(C.byref(out))


class Index:
    """Owning wrapper of a `sbwtgpu_index*` (one GPU)."""

    def __init__(self, handle: int, keepalive=None):
        self._h = C.c_void_p(handle)
        self._keep = keepalive
        info = IndexInfo()
        _check(lib().sbwtgpu_index_get_info(self._h, C.byref(info)))
        self.n_nodes, self.n_kmers, self.k = info.n_nodes, info.n_kmers, info.k
        self.precalc_k = info.precalc_k
        self.C = [info.C[i] for i in range(4)]
        self.has_streaming_support = bool(info.has_streaming_support)
        self.device = info.device
        self.device_precalc_k = info.device_precalc_k
        self.blob_bytes = info.blob_bytes
        self.image_level = info.image_level
        self.n_paths, self.n_branch, self.default_search_variant = info.n_paths, info.n_branch, info.default_search_variant

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    @classmethod
    def create(cls, A, Cb, G, T, ssup, n_nodes: int, k: int, n_kmers: int = 0, precalc_k: int = 0,
               precalc=None, device: int = 0) -> "Index":
        A, Cb, G, T = _words(A), _words(Cb), _words(G), _words(T)
        s = _words(ssup) if ssup is not None else None
        pc = np.ascontiguousarray(precalc, dtype=np.int64) if precalc is not None else None
        d = IndexDesc(n_nodes, A.ctypes.data, Cb.ctypes.data, G.ctypes.data, T.ctypes.data,
                      s.ctypes.data if s is not None else None, k, n_kmers, precalc_k,
                      pc.ctypes.data if pc is not None else None)
        h = C.c_void_p()
        _check(lib().sbwtgpu_index_create(C.byref(d), device, C.byref(h)))
        return cls(h.value)

    @classmethod
    def adopt(cls, header: bytes, dev_ptr: int, blob_bytes: int, device: int, keepalive=None) -> "Index":
        h = C.c_void_p()
        buf = C.create_string_buffer(header, len(header))
        _check(lib().sbwtgpu_index_adopt(buf, len(header), C.c_void_p(dev_ptr), blob_bytes, device, C.byref(h)))
        return cls(h.value, keepalive=keepalive)

    def close(self) -> None:
        if self._h:
            lib().sbwtgpu_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- replication helpers ----
    def export_header(self) -> bytes:
        n = C.c_int64(0)
        _check(lib().sbwtgpu_index_export_header(self._h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(lib().sbwtgpu_index_export_header(self._h, buf, n.value, C.byref(n)))
        return buf.raw[: n.value]

    def blob(self):
        p, n = C.c_void_p(), C.c_int64(0)
        _check(lib().sbwtgpu_index_blob(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def blob_tensor(self):
        """The device image as a uint8 torch tensor that ALIASES it (no copy): what rank 0 hands to the broadcast.  The
        tensor is valid while this Index is."""
        import torch
        ptr, n = self.blob()

        class _Alias:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}
        t = torch.as_tensor(_Alias(), device=torch.device("cuda", self.device))
        t._sbwt_owner = self          # keeps the image alive as long as the tensor
        return t

    def copy_blob(self, dst_dev_ptr: int, nbytes: int, stream: int = 0) -> None:
        _check(lib().sbwtgpu_index_copy_blob(self._h, dst_dev_ptr, nbytes, stream))

    def get_precalc(self) -> np.ndarray:
        out = np.zeros((4 ** self.precalc_k if self.precalc_k else 0, 2), dtype=np.int64)
        if self.precalc_k:
            _check(lib().sbwtgpu_index_get_precalc(self._h, out.ctypes.data))
        return out

    # ---- host-buffer queries ----
    def rank(self, pos, sym) -> np.ndarray:
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        out = np.empty(len(pos), dtype=np.int64)
        _check(lib().sbwtgpu_rank_batch(self._h, pos.ctypes.data, sym.ctypes.data, len(pos), out.ctypes.data))
        return out

    def _search(self, fn, bases, read_off, out_off=None):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        if out_off is None:
            out_off = out_offsets(read_off, self.k)
        out_off = np.ascontiguousarray(out_off, dtype=np.int64)
        out = np.full(int(out_off[-1]) if len(out_off) else 0, -12345, dtype=np.int64)
        _check(fn(self._h, bases.ctypes.data, read_off.ctypes.data, len(read_off) - 1, out.ctypes.data,
                  out_off.ctypes.data))
        return out, out_off

    def streaming_search(self, bases, read_off, out_off=None):
        return self._search(lib().sbwtgpu_streaming_search_batch, bases, read_off, out_off)

    def search(self, bases, read_off, out_off=None):
        return self._search(lib().sbwtgpu_search_batch, bases, read_off, out_off)

    def search_i32(self, bases, read_off, streaming: bool = True, out_off=None):
        """The same with int32 results (sbwtgpu_*_batch_i32: half the bytes over PCIe; indexes of fewer than 2^31 columns)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        if out_off is None:
            out_off = out_offsets(read_off, self.k)
        out_off = np.ascontiguousarray(out_off, dtype=np.int64)
        out = np.full(int(out_off[-1]) if len(out_off) else 0, -12345, dtype=np.int32)
        fn = lib().sbwtgpu_streaming_search_batch_i32 if streaming else lib().sbwtgpu_search_batch_i32
        _check(fn(self._h, bases.ctypes.data, read_off.ctypes.data, len(read_off) - 1, out.ctypes.data, out_off.ctypes.data))
        return out, out_off

    def streaming_search_reads(self, reads: Sequence[bytes]):
        bases, off = concat_reads(reads)
        out, oo = self.streaming_search(bases, off)
        return [out[oo[i]:oo[i + 1]] for i in range(len(reads))]

    def search_reads(self, reads: Sequence[bytes]):
        bases, off = concat_reads(reads)
        out, oo = self.search(bases, off)
        return [out[oo[i]:oo[i + 1]] for i in range(len(reads))]

    def search_text(self, bases, read_off, streaming: bool = True):
        """The formatted `sbwt search` output of a batch (device-side print_vector, pipelined host path):
        returns (text bytes, number of k-mers searched)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        p, n, q = C.c_void_p(), C.c_int64(0), C.c_int64(0)
        _check(lib().sbwtgpu_search_text_batch(self._h, bases.ctypes.data, read_off.ctypes.data, len(read_off) - 1,
                                               int(streaming), C.byref(p), C.byref(n), C.byref(q)))
        try:
            return C.string_at(p.value, n.value), q.value
        finally:
            lib().sbwtgpu_free_host(p)

    def update_interval(self, bases, off, first, second):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int64)
        first = np.array(first, dtype=np.int64)
        second = np.array(second, dtype=np.int64)
        _check(lib().sbwtgpu_update_interval_batch(self._h, bases.ctypes.data, off.ctypes.data, len(first),
                                                   first.ctypes.data, second.ctypes.data))
        return first, second

    def forward(self, node, sym) -> np.ndarray:
        node = np.ascontiguousarray(node, dtype=np.int64)
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        out = np.empty(len(node), dtype=np.int64)
        _check(lib().sbwtgpu_forward_batch(self._h, node.ctypes.data, sym.ctypes.data, len(node), out.ctypes.data))
        return out

    def partial_search(self, bases, off):
        """SBWT::partial_search for every query: (first, second, matched_len) arrays."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int64)
        n = len(off) - 1
        first, second, matched = (np.empty(n, dtype=np.int64) for _ in range(3))
        _check(lib().sbwtgpu_partial_search_batch(self._h, bases.ctypes.data, off.ctypes.data, n, first.ctypes.data,
                                                  second.ctypes.data, matched.ctypes.data))
        return first, second, matched

    def get_kmers(self, colex_ranks) -> np.ndarray:
        """SBWT::get_kmer for every column: an (n, k) uint8 array of ASCII chars ('$' = dummy prefix)."""
        cr = np.ascontiguousarray(colex_ranks, dtype=np.int64)
        out = np.empty((len(cr), self.k), dtype=np.uint8)
        _check(lib().sbwtgpu_get_kmer_batch(self._h, cr.ctypes.data, len(cr), out.ctypes.data))
        return out

    def select(self, j, sym) -> np.ndarray:
        j = np.ascontiguousarray(j, dtype=np.int64)
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        out = np.empty(len(j), dtype=np.int64)
        _check(lib().sbwtgpu_select_batch(self._h, j.ctypes.data, sym.ctypes.data, len(j), out.ctypes.data))
        return out

    # ---- device-buffer queries (raw pointers; torch tensors' data_ptr() go here) ----
    def streaming_search_dev(self, d_bases: int, total_bases: int, d_read_off: int, n_reads: int, d_out: int,
                             d_out_off: int, d_ws: int, ws_bytes: int, stream: int = 0, streaming: bool = True):
        fn = lib().sbwtgpu_streaming_search_dev if streaming else lib().sbwtgpu_search_dev
        _check(fn(self._h, d_bases, total_bases, d_read_off, n_reads, d_out, d_out_off, d_ws, ws_bytes, stream))

    def streaming_search_dev_i32(self, d_bases: int, total_bases: int, d_read_off: int, n_reads: int, d_out32: int,
                                 d_out_off: int, d_ws: int, ws_bytes: int, stream: int = 0, streaming: bool = True):
        """The same with an int32 result array on the device (sbwtgpu_*_dev_i32)."""
        fn = lib().sbwtgpu_streaming_search_dev_i32 if streaming else lib().sbwtgpu_search_dev_i32
        _check(fn(self._h, d_bases, total_bases, d_read_off, n_reads, d_out32, d_out_off, d_ws, ws_bytes, stream))

    def encode_bases_dev(self, d_bases: int, total_bases: int, d_ws: int, ws_bytes: int, stream: int = 0):
        _check(lib().sbwtgpu_encode_bases_dev(self._h, d_bases, total_bases, d_ws, ws_bytes, stream))

    def search_encoded_dev(self, total_bases: int, d_read_off: int, n_reads: int, d_out: int, d_out_off: int,
                           d_ws: int, ws_bytes: int, streaming: bool = True, stream: int = 0):
        _check(lib().sbwtgpu_search_encoded_dev(self._h, total_bases, d_read_off, n_reads, d_out, d_out_off, d_ws,
                                                ws_bytes, int(streaming), stream))

    def workspace_stats(self, d_ws: int, stream: int = 0):
        """(n_stream, n_search, n_lf, n_tab_hit, n_ext) of the last search on this workspace."""
        st = (C.c_int64 * 8)()
        _check(lib().sbwtgpu_workspace_stats(d_ws, stream, st))
        return tuple(int(x) for x in st[:5])

    def workspace_bridges(self, d_ws: int, stream: int = 0) -> int:
        """Substitutions bridged along a path (M_BRIDGE) by the last search on this workspace."""
        st = (C.c_int64 * 8)()
        _check(lib().sbwtgpu_workspace_stats(d_ws, stream, st))
        return int(st[5])

    def workspace_status(self, d_ws: int, stream: int = 0) -> int:
        st = C.c_int(0)
        _check(lib().sbwtgpu_workspace_status(d_ws, stream, C.byref(st)))
        return st.value


def kernel_times() -> list:
    """Durations (ms) of the dominant kernel of the search calls since set_tuning("kernel_events", 1)."""
    buf = (C.c_double * 256)()
    n = C.c_int64(0)
    _check(lib().sbwtgpu_kernel_times(buf, 256, C.byref(n)))
    return [float(buf[i]) for i in range(n.value)]


def search_workspace_bytes(total_bases: int) -> int:
    return int(lib().sbwtgpu_search_workspace_bytes(total_bases))
