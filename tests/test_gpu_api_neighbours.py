"""GPU parity tests of the batched API neighbours of the search path, through the C ABI:
SubsetMatrixRank::rank on arbitrary bit vectors (the 4-pairs-per-lane kernel, its scalar tail, the 64-bit count
layout), SBWT::partial_search, SBWT::get_kmer / get_kmer_fast and SubsetMatrixSelectSupport::select."""
import numpy as np
import pytest

from oracle import OracleIndex
from sbwt_amd import capi, hostlib, synth

pytestmark = pytest.mark.gpu


def gpu_index_from_oracle(orc: OracleIndex) -> capi.Index:
    cols = orc.columns()
    return capi.Index.create(cols[0], cols[1], cols[2], cols[3], orc.ssup_words(), orc.n_nodes, orc.k,
                             orc.n_kmers, orc.precalc_k, None)


@pytest.fixture(scope="module")
def case():
    k = 31
    genomes = [synth.random_genome(150_000, 11)]
    genomes.append(synth.mutate(genomes[0], 0.03, 12))
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 6)
    return genomes, orc


@pytest.mark.parametrize("force_mega", [0, 1])
@pytest.mark.parametrize("n_bits", [1, 63, 64, 65, 4097, 1_000_003])
def test_rank_arbitrary_bit_vectors(gpu, n_bits, force_mega):
    """SubsetMatrixRank as a stand-alone structure (SubsetMatrixRank.hh:52-58): four unrelated random rows."""
    rng = np.random.default_rng(n_bits)
    nw = (n_bits + 63) // 64
    cols = [rng.integers(0, 2**64, size=nw, dtype=np.uint64) for _ in range(4)]
    capi.set_tuning("force_mega", force_mega)
    try:
        idx = capi.Index.create(cols[0], cols[1], cols[2], cols[3], None, n_bits, 1, 0, 0)
    finally:
        capi.set_tuning("force_mega", 0)
    orc = OracleIndex.from_bits(cols[0], cols[1], cols[2], cols[3], None, n_bits, 1, 0, 0)
    for n in (1, 2, 3, 4, 5, 7, 8, 1023, 20_001):      # n % 4 != 0 exercises the scalar tail next to the 4-wide kernel
        pos = rng.integers(0, n_bits + 1, size=n)
        pos[0] = n_bits
        pos[-1] = 0
        sym = rng.choice(np.frombuffer(b"ACGTNacgt$\x00\xff", dtype=np.uint8), size=n)
        got = idx.rank(pos, sym)
        want, _ = orc.batch_rank(pos, sym, 1)
        assert np.array_equal(got, want), (n_bits, n)


def test_rank_dev_misaligned_buffers(gpu, case):
    """The device entry point takes any 8-byte aligned pos/out and any sym pointer: misaligned ones take the scalar kernel."""
    import torch
    _, orc = case
    idx = gpu_index_from_oracle(orc)
    rng = np.random.default_rng(5)
    n = 10_001
    pos = rng.integers(0, orc.n_nodes + 1, size=n)
    sym = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
    want, _ = orc.batch_rank(pos, sym, 1)
    dev = torch.device("cuda", 0)
    for shift_pos, shift_sym in ((0, 0), (1, 0), (0, 1), (1, 3)):
        d_pos = torch.zeros(n + 2, dtype=torch.int64, device=dev)
        d_sym = torch.zeros(n + 8, dtype=torch.uint8, device=dev)
        d_out = torch.full((n + 2,), -9, dtype=torch.int64, device=dev)
        d_pos[shift_pos:shift_pos + n] = torch.from_numpy(pos).to(dev)
        d_sym[shift_sym:shift_sym + n] = torch.from_numpy(sym).to(dev)
        capi._check(capi.lib().sbwtgpu_rank_dev(idx.handle, d_pos.data_ptr() + 8 * shift_pos, d_sym.data_ptr() + shift_sym,
                                                n, d_out.data_ptr() + 8 * shift_pos,
                                                torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert np.array_equal(d_out[shift_pos:shift_pos + n].cpu().numpy(), want)


def test_partial_search_batch(gpu, case):
    genomes, orc = case
    idx = gpu_index_from_oracle(orc)
    rng = np.random.default_rng(9)
    g = genomes[0]
    queries = []
    for _ in range(3000):
        L = int(rng.integers(0, 80))
        s = int(rng.integers(0, len(g) - 100))
        q = bytearray(g[s:s + L].tobytes())
        r = rng.random()
        if L and r < 0.3:
            q[int(rng.integers(0, L))] = ord("ACGT"[int(rng.integers(0, 4))])     # a substitution somewhere
        elif L and r < 0.4:
            q[int(rng.integers(0, L))] = ord("N")
        elif L and r < 0.5:
            q = bytearray(bytes(q).lower())                                         # lower case matches (SBWT.hh:529)
        queries.append(bytes(q))
    queries += [b"", b"N", b"a", b"$", b"ACGT" * 30]
    bases, off = capi.concat_reads(queries)
    first, second, matched = idx.partial_search(bases, off)
    for q, a, c, m in zip(queries, first, second, matched):
        (wl, wr), wm = orc.partial_search(q)
        assert (a, c, m) == (wl, wr, wm), q


def test_get_kmer_and_select(gpu, case):
    genomes, orc = case
    idx = gpu_index_from_oracle(orc)
    rng = np.random.default_rng(10)
    n = orc.n_nodes
    cols = np.concatenate([rng.integers(0, n, size=3000), np.array([0, 1, 2, n - 1, n - 2])])
    got = idx.get_kmers(cols)
    for v, row in zip(cols, got):
        assert row.tobytes() == orc.get_kmer(int(v)), int(v)
    # a found k-mer spells itself
    bases, off = synth.sample_reads(genomes, 50, 31, 0.0, 3)
    res, _ = idx.search(bases, off)
    assert (res >= 0).all()
    back = idx.get_kmers(res)
    for r in range(50):
        assert back[r].tobytes() == bases[off[r]:off[r + 1]].tobytes()
    # select inverts rank on set bits; first and last one of every row; non-ACGT -> 0
    words = orc.columns()
    for ci, ch in enumerate(b"ACGT"):
        bits = np.unpackbits(words[ci].view(np.uint8), bitorder="little")[:n]
        ones = np.flatnonzero(bits)
        pick = np.unique(np.concatenate([[0, len(ones) - 1], rng.integers(0, len(ones), size=2000)]))
        got = idx.select(pick + 1, np.full(len(pick), ch, dtype=np.uint8))
        assert np.array_equal(got, ones[pick])
        with pytest.raises(capi.SbwtGpuError):
            idx.select(np.array([len(ones) + 1]), np.array([ch], dtype=np.uint8))
    assert list(idx.select(np.array([5, 6]), np.frombuffer(b"N$", dtype=np.uint8))) == [0, 0]


def test_small_calls_equal_large_calls(gpu, case):
    """Batches of one go through the per-thread small-call slots, big batches through temporary buffers: same bits."""
    genomes, orc = case
    idx = gpu_index_from_oracle(orc)
    bases, off = synth.sample_reads(genomes, 3000, 150, 0.01, 77)
    bases = synth.inject(bases, 40, ord("N"), 7)
    big, oo = idx.streaming_search(bases, off)                     # one large call
    for r in list(range(40)) + [2999]:
        one, _ = idx.streaming_search(bases[off[r]:off[r + 1]], np.array([0, off[r + 1] - off[r]]))
        assert np.array_equal(one, big[oo[r]:oo[r + 1]])
        kmer = bases[off[r] + 7:off[r] + 7 + orc.k]
        got, _ = idx.search(kmer, np.array([0, orc.k]))
        assert got[0] == orc.search(kmer.tobytes())
    # a call larger than the slot (1 MiB) right after small ones, and small ones again
    big2, _ = idx.streaming_search(bases, off)
    assert np.array_equal(big, big2)
    one, _ = idx.streaming_search(bases[:150], np.array([0, 150]))
    assert np.array_equal(one, big[:oo[1]])
    capi.lib().sbwtgpu_release_cached_buffers()
    one, _ = idx.streaming_search(bases[:150], np.array([0, 150]))
    assert np.array_equal(one, big[:oo[1]])


def test_image_levels_and_memory_cap(gpu, case):
    """The derived structures are optional: a capped / minimal image gives the same bits."""
    genomes, orc = case
    bases, off = synth.sample_reads(genomes, 3000, 150, 0.02, 99)
    bases = synth.inject(bases, 50, ord("N"), 3)
    want = np.concatenate([orc.streaming_search(bases[off[r]:off[r + 1]].tobytes()) for r in range(3000)])
    sizes = []
    for level in (0, 1, 2):
        capi.set_tuning("image_level", level)
        try:
            idx = gpu_index_from_oracle(orc)
        finally:
            capi.set_tuning("image_level", 0)
        assert idx.image_level == level
        sizes.append(idx.blob_bytes)
        assert np.array_equal(idx.streaming_search(bases, off)[0], want)
        assert np.array_equal(idx.search(bases, off)[0], want)
    assert sizes[0] > sizes[1]          # level 2 carries a deeper dense table instead: smaller only for large indexes
    # a cap selects the first level whose image fits; at level 2 the dense table gets as shallow as it has to
    for cap, expect in ((sizes[0] - 1, 1), (sizes[1] - 1, 2), (sizes[0], 0), (2_000_000, 2)):
        capi.set_tuning("max_image_bytes", cap)
        try:
            idx = gpu_index_from_oracle(orc)
        finally:
            capi.set_tuning("max_image_bytes", 0)
        assert idx.image_level == expect and idx.blob_bytes <= cap, (cap, idx.image_level, idx.blob_bytes)
        assert np.array_equal(idx.streaming_search(bases, off)[0], want)
    capi.set_tuning("max_image_bytes", 1000)
    try:
        with pytest.raises(capi.SbwtGpuError) as ei:
            gpu_index_from_oracle(orc)
        assert ei.value.code == capi.ERR_OOM
    finally:
        capi.set_tuning("max_image_bytes", 0)


def test_text_stream_pieces_and_kernel_events(gpu):
    """sbwtgpu_search_text_stream hands out the text of sbwtgpu_search_text_batch piece by piece, in order; the library's
    kernel events time the fused kernel of every device-pointer call (what bench.py reports as roofline.kernel_ms)."""
    import ctypes as C
    import torch
    genomes = synth.coli3_like(200_000)
    bits = hostlib.build_bits([g.tobytes() for g in genomes], 30, False, True, n_threads=8)
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, 30, bits.n_kmers, 8)
    bases, off = synth.sample_reads(genomes, 300_000, 100, 0.01, 11)        # 30 M bases: several pipeline chunks
    bases = synth.inject(bases, 500, ord("N"), 3)
    want, nq = idx.search_text(bases, off, True)
    pieces = []
    SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)

    def sink(_ctx, text, n):
        pieces.append(C.string_at(text, n))
        return 0
    cb = SINK(sink)
    n_q = C.c_int64(0)
    L = capi.lib()
    L.sbwtgpu_search_text_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, SINK, C.c_void_p,
                                             C.POINTER(C.c_int64)]
    capi._check(L.sbwtgpu_search_text_stream(idx.handle, bases.ctypes.data, off.ctypes.data, len(off) - 1, 1, cb, None,
                                             C.byref(n_q)))
    assert len(pieces) >= 3 and b"".join(pieces) == want and n_q.value == nq
    # a sink that gives up ends the call with an error instead of going on
    bad = SINK(lambda _c, _t, _n: 1)
    assert L.sbwtgpu_search_text_stream(idx.handle, bases.ctypes.data, off.ctypes.data, len(off) - 1, 1, bad, None,
                                        C.byref(n_q)) != 0
    # kernel events
    dev = torch.device("cuda:0")
    n = 200_000
    d_b = torch.from_numpy(bases[: n * 100].copy()).to(dev)
    d_ro = torch.arange(n + 1, dtype=torch.int64, device=dev) * 100
    d_oo = torch.arange(n + 1, dtype=torch.int64, device=dev) * 71
    d_out = torch.empty(n * 71, dtype=torch.int64, device=dev)
    wsb = capi.search_workspace_bytes(d_b.numel())
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    capi.set_tuning("kernel_events", 1)
    try:
        for _ in range(3):
            idx.streaming_search_dev(d_b.data_ptr(), d_b.numel(), d_ro.data_ptr(), n, d_out.data_ptr(), d_oo.data_ptr(),
                                     d_ws.data_ptr(), wsb, st, True)
        kt = capi.kernel_times()
    finally:
        capi.set_tuning("kernel_events", 0)
    assert len(kt) == 3 and all(0 < t < 1000 for t in kt)
    ref, _ = idx.streaming_search(bases[: n * 100], off[: n + 1])
    assert np.array_equal(d_out.cpu().numpy(), ref)


def test_int32_results_small_and_pipelined(gpu):
    """sbwtgpu_streaming_search_batch_i32 / sbwtgpu_search_batch_i32 (result compaction): the same values as
    the int64 calls, written as int32 by the kernels -- a small batch (host loop) and one large enough for the two-stream pipeline,
    pageable and pinned destinations, N and short reads among them."""
    import torch
    genomes = [synth.random_genome(300_000, 3)]
    genomes.append(synth.mutate(genomes[0], 0.05, 4))
    bits = capi.build_bits_gpu([g.tobytes() for g in genomes], 30, False, True)
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, 30, bits.n_kmers, 8)
    for n_reads in (300, 200_000):                       # 200 000 x 121 x 4 B = 97 MB of int32 results: pipelined
        bases, off = synth.sample_reads(genomes, n_reads, 150, 0.01, 9)
        bases = synth.inject(bases, 50, ord("N"), 2)
        want, oo = idx.streaming_search(bases, off)
        for streaming in (True, False):
            got, oo2 = idx.search_i32(bases, off, streaming)
            assert got.dtype == np.int32 and np.array_equal(oo, oo2)
            assert np.array_equal(got.astype(np.int64), want if streaming else idx.search(bases, off)[0])
    # pinned destination: the DMA's target itself
    hb = torch.from_numpy(bases).pin_memory()
    ho = torch.empty(int(oo[-1]), dtype=torch.int32).pin_memory()
    capi._check(capi.lib().sbwtgpu_streaming_search_batch_i32(idx.handle, hb.data_ptr(), off.ctypes.data, len(off) - 1, ho.data_ptr(),
                                                              oo.ctypes.data))
    assert np.array_equal(ho.numpy().astype(np.int64), want)


@pytest.mark.parametrize("k,streaming", [(30, True), (31, False), (63, False)])
def test_int32_results_on_the_device_every_route(gpu, k, streaming):
    """sbwtgpu_streaming_search_dev_i32 / sbwtgpu_search_dev_i32: every kernel of every route writes int32 results into the
    caller's device array -- the same values as the int64 call, nothing outside the result range touched.  Reads of one
    length, ragged ones (some shorter than k), reads of 250 and 5 000 bases (pieces / zones), N and lower case (handed on to
    the general kernel), with the results poisoned first."""
    import torch
    dev = torch.device("cuda:0")
    genomes = [synth.random_genome(200_000, 21)]
    genomes.append(synth.mutate(genomes[0], 0.05, 22))
    bits = capi.build_bits_gpu([g.tobytes() for g in genomes], k, False, streaming)
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k, bits.n_kmers, 8)
    batches = [synth.sample_reads(genomes, 6000, 150, 0.01, 31), synth.ragged_reads(genomes, 5000, 20, 250, 0.01, 32),
               synth.sample_reads(genomes, 3000, 250, 0.01, 33), synth.sample_reads(genomes, 40, 5000, 0.01, 34)]
    b4, o4 = synth.sample_reads(genomes, 4000, 150, 0.01, 35)
    b4 = synth.inject(synth.inject(b4, 300, ord("N"), 5), 300, ord("a"), 6)
    batches.append((b4, o4))
    st = torch.cuda.current_stream().cuda_stream
    capi.set_tuning("poison_results", 1)
    try:
        for bases, off in batches:
            ooff = capi.out_offsets(off, k)
            n_out = int(ooff[-1])
            d_b, d_ro, d_oo = torch.from_numpy(bases).to(dev), torch.from_numpy(off).to(dev), torch.from_numpy(ooff).to(dev)
            wsb = capi.search_workspace_bytes(d_b.numel())
            d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            d64 = torch.empty(n_out, dtype=torch.int64, device=dev)
            for variant in (5, 4, 1, 0):
                capi.set_tuning("search_variant", variant)
                try:
                    idx.streaming_search_dev(d_b.data_ptr(), d_b.numel(), d_ro.data_ptr(), len(off) - 1, d64.data_ptr(), d_oo.data_ptr(),
                                             d_ws.data_ptr(), wsb, st, streaming)
                    d32 = torch.full((n_out + 64,), 77, dtype=torch.int32, device=dev)     # 64 guard values behind the results
                    idx.streaming_search_dev_i32(d_b.data_ptr(), d_b.numel(), d_ro.data_ptr(), len(off) - 1, d32.data_ptr(),
                                                 d_oo.data_ptr(), d_ws.data_ptr(), wsb, st, streaming)
                finally:
                    capi.set_tuning("search_variant", -1)
                torch.cuda.synchronize()
                assert torch.equal(d32[:n_out].to(torch.int64), d64), (variant, len(off))
                assert bool((d32[n_out:] == 77).all()), variant
    finally:
        capi.set_tuning("poison_results", 0)
