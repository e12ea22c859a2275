"""Concurrent readers of ONE index handle.  The reference's query methods are const and hold no mutable state
(SBWT.hh:175-245: re-entrant for concurrent readers); the C ABI promises the same -- a handle is immutable after create, batch
calls may run concurrently, one HIP stream per call / thread (include/sbwtgpu.h).  Two shapes: host threads calling the
host-buffer entry points at the same time, and device-pointer calls queued on two streams with two workspaces (two batches
in flight, bench.py's `two_batches_in_flight`)."""
import threading

import numpy as np
import pytest

from oracle import OracleIndex
from sbwt_amd import capi, synth

pytestmark = pytest.mark.gpu


def _index(k, streaming=True):
    g0 = synth.random_genome(150_000, 11)
    genomes = [g0, synth.mutate(g0, 0.05, 12)]
    bits = capi.build_bits_gpu([g.tobytes() for g in genomes], k, False, streaming)
    orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k, bits.n_kmers, 8)
    idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, k, bits.n_kmers, 8)
    return genomes, orc, idx


def _want(orc, bases, off, streaming=True):
    f = orc.streaming_search if streaming else (lambda s: orc.search_all(s))
    out = [f(bases[off[r]:off[r + 1]].tobytes()) for r in range(len(off) - 1)]
    return np.concatenate(out) if out else np.zeros(0, np.int64)


@pytest.mark.parametrize("k", [30, 63])
def test_host_threads_share_one_handle(gpu, k):
    genomes, orc, idx = _index(k)
    batches = []
    for t in range(4):
        if t == 3:
            bases, off = synth.ragged_reads(genomes, 1500, 20, 250, 0.01, 70 + t)       # another route through the library
        else:
            bases, off = synth.sample_reads(genomes, 2000, 150, 0.01, 70 + t)
        batches.append((bases, off, _want(orc, bases, off)))
    got = [None] * len(batches)
    errs = []

    def worker(t):
        try:
            for rep in range(3):                   # several calls per thread, so that calls of different threads overlap
                bases, off, _ = batches[t]
                if t == 1:
                    got[t] = idx.search_i32(bases, off)[0].astype(np.int64)
                else:
                    got[t] = idx.streaming_search(bases, off)[0]
        except Exception as e:                     # noqa: BLE001 - reported below
            errs.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(len(batches))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    for t, (_, _, want) in enumerate(batches):
        assert np.array_equal(got[t], want), t


@pytest.mark.parametrize("k,streaming", [(30, True), (63, False)])
def test_two_batches_in_flight_on_two_streams(gpu, k, streaming):
    import torch
    genomes, orc, idx = _index(k, streaming)
    dev = torch.device("cuda", 0)
    sets = []
    for q in range(2):
        bases, off = synth.sample_reads(genomes, 40_000, 150, 0.01, 80 + q)
        ooff = capi.out_offsets(off, k)
        d = {"bases": torch.from_numpy(bases).to(dev), "off": torch.from_numpy(off).to(dev), "ooff": torch.from_numpy(ooff).to(dev),
             "out": torch.full((int(ooff[-1]),), -7, dtype=torch.int64, device=dev), "n": len(off) - 1,
             "stream": torch.cuda.Stream(device=dev), "host": (bases, off)}
        d["wsb"] = capi.search_workspace_bytes(d["bases"].numel())
        d["ws"] = torch.empty(d["wsb"], dtype=torch.uint8, device=dev)
        sets.append(d)
    torch.cuda.synchronize()
    for rep in range(6):                           # launches of the two batches alternate; nothing waits in between
        d = sets[rep & 1]
        idx.streaming_search_dev(d["bases"].data_ptr(), d["bases"].numel(), d["off"].data_ptr(), d["n"], d["out"].data_ptr(),
                                 d["ooff"].data_ptr(), d["ws"].data_ptr(), d["wsb"], d["stream"].cuda_stream, streaming)
    torch.cuda.synchronize()
    for d in sets:
        bases, off = d["host"]
        sample = 1500                              # the oracle on the first reads, the library's own single call on all of them
        want = _want(orc, bases, off[:sample + 1], streaming)
        got = d["out"].cpu().numpy()
        assert np.array_equal(got[:len(want)], want)
        alone = (idx.streaming_search if streaming else idx.search)(bases, off)[0]
        assert np.array_equal(got, alone)
