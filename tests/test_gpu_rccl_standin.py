"""GPU, one device: the single-process RCCL call sequence of sbwtgpu_index_bcast.

The code between `dlopen` and the last `ncclCommDestroy` of sbwt_amd/csrc/sbwtgpu_capi.cpp needs two distinct devices
to run against RCCL itself.  Here it runs on ONE: SBWTGPU_BCAST_NO_DEDUP=1 makes every entry of devs[] a rank of its own
and SBWTGPU_RCCL_LIB points at tests/cpp/rccl_standin.cpp (ncclBroadcast = one hipMemcpyAsync per non-root rank, issued at
ncclGroupEnd on that rank's stream).  Exercised: the per-rank streams and allocations, ncclCommInitAll, the grouped
broadcast, the synchronisation, the clean-up on success and on a failure of each of the three calls, the handles the
caller gets, `SBWT::use_devices` and the sharded search through them.  NOT exercised: RCCL's transport over xGMI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import OracleIndex, print_vector
from sbwt_amd import capi, synth
from test_gpu_cli import run, write_fasta, write_fastq

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def standin(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("rccl") / "librccl_standin.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "tests", "cpp", "rccl_standin.cpp")])
    old = {k: os.environ.get(k) for k in ("SBWTGPU_RCCL_LIB", "SBWTGPU_BCAST_NO_DEDUP", "RCCL_STANDIN_FAIL")}
    os.environ["SBWTGPU_RCCL_LIB"] = so
    os.environ["SBWTGPU_BCAST_NO_DEDUP"] = "1"
    yield so
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _small_index(k=30, seed=3):
    genomes = [synth.random_genome(60_000, seed)]
    orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 6)
    cols = orc.columns()
    root = capi.Index.create(cols[0], cols[1], cols[2], cols[3], orc.ssup_words(), orc.n_nodes, k, orc.n_kmers, 6)
    return genomes, orc, root


def _blob_bytes(index):
    return index.blob_tensor().cpu().numpy().copy()


def test_replicas_through_the_rccl_call_sequence_equal_the_root(gpu, standin):
    genomes, orc, root = _small_index()
    bases, off = synth.sample_reads(genomes, 500, 120, 0.02, 8)
    want, _ = root.streaming_search(bases, off)
    live = C.CDLL(standin).rccl_standin_live_comms
    devs = (C.c_int * 3)(0, 0, 0)
    outs = (C.c_void_p * 3)()
    capi._check(capi.lib().sbwtgpu_index_bcast(root.handle, 3, devs, outs))
    # three ranks: the root's entry is the root, the other two are images of their own
    assert outs[0] == root.handle.value and outs[1] != outs[0] and outs[2] != outs[0] and outs[1] != outs[2]
    assert live() == 0                                   # every communicator was destroyed
    ref = _blob_bytes(root)
    for h in (outs[1], outs[2]):
        rep = capi.Index(h)
        assert np.array_equal(_blob_bytes(rep), ref)     # bits identical to the root's image
        assert rep.device == 0 and rep.n_nodes == root.n_nodes
        got, _ = rep.streaming_search(bases, off)
        assert np.array_equal(got, want)
        rep.close()
    # and against the oracle, through a replica's own search
    sample = [orc.streaming_search(bases[off[r]:off[r + 1]].tobytes()) for r in range(0, 500, 50)]
    m = 120 - 30 + 1
    for q, r in enumerate(range(0, 500, 50)):
        assert np.array_equal(want[r * m:(r + 1) * m], np.asarray(sample[q], dtype=np.int64))


@pytest.mark.parametrize("where,text", [("init", "ncclCommInitAll failed"), ("bcast", "ncclBroadcast failed"),
                                        ("groupend", "ncclGroupEnd failed")])
def test_a_failing_rccl_call_frees_everything_and_is_named(gpu, standin, where, text):
    import torch
    genomes, orc, root = _small_index(seed=5)
    live = C.CDLL(standin).rccl_standin_live_comms
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    os.environ["RCCL_STANDIN_FAIL"] = where
    try:
        devs = (C.c_int * 3)(0, 0, 0)
        outs = (C.c_void_p * 3)()
        rc = capi.lib().sbwtgpu_index_bcast(root.handle, 3, devs, outs)
    finally:
        os.environ.pop("RCCL_STANDIN_FAIL", None)
    assert rc != 0
    assert text in capi.lib().sbwtgpu_last_error().decode(errors="replace")
    assert live() == 0                                   # communicators destroyed on the way out (none were made for "init")
    assert outs[1] is None and outs[2] is None           # no handle is handed out
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free1 >= free0 - (4 << 20), (free0, free1)    # the replicas' device memory came back
    # the root is untouched and the next replication works
    outs = (C.c_void_p * 2)()
    capi._check(capi.lib().sbwtgpu_index_bcast(root.handle, 2, (C.c_int * 2)(0, 0), outs))
    rep = capi.Index(outs[1])
    bases, off = synth.sample_reads(genomes, 50, 100, 0.02, 8)
    assert np.array_equal(rep.streaming_search(bases, off)[0], root.streaming_search(bases, off)[0])
    rep.close()


def test_cli_shards_a_batch_in_order_through_use_devices(gpu, standin, tmp_path):
    """`sbwt search --gpu-list 0,0` = SBWT::use_devices({0, 0}) -> sbwtgpu_index_bcast -> the stand-in: two images on the one
    device, every batch cut into two contiguous shards, the output byte-identical to the one-device run and the oracle."""
    d = str(tmp_path)
    genomes = [synth.random_genome(30_000, 4)]
    write_fasta(d + "/g.fna", [genomes[0].tobytes()])
    run("build", "-i", d + "/g.fna", "-o", d + "/st.sbwt", "-k", "31", "-p", "5", "-t", "4")
    bases, off = synth.sample_reads(genomes, 900, 120, 0.02, 4)
    reads = [bases[off[r]:off[r + 1]].tobytes() for r in range(900)] + [b"ACGT", b"A" * 31]
    write_fastq(d + "/r.fastq", reads)
    orc = OracleIndex.build([genomes[0].tobytes()], 31, True, False, 5)
    want = b"".join(print_vector(orc.streaming_search(r)) for r in reads)
    p = run("search", "-o", d + "/one.out", "-i", d + "/st.sbwt", "-q", d + "/r.fastq")
    assert open(d + "/one.out", "rb").read() == want
    p = run("search", "-o", d + "/two.out", "-i", d + "/st.sbwt", "-q", d + "/r.fastq", "--gpu-list", "0,0", "--batch-bases", "20000")
    assert b"replicated onto 2 GPU contexts" in p.stderr
    assert open(d + "/two.out", "rb").read() == want
    # a failing broadcast ends the command with the named error, not with a partial output
    os.environ["RCCL_STANDIN_FAIL"] = "groupend"
    try:
        p = run("search", "-o", d + "/bad.out", "-i", d + "/st.sbwt", "-q", d + "/r.fastq", "--gpu-list", "0,0", check=False)
    finally:
        os.environ.pop("RCCL_STANDIN_FAIL", None)
    assert p.returncode == 1 and b"ncclGroupEnd failed" in p.stderr
