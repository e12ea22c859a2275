"""GPU: the N > 1 paths.  (1) `python bench.py --gpus 2` with no launcher must start two ranks itself and report
n_gpus == 2 with bit-exact parity (on a 1-GPU box the two ranks share the device over gloo; with >= 2 devices they run
one per GPU over RCCL).  (2) a rank count that does not match --gpus is an error, not a warning.  (3) the
single-process RCCL replication sbwtgpu_index_bcast with two DISTINCT devices and with duplicates (needs >= 2 GPUs)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import OracleIndex
from sbwt_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_env(n_dev):
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.pop("LOCAL_RANK", None)
    if n_dev < 2:                       # one GPU: both ranks on device 0, index image over gloo
        env["SBWT_BENCH_FORCE_DEVICE"] = "0"
        env["SBWT_BENCH_BACKEND"] = "gloo"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


@pytest.mark.parametrize("replicate", ["image", "rebuild"])
def test_bench_gpus2_launches_two_ranks(gpu, replicate):
    n_dev = capi.device_count()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "200000", "--genome-len", "300000",
           "--steps", "2", "--warmup", "1", "--check-ranks", "--replicate", replicate]
    p = subprocess.run(cmd, env=_bench_env(n_dev), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2
    assert res["config"]["reads_per_gpu"] == 200000
    assert res["rank_parity"] == [True, True], res.get("rank_parity")
    assert res["rank_reads_differ"] is True
    assert res["value"] > 0 and res["index_broadcast_s"] >= 0 and res["index_replication"] == replicate


def _run_bench(args, n_dev):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, env=_bench_env(n_dev), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_gpus2_config4_index_type(gpu):
    # BASELINE config 4's KIND of index at N = 2: k = 31 pan-genome (a core + derived genomes: core-following paths,
    # branching columns, negative transition entries), image broadcast + adopt, every rank against the oracle
    res = _run_bench(["--gpus", "2", "--config", "3", "--derived", "4", "--genome-len", "300000", "--reads", "200000",
                      "--steps", "2", "--warmup", "1", "--check-ranks"], capi.device_count())
    assert res["n_gpus"] == 2 and res["config"]["k"] == 31
    assert res["rank_parity"] == [True, True], res.get("rank_parity")
    assert res["rank_reads_differ"] is True
    assert len(res["rank_kernel_ms"]) == 2 and min(res["rank_kernel_ms"]) > 0
    assert res["index_broadcast_GBps"] > 0
    assert res["roofline"]["work_per_launch"]["stream_steps"] > 0          # reads did leave their paths


def test_bench_strong_scaling_preserves_order(gpu):
    # north_star: "reads sharded ... preserving order": ONE read set cut into contiguous shards, rank r searches shard r;
    # the ranks' outputs concatenated in rank order are bit for bit the output of the whole set on one GPU
    res = _run_bench(["--gpus", "2", "--scaling", "strong", "--reads", "300001", "--genome-len", "300000", "--steps", "2",
                      "--warmup", "1", "--check-ranks"], capi.device_count())
    assert res["n_gpus"] == 2 and res["scaling"] == "strong"
    assert res["strong_concat_equals_single"] is True
    assert res["rank_parity"] == [True, True]
    assert res["config"]["kmers_per_gpu"] in (150000 * 121, 150001 * 121)


def test_bench_rejects_world_mismatch(gpu):
    env = _bench_env(1)
    env["RANK"], env["WORLD_SIZE"], env["LOCAL_RANK"] = "0", "1", "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "1000"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "WORLD_SIZE" in p.stderr


def test_index_bcast_distinct_devices(gpu):
    import ctypes as C
    n_dev = capi.device_count()
    genomes = [synth.random_genome(60_000, 3)]
    orc = OracleIndex.build([g.tobytes() for g in genomes], 30, True, False, 6)
    cols = orc.columns()
    root = capi.Index.create(cols[0], cols[1], cols[2], cols[3], orc.ssup_words(), orc.n_nodes, 30, orc.n_kmers, 6)
    bases, off = synth.sample_reads(genomes, 500, 120, 0.02, 8)
    want, _ = root.streaming_search(bases, off)
    # duplicates of the root's device share the root's handle (no RCCL involved)
    devs = (C.c_int * 3)(0, 0, 0)
    outs = (C.c_void_p * 3)()
    capi._check(capi.lib().sbwtgpu_index_bcast(root.handle, 3, devs, outs))
    assert outs[0] == outs[1] == outs[2] == root.handle.value
    if n_dev < 2:
        pytest.skip("sbwtgpu_index_bcast over RCCL needs two GPUs (this box has one)")
    # two distinct devices + a duplicate: RCCL must load and run (a failure here fails the test)
    devs = (C.c_int * 3)(0, 1, 1)
    outs = (C.c_void_p * 3)()
    capi._check(capi.lib().sbwtgpu_index_bcast(root.handle, 3, devs, outs))
    assert outs[0] == root.handle.value and outs[1] == outs[2] and outs[1] != outs[0]
    rep = capi.Index(outs[1])
    assert rep.device == 1 and rep.n_nodes == root.n_nodes
    got, _ = rep.streaming_search(bases, off)
    assert np.array_equal(got, want)


def test_image_broadcast_failure_falls_back_to_rebuild(gpu):
    """when the image broadcast fails on any rank (here: the last rank raises instead of joining it,
    SBWT_BENCH_FAIL_BCAST), every rank learns it over the control group and they all take the same fall-back IN PROCESS --
    the five bit vectors over gloo, every rank derives its own image -- and the line still comes, with parity, and says so."""
    n_dev = capi.device_count()
    env = _bench_env(n_dev)
    env["SBWT_BENCH_FAIL_BCAST"] = "1"
    env["SBWT_BENCH_DIST_TIMEOUT"] = "120"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "100000", "--genome-len", "200000",
           "--steps", "2", "--warmup", "1", "--check-ranks"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    # (round 6: the failure is counted in a pre-flight over the control group, before anyone enters the data collective --
    # with one device over gloo and with two over RCCL alike)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert "image broadcast failed on 1 rank(s)" in res["index_replication_fallback"]
    assert res["rank_parity"] == [True, True]


def test_a_failing_run_names_its_phase(gpu):
    """Every way out of bench.py prints the JSON line or ONE line naming the phase, and exits non-zero: a batch that cannot
    fit says what it needed before anything is allocated."""
    env = _bench_env(2)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "2000000000", "--genome-len", "200000", "--steps", "1",
           "--warmup", "0", "--no-cpu-baseline", "--no-end-to-end"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "FAILED" in p.stderr and "allocating reads and results" in p.stderr and "cannot allocate" in p.stderr, p.stderr[-1500:]
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
