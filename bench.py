#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X k-mer streaming-search path (BASELINE.json).

A "step" is one pass of the hot path (2-bit re-encoding of the bases + SBWT::streaming_search of
every read) over this rank's batch of synthetic reads, inputs and outputs resident in HBM.

Workload (BASELINE.json configs[1]): coli3-like synthetic genomes (3 x 5 Mbp, 5 % divergence;
coli3.fna itself is not available offline), k=30, plain-matrix index with precalc 8 and streaming
support, 10 M synthetic 150 bp reads per GPU with 1 % substitutions.  Reads are sharded across
ranks (weak scaling: 10 M per GPU), the read-only index is built on rank 0 and replicated with one
RCCL broadcast at load time; there are no per-step collectives.

  python bench.py [--gpus N] [--steps K] [--warmup W]        (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and, at N=1,
`cpu_baseline` (the oracle = restated reference CPU path, timed on the host cores on a bounded
sample of the same reads; the same sample doubles as a bit-exact parity check of the GPU output).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch and the HIP library are imported in main(), AFTER the decision whether this process is a launcher
# (python bench.py --gpus N without a launcher starts the N ranks itself, before anything touches the GPU)
torch = dist = capi = hostlib = synth = sdist = None

K = 30                          # overridden by --config (3: k=31 pan-genome, 5: k=63 non-streaming)
PRECALC = 8
READ_LEN = 150
SUB_RATE = 0.01
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
# SURVEY 8d algorithmic bytes, per operation
B_STREAM = 89                   # streaming step: 1 base + 8 ssup word + (8 count + 64 block bits) + 8 out
B_TABLE = 16                    # one prefix-table entry per walk (+ the bases of its window)
B_LF = 2 * 72                   # interval update: two ranks, (8 count + 64 block bits) each
B_OUT = 8                       # one int64 result
# the path-order kernel's own operations: what they must read and write
B_RUN = 8 + 4 + 0.5 + 1         # k-mer answered along a path run: int64 out + col[t] + its share of the path quad + 1 base
B_TRANS = 16 + 1 + 8            # streaming step at a branch point: one 16-byte transition quad + 1 base + out


def effective_cores() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)


def log(msg: str) -> None:
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


_T0 = time.time()
_PHASE = "start-up"             # what this process was doing last: a failure names it (main())


def phase(rank: int, world: int, what: str) -> None:
    """Multi-rank runs say where they are (rank 0 always; every rank with SBWT_BENCH_VERBOSE): a stalled 8-GPU run then
    names the phase -- columns, image, broadcast, reads, warm-up, timed loop -- instead of just timing out."""
    global _PHASE
    _PHASE = what
    if world > 1 and (rank == 0 or os.environ.get("SBWT_BENCH_VERBOSE")):
        log("rank %d/%d t+%.1fs: %s" % (rank, world, time.time() - _T0, what))


def dist_timeout():
    """Bound on every collective of a multi-rank run (process-group set-up, the image broadcast, the barriers): a rank
    that never arrives ends the run with an error that names the phase instead of hanging the node.  SBWT_BENCH_DIST_TIMEOUT
    seconds (default 900: the 13.4 GB image of config 3 moves in well under a minute on any fabric)."""
    import datetime
    return datetime.timedelta(seconds=int(os.environ.get("SBWT_BENCH_DIST_TIMEOUT", "900")))


def device_memory_check(rank: int, what: str, need_bytes: int, dev) -> None:
    """Before a large allocation: does it fit next to what this rank already holds (rank 0: the image, and whatever the
    builders' caches still keep)?  A clear message beats an out-of-memory box (N = 8, config 3: 13.4 GB image + 15 GB reads
    + 96 GB results + 2 GB workspace per rank)."""
    free, total = torch.cuda.mem_get_info(dev)
    if need_bytes > free:
        raise SystemExit("bench.py: rank %d cannot allocate %s: %.1f GB needed, %.1f GB of %.1f GB free on device %s "
                         "(smaller --reads, or a lower --image-level)" % (rank, what, need_bytes / 1e9, free / 1e9, total / 1e9, dev))


def gpu_reads(genomes, n_reads: int, seed: int, dev: torch.device, in_genome_order: bool = False) -> torch.Tensor:
    """n_reads x READ_LEN substrings at uniform (genome, offset) with per-base substitutions,
    generated on the GPU (seeded) so that the inputs are resident in HBM.  in_genome_order (experiments only,
    tools/ab_step.py SORTED=1): the same kind of reads, but each chunk of 2^20 laid out by (genome, offset) and the chunks
    covering consecutive slices of the genomes -- what a batch sorted by locus looks like."""
    import torch
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    lens = torch.tensor([len(g) for g in genomes], device=dev)
    starts = torch.tensor(np.concatenate([[0], np.cumsum([len(g) for g in genomes])[:-1]]), device=dev)
    cat = torch.from_numpy(np.concatenate(genomes)).to(dev)
    code = torch.zeros(256, dtype=torch.uint8, device=dev)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    code[acgt.long()] = torch.arange(4, dtype=torch.uint8, device=dev)
    out = torch.empty(n_reads * READ_LEN, dtype=torch.uint8, device=dev)
    ar = torch.arange(READ_LEN, device=dev)
    chunk = 1 << 20
    for lo in range(0, n_reads, chunk):
        n = min(chunk, n_reads - lo)
        which = torch.randint(0, len(genomes), (n,), device=dev, generator=gen)
        u = torch.rand(n, device=dev, generator=gen, dtype=torch.float64)
        off = (u * (lens[which] - READ_LEN + 1)).long() + starts[which]
        if in_genome_order:
            total = int(lens.sum().item()) - READ_LEN
            a, b = total * lo // n_reads, total * (lo + n) // n_reads
            off = torch.sort((u * (b - a)).long() + a).values
        r = cat[(off[:, None] + ar[None, :]).reshape(-1)]
        hit = torch.rand(n * READ_LEN, device=dev, generator=gen) < SUB_RATE
        shift = torch.randint(1, 4, (n * READ_LEN,), device=dev, generator=gen, dtype=torch.uint8)
        sub = acgt[((code[r.long()] + shift) & 3).long()]
        out[lo * READ_LEN:(lo + n) * READ_LEN] = torch.where(hit, sub, r)
    return out


SEARCH_SOURCES = ("sbwt_search_fused.hip", "sbwt_search_fused_loop.inc", "sbwt_search.hip", "sbwt_derived.hip", "sbwt_sort.hip", "sbwt_kernels_common.h",
                  "sbwt_device.h", "sbwtgpu_capi.cpp")


def kernel_source_sha16() -> str:
    """Identifies the code a search profile was taken on: sha256 of the sources that decide the search kernels' traffic --
    the kernels, the derived structures (path order, transition table, sparse table), the read sorting, and the C ABI with
    its defaults (which kernel, which image) -- plus the tuning overrides in the environment; first 16 hex digits.  (The
    builder, the rank / API kernels and the formatter do not take part in a search step.)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sbwt_amd", "csrc")
    for f in SEARCH_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    for key in sorted(os.environ):
        if key.startswith("SBWTGPU_") and key != "SBWTGPU_LIB":
            h.update(("%s=%s" % (key, os.environ[key])).encode())
    return h.hexdigest()[:16]


def dominant_kernel(variant: int, image_level: int) -> str:
    """The kernel that does a search step's work for (search variant, image level)."""
    if variant == 0:
        return "k_search"
    if image_level != 0:
        return "k_search_cert"
    return {5: "k_search_fused", 4: "k_search_cert<PATH,SEG>", 2: "k_search_cert<PATH,SEG>", 3: "k_search_cert<PATH,SEG>"}.get(
        variant, "k_search_cert")


def traffic_key(config: int, image_level: int, variant: int) -> str:
    return "config%d_level%d_variant%d" % (config, image_level, variant)


def load_traffic(config: int, n_reads: int, image_level: int, variant: int, kernel: str):
    """profiles/traffic.json entry for this workload, if it was measured on the current kernel sources, the same image
    level, the same search route and the same dominant kernel -- a counter value never travels to another kernel's line."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(path))
    except Exception:
        return None
    ent = tj.get(traffic_key(config, image_level, variant))
    if not isinstance(ent, dict):
        return None
    if (ent.get("kernel_source_sha16") != kernel_source_sha16() or ent.get("reads_per_gpu") != n_reads or
            ent.get("kernel") != kernel or ent.get("image_level") != image_level or ent.get("search_variant") != variant):
        return None
    return ent


# What the memory system delivers per access shape on this chip (tools/micro/ceilings.hip, gather_modes.hip, mixed_rw.hip;
# measured, profiles/r02_ceilings.txt, r03_gather_modes.txt, r04_mixed_rw.txt): a random read leaves an XCD as ONE 128-byte
# fabric request whatever the load width -- 56 G requests/s --; coalesced nontemporal 16-byte stores run at 6.46 TB/s = 101 G
# 64-byte requests/s; a kernel that does both takes the SUM of the two times (mixes of gathers and stores run at 90-96 % of
# that additive model), so reads and writes do not hide behind each other.
READ_REQ_CEILING = 55.7e9
WRITE_REQ_CEILING = 6.46e12 / 64


def request_model(read_requests: float, write_requests: float, kernel_ms: float) -> dict:
    """The kernel's time against the request rates the memory system delivers: reads and writes share the fabric, so the
    floor for a launch is read_requests / 56 G/s + write_requests / 80 G/s.  `frac_of_deliverable` near 1 says the kernel
    runs at what the memory system gives this access pattern -- the only way on is fewer lines per read."""
    floor_ms = (read_requests / READ_REQ_CEILING + write_requests / WRITE_REQ_CEILING) * 1e3
    return {"read_requests_128B": read_requests, "write_requests_64B": write_requests,
            "read_ceiling_Greq_s": READ_REQ_CEILING / 1e9, "write_ceiling_Greq_s": WRITE_REQ_CEILING / 1e9,
            "floor_ms": floor_ms, "frac_of_deliverable": floor_ms / kernel_ms if kernel_ms > 0 else None,
            "ceilings_source": "profiles/r03_gather_modes.txt, profiles/r04_mixed_rw.txt (tools/micro/): measured ceilings; the "
                               "micro benchmark's own gather + store mixes reach 90-96 % of this additive floor"}


RANK_B_SURVEY = 72              # one rank = 8 B count + 64 B block bits
RANK_B_LAYOUT = 8 + 1 + 16 + 8  # this layout: pos + symbol + one 16-byte quad {bits, count} + result


def rank_bench(args, rank: int, world: int, local_rank: int, dev) -> int:
    """bench.py --kernel rank: batched SubsetMatrixRank::rank (SubsetMatrixRank.hh:31-37) on uniform random
    (pos, symbol) pairs, inputs and outputs resident in HBM.  Two images: four random bit vectors of --rank-columns
    columns (HBM resident, the headline) and the config-2 index (cache resident).  Every rank builds its own images
    and ranks its own queries (replicas; no collective)."""
    stream = torch.cuda.current_stream().cuda_stream
    Q = args.rank_queries
    cores = effective_cores()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7 + rank)

    def measure(index, n_cols, cols, label):
        d_pos = torch.randint(0, n_cols + 1, (Q,), dtype=torch.int64, device=dev, generator=gen)
        d_sym = acgt[torch.randint(0, 4, (Q,), device=dev, generator=gen)]
        d_out = torch.empty(Q, dtype=torch.int64, device=dev)

        def step():
            capi._check(capi.lib().sbwtgpu_rank_dev(index.handle, d_pos.data_ptr(), d_sym.data_ptr(), Q, d_out.data_ptr(),
                                                    stream))
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b in evs:
            a.record()
            step()
            b.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            elapsed = sdist.max_over_ranks(elapsed, dev)
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        res = {"label": label, "columns": n_cols, "image_bytes": index.blob_bytes, "queries_per_gpu": Q,
               "ms_per_step": elapsed / args.steps * 1e3, "kernel_ms": kernel_ms,
               "ranks_per_s": Q * world * args.steps / elapsed, "kernel_only_ranks_per_s": Q / (kernel_ms * 1e-3)}
        cpu = None
        if rank == 0 and not args.no_cpu_baseline:
            from oracle import OracleIndex
            orc = OracleIndex.from_bits(cols[0], cols[1], cols[2], cols[3], None, n_cols, 1, 0, 0)
            sample = min(Q, 4_000_000 * cores)
            h_pos = d_pos[:sample].cpu().numpy()
            h_sym = d_sym[:sample].cpu().numpy()
            best = None
            for _ in range(2):
                want, secs = orc.batch_rank(h_pos, h_sym, cores)
                best = secs if best is None else min(best, secs)
            parity = bool(np.array_equal(want, d_out[:sample].cpu().numpy()))
            cpu = {"value": sample / best, "unit": "ranks/s", "cores": cores, "kind": "port",
                   "sample": "first %d of the same (pos, symbol) pairs, oracle rank (rank_support_v5-shaped directory), "
                             "%d threads over contiguous ranges, slowest thread's loop time, best of 2" % (sample, cores),
                   "gpu_output_bit_identical_on_sample": parity}
            if not parity:
                raise SystemExit("PARITY FAILURE: GPU ranks differ from the oracle (%s)" % label)
        return res, cpu

    # (1) HBM resident: four unrelated random rows (SubsetMatrixRank as a stand-alone structure)
    n_cols = args.rank_columns
    rng = np.random.Generator(np.random.PCG64(70 + rank))
    nw = (n_cols + 63) // 64
    cols = [rng.integers(0, 2**64, size=nw, dtype=np.uint64) for _ in range(4)]
    t0 = time.time()
    big = capi.Index.create(cols[0], cols[1], cols[2], cols[3], None, n_cols, 1, 0, 0, None, device=local_rank)
    log(f"rank image: {n_cols} columns, {big.blob_bytes / 1e6:.0f} MB ({time.time() - t0:.1f} s)")
    hbm, cpu_hbm = measure(big, n_cols, cols, "four random bit vectors, HBM resident")
    big.close()
    del cols
    # (2) cache resident: the config-2 index
    genomes = synth.coli3_like(args.genome_len)
    bits = hostlib.build_bits([g.tobytes() for g in genomes], 30, False, True, n_threads=cores)
    small = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, 30,
                              bits.n_kmers, PRECALC, None, device=local_rank)
    cache, cpu_cache = measure(small, bits.n_nodes, bits.cols, "config-2 index (coli3-like, k=30), cache resident")
    if rank == 0:
        ach = RANK_B_SURVEY * Q / (hbm["kernel_ms"] * 1e-3) / 1e9
        result = {
            "metric": "ranks/sec (whole node), batched SubsetMatrixRank::rank, plain-matrix",
            "value": hbm["ranks_per_s"], "unit": "ranks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": hbm["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "rank kernel: %d uniform random (pos, symbol) pairs per GPU on four random bit vectors "
                                   "of %d columns (image %d bytes, HBM resident)" % (Q, n_cols, hbm["image_bytes"]),
                       "parallelism": "replicas x%d (independent queries, no collective)" % world},
            "roofline": {"bound": "hbm", "kernel": "k_rank4", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": None, "kernel_ms": hbm["kernel_ms"],
                         "algorithmic_bytes_per_launch": RANK_B_SURVEY * Q,
                         "pricing": "SURVEY 8d: 72 B per rank (8 B count + 64 B block bits); this layout moves %d B per "
                                    "rank (pos 8 + symbol 1 + one 16-byte quad + result 8): layout_priced_GBps"
                                    % RANK_B_LAYOUT,
                         "layout_priced_GBps": RANK_B_LAYOUT * Q / (hbm["kernel_ms"] * 1e-3) / 1e9,
                         "kernel_only_ranks_per_s": hbm["kernel_only_ranks_per_s"]},
            "cache_resident": cache,
        }
        if cpu_hbm:
            result["cpu_baseline"] = cpu_hbm
            result["cache_resident"]["cpu_baseline"] = cpu_cache
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def launch_ranks(n: int) -> int:
    """One process per GPU through torch.distributed.run (the command line the driver itself uses), as a CHILD
    process; its exit code is ours."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching %d ranks: %s" % (n, " ".join(cmd)))
    return subprocess.call(cmd)


def main() -> int:
    """Every way out prints either the JSON line (rank 0) or ONE line that names the phase that failed, and a non-zero
    exit code: a multi-GPU run that dies must say where."""
    try:
        return run()
    except SystemExit as ex:
        if ex.code not in (0, None):
            log("FAILED (rank %s) in phase '%s': %s" % (os.environ.get("RANK", "0"), _PHASE, ex.code))
        raise
    except BaseException as ex:                # noqa: BLE001 -- anything: the line is the point
        log("FAILED (rank %s) in phase '%s': %s: %s" % (os.environ.get("RANK", "0"), _PHASE, type(ex).__name__,
                                                       str(ex).replace("\n", " | ")[:600]))
        return 1


def run() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("SBWT_BENCH_READS", 10_000_000)),
                    help="reads per GPU")
    ap.add_argument("--genome-len", type=int, default=int(os.environ.get("SBWT_BENCH_GENOME", 5_000_000)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the end-to-end leg (host buffers over PCIe, and the `sbwt search` CLI on a FASTQ file)")
    ap.add_argument("--e2e-reads", type=int, default=2_000_000, help="reads of the end-to-end leg")
    ap.add_argument("--no-int32-leg", action="store_true", help="skip the extra leg with int32 results on the device")
    ap.add_argument("--no-two-in-flight", action="store_true", help="skip the extra leg that issues the same steps on two streams")
    ap.add_argument("--no-cli-full", action="store_true", help="end-to-end leg: skip the `sbwt search` run on the whole batch")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 5, 6],
                    help="BASELINE.json config: 2 = coli3-like k=30 (headline, default); 3 = pan-genome-like "
                         "k=31 (65 genomes, 100 M reads unless --reads); 5 = k=63 without streaming support")
    ap.add_argument("--derived", type=int, default=64, help="config 3: number of derived genomes")
    ap.add_argument("--kernel", choices=["search", "rank"], default="search",
                    help="search (default): the headline streaming-search pass; rank: the batched "
                         "SubsetMatrixRank::rank kernel (SubsetMatrixRank.hh:31-37) on random (pos, symbol) pairs")
    ap.add_argument("--rank-queries", type=int, default=1 << 30, help="--kernel rank: (pos, symbol) pairs per launch and GPU")
    ap.add_argument("--rank-columns", type=int, default=(1 << 31) - 128,
                    help="--kernel rank: columns of the four random bit vectors of the HBM-resident case (the image is "
                         "1 byte per column: the default 2 GiB defeats the 256 MB Infinity Cache); the config-2 index "
                         "(12.8 MB of blocks, cache resident) is measured beside it")
    ap.add_argument("--hbm-k", type=int, default=0, help="config 6: k (default 31)")
    ap.add_argument("--hbm-genome-len", type=int, default=1_000_000_000,
                    help="--config 6: length of the one random sequence (10^9 columns = a 139 GB image)")
    ap.add_argument("--image-level", type=int, default=0, choices=[0, 1, 2],
                    help="device image: 0 = all derived structures (default), 1 = no path order, 2 = blocks + dense table")
    ap.add_argument("--replicate", choices=["image", "rebuild"], default="image",
                    help="N > 1: broadcast the finished device image (default), or broadcast the five bit vectors "
                         "(0.7 bytes per column) and let every rank derive its own image")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --reads reads per GPU; strong: ONE set of --reads reads, cut into contiguous shards "
                         "balanced by bases (sbwt_amd/dist.py::contiguous_shard), rank r searches shard r -- the concatenation "
                         "of the ranks' outputs in rank order is the output of the whole set (checked with --check-ranks)")
    ap.add_argument("--check-ranks", action="store_true",
                    help="every rank's first 2000 reads are compared with the oracle on rank 0 (rank_parity in the JSON "
                         "line); used by the N > 1 tests")
    args = ap.parse_args()

    # ---- launcher: `python bench.py --gpus N` without RANK in the environment starts the N ranks itself.
    # Decided here, before torch / HIP are even imported: a process that has touched the GPU is never re-executed.
    if args.gpus > 1 and "RANK" not in os.environ:
        # (counting devices does not initialise the GPU on this image; SBWT_BENCH_FORCE_DEVICE = several ranks on one GPU)
        if "SBWT_BENCH_FORCE_DEVICE" not in os.environ:
            import torch as _t
            have = _t.cuda.device_count()
            if args.gpus > have:
                log(f"error: --gpus {args.gpus} but only {have} HIP device(s) are visible on this node")
                return 2
        return launch_ranks(args.gpus)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        log(f"error: --gpus {args.gpus} but WORLD_SIZE={world}: start {args.gpus} ranks "
            f"(python bench.py --gpus {args.gpus} does it itself when RANK is unset)")
        return 2

    global K, torch, dist, capi, hostlib, synth, sdist
    import torch as _torch
    import torch.distributed as _dist
    from sbwt_amd import capi as _capi, hostlib as _hostlib, synth as _synth, dist as _sdist
    torch, dist, capi, hostlib, synth, sdist = _torch, _dist, _capi, _hostlib, _synth, _sdist

    streaming = True
    if args.config == 3:
        K = 31
        if "SBWT_BENCH_READS" not in os.environ and "--reads" not in sys.argv:
            args.reads = 100_000_000
    elif args.config == 5:
        K = 63
        streaming = False
    elif args.config == 6:
        K = 31                      # SURVEY 8d "G-hbm": one uniform-random sequence, an image far beyond the Infinity Cache
        if args.hbm_k:              # (--hbm-k 32: the two-level tables of 31 < k <= 63 at that size; the GPU builder's keys fit 64 bits up to 32)
            K = args.hbm_k

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the search path has no CPU fallback")
    if "SBWT_BENCH_FORCE_DEVICE" not in os.environ and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants device {local_rank}, but only {torch.cuda.device_count()} HIP "
                         f"device(s) are visible (--gpus {args.gpus})")
    # test hooks: run several ranks on one GPU (SBWT_BENCH_FORCE_DEVICE) over gloo (SBWT_BENCH_BACKEND) to
    # exercise the N>1 code path on a single-GPU box; the driver's multi-GPU run uses neither
    if "SBWT_BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["SBWT_BENCH_FORCE_DEVICE"])
    backend = os.environ.get("SBWT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctl = None                      # a gloo group beside RCCL's: agreement on failures, and the fall-back route of the replication
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        phase(rank, world, "process group set-up (backend %s)" % backend)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=dist_timeout())
            ctl = dist.new_group(backend="gloo", timeout=dist_timeout())
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=dist_timeout())

    if args.kernel == "rank":
        return rank_bench(args, rank, world, local_rank, dev)

    phase(rank, world, "process group up (backend %s), device %d" % (backend if world > 1 else "-", local_rank))
    # ---- index: built once on rank 0 (host sort-based builder), replicated by one broadcast ----
    if args.config == 6:
        genomes = [synth.random_genome(args.hbm_genome_len, 7)]
    else:
        genomes = synth.pan_like(args.derived, args.genome_len) if args.config == 3 else synth.coli3_like(args.genome_len)
    t0 = time.time()
    bits = None
    if rank == 0:
        # columns: on the GPU for k <= 64 (sbwtgpu_build_plain_matrix), host sort-based builder beyond
        columns_on = "gpu" if K <= 64 else "host"
        if K <= 64:
            try:
                bits = capi.build_bits_gpu([g.tobytes() for g in genomes], K, False, streaming, device=local_rank)
            except capi.SbwtGpuError as ex:
                if ex.code != capi.ERR_OOM:         # a kernel fault is an error, not a reason to fall back
                    raise
                log(f"device builder: {ex.msg}; building the columns on the host instead")
                columns_on = "host (device builder did not fit)"
                bits = hostlib.build_bits([g.tobytes() for g in genomes], K, False, streaming, n_threads=effective_cores())
        else:
            bits = hostlib.build_bits([g.tobytes() for g in genomes], K, False, streaming, n_threads=effective_cores())
        t_cols = time.time() - t0
        phase(rank, world, "columns built (%.1f s)" % t_cols)
    capi.set_tuning("image_level", args.image_level)
    build_times = None
    t_bcast = None
    replicate_fallback = None
    if world > 1 and args.replicate == "rebuild":
        # the five bit vectors travel (RCCL / gloo broadcast of one uint64 tensor), every rank derives its own image
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.time()
        meta = torch.zeros(4, dtype=torch.int64, device=dev)
        if rank == 0:
            meta[0], meta[1], meta[2] = bits.n_nodes, bits.n_kmers, 1 if bits.ssup is not None else 0
        dist.broadcast(meta, src=0)
        n_nodes_b, n_kmers_b, has_ssup_b = int(meta[0]), int(meta[1]), int(meta[2])
        nw = (n_nodes_b + 63) // 64
        rows = torch.empty((4 + has_ssup_b) * nw, dtype=torch.int64, device=dev)
        if rank == 0:
            parts = list(bits.cols) + ([bits.ssup] if has_ssup_b else [])
            rows.copy_(torch.from_numpy(np.concatenate([np.asarray(p)[:nw] for p in parts]).view(np.int64)))
        dist.broadcast(rows, src=0)
        torch.cuda.synchronize()
        t_bcast = time.time() - tb
        if rank != 0:
            h = rows.cpu().numpy().view(np.uint64)
            bits = capi.BuiltBits([h[c * nw:(c + 1) * nw] for c in range(4)], h[4 * nw:5 * nw] if has_ssup_b else None,
                                  n_nodes_b, n_kmers_b, K)
        del rows
    if rank == 0 or (world > 1 and args.replicate == "rebuild"):
        t1 = time.time()
        index = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, K,
                                  bits.n_kmers, PRECALC, None, device=local_rank)
        t_img = time.time() - t1
    if rank == 0:
        log(f"index: n_nodes={index.n_nodes} n_kmers={index.n_kmers} image={index.blob_bytes / 1e6:.1f} MB level={index.image_level} "
            f"device_precalc={index.device_precalc_k} (columns {t_cols:.2f} s + image {t_img:.2f} s)")
        build_times = {"columns_s": t_cols, "image_s": t_img, "columns_on": columns_on,
                       "image_level": index.image_level, "image_bytes_per_column": index.blob_bytes / index.n_nodes,
                       "paths": index.n_paths, "branching_columns": index.n_branch,
                       "search_variant": index.default_search_variant}
    phase(rank, world, "image ready on rank 0" if rank == 0 else "waiting for the image")
    if world > 1 and args.replicate == "image":
        hdr, blob = None, None
        if rank == 0:
            hdr = index.export_header()
            blob = index.blob_tensor()           # the image itself (an alias, no second copy on the root)
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.time()
        phase(rank, world, "image broadcast (RCCL)" if backend == "nccl" else "image broadcast (%s)" % backend)
        # Pre-flight over the CONTROL group (gloo beside RCCL): header and length travel there, every rank allocates its
        # receiving tensor, the test hook fires, and all ranks count the failures BEFORE anyone enters the data collective --
        # a rank that fails on its own can then never leave the others blocked inside RCCL.  With any failure all
        # ranks take the same fall-back in process: the five bit vectors over the control group, every rank derives its own
        # image (--replicate rebuild's work; a rank that has touched the GPU is never re-executed).
        hook = bool(os.environ.get("SBWT_BENCH_FAIL_BCAST")) and rank == world - 1     # test hook: one rank cannot take part
        if hook:
            log("rank %d: image broadcast failed (SBWT_BENCH_FAIL_BCAST)" % rank)
        hdr, blob, n_failed = sdist.blob_preflight(hdr, blob, dev, ctl, src=0, fail=hook)
        why = "image broadcast failed on %d rank(s)" % n_failed
        if n_failed == 0:
            # the data collective itself: ONE broadcast of the image (RCCL over xGMI, load time only).  A failure INSIDE it is
            # fatal -- the other ranks are in the same collective and cannot be told: main() prints the phase, exit non-zero
            dist.broadcast(blob, src=0)
            torch.cuda.synchronize()
            adopt_err = None
            if rank != 0:
                try:
                    index = capi.Index.adopt(hdr, blob.data_ptr(), blob.numel(), local_rank, keepalive=blob)
                except Exception as ex:                                        # noqa: BLE001
                    adopt_err = "%s: %s" % (type(ex).__name__, str(ex).replace("\n", " | ")[:300])
                    log("rank %d: adopting the image failed (%s)" % (rank, adopt_err))
            n_failed = sdist.count_failures(adopt_err is not None, ctl)    # (every rank is out of the collective: safe)
            why = "adopting the image failed on %d rank(s)" % n_failed
        if n_failed > 0:
            replicate_fallback = "%s; bit vectors over %s, every rank derived its image" % (why, "gloo" if ctl is not None else backend)
            phase(rank, world, "fall-back: " + replicate_fallback)
            blob = None
            bits = sdist.broadcast_bits_cpu(bits if rank == 0 else None, K, ctl, capi.BuiltBits)
            if rank != 0:
                index = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, K,
                                          bits.n_kmers, PRECALC, None, device=local_rank)
        t_bcast = time.time() - tb
        phase(rank, world, "image replicated: %.2f GB in %.2f s" % (index.blob_bytes / 1e9, t_bcast))

    # ---- this rank's reads, resident in HBM ----
    m = READ_LEN - K + 1
    phase(rank, world, "allocating reads and results")
    searched_here = args.reads if args.scaling != "strong" else -(-args.reads // world) + 1     # (strong scaling generates the whole set, searches a shard)
    device_memory_check(rank, "%d reads, the int64 results of %d of them and the workspace" % (args.reads, searched_here),
                        args.reads * READ_LEN + searched_here * m * 8 + capi.search_workspace_bytes(searched_here * READ_LEN)
                        + (1 << 28), dev)
    strong = args.scaling == "strong" and world >= 1
    d_all = None
    if args.scaling == "strong":
        # one read set (the same bytes on every rank: same seed), cut into contiguous shards balanced by bases
        d_all = gpu_reads(genomes, args.reads, 42, dev)
        lo, hi = sdist.contiguous_shard(np.arange(args.reads + 1, dtype=np.int64) * READ_LEN, rank, world)
        n_reads = hi - lo
        d_bases = d_all[lo * READ_LEN:hi * READ_LEN]
        shard_lo = lo
    else:
        n_reads = args.reads
        d_bases = gpu_reads(genomes, n_reads, 42 + rank, dev)
    total_bases = d_bases.numel()
    n_kmers = n_reads * m
    d_roff = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * READ_LEN
    d_ooff = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * m
    d_out = torch.empty(n_kmers, dtype=torch.int64, device=dev)
    ws_bytes = capi.search_workspace_bytes(total_bases)
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    # the kernel that runs: the index's default unless SBWTGPU_SEARCH_VARIANT overrides it.  Variant 5 (fused route) is
    # ONE call per step -- sbwtgpu_streaming_search_dev: k_check_uniform2 + k_search_fused (encodes the bases itself) + the
    # two kernels chained behind it for reads it hands on (none here) -- and the library records HIP events around
    # k_search_fused on the launch stream ("kernel_events"); the older variants are the two calls encode + search, with
    # this script's events around the search call.
    variant = (index.default_search_variant if os.environ.get("SBWTGPU_SEARCH_VARIANT") is None
               else int(os.environ["SBWTGPU_SEARCH_VARIANT"]))
    one_call = (variant == 5 and index.image_level == 0)

    def step(ev=None):
        if one_call:
            if ev is not None:
                ev[0].record()
            index.streaming_search_dev(d_bases.data_ptr(), total_bases, d_roff.data_ptr(), n_reads, d_out.data_ptr(),
                                       d_ooff.data_ptr(), d_ws.data_ptr(), ws_bytes, stream, streaming)
            if ev is not None:
                ev[1].record()
            return
        index.encode_bases_dev(d_bases.data_ptr(), total_bases, d_ws.data_ptr(), ws_bytes, stream)
        if ev is not None:
            ev[0].record()
        index.search_encoded_dev(total_bases, d_roff.data_ptr(), n_reads, d_out.data_ptr(), d_ooff.data_ptr(),
                                 d_ws.data_ptr(), ws_bytes, streaming, stream)
        if ev is not None:
            ev[1].record()

    phase(rank, world, "reads resident (%d reads), warm-up" % n_reads)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    phase(rank, world, "timed loop: %d steps" % args.steps)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if one_call:
        capi.set_tuning("kernel_events", 1)
    t_start = time.perf_counter()
    for s in range(args.steps):
        step(events[s])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        elapsed = sdist.max_over_ranks(elapsed, dev)
    phase(rank, world, "timed loop done: %.2f ms per step (max over ranks)" % (elapsed / args.steps * 1e3))

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))     # the whole call (two-call routes: the search call)
    if one_call:
        kt = capi.kernel_times()                    # ... the fused kernel alone, when the fused route took the batch
        capi.set_tuning("kernel_events", 0)
        if len(kt) >= args.steps:
            kernel_ms = float(np.mean(kt[-args.steps:]))
    n_stream, n_search, n_lf, n_tab, n_ext = index.workspace_stats(d_ws.data_ptr(), stream)
    status = index.workspace_status(d_ws.data_ptr(), stream)
    if status != 0:
        raise SystemExit(f"search kernel reported status {status}")
    # The same K steps with int32 results on the device (sbwtgpu_*_dev_i32; n_nodes < 2^31): every kernel writes 4 bytes per
    # k-mer, half the write requests of a launch.  Reported beside the line: `value` is the reference's int64 interface.
    int32_leg = None
    if one_call and world == 1 and not args.no_int32_leg and index.n_nodes < (1 << 31) and n_kmers * 8 <= (24 << 30):
        d32 = torch.empty(n_kmers, dtype=torch.int32, device=dev)

        def step32():
            index.streaming_search_dev_i32(d_bases.data_ptr(), total_bases, d_roff.data_ptr(), n_reads, d32.data_ptr(),
                                           d_ooff.data_ptr(), d_ws.data_ptr(), ws_bytes, stream, streaming)
        step32()
        torch.cuda.synchronize()
        capi.set_tuning("kernel_events", 1)
        t3 = time.perf_counter()
        for _ in range(args.steps):
            step32()
        torch.cuda.synchronize()
        e3 = time.perf_counter() - t3
        kt3 = capi.kernel_times()
        capi.set_tuning("kernel_events", 0)
        same = True
        CH = 1 << 28                                    # compared in pieces: no second 10 GB array
        for lo in range(0, n_kmers, CH):
            same = same and bool(torch.equal(d32[lo:lo + CH].to(torch.int64), d_out[lo:lo + CH]))
        int32_leg = {"ms_per_step": e3 / args.steps * 1e3, "value": n_kmers * args.steps / e3, "unit": "k-mers/s",
                     "kernel_ms": float(np.mean(kt3[-args.steps:])) if len(kt3) >= args.steps else None,
                     "results_identical": same,
                     "note": "sbwtgpu_streaming_search_dev_i32 / sbwtgpu_search_dev_i32: the same search writing int32 results (4 bytes per k-mer instead of 8)"}
        del d32
        if not same:
            raise SystemExit("int32 results differ from the int64 results")
        # (the workspace counters below are this leg's: the same work)
    # The same K steps once more with TWO batches in flight (two streams, workspaces and result buffers): a launch ends in
    # ~0.5 ms in which its waves leave one by one (tools/timeline_fused.py); with a second launch queued behind, the chip
    # stays full.  Reported beside the line, never as `value`: `value` is one launch at a time, like its kernel_ms.
    two_in_flight = None
    if one_call and world == 1 and not args.no_two_in_flight and n_kmers * 8 <= (24 << 30):
        side = torch.cuda.Stream(device=dev)
        outs, wss = [d_out, torch.empty_like(d_out)], [d_ws, torch.empty_like(d_ws)]
        sts = [torch.cuda.current_stream(), side]

        def steps2(n):
            for s in range(n):
                q = s & 1
                with torch.cuda.stream(sts[q]):
                    index.streaming_search_dev(d_bases.data_ptr(), total_bases, d_roff.data_ptr(), n_reads, outs[q].data_ptr(),
                                               d_ooff.data_ptr(), wss[q].data_ptr(), ws_bytes, sts[q].cuda_stream, streaming)
        steps2(4)                                   # (two calls per workspace: the kernel choice settles after two)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        steps2(args.steps)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t2
        same = bool(torch.equal(outs[0], outs[1]))
        two_in_flight = {"ms_per_step": e2 / args.steps * 1e3, "value": n_kmers * args.steps / e2, "unit": "k-mers/s",
                         "results_identical": same,
                         "note": "the same steps issued on two streams alternately (two workspaces, two result buffers): the next launch fills the chip while the waves of the last one leave"}
        del outs, wss
        if not same:
            raise SystemExit("two batches in flight: the two result buffers differ")
    # Algorithmic bytes of the work the kernel EXECUTED, priced per operation at SURVEY 8d's figures:
    # streaming steps, walks (prefix-table entry + its window of bases), interval updates, and the
    # result of every k-mer that did not come from a streaming step.  (The reference's own order of
    # searches would execute ~3.5x more interval updates for the same output; DESIGN.md.)
    n_streamed = n_stream + n_ext                # k-mers answered by streaming semantics (SBWT.hh:562-575)
    if n_ext:
        # path-order kernel: run k-mers and transition steps priced at what THIS algorithm must move;
        # pricing them at the reference's 89 B/step would credit bytes the kernel never touches
        alg_bytes = (B_RUN * n_ext + B_TRANS * n_stream + (B_TABLE + index.device_precalc_k) * n_search + B_LF * n_lf
                     + B_OUT * (n_kmers - n_streamed))
    else:
        alg_bytes = (B_STREAM * n_stream + (B_TABLE + index.device_precalc_k) * n_search + B_LF * n_lf
                     + B_OUT * (n_kmers - n_stream))
    # the same work priced entirely at SURVEY 8d's per-operation figures (the reference's data structure)
    survey_priced_bytes = (B_STREAM * n_streamed + (B_TABLE + index.device_precalc_k) * n_search + B_LF * n_lf
                           + B_OUT * (n_kmers - n_streamed))
    # SURVEY 8d's nominal formula (every full search priced at all k-p interval updates)
    n_full = n_kmers - n_streamed
    nominal_bytes = B_STREAM * n_streamed + (K + 16 + 8 + B_LF * (K - PRECALC)) * n_full

    if world > 1:
        total_kmers = int(sdist.sum_over_ranks([n_kmers], dev)[0])
        rank_kernel_ms = [float(x) for x in sdist.gather_floats(kernel_ms, dev)]
    else:
        total_kmers, rank_kernel_ms = n_kmers, [kernel_ms]
    value = total_kmers * args.steps / elapsed
    result = {
        "metric": "k-mers/sec (whole node), plain-matrix k=%d %s" % (K, "streaming search" if streaming else
                                                                       "search (no streaming support)"),
        "value": value,
        "unit": "k-mers/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "rank_kernel_ms": rank_kernel_ms,
        "config": {
            "workload": ("config %d: " % args.config) + (
                "pan-genome-like synthetic genomes (1 + %d x %d bp, 2%% divergence)" % (args.derived, args.genome_len)
                if args.config == 3 else
                "G-hbm (SURVEY 8d extra): one uniform-random sequence of %d bp (seed 7)" % args.hbm_genome_len
                if args.config == 6 else
                "coli3-like synthetic genomes (3 x %d bp, 5%% divergence)" % args.genome_len) +
                " k=%d plain-matrix precalc=8 %s; %d synthetic 150bp reads per GPU, 1%% substitutions; %s of every read"
                % (K, "streaming support" if streaming else "NO streaming support", n_reads,
                   "SBWT::streaming_search" if streaming else "SBWT::search of every k-mer"),
            "k": K, "precalc_k": PRECALC, "read_len": READ_LEN, "reads_per_gpu": n_reads,
            "kmers_per_gpu": n_kmers, "n_nodes": index.n_nodes, "index_image_bytes": index.blob_bytes,
            "parallelism": "reads sharded x%d, index replicated (one RCCL broadcast at load)" % world,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": dominant_kernel(variant, index.image_level),
            "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "traffic": None,
            "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes,
            "pricing": ("per operation of the path-order algorithm: run k-mer 13.5 B, transition 25 B, walk start 16 B + p, "
                        "interval update 144 B, other result 8 B (DESIGN.md section 4); the same work at SURVEY 8d's 89 B per "
                        "streaming step is survey_8d_priced_*") if n_ext else "SURVEY 8d per-operation figures",
            "survey_8d_priced_bytes_per_launch": survey_priced_bytes,
            "survey_8d_priced_GBps": survey_priced_bytes / (kernel_ms * 1e-3) / 1e9,
            "nominal_bytes_per_launch_survey_8d": nominal_bytes,
            "work_per_launch": {"path_run_kmers": n_ext, "stream_steps": n_stream, "walks": n_search,
                                "bridged_substitutions": index.workspace_bridges(d_ws.data_ptr(), stream),
                                "interval_updates": n_lf, "table_hits": n_tab, "kmers_not_streamed": n_full},
            "kernel_only_kmers_per_s": n_kmers / (kernel_ms * 1e-3),
        },
    }
    if int32_leg is not None:
        result["int32_results_on_device"] = int32_leg
    if two_in_flight is not None:
        result["two_batches_in_flight"] = two_in_flight
    if replicate_fallback is not None:
        result["index_replication_fallback"] = replicate_fallback
    if t_bcast is not None:
        result["index_broadcast_s"] = t_bcast
        result["index_replication"] = args.replicate
        if args.replicate == "image":
            result["index_broadcast_GBps"] = index.blob_bytes / max(t_bcast, 1e-9) / 1e9
    if rank == 0:
        result["index_build"] = build_times
    # HBM-side bytes per launch from the PMC counters: measured by tools/profile.sh in separate rocprofv3 passes (a
    # bench run cannot collect counters on itself) and attached ONLY while the profile belongs to the kernels that
    # just ran (same source hash, same workload); otherwise null, never a stale number
    tj = load_traffic(args.config, n_reads, index.image_level, variant, result["roofline"]["kernel"])
    result["roofline"]["traffic_over_algorithmic"] = None
    result["roofline"]["read_lines_per_read"] = None
    if tj is not None:
        result["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
        result["roofline"]["traffic_source"] = tj.get("source")
        result["roofline"]["traffic_over_algorithmic"] = tj.get("hbm_bytes_per_launch") / alg_bytes
        if tj.get("read_requests_128B"):
            result["roofline"]["read_lines_per_read"] = tj["read_requests_128B"] / n_reads      # 128-byte fabric requests
            result["roofline"]["write_requests_per_read"] = (tj.get("write_requests_64B") or 0) / n_reads
            result["roofline"]["request_model"] = request_model(tj["read_requests_128B"], tj.get("write_requests_64B") or 0, kernel_ms)
        # what the memory system actually moved per second during the kernel (every gather drags a 128-byte line):
        result["roofline"]["traffic_GBps"] = tj.get("hbm_bytes_per_launch") / (kernel_ms * 1e-3) / 1e9
        result["roofline"]["traffic_frac_of_peak"] = result["roofline"]["traffic_GBps"] / HBM_PEAK_GBPS
    result["roofline"]["kernel_source_sha16"] = kernel_source_sha16()

    # ---- strong scaling: the ranks' outputs, concatenated in rank order, are the output of the whole set ----
    if args.check_ranks and args.scaling == "strong":
        counts = sdist.gather_floats(float(n_kmers), dev) if world > 1 else [float(n_kmers)]
        if world > 1:
            mx = int(max(counts))
            pad = torch.full((mx,), -7, dtype=torch.int64, device=dev)
            pad[:n_kmers] = d_out
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad)
        else:
            parts = [d_out]
        if rank == 0:
            whole = torch.cat([parts[r][: int(counts[r])] for r in range(world)])
            ref = torch.empty(args.reads * m, dtype=torch.int64, device=dev)
            ws1 = torch.empty(capi.search_workspace_bytes(d_all.numel()), dtype=torch.uint8, device=dev)
            ro1 = torch.arange(args.reads + 1, dtype=torch.int64, device=dev) * READ_LEN
            oo1 = torch.arange(args.reads + 1, dtype=torch.int64, device=dev) * m
            index.streaming_search_dev(d_all.data_ptr(), d_all.numel(), ro1.data_ptr(), args.reads, ref.data_ptr(), oo1.data_ptr(),
                                       ws1.data_ptr(), ws1.numel(), stream, streaming)
            torch.cuda.synchronize()
            result["strong_concat_equals_single"] = bool(whole.numel() == ref.numel() and torch.equal(whole, ref))
            del ref, ws1, whole

    # ---- N > 1 tests: every rank's first reads against the oracle (rank 0 holds the bits) ----
    if args.check_ranks:
        sample = min(n_reads, 2000)
        mine_b = d_bases[: sample * READ_LEN].contiguous()
        mine_o = d_out[: sample * m].contiguous()
        if world > 1:
            gb = [torch.empty_like(mine_b) for _ in range(world)]
            go = [torch.empty_like(mine_o) for _ in range(world)]
            dist.all_gather(gb, mine_b)
            dist.all_gather(go, mine_o)
        else:
            gb, go = [mine_b], [mine_o]
        if rank == 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle import OracleIndex
            orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes,
                                        K, bits.n_kmers, PRECALC)
            roff = np.arange(sample + 1, dtype=np.int64) * READ_LEN
            ooff = np.arange(sample + 1, dtype=np.int64) * m
            par = []
            for r in range(world):
                want, _ = orc.batch_search(gb[r].cpu().numpy(), roff, ooff, effective_cores())
                par.append(bool(np.array_equal(want, go[r].cpu().numpy())))
            result["rank_parity"] = par
            if world > 1:    # the ranks searched different reads
                result["rank_reads_differ"] = bool(not torch.equal(gb[0], gb[1]))

    # ---- CPU baseline (restated reference CPU path = the oracle), rank 0 at N=1 only ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import OracleIndex  # test infrastructure: used here only as baseline + checker
        orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes,
                                    K, bits.n_kmers, PRECALC)   # without ssup it runs the per-k-mer search loop
        cores = effective_cores()
        probe = min(n_reads, 20_000)
        h_bases = d_bases[: probe * READ_LEN].cpu().numpy()
        roff = np.arange(probe + 1, dtype=np.int64) * READ_LEN
        ooff = np.arange(probe + 1, dtype=np.int64) * m
        _, secs1 = orc.batch_search(h_bases, roff, ooff, 1)
        per_read_1t = secs1 / probe
        # bounded sample: ~1 s of work per thread (about `cores` core-seconds), at most all reads
        sample = int(min(n_reads, max(probe, cores * 1.0 / per_read_1t)))
        h_bases = d_bases[: sample * READ_LEN].cpu().numpy()
        roff = np.arange(sample + 1, dtype=np.int64) * READ_LEN
        ooff = np.arange(sample + 1, dtype=np.int64) * m
        cpu_out = np.full(sample * m, -3, dtype=np.int64)           # pre-faulted
        best = None
        for _ in range(2):                                           # second pass runs warm
            t1 = time.perf_counter()
            _, busy = orc.batch_search(h_bases, roff, ooff, cores, out=cpu_out)
            wall_i = time.perf_counter() - t1
            if best is None or busy < best[0]:
                best = (busy, wall_i)
        wall = best[0]      # the slowest thread's summed query time (sbwt_search.cpp:54-56 style)
        gpu_out = d_out[: sample * m].cpu().numpy()
        parity = bool(np.array_equal(cpu_out, gpu_out))
        result["cpu_baseline"] = {
            "value": sample * m / wall,
            "unit": "k-mers/s",
            "cores": cores,
            "kind": "port",
            "sample": "first %d of the same reads (%d k-mers), oracle streaming_search, %d threads over "
                      "contiguous read ranges; time = slowest thread's summed per-read query time "
                      "(sbwt_search.cpp:54-56 style, best of 2 passes, output pre-faulted); wall clock of that "
                      "pass %.3f s" % (sample, sample * m, cores, best[1]),
            "value_1thread": m / per_read_1t,
            "value_per_core": sample * m / wall / cores,
            "cores_note": "cores = what this process may use (affinity mask capped by the cgroup CPU quota), not the host's "
                          "%d logical CPUs; the GPU / CPU ratio below is against THESE cores -- scale value_per_core by a "
                          "core count yourself, it is not measured beyond them" % (os.cpu_count() or 0),
            "host_logical_cpus": os.cpu_count(),
            "gpu_output_bit_identical_on_sample": parity,
        }
        result["gpu_vs_cpu"] = value / (sample * m / wall)
        if not parity:
            print(json.dumps(result))
            raise SystemExit("PARITY FAILURE: GPU output differs from the oracle on the baseline sample")

    # ---- end to end (SURVEY 8d "Timing"; sbwt_search.cpp:54-56,145,255-256): rank 0 at N=1 only ----
    if rank == 0 and world == 1 and not args.no_end_to_end and args.config in (2, 5):
        try:
            result["end_to_end"] = end_to_end_leg(args, index, genomes, d_bases, m, streaming, dev)
        except Exception as ex:      # the headline must not die with its side measurement
            result["end_to_end"] = {"error": repr(ex)}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def end_to_end_leg(args, index, genomes, d_bases, m, streaming, dev):
    """What a caller with HOST buffers gets, PCIe included: (1) sbwtgpu_streaming_search_batch / sbwtgpu_search_batch on
    pinned host reads -> int64 results in pinned host memory (median of 3; chunks pipelined over two streams inside the
    library), beside the rate PCIe allows (8 bytes of results per k-mer at the measured device-to-host copy rate); (2) the
    `sbwt search` command on a FASTQ file of the same reads, index file -> output text file, wall clock (best of 2)."""
    import ctypes as C
    import shutil
    import subprocess
    import tempfile
    E = min(args.e2e_reads, args.reads)
    n_k = E * m
    h_bases = torch.empty(E * READ_LEN, dtype=torch.uint8, pin_memory=True)
    h_bases.copy_(d_bases[: E * READ_LEN])
    h_out = torch.empty(n_k, dtype=torch.int64, pin_memory=True)
    roff = np.arange(E + 1, dtype=np.int64) * READ_LEN
    ooff = np.arange(E + 1, dtype=np.int64) * m
    fn = capi.lib().sbwtgpu_streaming_search_batch if streaming else capi.lib().sbwtgpu_search_batch
    times = []
    for _ in range(4):
        t0 = time.perf_counter()
        capi._check(fn(index.handle, h_bases.data_ptr(), roff.ctypes.data, E, h_out.data_ptr(), ooff.ctypes.data))
        times.append(time.perf_counter() - t0)
    host_rate = n_k / float(np.median(times[1:]))
    # the same with int32 results (sbwtgpu_*_batch_i32: the device narrows, 4 bytes of PCIe per k-mer)
    host_rate_i32 = None
    if index.n_nodes < (1 << 31):
        h_out32 = torch.empty(n_k, dtype=torch.int32, pin_memory=True)
        fn32 = capi.lib().sbwtgpu_streaming_search_batch_i32 if streaming else capi.lib().sbwtgpu_search_batch_i32
        t32 = []
        for _ in range(4):
            t0 = time.perf_counter()
            capi._check(fn32(index.handle, h_bases.data_ptr(), roff.ctypes.data, E, h_out32.data_ptr(), ooff.ctypes.data))
            t32.append(time.perf_counter() - t0)
        host_rate_i32 = n_k / float(np.median(t32[1:]))
        if not torch.equal(h_out32.to(torch.int64), h_out):
            raise RuntimeError("int32 results differ from the int64 results")
        del h_out32
    # the copy rate PCIe gives this box, device -> pinned host, 1 GiB
    probe = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    hprobe = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hprobe.copy_(probe, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    d2h = (1 << 30) / best
    del probe, hprobe
    out = {"batch_reads": E, "host_buffers_kmers_per_s": host_rate, "host_buffers_s": float(np.median(times[1:])),
           "d2h_GBps": d2h / 1e9, "pcie_bound_kmers_per_s": d2h / 8.0,
           "host_buffers_frac_of_pcie_bound": host_rate / (d2h / 8.0),
           "host_buffers": "pinned bases + pinned int64 results, offsets pageable; median of 3 after one warm-up call",
           "host_buffers_i32_kmers_per_s": host_rate_i32,
           "host_buffers_i32": "sbwtgpu_*_batch_i32: the same call with int32 results (narrowed on the device, 4 bytes of PCIe per "
                               "k-mer; indexes of fewer than 2^31 columns)"}
    # the CLI: index file, FASTQ file, output text file
    sbwt = os.path.join(ROOT, "sbwt_amd", "bin", "sbwt")
    d = tempfile.mkdtemp(prefix="sbwt_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        with open(d + "/g.fna", "wb") as f:
            for i, g in enumerate(genomes):
                f.write(b">g%d\n" % i + g.tobytes() + b"\n")
        cmd = [sbwt, "build", "-i", d + "/g.fna", "-o", d + "/i.sbwt", "-k", str(K), "-t", str(effective_cores())]
        if not streaming:
            cmd.append("--no-streaming-support")
        subprocess.run(cmd, check=True, capture_output=True)
        def write_fastq(path, d_src, n):
            """n reads as a 4-line FASTQ file, written in slices (10 M reads are 3 GB)"""
            with open(path, "wb") as f:
                for lo in range(0, n, 1_000_000):
                    hi = min(n, lo + 1_000_000)
                    rows = d_src[lo * READ_LEN:hi * READ_LEN].cpu().numpy().reshape(hi - lo, READ_LEN)
                    rec = np.empty((hi - lo, 7 + 2 * READ_LEN), dtype=np.uint8)                # "@r\n" seq "\n+\n" qual "\n"
                    rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
                    rec[:, 3:3 + READ_LEN] = rows
                    rec[:, 3 + READ_LEN:6 + READ_LEN] = np.frombuffer(b"\n+\n", dtype=np.uint8)
                    rec[:, 6 + READ_LEN:6 + 2 * READ_LEN] = ord("I")
                    rec[:, 6 + 2 * READ_LEN] = ord("\n")
                    rec.tofile(f)

        def run_cli(fastq, n, repeats):
            best, stages = None, None
            for _ in range(repeats):
                t0 = time.perf_counter()
                p = subprocess.run([sbwt, "search", "-i", d + "/i.sbwt", "-q", fastq, "-o", d + "/out.txt"], capture_output=True,
                                   env=dict(os.environ, SBWT_CLI_TIMING="1"))
                dt = time.perf_counter() - t0
                if p.returncode != 0:
                    raise RuntimeError("sbwt search failed: " + p.stderr.decode(errors="replace")[-300:])
                if best is None or dt < best:
                    best = dt
                    stages = [l[len("timing: "):] for l in p.stderr.decode().splitlines() if l.startswith("timing: ")]
            logs = [l.split("us/query")[1].strip() for l in p.stderr.decode().splitlines() if "us/query" in l]
            return best, logs, stages

        write_fastq(d + "/r.fastq", d_bases, E)
        best, logs, stages = run_cli(d + "/r.fastq", E, 2)
        out.update({"cli_kmers_per_s": n_k / best, "cli_wall_s": best, "cli_us_per_query_lines": logs,
                    "cli_fastq_bytes": os.path.getsize(d + "/r.fastq"), "cli_output_bytes": os.path.getsize(d + "/out.txt"),
                    "cli_stage_marks": stages,
                    "cli": "sbwt search -i index -q reads.fastq -o out.txt, process start to exit (index load, parse, search, format on the GPU, write), best of 2"})
        # ... what ONE call of the reference's scalar API costs through the C++ mirror (host/SBWT.hh): search of
        # one k-mer and rank of one position answer on the host since round 5, streaming_search of one read is a GPU
        # batch of one.  tools/scalar_api_bench.cpp, compiled here against the host headers.
        try:
            lib = os.path.join(ROOT, "sbwt_amd", "lib")
            exe = d + "/scalar_api_bench"
            subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "sbwt_amd", "csrc", "host"),
                            os.path.join(ROOT, "tools", "scalar_api_bench.cpp"), "-o", exe, "-L" + lib, "-lsbwtgpu", "-lz",
                            "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, timeout=300)
            one_read = d_bases[:READ_LEN].cpu().numpy().tobytes().decode()
            p = subprocess.run([exe, d + "/i.sbwt", one_read], capture_output=True, timeout=600)
            if p.returncode != 0:
                raise RuntimeError(p.stderr.decode(errors="replace")[-300:])
            out["scalar_api"] = json.loads(p.stdout.decode().strip().splitlines()[-1])
            out["scalar_api"]["what"] = ("one call through sbwt_amd/csrc/host/SBWT.hh: SBWT::search(k-mer) and SubsetMatrixRank::rank on "
                                         "the host (rank_support_v5 directory), the same as GPU batches of one, and "
                                         "streaming_search of one %d-base read (a GPU batch of one)" % READ_LEN)
        except Exception as ex:                                            # noqa: BLE001
            out["scalar_api"] = {"error": repr(ex)[:300]}
        # ... the same command on the whole batch (config 2: 10 M reads, 3 GB of FASTQ in, 8 GB of text out) when the scratch
        # directory has the room: fixed costs (process and HIP start-up, index load) no longer dominate
        big = args.reads
        need = big * (7 + 2 * READ_LEN) + big * m * 8
        if big > E and shutil.disk_usage(d).free > need + (4 << 30) and not args.no_cli_full:
            os.remove(d + "/out.txt")
            write_fastq(d + "/r.fastq", d_bases, big)
            best_b, logs_b, stages_b = run_cli(d + "/r.fastq", big, 1)
            out.update({"cli_full_reads": big, "cli_full_kmers_per_s": big * m / best_b, "cli_full_wall_s": best_b,
                        "cli_full_fastq_bytes": os.path.getsize(d + "/r.fastq"), "cli_full_output_bytes": os.path.getsize(d + "/out.txt"),
                        "cli_full_stage_marks": stages_b})
            os.remove(d + "/out.txt")
        # ... and what the reference's own CLI loop costs on the host: the oracle's restatement of run_file + print_vector
        # (sbwt_search.cpp:21-105: ONE thread, read by read), on a bounded sample of the same file
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle import OracleIndex
            f = hostlib.read_index_file(d + "/i.sbwt")
            orc = OracleIndex.from_bits(f.cols[0], f.cols[1], f.cols[2], f.cols[3], f.ssup, f.n_nodes, f.k, f.n_kmers, f.precalc_k)
            S = min(E, 200_000)
            write_fastq(d + "/s.fastq", d_bases, S)
            wall, qsecs, nr, nk = orc.search_file(d + "/s.fastq", d + "/s_out.txt")
            out.update({"cpu_cli_port": {"reads": nr, "kmers": nk, "wall_s": wall, "query_s": qsecs, "kmers_per_s": nk / wall,
                                         "threads": 1,
                                         "what": "the oracle's restatement of the reference CLI loop (sbwt_search.cpp:21-105: parse, "
                                                 "streaming_search / search per read, print_vector, write) on the first %d reads of "
                                                 "the same FASTQ file, one thread like the reference; index load not included" % S}})
            out["cli_vs_cpu_cli_port"] = out["cli_kmers_per_s"] / (nk / wall)
        except Exception as ex:
            out["cpu_cli_port"] = {"error": repr(ex)}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


if __name__ == "__main__":
    sys.exit(main())
