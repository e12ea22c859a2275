"""An index of more than 2^31 columns (one random sequence, k = 31): which image it gets, how long that takes, and whether
every route gives the same results as the reference-order kernel and (on a sample) the oracle.  Env: L (bases, default
2.25e9), NREADS (default 200000), ORACLE (reads compared with the oracle, default 600; 0: skip)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from sbwt_amd import capi, synth

L = int(float(os.environ.get("L", 2.25e9)))
n_reads = int(os.environ.get("NREADS", 200000))
n_orc = int(os.environ.get("ORACLE", 600))
t0 = time.time()
genome = synth.random_genome(L, 7)
print("genome %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
bits = capi.build_bits_gpu([genome.tobytes()], 31, False, True)
print("columns: n_nodes %d (%.3f x 2^31), %.1f s" % (bits.n_nodes, bits.n_nodes / 2**31, time.time() - t0), flush=True)
t0 = time.time()
idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, 31, bits.n_kmers, 8)
print("image: level %d, %.1f GB (%.1f B per column), paths %d, %.1f s; free device memory %.1f GB" % (
    idx.image_level, idx.blob_bytes / 1e9, idx.blob_bytes / idx.n_nodes, idx.n_paths, time.time() - t0,
    torch.cuda.mem_get_info()[0] / 1e9), flush=True)
bases, off = synth.sample_reads([genome], n_reads, 150, 0.01, 5)
bases = synth.inject(bases, max(40, n_reads // 100), ord("N"), 6)
bases = synth.inject(bases, max(10, n_reads // 1000), ord("a"), 7)
res = {}
for v in (-1, 1, 0):
    capi.set_tuning("search_variant", v)
    t0 = time.time()
    res[v], _ = idx.streaming_search(bases, off)
    res[(v, "search")], _ = idx.search(bases, off)
    print("variant %2d: %.2f s, found %.3f, above 2^31: %d" % (v, time.time() - t0, float((res[v] >= 0).mean()),
                                                             int((res[v] >= (1 << 31)).sum())), flush=True)
capi.set_tuning("search_variant", -1)
ok = True
for key, r in res.items():
    ref = res[0] if not isinstance(key, tuple) else res[(0, "search")]
    same = np.array_equal(r, ref)
    ok = ok and same
    if not same:
        bad = np.flatnonzero(r != ref)
        print("DIFF", key, len(bad), "first at", bad[:5], r[bad[:5]], ref[bad[:5]], flush=True)
print("all routes equal the reference-order kernel:", ok, flush=True)
if n_orc:
    from oracle import OracleIndex
    t0 = time.time()
    orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, 31, bits.n_kmers, 8)
    want = np.concatenate([orc.streaming_search(bases[off[r]:off[r + 1]].tobytes()) for r in range(n_orc)])
    same = np.array_equal(res[-1][:len(want)], want)
    ok = ok and same
    print("oracle on %d reads: %s (%.1f s)" % (n_orc, same, time.time() - t0), flush=True)
# the device-resident rate
dev = torch.device("cuda:0")
n_big = int(os.environ.get("BENCH_READS", 4_000_000))
bb, oo = synth.sample_reads([genome], n_big, 150, 0.01, 9)
d_b, d_ro = torch.from_numpy(bb).to(dev), torch.from_numpy(oo).to(dev)
ooff = capi.out_offsets(oo, 31)
d_oo = torch.from_numpy(ooff).to(dev)
d_out = torch.empty(int(ooff[-1]), dtype=torch.int64, device=dev)
wsb = capi.search_workspace_bytes(d_b.numel())
d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    idx.streaming_search_dev(d_b.data_ptr(), d_b.numel(), d_ro.data_ptr(), n_big, d_out.data_ptr(), d_oo.data_ptr(), d_ws.data_ptr(), wsb, st, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
print("device-resident: %d reads in %.2f ms = %.1f G k-mers/s" % (n_big, ms, int(ooff[-1]) / ms / 1e6), flush=True)
sys.exit(0 if ok else 1)
