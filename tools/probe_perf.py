"""Scratch perf probe (not the bench): oracle-built index, GPU streaming search via the dev API."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from sbwt_amd import capi, synth
from oracle import OracleIndex

glen = int(os.environ.get("GLEN", 2_000_000))
n_reads = int(os.environ.get("NREADS", 2_000_000))
sub = float(os.environ.get("SUB", 0.01))
k = 30
t0 = time.time()
genomes = synth.coli3_like(glen)
orc = OracleIndex.build([g.tobytes() for g in genomes], k, True, False, 8)
print("oracle build", time.time() - t0, "n_nodes", orc.n_nodes, flush=True)
cols = orc.columns()
idx = capi.Index.create(cols[0], cols[1], cols[2], cols[3], orc.ssup_words(), orc.n_nodes, k, orc.n_kmers, 8)
print("gpu index: blob MB", idx.blob_bytes / 1e6, "p_dev", idx.device_precalc_k, flush=True)
bases, off = synth.sample_reads(genomes, n_reads, 150, sub, 42)
oo = capi.out_offsets(off, k)
dev = torch.device("cuda:0")
d_bases = torch.from_numpy(bases).to(dev)
d_off = torch.from_numpy(off).to(dev)
d_oo = torch.from_numpy(oo).to(dev)
d_out = torch.empty(int(oo[-1]), dtype=torch.int64, device=dev)
wsb = capi.search_workspace_bytes(len(bases))
d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
def run(streaming=True):
    idx.streaming_search_dev(d_bases.data_ptr(), len(bases), d_off.data_ptr(), n_reads, d_out.data_ptr(),
                             d_oo.data_ptr(), d_ws.data_ptr(), wsb, stream, streaming)
for mode in (True, False):
    run(mode); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): run(mode)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"streaming={mode}: {ms:.3f} ms/pass  {int(oo[-1]) / ms / 1e6:.3f} G kmers/s", flush=True)
run(True); torch.cuda.synchronize()
out = d_out.cpu().numpy()
print("hit fraction", (out >= 0).mean())
# check first 3000 reads against oracle
want = np.concatenate([orc.streaming_search(bases[off[r]:off[r+1]].tobytes()) for r in range(3000)])
print("parity(3000 reads):", np.array_equal(out[:len(want)], want))
ns, nf, nl = orc.count_work(bases[: 20000 * 150], off[:20001])
print("work per kmer (sample): stream", ns / (ns + nf), "searches", nf / (ns + nf), "lf/search", nl / max(nf, 1))
