"""Lane-iterations by kind (tools/lane_stats.py) for the robustness workloads: needs a library built with -DSBWT_STATS
(SBWTGPU_LIB=...).  Prints per read: fetch / reload / init / step / trans / bridge / ext / idle, lanes busy per iteration."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sbwt_amd import capi, synth

K, PRE = 30, 8
n_reads = int(os.environ.get("NREADS", 4_000_000))
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
base = synth.coli3_like(5_000_000)
bits = capi.build_bits_gpu([g.tobytes() for g in base], K, False, True)
idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, K, bits.n_kmers, PRE)
names = ["fetch", "reload", "init", "step", "trans", "bridge", "ext", "idle"]


def run(name, bases, off):
    d_bases = torch.from_numpy(bases).to(dev)
    ooff = capi.out_offsets(off, K)
    d_roff, d_ooff = torch.from_numpy(off).to(dev), torch.from_numpy(ooff).to(dev)
    d_out = torch.empty(int(ooff[-1]), dtype=torch.int64, device=dev)
    wsb = capi.search_workspace_bytes(len(bases))
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ts = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        idx.streaming_search_dev(d_bases.data_ptr(), len(bases), d_roff.data_ptr(), len(off) - 1, d_out.data_ptr(),
                                 d_ooff.data_ptr(), d_ws.data_ptr(), wsb, st, True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    hdr = d_ws[:256].cpu().numpy().view("uint64")
    nr = len(off) - 1
    mix = {n: round(int(hdr[13 + 8 + q]) / nr, 2) for q, n in enumerate(names)}
    tot = sum(int(hdr[13 + 8 + q]) for q in range(8))
    print(json.dumps({"workload": name, "kmers_per_read": round(int(ooff[-1]) / nr, 1), "lane_iterations_per_read": mix,
                      "all": round(tot / nr, 2), "wave_iterations": int(hdr[13 + 16]),
                      "lanes_busy": round((tot - int(hdr[13 + 8 + 7])) / max(1, int(hdr[13 + 16])), 1),
                      "ms": round(min(ts[1:]), 3), "G_kmers_per_s": round(int(ooff[-1]) / min(ts[1:]) / 1e6, 1)}), flush=True)


b, o = synth.sample_reads(base, n_reads, 150, 0.01, 42)
run("150 bp", b, o)
b, o = synth.ragged_reads(base, n_reads, 80, 250, 0.01, 43)
run("ragged 80-250", b, o)
b, o = synth.sample_reads(base, n_reads, 250, 0.01, 48)
run("250 bp", b, o)
b, o = synth.indel_reads(base, n_reads, 150, 0.01, 0.002, 44)
run("indels", b, o)
b, o = synth.ragged_reads(base, n_reads, 149, 151, 0.01, 49)
run("ragged 149-151", b, o)
b, o = synth.ragged_reads(base, n_reads, 100, 128, 0.01, 50)
run("ragged 100-128", b, o)
