"""Lane-iterations by kind for BASELINE config 5 (k=63, no streaming support; SBWT::search of every k-mer).
Needs a library built with -DSBWT_STATS (SBWTGPU_LIB=...)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sbwt_amd import capi, hostlib, synth
import bench as B
K = 63
n_reads = int(os.environ.get("NREADS", 10_000_000))
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
genomes = synth.coli3_like(5_000_000)
bits = hostlib.build_bits([g.tobytes() for g in genomes], K, False, False, n_threads=os.cpu_count())
idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], None, bits.n_nodes, K, bits.n_kmers, 8)
d_bases = B.gpu_reads(genomes, n_reads, 42, dev)
m = 150 - K + 1
d_roff = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * 150
d_ooff = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * m
d_out = torch.empty(n_reads * m, dtype=torch.int64, device=dev)
wsb = capi.search_workspace_bytes(d_bases.numel())
d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
idx.streaming_search_dev(d_bases.data_ptr(), d_bases.numel(), d_roff.data_ptr(), n_reads, d_out.data_ptr(), d_ooff.data_ptr(),
                         d_ws.data_ptr(), wsb, st, False)
torch.cuda.synchronize()
hdr = d_ws[:256].cpu().numpy().view("uint64")
names = ["fetch", "reload", "init", "step", "trans", "bridge", "ext", "idle"]
tot = sum(int(hdr[13 + 8 + q]) for q in range(8))
print(json.dumps({"lane_iterations_per_read": {n: round(int(hdr[13 + 8 + q]) / n_reads, 2) for q, n in enumerate(names)},
                  "all": round(tot / n_reads, 2), "wave_iterations": int(hdr[13 + 16]),
                  "stats": idx.workspace_stats(d_ws.data_ptr(), st)}))
