"""sbwtgpu_streaming_search_batch on pinned host buffers (2 M reads of config 2): seconds per call, G k-mers/s, fraction of
what the device-to-host copy rate allows.  SBWTGPU_PIPE_CHUNK_MB sets the pipeline's chunk size (results per chunk)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sbwt_amd import capi, synth
K, L, E = 30, 150, int(os.environ.get("NREADS", 2_000_000))
genomes = synth.coli3_like(5_000_000)
bits = capi.build_bits_gpu([g.tobytes() for g in genomes], K, False, True)
idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, K, bits.n_kmers, 8)
hb, ho = synth.sample_reads(genomes, E, L, 0.01, 42)
m = L - K + 1
h_bases = torch.empty(E * L, dtype=torch.uint8, pin_memory=True); h_bases.numpy()[:] = hb
h_out = torch.empty(E * m, dtype=torch.int64, pin_memory=True)
roff = np.arange(E + 1, dtype=np.int64) * L
ooff = np.arange(E + 1, dtype=np.int64) * m
fn = capi.lib().sbwtgpu_streaming_search_batch
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    capi._check(fn(idx.handle, h_bases.data_ptr(), roff.ctypes.data, E, h_out.data_ptr(), ooff.ctypes.data))
    ts.append(time.perf_counter() - t0)
d = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0"); hp = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
torch.cuda.synchronize(); t0 = time.perf_counter(); hp.copy_(d); torch.cuda.synchronize(); rate = (1 << 30) / (time.perf_counter() - t0)
best = float(np.median(ts[1:]))
print(f"chunk {os.environ.get('SBWTGPU_PIPE_CHUNK_MB', 'default')} MB: {best * 1e3:.1f} ms -> {E * m / best / 1e9:.2f} G k-mers/s; D2H {rate / 1e9:.1f} GB/s "
      f"-> bound {rate / 8 / 1e9:.2f} G k-mers/s, fraction {E * m / best / (rate / 8):.3f}")
