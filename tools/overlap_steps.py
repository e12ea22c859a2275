"""Whole search steps with TWO batches in flight (two streams, two workspaces, two result buffers): what the tail of one
launch -- the ~0.5 ms in which its waves leave one by one -- is worth when the next launch can fill the chip behind it.
Env of tools/ab_step.py (NREADS, K, GENOMES ..); prints serial and overlapped ms per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CONFIGS", "[[5,0]]"); os.environ.setdefault("ROUNDS", "3")
import numpy as np
import torch
import tools.ab_step as ab

dev = ab.dev
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
outs = [ab.d_out, torch.empty_like(ab.d_out)]
wss = [ab.d_ws, torch.empty_like(ab.d_ws)]
steps = int(os.environ.get("STEPS", 20))


def run(n_streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        q = s % n_streams
        with torch.cuda.stream(streams[q]):
            ab.idx.streaming_search_dev(ab.d_bases.data_ptr(), ab.d_bases.numel(), ab.d_roff.data_ptr(), ab.n_reads, outs[q].data_ptr(),
                                        ab.d_ooff.data_ptr(), wss[q].data_ptr(), ab.wsb, streams[q].cuda_stream, bool(ab.streaming))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


run(1); run(2)
a = [run(1) for _ in range(3)]
b = [run(2) for _ in range(3)]
assert torch.equal(outs[0], outs[1])
print("reads %d: one batch in flight %.3f ms per step (%.1f G k-mers/s); two in flight %.3f ms per step (%.1f G k-mers/s)" %
      (ab.n_reads, min(a), ab.n_kmers / min(a) / 1e6, min(b), ab.n_kmers / min(b) / 1e6))
