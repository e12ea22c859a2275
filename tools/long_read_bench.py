"""Long reads through the device-pointer entry point: whole genomes as single reads, and 10 kbp reads, with the device-side
cut into pieces (SbwtPieceTab, "split_long" = 1, the default) and without.  Prints ms per call and G k-mers/s."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sbwt_amd import capi, synth

K = int(os.environ.get("K", 30))            # K=63: the second-level table and the aligned compare
SUBS = float(os.environ.get("SUBS", 0.01))  # substitution rate of the reads
dev = torch.device("cuda", 0)
genomes = synth.coli3_like(int(os.environ.get("GLEN", 5_000_000)))
bits = capi.build_bits_gpu([g.tobytes() for g in genomes], K, False, True)
idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, K, bits.n_kmers, 8)


def run(name, bases, off, split, table=1):
    lens = np.diff(off)
    oo = np.concatenate([[0], np.cumsum(np.maximum(lens - K + 1, 0))]).astype(np.int64)
    d_b = torch.from_numpy(bases).to(dev)
    d_ro = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_oo = torch.from_numpy(oo).to(dev)
    d_out = torch.empty(int(oo[-1]), dtype=torch.int64, device=dev)
    wsb = capi.search_workspace_bytes(d_b.numel())
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    capi.set_tuning("split_long", split)
    capi.set_tuning("fused_table", table)        # (round 6: the fused kernel's ticket table for batches of long reads)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        idx.streaming_search_dev(d_b.data_ptr(), d_b.numel(), d_ro.data_ptr(), len(off) - 1, d_out.data_ptr(), d_oo.data_ptr(),
                                 d_ws.data_ptr(), wsb, st, True)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    capi.set_tuning("split_long", 1)
    capi.set_tuning("fused_table", 1)
    chk = int((d_out * (torch.arange(d_out.numel(), device=dev) % 1009 + 1)).sum().item())
    print(f"k={K} subs={SUBS} {name} split={split} table={table}: {min(ts[1:]) * 1e3:.2f} ms -> {oo[-1] / min(ts[1:]) / 1e9:.2f} G k-mers/s, found "
          f"{int((d_out >= 0).sum().item())} of {oo[-1]}, checksum {chk}", flush=True)


whole = np.concatenate(genomes)
woff = np.concatenate([[0], np.cumsum([len(g) for g in genomes])]).astype(np.int64)
mut = synth.mutate(whole, SUBS, 3)
for table in (1, 0):
    run("3 genomes as 3 reads", mut, woff, 1, table)
for L in (1000, 10000):
    nL = len(mut) // L
    offL = np.arange(nL + 1, dtype=np.int64) * L
    reps = max(1, int(os.environ.get("GBASES", 1)) * 1_000_000_000 // (nL * L))       # (>= GBASES x 10^9 bases per batch)
    b = np.tile(mut[: nL * L], reps)
    o = np.arange(nL * reps + 1, dtype=np.int64) * L
    for table in (1, 0):
        run("%d bp reads (%d of them)" % (L, nL * reps), b, o, 1, table)
