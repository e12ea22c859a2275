"""Kernel-only throughput of the search path on harder inputs than the headline workload:
repeated content in the genome, indels in the reads, ragged read lengths.  Index size as in config 2 (3 x 5 Mbp).
Prints one line per workload: k-mers/s (encode + search kernels, HIP events, median of 5) and the work mix per read."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sbwt_amd import capi, synth

K, PRE = 30, 8
n_reads = int(os.environ.get("NREADS", 4_000_000))
glen = int(os.environ.get("GLEN", 5_000_000))
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream


def make_index(genomes, rc=False):
    bits = capi.build_bits_gpu([g.tobytes() for g in genomes], K, rc, True)
    return capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, K, bits.n_kmers, PRE)


def run(name, idx, bases, off):
    d_bases = torch.from_numpy(bases).to(dev)
    ooff = capi.out_offsets(off, K)
    d_roff, d_ooff = torch.from_numpy(off).to(dev), torch.from_numpy(ooff).to(dev)
    n_k = int(ooff[-1])
    d_out = torch.empty(n_k, dtype=torch.int64, device=dev)
    wsb = capi.search_workspace_bytes(len(bases))
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ts = []
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        idx.streaming_search_dev(d_bases.data_ptr(), len(bases), d_roff.data_ptr(), len(off) - 1, d_out.data_ptr(),
                                 d_ooff.data_ptr(), d_ws.data_ptr(), wsb, st, True)
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1))
    s = idx.workspace_stats(d_ws.data_ptr(), st)
    nr = len(off) - 1
    hit = float((d_out >= 0).double().mean().item())
    res = {"workload": name, "reads": nr, "kmers": n_k, "ms": float(np.median(ts)), "G_kmers_per_s": n_k / np.median(ts) / 1e6,
           "hit_rate": round(hit, 4), "per_read": {"run_kmers": s[4] / nr, "transition_steps": s[0] / nr, "walks": s[1] / nr,
                                                   "interval_updates": s[2] / nr}}
    print(json.dumps(res), flush=True)
    return d_out


base = synth.coli3_like(glen)
idx0 = make_index(base)
b, o = synth.sample_reads(base, n_reads, 150, 0.01, 42)
run("config 2: 3 strains 5% apart, 150 bp reads, 1% substitutions", idx0, b, o)
b, o = synth.ragged_reads(base, n_reads, 80, 250, 0.01, 43)
run("ragged read lengths 80-250", idx0, b, o)
# reads of different lengths that the fused kernel can take (<= 160 bases): trimmed reads; with "fused_ragged" = 0 the
# two-pass route as before
b, o = synth.ragged_reads(base, n_reads, 80, 150, 0.01, 49)
run("trimmed reads: lengths 80-150", idx0, b, o)
capi.set_tuning("fused_ragged", 0)
run("trimmed reads: lengths 80-150, two-pass route", idx0, b, o)
capi.set_tuning("fused_ragged", 1)
b, o = synth.sample_reads(base, n_reads, 150, 0.01, 42)
b = np.concatenate([b, b[:151]]); o = np.concatenate([o, [o[-1] + 151]])
run("150 bp reads and one of 151", idx0, b, o)
capi.set_tuning("fused_ragged", 0)
run("150 bp reads and one of 151, two-pass route", idx0, b, o)
capi.set_tuning("fused_ragged", 1)
# reads of more than 160 bases: up to three pieces of the fused kernel by default (round 5; the two-pass route before)
b, o = synth.sample_reads(base, n_reads, 250, 0.01, 48)
run("250 bp reads, 1% substitutions", idx0, b, o)
capi.set_tuning("fused_pieces", 1)
run("250 bp reads, two-pass route (general kernel)", idx0, b, o)
b2, o2 = synth.ragged_reads(base, n_reads, 80, 250, 0.01, 43)
run("ragged read lengths 80-250, two-pass route (general kernel)", idx0, b2, o2)
capi.set_tuning("fused_pieces", -1)
b, o = synth.indel_reads(base, n_reads, 150, 0.01, 0.002, 44)
run("0.2% indels + 1% substitutions", idx0, b, o)
b, o = synth.random_reads(n_reads, 150, 45)
run("random reads (every k-mer absent)", idx0, b, o)
g0 = synth.repeat_genome(glen, 5)
rep = [g0, synth.mutate(g0, 0.05, 2), synth.mutate(g0, 0.05, 3)]
idx1 = make_index(rep)
print(json.dumps({"repeat_index": {"n_nodes": idx1.n_nodes, "image_MB": idx1.blob_bytes / 1e6}}), flush=True)
b, o = synth.sample_reads(rep, n_reads, 150, 0.01, 46)
run("5% repeated content (IS-like, rRNA x7, homopolymer/tandem tracts), 150 bp, 1% substitutions", idx1, b, o)
single = [synth.random_genome(3 * glen, 9)]
idx2 = make_index(single)
b, o = synth.sample_reads(single, n_reads, 150, 0.01, 47)
run("one 15 Mbp strain (long unbranched paths)", idx2, b, o)

# the normal use of the tool: an index with reverse complements (tests/test_CLI.hh:43), reads from both strands
del idx1, idx2
idx3 = make_index(base, rc=True)
print(json.dumps({"revcomp_index": {"n_nodes": idx3.n_nodes, "image_MB": idx3.blob_bytes / 1e6, "paths": idx3.n_paths}}), flush=True)
b, o = synth.both_strand_reads(base, n_reads, 150, 0.01, 50)
run("reverse-complement index (2 x the k-mers), reads from both strands, 150 bp, 1% substitutions", idx3, b, o)

# k > 31: reads of more than 160 bases go through the fused kernel as pieces by default (F_CMP; the general kernel has neither
# bridges nor anchors there)
del idx3
K = 63
idx5 = make_index(base)
b, o = synth.sample_reads(base, n_reads, 250, 0.01, 51)
run("k=63 index, 250 bp reads (fused kernel, pieces: the default for k > 31)", idx5, b, o)
capi.set_tuning("fused_pieces", 1)
run("k=63 index, 250 bp reads, two-pass route (general kernel)", idx5, b, o)
capi.set_tuning("fused_pieces", -1)
