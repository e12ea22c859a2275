"""A/B timing of whole search STEPS (sbwtgpu_streaming_search_dev: every kernel of the route, encode included) in ONE
process, interleaved rounds.  Env: NREADS, GLEN, ROUNDS, READLEN, K, CONFIGS = json list of [variant, debug(, fused_sort)] entries,
GENOMES = coli3 | pan<N> | single.  Prints per config: median / min step ms, the fused kernel's own ms (variant 5), the
work counters, and a checksum that must be the same for every config."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sbwt_amd import capi, hostlib, synth
import bench as B

n_reads = int(os.environ.get("NREADS", 10_000_000))
glen = int(os.environ.get("GLEN", 5_000_000))
K = int(os.environ.get("K", 30))
L = int(os.environ.get("READLEN", 150))
configs = json.loads(os.environ.get("CONFIGS", "[[5,0],[4,0]]"))
rounds = int(os.environ.get("ROUNDS", 5))
streaming = int(os.environ.get("STREAMING", 1))
gsel = os.environ.get("GENOMES", "coli3")
dev = torch.device("cuda", 0)
if gsel.startswith("pan"):
    genomes = synth.pan_like(int(gsel[3:] or 64), glen)
else:
    genomes = synth.coli3_like(glen)
    if gsel == "single":
        genomes = genomes[:1]
B.K, B.READ_LEN = K, L
if K <= 64:
    bits = capi.build_bits_gpu([g.tobytes() for g in genomes], K, False, bool(streaming))
else:
    bits = hostlib.build_bits([g.tobytes() for g in genomes], K, False, bool(streaming), n_threads=os.cpu_count())
idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, K, bits.n_kmers, 8)
print("index: n_nodes", idx.n_nodes, "image MB", idx.blob_bytes / 1e6, "B/col", idx.blob_bytes / idx.n_nodes, "paths", idx.n_paths,
      "branching", idx.n_branch, flush=True)
if os.environ.get("SUBS"):          # SUBS=0.0 / 0.05: the reads' substitution rate (default: bench.py's 1 %)
    B.SUB_RATE = float(os.environ["SUBS"])
d_bases = B.gpu_reads(genomes, n_reads, 42, dev, in_genome_order=bool(os.environ.get("SORTED")))
if os.environ.get("ABSENT"):        # ABSENT=1: uniform-random reads (nothing of them is in the index: searches only, no path is followed)
    g = torch.Generator(device=dev); g.manual_seed(9)
    d_bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n_reads * L,), device=dev, generator=g)]
m = L - K + 1
d_roff = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * L
d_ooff = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * m
ostride = int(os.environ.get("OSTRIDE", 0))        # OSTRIDE=128: every read's results start at a multiple of 128 slots (1 KB)
if ostride:
    d_ooff = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * ostride
if os.environ.get("RAGGED"):       # RAGGED=lo: reads of lo .. READLEN bases (host generator), checksum over a flat view
    hb, ho = synth.ragged_reads(genomes, n_reads, int(os.environ["RAGGED"]), L, 0.01, 43)
    d_bases = torch.from_numpy(hb).to(dev)
    d_roff = torch.from_numpy(ho).to(dev)
    d_ooff = torch.from_numpy(capi.out_offsets(ho, K)).to(dev)
n_kmers = int(d_ooff[-1].item())
I32 = bool(os.environ.get("I32"))               # I32=1: the int32-result entry point (sbwtgpu_*_dev_i32)
d_out = torch.empty(n_kmers, dtype=torch.int32 if I32 else torch.int64, device=dev)
if ostride:
    n_kmers = n_reads * m                         # (the rate counts k-mers, not slots)
wsb = capi.search_workspace_bytes(d_bases.numel())
d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
w = torch.arange(1, m + 1, device=dev, dtype=torch.int64)
if os.environ.get("SORTED") == "2" and not os.environ.get("RAGGED"):
    # the batch laid out by the PATH POSITION of each read's first k-mer that the index holds (what a pre-pass could do):
    # pos[] out of the image (SbwtBlobHeader.off_pos is the int64 at byte 176), one search for the columns
    import struct
    off_pos = struct.unpack_from("<q", idx.export_header(), 176)[0]
    blob = idx.blob_tensor()
    pos = blob[off_pos:off_pos + 4 * idx.n_nodes].view(torch.int32)
    idx.streaming_search_dev(d_bases.data_ptr(), d_bases.numel(), d_roff.data_ptr(), n_reads, d_out.data_ptr(), d_ooff.data_ptr(),
                             d_ws.data_ptr(), wsb, st, bool(streaming))
    torch.cuda.synchronize()
    res = d_out.view(n_reads, m)
    first = (res >= 0).to(torch.int8).argmax(dim=1)
    col0 = res.gather(1, first[:, None])[:, 0]
    key = torch.where(col0 >= 0, pos[col0.clamp(min=0)].long() - first, torch.full_like(col0, 1 << 40))
    shift = int(os.environ.get("SORT_SHIFT", 0))       # SORT_SHIFT=s: only by the key's bits above s (a bucket sort's order)
    perm = torch.argsort(key >> shift, stable=True)
    d_bases = d_bases.view(n_reads, L)[perm].contiguous().view(-1)
    del res, first, col0, key, perm
    print("reads laid out by path position (shift %d)" % shift, flush=True)
ref = None
times = {tuple(c): [] for c in configs}
ktimes = {tuple(c): [] for c in configs}
capi.set_tuning("kernel_events", 1)
for rnd in range(rounds + 1):
    for c in configs:
        capi.set_tuning("search_variant", c[0]); capi.set_tuning("debug", c[1] if len(c) > 1 else 0)
        # (third entry: "fused_sort" -- 0 the unsorted kernel, 7728 = 3632 | 4096 the sorted one whatever the workspace's hint says,
        # 3632 as shipped: by the hint the call before left; without a third entry the library's default, 3632)
        try:
            capi.set_tuning("fused_sort", c[2] if len(c) > 2 else int(os.environ.get("SBWTGPU_FUSED_SORT", 3632)))
        except capi.SbwtGpuError:
            pass                                                       # (a library of an earlier round)
        if rnd == 0:
            d_out.fill_(-7)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        (idx.streaming_search_dev_i32 if I32 else idx.streaming_search_dev)(
            d_bases.data_ptr(), d_bases.numel(), d_roff.data_ptr(), n_reads, d_out.data_ptr(), d_ooff.data_ptr(),
            d_ws.data_ptr(), wsb, st, bool(streaming))
        e1.record(); torch.cuda.synchronize()
        if rnd == 0:
            if os.environ.get("RAGGED"):
                chk = int((d_out.long() * (torch.arange(n_kmers, device=dev) % 1009 + 1)).sum().item())
            else:
                dv = d_out.view(n_reads, ostride)[:, :m] if ostride else d_out.view(n_reads, m)
                chk = int(((dv.long() * w).sum(dim=1) * torch.arange(1, n_reads + 1, device=dev)).sum().item())
            if ref is None: ref = chk
            if len(c) > 2 and c[2]:
                hdr = d_ws[:256].cpu().numpy().view("uint64")      # ws->pad[11..14]: wave-iterations and busy lanes per wave class
                it_s, b_s, it_p, b_p = (int(hdr[16 + q]) for q in (11, 12, 13, 14))
                print("   SORT: searcher waves %d iterations x %.1f busy lanes, follower waves %d x %.1f; lists written %d, not a read's last %d" % (
                    it_s, b_s / max(1, it_s), it_p, b_p / max(1, it_p), int(hdr[16 + 9]), int(hdr[16 + 10])), flush=True)
            print("config", c, "checksum", chk, "same" if chk == ref else "DIFFERENT", "stats", idx.workspace_stats(d_ws.data_ptr(), st),
                  "bridges", idx.workspace_bridges(d_ws.data_ptr(), st), flush=True)
        else:
            times[tuple(c)].append(e0.elapsed_time(e1))
            if c[0] == 5:
                ktimes[tuple(c)].append(capi.kernel_times()[-1])
for c, v in times.items():
    kt = ktimes[c]
    if os.environ.get("PRINT_ALL"):      # the steps in order: does the box slow down while it runs?
        print("steps ms:", " ".join("%.2f" % x for x in v))
    print(f"variant={c[0]} debug={c[1] if len(c) > 1 else 0} sort={c[2] if len(c) > 2 else 'default'}: step median {np.median(v):.3f} ms min {min(v):.3f} ms -> "
          f"{n_kmers / np.median(v) / 1e6:.2f} G kmers/s" + (f"; fused kernel median {np.median(kt):.3f} ms" if kt else ""))
