"""A/B timing of search-kernel variants in ONE process, interleaved rounds (guide rule 24)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sbwt_amd import capi, hostlib, synth
import bench as B
B.SUB_RATE = float(os.environ.get("SUB", B.SUB_RATE))

n_reads = int(os.environ.get("NREADS", 4_000_000))
glen = int(os.environ.get("GLEN", 5_000_000))
configs = json.loads(os.environ.get("CONFIGS", '[[0,-1],[1,-1],[1,0]]'))   # [variant, probe_len]
rounds = int(os.environ.get("ROUNDS", 3))
dev = torch.device("cuda", 0)
genomes = synth.coli3_like(glen)
if os.environ.get("SINGLE"):            # one strain only: long unbranched paths
    genomes = genomes[:1]
bits = hostlib.build_bits([g.tobytes() for g in genomes], 30, False, True, n_threads=os.cpu_count())
indexes = {}
def index_for(sparse, pfilter=1, path=1):   # configs may carry: [3] sparse table depth (0 = off), [4] probe filter, [5] path order
    key = (sparse, pfilter, path)
    if key not in indexes:
        capi.set_tuning("sparse_depth", sparse)
        capi.set_tuning("probe_filter", pfilter)
        capi.set_tuning("path_order", path)
        t0 = time.time()
        ix = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup, bits.n_nodes, 30, bits.n_kmers, 8)
        print("sparse/filter/path", key, "n_nodes", ix.n_nodes, "p_dev", ix.device_precalc_k, "blob MB", ix.blob_bytes / 1e6,
              "create s", round(time.time() - t0, 2), flush=True)
        indexes[key] = ix
    return indexes[key]
idx = index_for(31)
d_bases = B.gpu_reads(genomes, n_reads, 42, dev)
if os.environ.get("SORTED"):
    # experiment: the same kind of reads, but in genome order (what a position-sorting pre-pass would produce)
    gen = torch.Generator(device=dev); gen.manual_seed(42)
    L = B.READ_LEN
    cat = torch.from_numpy(np.concatenate(genomes)).to(dev)
    lens = torch.tensor([len(g) for g in genomes], device=dev)
    starts = torch.tensor(np.concatenate([[0], np.cumsum([len(g) for g in genomes])[:-1]]), device=dev)
    which = torch.randint(0, len(genomes), (n_reads,), device=dev, generator=gen)
    u = torch.rand(n_reads, device=dev, generator=gen, dtype=torch.float64)
    off = ((u * (lens[which] - L + 1)).long() + starts[which]).sort().values
    if os.environ.get("SORTED") == "block":      # sorted in blocks of 64K reads only (coarse bucket sort)
        pass
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    code = torch.zeros(256, dtype=torch.uint8, device=dev); code[acgt.long()] = torch.arange(4, dtype=torch.uint8, device=dev)
    ar = torch.arange(L, device=dev)
    for lo in range(0, n_reads, 1 << 20):
        n = min(1 << 20, n_reads - lo)
        r = cat[(off[lo:lo + n, None] + ar[None, :]).reshape(-1)]
        hit = torch.rand(n * L, device=dev, generator=gen) < B.SUB_RATE
        shift = torch.randint(1, 4, (n * L,), device=dev, generator=gen, dtype=torch.uint8)
        d_bases[lo * L:(lo + n) * L] = torch.where(hit, acgt[((code[r.long()] + shift) & 3).long()], r)
m = 121
d_roff = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * 150
d_ooff = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * m
d_out = torch.empty(n_reads * m, dtype=torch.int64, device=dev)
wsb = capi.search_workspace_bytes(d_bases.numel())
d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
idx.encode_bases_dev(d_bases.data_ptr(), d_bases.numel(), d_ws.data_ptr(), wsb, st)
ref = None
times = {tuple(c): [] for c in configs}
for rnd in range(rounds + 1):
    for c in configs:
        capi.set_tuning("search_variant", c[0]); capi.set_tuning("probe_len", c[1]); capi.set_tuning("debug", c[2] if len(c) > 2 else 0)
        capi.set_tuning("trans_ext", c[6] if len(c) > 6 else -1)      # [6]: run on from transitions (-1 = as the index says)
        idx = index_for(c[3] if len(c) > 3 else 31, c[4] if len(c) > 4 else 1, c[5] if len(c) > 5 else 1)
        if rnd == 0:
            d_out.fill_(-7)        # never-written results must not inherit the previous config's values
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        idx.search_encoded_dev(d_bases.numel(), d_roff.data_ptr(), n_reads, d_out.data_ptr(), d_ooff.data_ptr(), d_ws.data_ptr(), wsb, True, st)
        e1.record(); torch.cuda.synchronize()
        if rnd == 0:
            chk = int((d_out * torch.arange(1, d_out.numel() + 1, device=dev)).sum().item())
            if ref is None: ref = chk
            print("config", c, "checksum", chk, "same" if chk == ref else "DIFFERENT", "stats", idx.workspace_stats(d_ws.data_ptr(), st),
                  "bridges", idx.workspace_bridges(d_ws.data_ptr(), st) if hasattr(idx, "workspace_bridges") else "-", flush=True)
        else:
            times[tuple(c)].append(e0.elapsed_time(e1))
for c, v in times.items():
    print(f"variant={c[0]} probe={c[1]} rest={c[2:]}: median {np.median(v):.2f} ms min {min(v):.2f} ms -> {n_reads * m / np.median(v) / 1e6:.2f} G kmers/s")
