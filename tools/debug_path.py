"""Debug helper: compare search variants (0 = reference order, 1 = certificates, 2 = path order) on a small case
through the device API, with the result buffer pre-filled so that never-written results show."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sbwt_amd import capi, hostlib, synth

k = int(os.environ.get("K", 31))
ssup = int(os.environ.get("SSUP", 0))
genomes = [synth.random_genome(30_000, 4)]
bits = hostlib.build_bits([genomes[0].tobytes()], k, False, bool(ssup), n_threads=4)
idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup if ssup else None,
                        bits.n_nodes, k, bits.n_kmers, int(os.environ.get("P", 0)))
nr = int(os.environ.get("NR", 700))
bases, off = synth.sample_reads(genomes, nr, 120, 0.02, 4)
bases = synth.inject(bases, 15, ord("N"), 1)
dev = torch.device("cuda", 0)
d_bases = torch.from_numpy(bases).to(dev)
d_roff = torch.from_numpy(off.astype(np.int64)).to(dev)
ooff = np.concatenate([[0], np.cumsum(np.maximum(np.diff(off) - k + 1, 0))]).astype(np.int64)
d_ooff = torch.from_numpy(ooff).to(dev)
wsb = capi.search_workspace_bytes(len(bases))
d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
idx.encode_bases_dev(d_bases.data_ptr(), len(bases), d_ws.data_ptr(), wsb, st)
res = {}
for v in (0, 1, 2):
    capi.set_tuning("search_variant", v)
    capi.set_tuning("debug", int(os.environ.get("DEBUG", 0)))
    d_out = torch.full((int(ooff[-1]),), -7, dtype=torch.int64, device=dev)
    idx.search_encoded_dev(len(bases), d_roff.data_ptr(), nr, d_out.data_ptr(), d_ooff.data_ptr(), d_ws.data_ptr(), wsb, bool(ssup), st)
    torch.cuda.synchronize()
    res[v] = d_out.cpu().numpy()
    print("variant", v, "found", int((res[v] >= 0).sum()), "unwritten", int((res[v] == -7).sum()), "stats", idx.workspace_stats(d_ws.data_ptr(), st))
for v in (1, 2):
    d = np.nonzero(res[v] != res[0])[0]
    print("variant", v, "diffs", len(d))
    shown = set()
    for x in d:
        r = int(np.searchsorted(ooff, x, side="right") - 1)
        if r in shown or len(shown) >= 4:
            continue
        shown.add(r)
        lo, hi = ooff[r], ooff[r + 1]
        bad = np.nonzero(res[v][lo:hi] != res[0][lo:hi])[0]
        print("  read", r, "obase", lo, "obase%16", lo % 16, "bad kmers", bad.tolist())
        print("     want", res[0][lo:hi][bad][:12].tolist())
        print("     got ", res[v][lo:hi][bad][:12].tolist())
