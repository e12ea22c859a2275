"""Which reads take the most lane-iterations?  (stats build: SBWTGPU_LIB=.../lib_stats.so; env of tools/ab_step.py)
Prints the slowest reads the kernel saw with their differences from the genome they were sampled from."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CONFIGS", "[[5,0]]"); os.environ.setdefault("ROUNDS", "1")
import numpy as np
import tools.ab_step as ab
from sbwt_amd import capi
mx = (ctypes.c_ulonglong * 8)()
assert capi.lib().sbwtgpu_debug_iter_max(mx, 1) == 0
cat = np.concatenate(ab.genomes)
L, K = ab.L, ab.K
import torch
for v in sorted(set(mx), reverse=True)[:6]:
    if not v:
        continue
    its, rd = v >> 32, v & 0xFFFFFFFF
    read = ab.d_bases[rd * L:(rd + 1) * L].cpu().numpy()
    # where does it come from?  (exact search of its longest clean stretch is overkill: try every 16-mer)
    s = read.tobytes()
    best = None
    cs = cat.tobytes()
    for w in range(0, L - 20, 7):
        at = cs.find(s[w:w + 20])
        if at >= 0:
            best = at - w
            break
    diffs = None
    if best is not None and best >= 0 and best + L <= len(cat):
        diffs = np.flatnonzero(cat[best:best + L] != read).tolist()
    out = ab.d_out[rd * ab.m:(rd + 1) * ab.m].cpu().numpy()
    print("read", rd, "iterations", its, "differences at", diffs, "hits", int((out >= 0).sum()), "of", ab.m)
    print("  ", s.decode())
