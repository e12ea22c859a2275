"""Per-iteration trace of ONE read of the ab_step workload (library built with -DSBWT_TRACE: the fused kernel prints
lane 0's state every iteration when a launch has exactly one read).  RD=<read number>; env of tools/ab_step.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CONFIGS", "[[5,0]]"); os.environ.setdefault("ROUNDS", "1")
import numpy as np, torch
import tools.ab_step as ab
from sbwt_amd import capi
for rd in [int(x) for x in os.environ["RD"].split(",")]:
    read = ab.d_bases[rd * ab.L:(rd + 1) * ab.L].cpu().numpy()
    off = np.array([0, ab.L], dtype=np.int64)
    print("==== read", rd, read.tobytes().decode(), flush=True)
    fn = ab.idx.streaming_search if ab.streaming else ab.idx.search
    out, _ = fn(read, off)
    torch.cuda.synchronize()
    print("result", out.tolist(), flush=True)
