"""Lane-iterations of the search kernel by kind (fetch / reload / walk start / interval update / transition / path run).
Needs a library built with -DSBWT_STATS:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSBWT_STATS -o /tmp/lib_stats.so sbwt_amd/csrc/sbwt_search.hip sbwt_amd/csrc/sbwt_api_kernels.hip sbwt_amd/csrc/sbwt_derived.hip sbwt_amd/csrc/sbwt_format.hip sbwt_amd/csrc/sbwtgpu_capi.cpp -ldl
  SBWTGPU_LIB=/tmp/lib_stats.so python tools/lane_stats.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CONFIGS", "[[2,-1,0,31,1]]"); os.environ.setdefault("ROUNDS", "1")
import torch
import tools.ab_bench as ab   # runs the bench
raw = torch.zeros(32, dtype=torch.int64)
hdr = ab.d_ws[:256].cpu().numpy().view("uint64")
names = ["fetch", "reload", "init", "step", "trans", "pos/bridge", "ext", "idle"]
tot = sum(int(hdr[14 + 8 + q]) for q in range(8))
print({n: int(hdr[14 + 8 + q]) for q, n in enumerate(names)}, "lane-iterations", tot, "wave-iterations", int(hdr[14 + 16]), "lanes busy/iter", (tot - int(hdr[14 + 8 + 7])) / max(1, int(hdr[14 + 16])),
      "path-run iterations cut short by the window", int(hdr[14 + 17]))
