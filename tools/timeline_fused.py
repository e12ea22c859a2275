"""When the waves of ONE k_search_fused launch start, see the ticket counter run out and leave.  Needs a library built by
`tools/build_stats_lib.sh timeline`:
  SBWTGPU_LIB=$PWD/sbwt_amd/lib/lib_timeline.so python tools/timeline_fused.py        (env of tools/ab_step.py applies)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CONFIGS", "[[5,0]]"); os.environ.setdefault("ROUNDS", "3")
import torch
import tools.ab_step as ab
from sbwt_amd import capi
L = capi.lib()
buf = (ctypes.c_ulonglong * (8 + 1024))()
L.sbwtgpu_debug_timeline(None, 1)
ab.idx.streaming_search_dev(ab.d_bases.data_ptr(), ab.d_bases.numel(), ab.d_roff.data_ptr(), ab.n_reads, ab.d_out.data_ptr(),
                            ab.d_ooff.data_ptr(), ab.d_ws.data_ptr(), ab.wsb, ab.st, bool(ab.streaming))
torch.cuda.synchronize()
assert L.sbwtgpu_debug_timeline(buf, 0) == 0
t0, d_lo, d_hi, t_end, nw, s_tail, s_life, n_it = (int(buf[q]) for q in range(8))
us = lambda x: x / 100.0
print("reads", ab.n_reads, "waves", nw, "kernel (first start -> last exit) %.1f us" % us(t_end - t0))
print("ticket counter ran out for the first wave at %.1f us, for the last at %.1f us" % (us(d_lo - t0), us(d_hi - t0)))
print("a wave's time after that: mean %.1f us; a wave's life: mean %.1f us" % (us(s_tail / nw), us(s_life / nw)))
h = [int(buf[8 + q]) for q in range(1024)]
acc = 0
rows = []
for q, v in enumerate(h):
    acc += v
    if v: rows.append("%d: %d (cum %.3f)" % (q * 10, v, acc / nw))
print("waves leaving per 10 us after the first start:", "; ".join(rows))
print("before that: %.1f iterations per wave, %.2f us each" % (n_it / nw, us(s_life - s_tail) / max(1, n_it)))
