"""Lane-iterations of k_search_fused by kind.  Needs a library built with -DSBWT_STATS (tools/build_stats_lib.sh):
  SBWTGPU_LIB=$PWD/sbwt_amd/lib/lib_stats.so python tools/lane_stats_fused.py        (env of tools/ab_step.py applies)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CONFIGS", "[[5,0]]"); os.environ.setdefault("ROUNDS", "1")
import tools.ab_step as ab
hdr = ab.d_ws[:256].cpu().numpy().view("uint64")
names = ["sparse lookup", "filter probe", "dense table", "second level", "interval update", "path run", "transition", "bridge", "pos",
         "idle"]
base = 16        # pad[0] is the 17th 8-byte word of the header
vals = [int(hdr[base + q]) for q in range(10)]
tot = sum(vals)
nr = ab.n_reads
print({n: round(v / nr, 3) for n, v in zip(names, vals)})
print("lane-iterations per read", round(tot / nr, 2), "wave-iterations", int(hdr[base + 10]), "lanes busy per iteration",
      round((tot - vals[9]) / max(1, int(hdr[base + 10])), 1))
# read lines per read by category: one line per lane-iteration of a kind -- a lookup's two 16-byte loads share
# their 128-byte line --, 1.25 for path runs and bridges (their two quads straddle a line one time in four), two for an interval
# update; the result lists' col[] lines and the read's own bases are not lane-iterations (config 2: 5.4 + 1.2)
w = [1, 1, 1, 1, 2, 1.25, 1, 1.25, 1]
lines = {n: round(v * f / nr, 2) for n, v, f in zip(names, vals, w)}
print("read lines per read by category (gathers only):", lines, "sum", round(sum(lines.values()), 2))
print("after the tickets ran out: wave-iterations", int(hdr[base + 11]), "(per wave %.1f)" % (int(hdr[base + 11]) / 5120), "idle lane-iterations",
      int(hdr[base + 12]), "= %.2f per read" % (int(hdr[base + 12]) / nr))
print("splits", int(hdr[base + 13]), "donor-capable lane-iterations in the tail", int(hdr[base + 14]))

try:
    import ctypes
    from sbwt_amd import capi
    h = (ctypes.c_ulonglong * 64)()
    if capi.lib().sbwtgpu_debug_iter_hist(h, 1) == 0:
        tot = sum(h)
        acc = 0
        rows = []
        for q in range(64):
            acc += h[q]
            if h[q]:
                rows.append("%d-%d: %.4f (cum %.4f)" % (2 * q, 2 * q + 1, h[q] / tot, acc / tot))
        print("lane-iterations per read or piece (%d of them):" % tot, "; ".join(rows))
except Exception as ex:
    print("no iteration histogram:", ex)

try:
    t = (ctypes.c_ulonglong * 192)()
    if capi.lib().sbwtgpu_debug_tail_prof(t, 1) == 0 and t[0]:
        print("the tail by iteration since a wave saw the tickets run out -- waves still running (of %d), busy lanes per wave, lanes that could give work away:" % t[0])
        print("; ".join("%d: %d %.1f %.1f" % (q, t[q], t[64 + q] / t[q], t[128 + q] / t[q]) for q in range(64) if t[q]))
except Exception as ex:
    print("no tail profile:", ex)

try:
    w = (ctypes.c_ulonglong * 24)()
    if capi.lib().sbwtgpu_debug_why(w, 1) == 0 and any(w):
        names = ["path runs that stop (mismatch or end of path)", "... at a step with other successors or the path's end -> transition lookup",
                 "... at a substitution-safe step -> bridge", "...... but a second difference within the window: certificates", "... at an only-successor step that is not safe: certificates",
                 "transition lookups: free slot", "negative entry, safe for this char -> bridge", "negative entry, not safe: certificates", "successor found",
                 "... whose quoted steps stop: absent", "... transition again", "... bridge",
                 "bridges that hold (k-1 bases)", "bridges that hold to the read's end", "bridges that fail (another difference within k-1 bases)",
                 "(of the stops at steps with other successors: mismatches, not path ends)"]
        launches = 2
        print("why substitutions are (not) bridged, per read:")
        for q in range(16):
            print("  %-90s %.3f" % (names[q], w[q] / launches / nr))
        if any(w[16:20]):
            print("k > 31, second-level misses (the k-mer is absent, its 31-prefix is there) by the prefix's columns, per read: one %.3f, two %.3f, three %.3f, more %.3f"
                  % tuple(w[q] / launches / nr for q in range(16, 20)))
except Exception as ex:
    print("no why-counters:", ex)

try:
    pl = (ctypes.c_ulonglong * 48)()
    if capi.lib().sbwtgpu_debug_plan(pl, 1) == 0 and any(pl):
        launches = 2
        names = {0: "plan: k-mer i's own lookup (nothing known about its window)", 1: "plan: own lookup, forced (a probe was inconclusive)",
                 2: "plan: own lookup, blind mode", 8: "plan: own lookup although b lies in the window", 3: "plan: filter window ENDING at b",
                 4: "plan: filter window STARTING at b", 9: "plan: filter window elsewhere", 5: "plan: range probe", 6: "plan: hinted probe",
                 7: "plan: dense-table walk", 10: "filter window: absent", 11: "filter window: (perhaps) present", 12: "range probe: absent",
                 13: "range probe: present", 14: "hinted probe: absent", 15: "hinted probe: present", 16: "own lookup (k-mer i): found",
                 17: "own lookup (k-mer i): not there", 18: "sparse bucket overflow: next bucket", 19: "lookup elsewhere: found",
                 20: "lookup elsewhere: not there", 21: "a read's first lookup: found", 22: "a read's first lookup: not there"}
        print("what the planner starts and what comes of it, per read:")
        for q in sorted(names):
            print("  %-70s %.3f" % (names[q], pl[q] / launches / nr))
        nprobe = pl[10] + pl[12] + pl[14]
        print("  k-mers certified absent per absent filter window: %.2f" % (pl[24] / max(1, nprobe)))
except Exception as ex:
    print("no planner counters:", ex)

try:
    t = (ctypes.c_ulonglong * 16)()
    if capi.lib().sbwtgpu_debug_sort_stats(t, 1) == 0 and t[0]:
        nm = ["searcher", "follower"]
        for c in (0, 1):
            it = max(1, t[4 * c])
            print("SORT %s waves: %d wave-iterations, per iteration %.1f lanes busy, %.1f waiting at the ring, %.1f idle with a slot" % (
                nm[c], t[4 * c], t[4 * c + 1] / it, t[4 * c + 2] / it, t[4 * c + 3] / it))
        print("SORT per read: %.2f hand-overs to the followers, %.2f to the searchers, %.2f free slots returned, %.3f hand-overs that waited "
              "for the writer; %d skipped wave-iterations" % (t[8] / nr, t[9] / nr, t[10] / nr, t[11] / nr, t[12]))
except Exception as ex:
    print("no SORT statistics:", ex)
