"""Condenses a tools/profile.sh output directory into a small text summary (what gets committed
under profiles/): per-kernel time stats from the kernel trace and per-kernel PMC averages."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(root, pattern), recursive=True))

print("== kernel trace (per kernel: calls, total ms, avg ms, min, max) ==")
for f in find("trace/**/*kernel_trace.csv"):
    agg = defaultdict(list)
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "?")
        agg[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name[:90]:90s} n={len(v):4d} total={sum(v):10.3f} avg={sum(v)/len(v):9.4f} min={min(v):9.4f} max={max(v):9.4f}")
    # (the trace's VGPR_Count / SGPR_Count columns are allocation granules of a dispatch, not the code object's numbers: the
    # table at the end of this summary is read from the code objects themselves, every instantiation by name)
print("== kernel stats csv ==")
for f in find("trace/**/*kernel_stats.csv"):
    print(open(f).read()[:3000])
print("== PMC (average per dispatch, by kernel) ==")
for f in find("pmc_*/**/*counter_collection.csv"):
    agg = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row.get("Kernel_Name", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, cs in agg.items():
        if "k_search" in name or "k_encode" in name or "k_rank" in name:
            for c, v in cs.items():
                print(f"{name[:60]:60s} {c:28s} n={len(v):3d} avg={sum(v)/len(v):.6g}")
# registers, spills, LDS and scratch of every search-kernel instantiation, from the code objects inside the library that ran
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sbwt_amd", "lib", "libsbwtgpu.so")
print("== code-object resources of %s (llvm-readelf --notes; per kernel instantiation) ==" % os.path.relpath(lib))
for ln in kernel_resources.table(lib, "k_search|k_rank|k_encode"):
    print(ln)
