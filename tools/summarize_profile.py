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
    rows = list(csv.DictReader(open(f)))
    if rows:
        r = [x for x in rows if "k_search_fused" in x.get("Kernel_Name", "")] or [x for x in rows if "k_search" in x.get("Kernel_Name", "")]
        if r:
            x = r[-1]
            print("k_search resources:", {k: x[k] for k in x if k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")})
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
