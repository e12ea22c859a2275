#!/bin/bash
# Calibrates FETCH_SIZE for this access pattern: random 16-byte gathers with a known count.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/calib
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- tools/micro/gather_bench > $OUT/gb.txt 2>/dev/null
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/r -- tools/micro/gather_bench > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
root = os.environ.get("OUT", os.getcwd() + "/gpurun_out/calib")
for sub in ("f", "r"):
    for f in glob.glob(root + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        # dispatches come in the order of gather_bench's loop; print the first config of every table size
        seen = {}
        for r in rows:
            key = (r["Kernel_Name"][:40], r["Counter_Name"])
            seen.setdefault(key, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r.get("Grid_Size")))
        for key, v in seen.items():
            v.sort()
            print(key, [(d, val, g) for d, val, g in v[:40]])
PY
