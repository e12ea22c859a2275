#!/bin/bash
# runs gather_modes plain and under the TCC request-size counters; prints both (GPU box, from the repo root)
export TMPDIR=/tmp
R=$PWD
timeout 120 $R/tools/micro/gather_modes ${1:-1024} 2>&1 | grep -v "^$"
cd /tmp
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  rm -rf /tmp/gm
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/gm -- $R/tools/micro/gather_modes ${1:-1024} > /dev/null 2>&1
  python3 - <<PY
import csv,glob
from collections import defaultdict
rows=[]
for f in glob.glob("/tmp/gm/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg=defaultdict(dict)
for r in rows:
    agg[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"])
    agg[int(r["Dispatch_Id"])]["k"]=r["Kernel_Name"][:40]
for d in sorted(agg):
    print(d, agg[d])
PY
done
