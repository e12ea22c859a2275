#!/bin/bash
# A few PMC groups on a bench.py run: tools/pmc_bench.sh <tag> <bench args...>
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcb_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 bench.py "$@" > $OUT/g$i.out 2> $OUT/g$i.err
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "g*/**/*counter_collection.csv"), recursive=True)):
    agg = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row.get("Kernel_Name", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, cs in agg.items():
        if "k_search" in name:
            for c, v in cs.items():
                print(f"{name[:44]:44s} {c:24s} n={len(v):3d} avg={sum(v)/len(v):.6g}")
PY
rm -rf $OUT/g*/
