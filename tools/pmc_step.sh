#!/bin/bash
# PMC counters of the search kernels of tools/ab_step.py (env as for ab_step.py: NREADS, CONFIGS, GENOMES, K, SORTED, ...).
# usage: tools/pmc_step.sh <tag> "<counter group>" ["<counter group>" ...]      (one rocprofv3 pass per group)
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcstep_$TAG
mkdir -p $OUT
export ROUNDS=${ROUNDS:-1}
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 tools/ab_step.py > $OUT/g$i.out 2> $OUT/g$i.err || echo "group $i ($grp) failed" >> $OUT/errors.txt
done
python3 - "$OUT" <<'PY' > $OUT/summary.txt
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "g*/**/*counter_collection.csv"), recursive=True)):
    agg = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row.get("Kernel_Name", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, cs in agg.items():
        if "k_search" in name or "k_presort" in name:
            for c, v in cs.items():
                print(f"{name[:72]:72s} {c:28s} n={len(v):3d} avg={sum(v)/len(v):.6g}")
PY
rm -rf $OUT/g*/
cat $OUT/summary.txt; cat $OUT/errors.txt 2>/dev/null
