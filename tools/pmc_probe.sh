#!/bin/bash
# Counter groups for the search kernel on a short A/B run (one index build, the product kernel only).
# usage: tools/pmc_probe.sh <tag> [counter groups, quoted ...]
set -u
TAG=${1:-probe}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export NREADS=${NREADS:-10000000} ROUNDS=${ROUNDS:-2} CONFIGS=${CONFIGS:-'[[2,-1,0]]'}
if [ $# -eq 0 ]; then
  set -- "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
fi
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 tools/ab_bench.py > $OUT/g$i.out 2> $OUT/g$i.err || echo "group $i ($grp) failed" >> $OUT/errors.txt
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
        if "k_search" in name:
            for c, v in cs.items():
                print(f"{name[:48]:48s} {c:28s} n={len(v):3d} avg={sum(v)/len(v):.6g}")
PY
rm -rf $OUT/g*/
cat $OUT/summary.txt; cat $OUT/errors.txt 2>/dev/null
