#!/bin/bash
# PC sampling of the search kernels of tools/ab_step.py (rocprofv3 --pc-sampling-beta-enabled): where the fused kernel's
# issue slots go, per instruction.  usage: tools/pc_sample.sh <tag> [stochastic|host_trap] [interval]
# Output: gpurun_out/pcs_<tag>/hist.txt (samples per code-object offset / instruction, descending) + the raw header.
set -u
TAG=$1; METHOD=${2:-stochastic}; INTERVAL=${3:-}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pcs_$TAG
mkdir -p $OUT
export ROUNDS=${ROUNDS:-3}
if [ "$METHOD" = stochastic ]; then UNIT=cycles; INTERVAL=${INTERVAL:-65536}; else UNIT=time; INTERVAL=${INTERVAL:-100}; fi
timeout ${PCS_TIMEOUT:-420} rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD \
    --pc-sampling-interval $INTERVAL --kernel-trace --output-format csv -d $OUT/raw -- python3 tools/ab_step.py > $OUT/run.out 2> $OUT/run.err
echo "rc $?" >> $OUT/run.out
find $OUT/raw -type f | xargs ls -la > $OUT/files.txt 2>&1
python3 - "$OUT" <<'PY' > $OUT/hist.txt 2> $OUT/hist.err
import csv, glob, os, sys, collections
root = sys.argv[1]
csv.field_size_limit(1 << 30)
for f in sorted(glob.glob(os.path.join(root, "raw/**/*pc_sampling*.csv"), recursive=True)):
    print("==", f)
    rd = csv.DictReader(open(f))
    print("columns:", rd.fieldnames)
    keys = [c for c in rd.fieldnames if any(t in c.lower() for t in ("offset", "instruction", "code_object", "stall", "type", "issued", "reason"))
            and "time" not in c.lower()]
    print("key columns:", keys)
    agg = collections.Counter()
    n = 0
    for row in rd:
        if n < 3: print("row:", row)
        n += 1
        agg[tuple(row[c] for c in keys)] += 1
    print("samples:", n)
    for kv, c in agg.most_common(6000):
        print(c, *kv, sep="\t")
PY
tail -c 2000 $OUT/run.err > $OUT/run.err.tail; rm -f $OUT/run.err
du -sh $OUT/raw > $OUT/raw_size.txt; rm -rf $OUT/raw
head -c 3000 $OUT/hist.txt; cat $OUT/run.out | tail -5; cat $OUT/run.err.tail | tail -20
