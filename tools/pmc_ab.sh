#!/bin/bash
# PMC counters of one ab_bench configuration (run on the GPU box): tools/pmc_ab.sh <tag> '<CONFIGS json>' "<counter group>" ...
set -u
TAG=$1; shift; CFG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/abpmc_$TAG
mkdir -p $OUT
export CONFIGS="$CFG" ROUNDS=${ROUNDS:-1} NREADS=${NREADS:-10000000}
for grp in "$@"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 tools/ab_bench.py > $OUT/$name.log 2>&1 || echo "pmc $grp failed"
done
python3 tools/summarize_profile.py $OUT 2>/dev/null | grep "k_search_cert" 
