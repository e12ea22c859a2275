#!/bin/bash
# A/B of two builds of libsbwtgpu.so on one box: tools/ab_libs.sh <other.so> [rounds] [configs]   (guide rule 24: same box, interleaved)
OTHER=$1; R=${2:-3}; CFG=${3:-'[[2,-1,0]]'}
for i in $(seq 1 $R); do
  NREADS=10000000 ROUNDS=5 CONFIGS="$CFG" python tools/ab_bench.py 2>&1 | grep "^variant" | sed 's/^/base : /'
  SBWTGPU_LIB=$OTHER NREADS=10000000 ROUNDS=5 CONFIGS="$CFG" python tools/ab_bench.py 2>&1 | grep "^variant" | sed 's/^/other: /'
done
