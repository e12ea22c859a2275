#!/bin/bash
# A/B of two builds of libsbwtgpu.so on one box: tools/ab_libs.sh <other.so> [rounds]   (guide rule 24: same box, interleaved)
OTHER=$1; R=${2:-3}
for i in $(seq 1 $R); do
  NREADS=10000000 ROUNDS=5 CONFIGS="[[2,-1,0]]" python tools/ab_bench.py 2>&1 | tail -1 | sed 's/^/base : /'
  SBWTGPU_LIB=$OTHER NREADS=10000000 ROUNDS=5 CONFIGS="[[2,-1,0]]" python tools/ab_bench.py 2>&1 | tail -1 | sed 's/^/other: /'
done
