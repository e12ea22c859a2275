#!/bin/bash
# A/B of two builds of libsbwtgpu.so on one box over the three bench configurations (whole steps, default route):
#   tools/ab_libs3.sh <other.so> [rounds]
OTHER=$1; R=${2:-2}
for i in $(seq 1 $R); do
  for lib in base other; do
    if [ $lib = other ]; then export SBWTGPU_LIB=$OTHER; else unset SBWTGPU_LIB; fi
    ROUNDS=9 CONFIGS='[[5,0]]' python tools/ab_step.py 2>&1 | grep "^variant" | sed "s/^/c2 $lib: /"
    K=63 STREAMING=0 ROUNDS=9 CONFIGS='[[5,0]]' python tools/ab_step.py 2>&1 | grep "^variant" | sed "s/^/c5 $lib: /"
    GENOMES=pan64 K=31 ROUNDS=5 CONFIGS='[[5,0]]' python tools/ab_step.py 2>&1 | grep "^variant" | sed "s/^/c3 $lib: /"
  done
done
