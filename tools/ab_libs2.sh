#!/bin/bash
# A/B of two builds of libsbwtgpu.so on one box, whole steps of the default route at several batch sizes:
#   tools/ab_libs2.sh <other.so> [rounds]        (same box, interleaved)
OTHER=$1; R=${2:-2}
for n in ${SIZES:-1000000 4000000 10000000}; do
  for i in $(seq 1 $R); do
    NREADS=$n ROUNDS=9 CONFIGS='[[5,0]]' python tools/ab_step.py 2>&1 | grep "^variant" | sed "s/^/$n base : /"
    SBWTGPU_LIB=$OTHER NREADS=$n ROUNDS=9 CONFIGS='[[5,0]]' python tools/ab_step.py 2>&1 | grep "^variant" | sed "s/^/$n other: /"
  done
done
for i in $(seq 1 $R); do
  RAGGED=80 NREADS=10000000 ROUNDS=9 CONFIGS='[[5,0]]' python tools/ab_step.py 2>&1 | grep "^variant" | sed "s/^/ragged base : /"
  SBWTGPU_LIB=$OTHER RAGGED=80 NREADS=10000000 ROUNDS=9 CONFIGS='[[5,0]]' python tools/ab_step.py 2>&1 | grep "^variant" | sed "s/^/ragged other: /"
done
