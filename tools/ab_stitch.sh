#!/bin/bash
# Same-box A/B of stitched chains (SBWTGPU_PATH_STITCH=1) against vertex-disjoint paths (=0): config 2 twice each
# (the stitched outcome must be the same both times), the pan-genome index once each, then a fuzz leg.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for s in 1 0; do
    echo "== c2 stitch=$s rep=$rep"
    SBWTGPU_PATH_STITCH=$s CONFIGS='[[5,0]]' ROUNDS=7 timeout 200 python tools/ab_step.py 2>&1 | grep -E "index:|stats|variant=" | cut -c1-330
  done
done
for s in 1 0; do
  echo "== pan stitch=$s"
  SBWTGPU_PATH_STITCH=$s GENOMES=pan64 K=31 CONFIGS='[[5,0]]' ROUNDS=5 timeout 300 python tools/ab_step.py 2>&1 | grep -E "index:|variant=" | cut -c1-250
done
SEED=11 timeout 120 python tools/fuzz_gpu.py 80 2>&1 | tail -1 | cut -c1-200
