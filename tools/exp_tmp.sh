#!/bin/bash
export ROUNDS=12
python3 tools/ab_step.py > /dev/null 2>&1
C='[[5,0],[5,327680]]'   # auto, 1280 workgroups
for n in 50000 100000 200000 400000 700000 1000000 1500000 2000000 3000000 4000000 6000000 10000000; do
echo "== c2 $n =="; NREADS=$n CONFIGS="$C" python3 tools/ab_step.py 2>&1 | grep "variant="
done
echo "== c5 1M =="; K=63 STREAMING=0 NREADS=1000000 CONFIGS="$C" python3 tools/ab_step.py 2>&1 | grep "variant="
echo "== c3 type 1M =="; GENOMES=pan64 K=31 NREADS=1000000 CONFIGS="$C" python3 tools/ab_step.py 2>&1 | grep "variant="
echo "== c2 250bp 1M =="; READLEN=250 NREADS=1000000 CONFIGS="$C" python3 tools/ab_step.py 2>&1 | grep "variant="
