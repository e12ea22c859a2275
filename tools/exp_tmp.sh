#!/bin/bash
export ROUNDS=8 CONFIGS='[[5,0],[5,128]]'
python3 tools/ab_step.py > /dev/null 2>&1
for pass in 1 2; do
echo "== c2 pass $pass =="; python3 tools/ab_step.py 2>&1 | grep "variant=\|checksum"
done
echo "== c5 =="; K=63 STREAMING=0 python3 tools/ab_step.py 2>&1 | grep "variant=\|checksum"
echo "== c3 =="; GENOMES=pan64 K=31 python3 tools/ab_step.py 2>&1 | grep "variant=\|checksum"
echo "== c2 1M =="; NREADS=1000000 python3 tools/ab_step.py 2>&1 | grep "variant="
