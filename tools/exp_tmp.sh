#!/bin/bash
L="sbwt_amd/lib/lib_base.so sbwt_amd/lib/lib_thin16_4.so sbwt_amd/lib/lib_thin32_2.so sbwt_amd/lib/lib_thin8_2.so"
SBWTGPU_LIB=$PWD/sbwt_amd/lib/lib_base.so ROUNDS=2 python3 tools/ab_step.py > /dev/null 2>&1   # warm-up
echo "== c2 10M =="; ROUNDS=6 tools/ab_libs5.sh "$L" 2 | grep "variant="
echo "== c2 1M =="; NREADS=1000000 ROUNDS=8 tools/ab_libs5.sh "$L" 2 | grep "variant="
echo "== c5 10M =="; K=63 STREAMING=0 ROUNDS=5 tools/ab_libs5.sh "$L" 1 | grep "variant="
echo "== c3 10M =="; GENOMES=pan64 K=31 ROUNDS=5 tools/ab_libs5.sh "$L" 1 | grep "variant="
