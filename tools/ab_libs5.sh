#!/bin/bash
# Several builds of the library on one box, interleaved: tools/ab_libs5.sh "<lib> <lib> ..." [passes]   (env as for ab_step.py)
# A lib is a path under sbwt_amd/lib/ (lib_r4.so = round 4's sources, built by the session that compares).  Prints one line
# per (pass, lib): the step and kernel medians of tools/ab_step.py.
set -u
LIBS=${1:-"sbwt_amd/lib/lib_r4.so sbwt_amd/lib/libsbwtgpu.so"}
PASSES=${2:-2}
export ROUNDS=${ROUNDS:-4} CONFIGS=${CONFIGS:-"[[5,0]]"}
for p in $(seq 1 $PASSES); do
  for lib in $LIBS; do
    SBWTGPU_LIB=$PWD/$lib python3 tools/ab_step.py 2>&1 | grep "variant=\|checksum" | sed "s|^|pass $p $(basename $lib): |"
  done
done
