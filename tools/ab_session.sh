#!/bin/bash
# One box, several builds of libsbwtgpu.so interleaved: tools/ab_session.sh <outdir> <rounds> name=lib[,ENV=VAL...] ...
OUT=$1; R=$2; shift 2
mkdir -p $OUT
for r in $(seq 1 $R); do
  for spec in "$@"; do
    name=${spec%%=*}; rest=${spec#*=}
    lib=${rest%%,*}; envs=""
    if [[ "$rest" == *,* ]]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
    env SBWTGPU_LIB=$lib $envs NREADS=${NREADS:-10000000} ROUNDS=5 CONFIGS="${CONFIGS:-[[4,-1,0]]}" python tools/ab_bench.py 2>&1 \
      | grep "^variant\|^config" | sed "s/^/$name: /" | tee -a $OUT/ab.log
  done
done
