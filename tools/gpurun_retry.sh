#!/bin/bash
# (container side) gpurun with retries while the pod's GPU slots are busy: tools/gpurun_retry.sh <timeout> '<command>'
T=$1; shift
for attempt in $(seq 1 30); do
  out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out" | grep -v "^\[gpurun\] send"
  exit $rc
done
echo "gpurun_retry: no slot after 30 attempts"; exit 3
