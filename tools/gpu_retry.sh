#!/bin/bash
# Developer helper (this container only): retry a gpurun call while the pod's GPU slots are busy.
#   tools/gpu_retry.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 30); do
  out=$(gpurun --timeout "$t" -- "$@" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 90; continue; fi
  echo "$out"; exit 0
done
echo "gpu_retry: no slot after 30 attempts"; exit 3
