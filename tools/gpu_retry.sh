#!/bin/bash
# gpu_retry.sh <timeout-seconds> <log> <command...>: one gpurun call, retried while the pool answers "transient" (no slot / no box)
t=$1; log=$2; shift 2
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > $log 2>&1
  if grep -q "status=transient" $log; then sleep 75; else break; fi
done
