#!/bin/bash
# flake_loop.sh <n> <pytest node id>: run one test n times, count the failures (flaky multi-process tests)
n=${1:-10}; shift
fail=0
for i in $(seq 1 $n); do
  if ! timeout 300 python -m pytest "$@" -x -q > /tmp/flake_$i.log 2>&1; then fail=$((fail+1)); echo "run $i FAILED"; grep -E "AssertionError" /tmp/flake_$i.log | head -2 | cut -c1-300; fi
done
echo "failures: $fail / $n"
