#!/bin/bash
# how often do two CONCURRENT ranks on one GPU (no turn-taking, native asynchronous gloo handles) differ bit for bit from the flat path?
# (the test records it as a warning; -W error turns the warning into a failure so that it can be counted)
n=${1:-12}; out=gpurun_out/r06_conc; mkdir -p $out
fail=0
for i in $(seq 1 $n); do
  if ! timeout 300 python -m pytest "tests/test_gpu_dist.py::test_per_bucket_adam_with_concurrent_ranks_and_asynchronous_handles" -x -q -m gpu -W error::UserWarning > $out/run_$i.log 2>&1; then
    fail=$((fail+1)); echo "run $i NOT bit-identical (or failed)"; grep -E "Error|Warning|assert" $out/run_$i.log | head -3 | cut -c1-300
  fi
done
echo "concurrent two-rank runs that were not bit-identical: $fail / $n" | tee $out/summary.txt
