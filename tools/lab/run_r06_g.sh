#!/bin/bash
# Adam with non-temporal loads / stores: step A/B + Adam kernel time
out=gpurun_out/r06_g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for i in 1 2; do
for v in base ant1 ant2 ant3; do
  if [ "$v" = base ]; then export AVA_HIP_LIB_TAG=; else export AVA_HIP_LIB_TAG=$v; fi
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 30 --warmup 20 --no-cpu-baseline --no-loader-path --global-batch 0 --no-roofline > $out/bench_$v.json 2> $out/bench_$v.err
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k_$v.csv \;
  rm -rf $out/prof
  echo "$v: $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$v.json)  adam avg ns: $(grep adam_flat $out/k_$v.csv | awk -F'",|,' '{print $(NF-5)}')  $(python3 tools/ab_diff.py $out/k_$v.csv $out/k_$v.csv | head -1)"
done
done
