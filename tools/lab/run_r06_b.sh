#!/bin/bash
# round 6, pass B: a library variant against the previous library (csrc/libava_hip_prev.so): step tests, then per-kernel traces
out=gpurun_out/r06_b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_autograd_semantics.py -x -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log
for v in prev base prev base; do
  if [ "$v" = base ]; then export AVA_HIP_LIB_TAG=; else export AVA_HIP_LIB_TAG=$v; fi
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_$v.json 2> $out/bench_$v.err
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k_$v.csv \;
  rm -rf $out/prof
  grep -o '"ms_per_step": [0-9.]*' $out/bench_$v.json
done
python3 tools/ab_diff.py $out/k_prev.csv $out/k_base.csv
