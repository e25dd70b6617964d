#!/bin/bash
# var_ab2.sh <tag> <variants...>: like var_ab.sh, but with the per-layer kernel parity tests first and the whole conv family printed
tag=$1; shift
out=gpurun_out/r05_var_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "$@"; do
  if [ "$v" = base ]; then export AVA_HIP_LIB_TAG=; else export AVA_HIP_LIB_TAG=$v; fi
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -x -q -k "conv_forward or backward_data_and_wgrad or flip_free or full_batch or bf16_activation" > $out/pytest_$v.log 2>&1; echo "pytest rc $?" >> $out/pytest_$v.log
  tail -n 2 $out/pytest_$v.log
  for rep in 1 2; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_$v.json 2> $out/bench_$v.err
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k_${v}_$rep.csv \;
  rm -rf $out/prof
  echo "== variant $v rep $rep"; python3 tools/kstats.py $out/k_${v}_$rep.csv 65 | grep -E "total" | cut -c1-150
  done
done
