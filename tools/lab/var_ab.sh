#!/bin/bash
# var_ab.sh <out-tag> <variant tags...>: same-box A/B of library variants (tools/lab/build_variant.sh; "base" = the product library):
# flip-free / full-batch step tests of each, then the step under the kernel trace (fused backward kernels + total)
tag=$1; shift
out=gpurun_out/r05_var_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "$@"; do
  if [ "$v" = base ]; then export AVA_HIP_LIB_TAG=; else export AVA_HIP_LIB_TAG=$v; fi
  timeout 900 python -m pytest tests/test_gpu_step.py -x -q -k "flip_free or full_batch" > $out/pytest_$v.log 2>&1; echo "pytest rc $?" >> $out/pytest_$v.log
  tail -n 2 $out/pytest_$v.log
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_$v.json 2> $out/bench_$v.err
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k_$v.csv \;
  rm -rf $out/prof
  echo "== variant $v"; python3 tools/kstats.py $out/k_$v.csv 25 | grep -E "${VAR_GREP:-bwd_fused_limb|total}" | cut -c1-150
done
