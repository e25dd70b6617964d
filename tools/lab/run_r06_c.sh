#!/bin/bash
# fold kernel on centred inputs: step tests, then forward-only kernel traces of the variants
out=gpurun_out/r06_c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_autograd_semantics.py -x -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log
for v in base wps2 prev base wps2 prev; do
  if [ "$v" = base ]; then export AVA_HIP_LIB_TAG=; else export AVA_HIP_LIB_TAG=$v; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o fwd --output-format csv -- python3 tools/lab/fwd_loop.py 256 30 > $out/fwd_$v.log 2>&1
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k_$v.csv \;
  rm -rf $out/prof
  echo "== $v: $(grep -h 'fold_kernel' $out/k_$v.csv | cut -d, -f1-4 | cut -c1-120)"
done
