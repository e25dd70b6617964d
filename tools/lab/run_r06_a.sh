#!/bin/bash
# round 6, pass A: the pinned-broadcast build -- immunity of the product forward beside the limb GEMM, tests, step A/B against the
# previous library (csrc/libava_hip_prev.so) with per-kernel traces
out=gpurun_out/r06_a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R=tools/lab/two_proc_fold
{
timeout 300 $R streams 2000 8 0 gemm
timeout 300 $R pair 600 64 0 gemm
timeout 300 $R lockstep 600 64 1
timeout 300 $R mixed 2000 8 0
} > $out/fold_immunity.log 2>&1
grep -v amdgpu.ids $out/fold_immunity.log | grep -v "partial rows" | grep "^mode"
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_autograd_semantics.py -x -q -m gpu -k "flip_free or full_batch or same_path or golden or scale" > $out/tests.log 2>&1; tail -3 $out/tests.log
for v in prev base prev base; do
  if [ "$v" = base ]; then export AVA_HIP_LIB_TAG=; else export AVA_HIP_LIB_TAG=$v; fi
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_$v.json 2> $out/bench_$v.err
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k_$v.csv \;
  rm -rf $out/prof
  grep -o '"ms_per_step": [0-9.]*' $out/bench_$v.json
done
python3 tools/ab_diff.py $out/k_prev.csv $out/k_base.csv
