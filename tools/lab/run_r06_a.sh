#!/bin/bash
# round 6, pass A: the pinned-broadcast build -- immunity of the product forward beside the limb GEMM, tests, step A/B
out=gpurun_out/r06_a; mkdir -p $out
R=tools/lab/two_proc_fold
{
timeout 300 $R streams 2000 8 0 gemm
timeout 300 $R pair 600 64 0 gemm
timeout 300 $R lockstep 600 64 1
timeout 300 $R mixed 2000 8 0
} > $out/fold_immunity.log 2>&1
grep -v amdgpu.ids $out/fold_immunity.log | grep -v "partial rows" | tail -20
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_autograd_semantics.py -x -q -m gpu > $out/tests.log 2>&1; tail -5 $out/tests.log
for i in 1 2; do
  AVA_HIP_LIB_TAG=prev timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_prev$i.json 2> $out/bench_prev$i.err
  timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_new$i.json 2> $out/bench_new$i.err
done
grep -o '"ms_per_step": [0-9.]*' $out/bench_prev1.json $out/bench_new1.json $out/bench_prev2.json $out/bench_new2.json
AVA_HIP_LIB_TAG=prev timeout 300 python tools/conv_bench.py > $out/conv_prev.log 2>&1
timeout 300 python tools/conv_bench.py > $out/conv_new.log 2>&1
paste <(cut -c1-40 $out/conv_prev.log) <(cut -c28-40 $out/conv_new.log) | head -n 46
