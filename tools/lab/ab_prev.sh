#!/bin/bash
# same-box A/B of the product library against a copy of the previous build (csrc/libava_hip_prev.so): gradients of a fixed
# step (expected bit-identical for a scheduling-only change), isolated conv kernels, step time
out=gpurun_out/r03_abprev; mkdir -p $out
for B in 8 256; do
  AB_B=$B AVA_HIP_LIB_TAG=prev timeout 300 python tools/ab_grads.py dump /tmp/a$B.npz > /dev/null 2>&1
  AB_B=$B timeout 300 python tools/ab_grads.py dump /tmp/b$B.npz > /dev/null 2>&1
  echo "B=$B prev vs new" >> $out/ab.log; python tools/ab_grads.py diff /tmp/a$B.npz /tmp/b$B.npz 2>&1 | head -n 4 >> $out/ab.log
done
cat $out/ab.log
AVA_HIP_LIB_TAG=prev timeout 300 python tools/conv_bench.py > $out/conv_prev.log 2>&1
timeout 300 python tools/conv_bench.py > $out/conv_new.log 2>&1
paste <(cut -c1-40 $out/conv_prev.log) <(cut -c28-40 $out/conv_new.log) | head -n 46
for i in 1 2; do
  AVA_HIP_LIB_TAG=prev timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_prev$i.json 2> $out/bench_prev$i.err
  timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_new$i.json 2> $out/bench_new$i.err
done
grep -o '"ms_per_step": [0-9.]*' $out/bench_prev1.json $out/bench_new1.json $out/bench_prev2.json $out/bench_new2.json
