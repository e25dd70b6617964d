#!/bin/bash
# limb MFMA also for the forward layers with 8 input channels (lab: AVA_CONV_LIMB=2): isolated kernels, step time, flip-sensitive tests
out=gpurun_out/r03_limb8; mkdir -p $out
export AVA_HIP_LIB_TAG=lab
for v in 1 2; do
  AVA_CONV_LIMB=$v timeout 300 python tools/conv_bench.py 2>&1 | grep "fwd" > $out/conv_$v.log
done
paste <(cut -c1-40 $out/conv_1.log) <(cut -c28-40 $out/conv_2.log)
for i in 1 2; do for v in 1 2; do
  AVA_CONV_LIMB=$v timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_${v}_$i.json 2> $out/bench_${v}_$i.err
  echo "AVA_CONV_LIMB=$v run $i: $(grep -o '"ms_per_step": [0-9.]*' $out/bench_${v}_$i.json)"
done; done
AVA_CONV_LIMB=2 timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_callers.py -q > $out/pytest_2.log 2>&1
echo "AVA_CONV_LIMB=2: $(tail -n 1 $out/pytest_2.log)"; grep -E "^FAILED" $out/pytest_2.log | head -5
