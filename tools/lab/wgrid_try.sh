#!/bin/bash
# fewer workgroups (= partial rows) in the 16 x 16 layers' weight-gradient kernels: step time and the flip-sensitive tests (lab build)
out=gpurun_out/r03_wgrid; mkdir -p $out
export AVA_HIP_LIB_TAG=lab
for g in 512 384 256 128; do
  AVA_WGRID=$g timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_$g.json 2> $out/bench_$g.err
  echo "AVA_WGRID=$g: $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$g.json)"
done
for g in 256 128; do
  AVA_WGRID=$g timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_callers.py -q > $out/pytest_$g.log 2>&1
  echo "AVA_WGRID=$g: $(tail -n 1 $out/pytest_$g.log)"; grep -E "^FAILED" $out/pytest_$g.log | head -5
done
