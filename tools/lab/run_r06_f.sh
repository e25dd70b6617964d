#!/bin/bash
# Adam of the two large buckets on a side stream under the encoder's backward: same-box A/B (AVA_OVERLAP_ADAM=0|1), tests, traces
out=gpurun_out/r06_f; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_autograd_semantics.py tests/test_gpu_callers.py -x -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log
for i in 1 2 3; do
  for v in 0 1; do
    AVA_OVERLAP_ADAM=$v timeout 600 python bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-loader-path --global-batch 0 --no-roofline > $out/bench_$v.$i.json 2> $out/bench_$v.$i.err
    echo "overlap=$v: $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$v.$i.json)"
  done
done
for v in 0 1; do
  AVA_OVERLAP_ADAM=$v timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path --global-batch 0 --no-roofline > $out/prof_$v.json 2> $out/prof_$v.err
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k_$v.csv \;
  rm -rf $out/prof
done
python3 tools/ab_diff.py $out/k_0.csv $out/k_1.csv
