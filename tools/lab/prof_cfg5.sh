#!/bin/bash
# kernel stats of configs[4] (256 x 256, z = 128, batch 64) in fp32 and in bf16 mode, same box
out=gpurun_out/r05_cfg5; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for d in f32 bf16; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --height 256 --width 256 --z-dim 128 --per-gpu-batch 64 --global-batch 0 --dtype $d --no-cpu-baseline --no-loader-path > $out/bench_$d.json 2> $out/bench_$d.err
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k_$d.csv \;
  rm -rf $out/prof
done
