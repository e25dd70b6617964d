#!/bin/bash
# same-box A/B of one lab switch on the step time:  tools/lab/ab_env.sh VAR A B   (alternating runs, lab library)
var=$1; a=$2; b=$3
out=gpurun_out/r03_abenv; mkdir -p $out
export AVA_HIP_LIB_TAG=lab
for i in 1 2 3; do
  for v in $a $b; do
    env $var=$v timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/bench_${v}_$i.json 2> $out/bench_${v}_$i.err
    echo "$var=$v run $i: $(grep -o '"ms_per_step": [0-9.]*' $out/bench_${v}_$i.json)"
  done
done
