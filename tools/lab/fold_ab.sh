#!/bin/bash
# same-box A/B of the convt7 fold (lab build, AVA_FOLD13=0|1) at 128x128 B=256 and 256x256 B=64: kernel stats of the step
out=gpurun_out/r05_foldab; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export AVA_HIP_LIB_TAG=lab
for f in 0 1; do for cfg in "128" "256"; do
  export AVA_FOLD13=$f
  if [ $cfg = 128 ]; then args="--global-batch 0"; else args="--height 256 --width 256 --z-dim 128 --per-gpu-batch 64 --global-batch 0"; fi
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path $args > $out/b${cfg}_fold$f.json 2> $out/b${cfg}_fold$f.err
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k${cfg}_fold$f.csv \;
  rm -rf $out/prof
  echo "== $cfg fold=$f"; python3 tools/kstats.py $out/k${cfg}_fold$f.csv 25 | grep -E "thin|total" | cut -c1-150
done; done
