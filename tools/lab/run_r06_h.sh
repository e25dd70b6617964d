#!/bin/bash
# configs[4] (256 x 256, z = 128, batch 64, bf16 conv arithmetic): larger forward tiles for the stride-1 layers
out=gpurun_out/r06_h; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for i in 1 2; do
for v in base bigs1; do
  if [ "$v" = base ]; then export AVA_HIP_LIB_TAG=; else export AVA_HIP_LIB_TAG=$v; fi
  for dt in bf16 f32; do
    timeout 400 python3 bench.py --height 256 --width 256 --z-dim 128 --per-gpu-batch 64 --global-batch 0 --steps 40 --warmup 20 --dtype $dt --no-cpu-baseline --no-loader-path --no-roofline > $out/b_${v}_$dt.json 2> $out/b_${v}_$dt.err
    echo "$v $dt: $(grep -o '"ms_per_step": [0-9.]*' $out/b_${v}_$dt.json)"
  done
done
done
