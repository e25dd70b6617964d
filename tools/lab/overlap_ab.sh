#!/bin/bash
# overlap_ab.sh [tag]: same-box A/B of the lab switch AVA_OVERLAP_DW (fc8's / fc1's weight gradients on the model's side stream beside
# the launches that follow them): step time of bench.py for 0 / 1 / 2 / 3, alternating, then the kernel trace of 0 and 3
tag=${1:-}
out=gpurun_out/r05_overlap$tag; mkdir -p $out
export AVA_HIP_LIB_TAG=lab
for rep in 1 2 3; do
  for v in 0 1 2 3; do
    AVA_OVERLAP_DW=$v timeout 300 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-loader-path --no-roofline --global-batch 0 > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err
    python3 -c "import json,sys; d=json.loads(open('$out/bench_${v}_$rep.json').read().strip().splitlines()[-1]); print('overlap $v rep $rep: %.4f ms/step  %.1f spectrograms/s' % (d['ms_per_step'], d['value']))"
  done
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in 0 3; do
  export AVA_OVERLAP_DW=$v
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path --global-batch 0 > $out/prof_$v.json 2> $out/prof_$v.err
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k_$v.csv \;
  rm -rf $out/prof
  echo "== kernel trace, overlap $v"; python3 tools/kstats.py $out/k_$v.csv 65 | grep -E "gemm_limb|skinny|fc_mid|24, 32, 0, 2|24, 24, 1, 1, 16|relu_mask|grouped|total" | cut -c1-150
done
