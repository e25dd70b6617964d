out=gpurun_out/r05_overlap_p; mkdir -p $out
for rep in 1 2 3; do
  for v in "" ov; do
    AVA_HIP_LIB_TAG=$v timeout 300 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-loader-path --no-roofline --global-batch 0 > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err
    python3 -c "import json,sys; d=json.loads(open('$out/bench_${v}_$rep.json').read().strip().splitlines()[-1]); print('product build, variant \"$v\" rep $rep: %.4f ms/step  %.1f spectrograms/s' % (d['ms_per_step'], d['value']))"
  done
done
