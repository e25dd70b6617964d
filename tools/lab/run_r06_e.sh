#!/bin/bash
# conv2's backward with the conv1 correlations: ablation (timing only) under the step trace
out=gpurun_out/r06_e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in prev base ccut1 ccut2; do
  if [ "$v" = base ]; then export AVA_HIP_LIB_TAG=; else export AVA_HIP_LIB_TAG=$v; fi
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loader-path --global-batch 0 --no-roofline > $out/bench_$v.json 2> $out/bench_$v.err
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k_$v.csv \;
  rm -rf $out/prof
  echo "== $v: $(grep -o '"ms_per_step": [0-9.]*' $out/bench_$v.json)"
  grep -h "bwd_fused_limb_kernel<8, 8, 1,\|thin_bwd_fused_1to8" $out/k_$v.csv | awk -F'","|",' '{print $1}' | cut -c1-10 > /dev/null
  python3 - $out/k_$v.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=[float(r['Calls']) for r in rows if r['Name'].startswith('adam_flat')][0]
for r in rows:
    if 'bwd_fused_limb_kernel<8, 8, 1,' in r['Name'] or 'thin_bwd_fused_1to8' in r['Name']:
        print("   %7.1f us  %s" % (int(r['TotalDurationNs'])/steps/1e3, r['Name'][:100]))
PY
done
