#!/bin/bash
# isolated fused backward kernels per tile variant (lab build, AVA_FUSED_VAR), rocprofv3 kernel durations
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/r03_fusedvar; mkdir -p $out
export AVA_HIP_LIB_TAG=lab
for v in 0 1 2; do
  export AVA_FUSED_VAR=$v
  rocprofv3 --kernel-trace --stats -d $out/prof_$v -o fb --output-format csv -- python3 tools/fused_bench.py > $out/prof_$v.log 2>&1
  find $out/prof_$v -name "*kernel_stats.csv" -exec cp {} $out/kstats_$v.csv \;
  rm -rf $out/prof_$v
done
python3 - <<'PY'
import csv
for v in (0,1,2):
    print("== AVA_FUSED_VAR=%d"%v)
    for r in sorted(csv.DictReader(open('gpurun_out/r03_fusedvar/kstats_%d.csv'%v)), key=lambda r:r['Name']):
        if 'bwd_fused_ws' in r['Name']: print("  %7.1f  %s"%(float(r['AverageNs'])/1000, r['Name'][:95]))
PY
