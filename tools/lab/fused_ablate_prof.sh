#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of the isolated fused backward kernels per ablation setting (lab build, AVA_FDBG)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/r03_fusedabl; mkdir -p $out
export AVA_HIP_LIB_TAG=lab
cols="0 1 2 3 4 8 12 15 16 32 63"
for d in $cols; do
  export AVA_FDBG=$d
  rocprofv3 --kernel-trace --stats -d $out/prof_$d -o fb --output-format csv -- python3 tools/fused_bench.py > $out/prof_$d.log 2>&1
  find $out/prof_$d -name "*kernel_stats.csv" -exec cp {} $out/kstats_$d.csv \;
  rm -rf $out/prof_$d
done
COLS="$cols" python3 - <<'PY'
import csv, os
cols=[int(c) for c in os.environ["COLS"].split()]
tab={}
for d in cols:
    for r in csv.DictReader(open('gpurun_out/r03_fusedabl/kstats_%d.csv'%d)):
        if 'fused' in r['Name']: tab.setdefault(r['Name'][:92],{})[d]=float(r['AverageNs'])/1000
print("%-92s"%"kernel"+"".join("%7s"%("d%d"%d) for d in cols))
for k,v in sorted(tab.items(), key=lambda kv:-kv[1].get(0,0)):
    print("%-92s"%k+"".join("%7.1f"%v.get(d,0) for d in cols))
PY
