#!/bin/bash
# kernel durations (rocprofv3 kernel trace, not host-launch-rate bound) of the isolated conv kernels per ablation setting
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/r03_convabl; mkdir -p $out
export AVA_HIP_LIB_TAG=lab
for d in 0 15 14 6 1; do
  export AVA_DBG=$d
  rocprofv3 --kernel-trace --stats -d $out/prof_$d -o cb --output-format csv -- python3 tools/conv_bench.py > $out/prof_$d.log 2>&1
  find $out/prof_$d -name "*kernel_stats.csv" -exec cp {} $out/kstats_$d.csv \;
  rm -rf $out/prof_$d
done
python3 - <<'PY'
import csv
cols=[0,15,14,6,1]
tab={}
for d in cols:
    for r in csv.DictReader(open('gpurun_out/r03_convabl/kstats_%d.csv'%d)):
        tab.setdefault(r['Name'][:100],{})[d]=float(r['AverageNs'])/1000
print("%-100s"%"kernel"+"".join("%8s"%("d%d"%d) for d in cols))
for k,v in sorted(tab.items(), key=lambda kv:-kv[1].get(0,0)):
    print("%-100s"%k+"".join("%8.1f"%v.get(d,0) for d in cols))
PY
