#!/bin/bash
# isolated fused backward kernels (tools/fused_bench.py) under rocprofv3: previous build (csrc/libava_hip_prev.so) vs this build
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/r03_fusedab; mkdir -p $out
for tag in prev new; do
  if [ $tag = prev ]; then export AVA_HIP_LIB_TAG=prev; else unset AVA_HIP_LIB_TAG; fi
  rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o fb --output-format csv -- python3 tools/fused_bench.py > $out/prof_$tag.log 2>&1
  find $out/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/kstats_$tag.csv \;
  rm -rf $out/prof_$tag
done
python3 - <<'PY'
import csv
tab={}
for d in ("prev","new"):
    for r in csv.DictReader(open('gpurun_out/r03_fusedab/kstats_%s.csv'%d)):
        if 'conv' in r['Name'] or 'thin' in r['Name']: tab.setdefault(r['Name'][:100],{})[d]=float(r['AverageNs'])/1000
for k,v in sorted(tab.items(), key=lambda kv:-kv[1].get("prev",0)):
    print("%-100s %7.1f %7.1f"%(k,v.get("prev",0),v.get("new",0)))
PY
