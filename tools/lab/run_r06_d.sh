#!/bin/bash
# larger forward tiles for the <= 32 x 32 layers: kernel tests of each variant, then forward-only kernel traces
out=gpurun_out/r06_d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in base big big2; do
  if [ "$v" = base ]; then export AVA_HIP_LIB_TAG=; else export AVA_HIP_LIB_TAG=$v; fi
  timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "forward or fwd" > $out/tests_$v.log 2>&1; tail -1 $out/tests_$v.log
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o fwd --output-format csv -- python3 tools/lab/fwd_loop.py 256 30 > $out/fwd_$v.log 2>&1
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/k_$v.csv \;
  rm -rf $out/prof
  tail -1 $out/fwd_$v.log
done
python3 - <<'PY'
import csv
def load(p):
    return {r['Name']: int(r['TotalDurationNs'])/30/1e3 for r in csv.DictReader(open(p))}
a=load('gpurun_out/r06_d/k_base.csv')
for v in ('big','big2'):
    b=load('gpurun_out/r06_d/k_%s.csv'%v)
    print("== %s: total %.1f -> %.1f us per forward"%(v,sum(a.values()),sum(b.values())))
    fa={n:t for n,t in a.items() if 'conv3x3_mfma' in n}; fb={n:t for n,t in b.items() if 'conv3x3_mfma' in n}
    print("   conv3x3_mfma* kernels: %.1f -> %.1f"%(sum(fa.values()),sum(fb.values())))
    for n in sorted(fb): 
        if n not in fa: print("   new  %6.1f %s"%(fb[n],n[:110]))
    for n in sorted(fa):
        if n not in fb: print("   gone %6.1f %s"%(fa[n],n[:110]))
PY
