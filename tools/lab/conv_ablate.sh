#!/bin/bash
# phase ablation of the wave-specialised forward / data-gradient conv kernels (lab build, AVA_DBG bits; timing only)
out=gpurun_out/r03_convabl; mkdir -p $out
for d in 0 1 2 4 8 6 14 15; do
  AVA_HIP_LIB_TAG=lab AVA_DBG=$d timeout 300 python tools/conv_bench.py > $out/dbg_$d.log 2>&1
done
python3 - <<'PY'
import re,glob
cols=[0,1,2,4,8,6,14,15]
tab={}
for d in cols:
    for ln in open('gpurun_out/r03_convabl/dbg_%d.log'%d):
        m=re.match(r'(\S+)\s+(fwd|bwd)\s+(\S+\s*\S+ @\s*\d+)\s+([\d.]+) us',ln)
        if m: tab.setdefault((m.group(1),m.group(2),m.group(3)),{})[d]=float(m.group(4))
print("%-8s %-4s %-14s"%("layer","","shape")+"".join("%8s"%("d%d"%d) for d in cols))
for k,v in tab.items():
    print("%-8s %-4s %-14s"%k+"".join("%8.1f"%v.get(d,0) for d in cols))
PY
