#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin) one kernel per line."""
import re, sys, subprocess
rows = []
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        rows.append([m.group(1), {}]); continue
    for k, short in (('VGPRs', 'v'), ('SGPRs', 's'), (r'ScratchSize \[bytes/lane\]', 'scr'),
                     (r'Occupancy \[waves/SIMD\]', 'occ'), (r'LDS Size \[bytes/block\]', 'lds')):
        m = re.search(r'remark:\s+' + k + r': (\d+)', line)
        if m and rows:
            rows[-1][1][short] = int(m.group(1))
seen = set()
for n, d in rows:
    if n in seen: continue
    seen.add(n)
    dn = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    dn = dn.replace('void ', '').split('(')[0]
    print(dn[:64].ljust(64), ' '.join('%s=%d' % kv for kv in d.items()))
