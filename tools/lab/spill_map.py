#!/usr/bin/env python3
"""spill_map.py <file.s> [name filter]: for every kernel of a hipcc -S dump, the sequence of code regions between barriers /
s_endpgm with their scratch (spill) loads+stores, MFMA and transposed-LDS-read counts -- shows WHICH wave role of a
role-split kernel spills, and whether inside its tile loop."""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:\s+; @', l)]
for si, st in enumerate(starts):
    en = starts[si + 1] if si + 1 < len(starts) else len(lines)
    name = lines[st].split(':')[0]
    if flt not in name:
        continue
    seg = lines[st:en]
    out, sc, mf, tr, vm = [], 0, 0, 0, 0
    for l in seg:
        if 'scratch_' in l: sc += 1
        elif 'v_mfma' in l: mf += 1
        elif 'ds_read_b64_tr' in l: tr += 1
        if 's_barrier' in l or 's_endpgm' in l or 's_setprio' in l:
            tag = 'B' if 's_barrier' in l else ('END' if 's_endpgm' in l else 'PRIO')
            out.append("[s%d m%d t%d]%s" % (sc, mf, tr, tag))
            sc = mf = tr = 0
    print(name[:110])
    print("   " + " ".join(out))
