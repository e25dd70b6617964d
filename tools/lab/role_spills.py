#!/usr/bin/env python3
"""role_spills.py <file.s>: registers and scratch (spill) operations INSIDE the tile loops (between the first and the last
barrier) of every fl_role<...> function of a hipcc -S dump of conv_fused_limb.hip with tools/lab/noinline_roles.patch applied
(the rejected noinline-roles experiment; fp32 activations, PRO_BWD)."""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z7fl_role\w+:\s+; @', l)]
for si, st in enumerate(starts):
    en = starts[si + 1] if si + 1 < len(starts) else len(lines)
    seg = []
    for l in lines[st:en]:
        seg.append(l)
        if l.startswith('.Lfunc_end'):
            break
    name = lines[st].split(':')[0]
    m = re.match(r'_Z7fl_roleILi(\d)ELi(\d+)ELi(\d+)ELi(\d)ELi(\d)ELi(\d+)ELi(\d+)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E(\w)Lb(\d)ELb(\d)EEv', name)
    if not m or m.group(12) != 'f' or m.group(5) != '1':
        continue
    regs, sc = [], 0
    for l in seg:
        if 'scratch_' in l:
            sc += 1
        if 's_barrier' in l:
            regs.append(sc)
            sc = 0
    regs.append(sc)
    nv = [l for l in lines[st:en] if '; NumVgprs:' in l]
    print("ci %s co %s mode %s tile %sx%s waves %s+%s+%s role %s  %s  in-loop scratch ops %d" % (
        m.group(2), m.group(3), m.group(4), m.group(6), m.group(7), m.group(8), m.group(9), m.group(10), m.group(1),
        nv[0].strip() if nv else '', sum(regs[1:-1])))
