#!/usr/bin/env python3
"""Find the code-generation pattern of DESIGN.md section 3, item 21 in a device assembly listing: inside loops, an exec-masked
block that reads memory and waits for it (`s_cbranch_execz ..; ds_read / global_load ..; s_waitcnt ..`) -- what
`cond ? f(memory) : 0` compiles to when the compiler may not speculate the read.  One such block per loop iteration is a
serialised round trip.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on --cuda-device-only -S csrc/conv_ws.hip -o /tmp/conv_ws.s
    python tools/lab/isa_scan.py /tmp/conv_ws.s            # kernels with the most such blocks first
"""
import re, subprocess, sys

for fn in sys.argv[1:]:
    s = open(fn).read()
    for m in re.finditer(r'^(_Z\S+):', s, re.M):
        name = m.group(1)
        i, j = m.start(), s.find('.Lfunc_end', m.start())
        if j < 0:
            continue
        lines = s[i:j].split('\n')
        inloop, cnt = False, 0
        for k, line in enumerate(lines):
            l = line.strip()
            if re.match(r'^\.LBB\d+_\d+:', l):
                inloop = 'Loop' in l
            if inloop and l.startswith('s_cbranch_execz'):
                rd = wait = False
                for t in lines[k + 1:k + 14]:
                    t = t.strip()
                    if re.match(r'^\.LBB', t):
                        break
                    if t.startswith(('ds_read', 'global_load', 'buffer_load')):
                        rd = True
                    if t.startswith('s_waitcnt') and rd:
                        wait = True
                cnt += rd and wait
        if cnt:
            dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:120]
            print("%4d conditional read-and-wait blocks in loops  %s" % (cnt, dn))
