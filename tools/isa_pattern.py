#!/usr/bin/env python3
"""Print the load / wait / barrier / shuffle / store sequence of kernels in a gfx950 .s file
(L global load, W s_waitcnt vmcnt, b branch, | barrier, p cross-lane op, S store, D ds_read/write)."""
import re, sys
txt = open(sys.argv[1]).read()
for k in sys.argv[2:]:
    m = re.search(r'^(\S*' + re.escape(k) + r'\S*):', txt, flags=re.M)
    if not m:
        print("not found", k); continue
    i = m.start(); j = txt.index('s_endpgm', i)
    lines = [l.strip() for l in txt[i:j].splitlines() if l.strip() and not l.strip().startswith(';')]
    seq = []
    for l in lines:
        if l.startswith(('global_load', 'buffer_load')): seq.append('L')
        elif 's_waitcnt' in l and 'vmcnt' in l: seq.append('W')
        elif l.startswith('s_cbranch'): seq.append('b')
        elif l.startswith('s_barrier'): seq.append('|')
        elif l.startswith('global_store'): seq.append('S')
        elif l.startswith(('ds_bpermute', 'ds_swizzle')) or 'dpp' in l: seq.append('p')
        elif l.startswith('ds_'): seq.append('D')
    print(m.group(1)[:80], 'ninstr', len(lines)); print(''.join(seq))
