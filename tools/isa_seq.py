#!/usr/bin/env python3
"""MFMA-kernel view of a gfx950 .s file: L global load, W wait incl. vmcnt, w lgkmcnt-only wait, M mfma,
r ds_read, d ds_write, | barrier, S global store, b branch.   usage: isa_seq.py file.s mangled-substring..."""
import re, sys
txt = open(sys.argv[1]).read()
for k in sys.argv[2:]:
    m = re.search(r'^(\S*' + re.escape(k) + r'\S*):', txt, flags=re.M)
    if not m:
        print("not found", k); continue
    i = m.start(); j = txt.index('s_endpgm', i)
    seq = []
    for l in txt[i:j].splitlines():
        l = l.strip()
        if l.startswith(('global_load', 'buffer_load')): seq.append('L')
        elif l.startswith('s_waitcnt'): seq.append('w' if 'lgkmcnt' in l and 'vmcnt' not in l else 'W')
        elif l.startswith('v_mfma'): seq.append('M')
        elif l.startswith('ds_read'): seq.append('r')
        elif l.startswith('ds_write'): seq.append('d')
        elif l.startswith('s_barrier'): seq.append('|')
        elif l.startswith('global_store'): seq.append('S')
        elif l.startswith('s_cbranch'): seq.append('b')
    print(m.group(1)[:90]); print(''.join(seq))
