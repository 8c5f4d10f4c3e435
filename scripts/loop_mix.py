#!/usr/bin/env python3
"""Instruction mix of the straight-line hot loop of an update kernel in a hipcc -S dump: the lines from the `Inner Loop Header` label to
the back edge that returns to it (the out-of-line renormalisation bodies sit behind it and are not counted).
usage: loop_mix.py file.s kernel-regex steps-per-loop"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
m = re.search(r'^(' + sys.argv[2] + r'):', s, re.M)
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
body = s[m.end():s.index('.Lfunc_end', m.end())].split('\n')
hdr = max((n for n, l in enumerate(body) if 'Loop Header' in l and 'Depth=1' in l), key=lambda n: 0) if False else None
best = None
for n, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):.*Loop Header: Depth=1', l)
    if not mm: continue
    for k in range(n + 1, len(body)):
        if re.search(r's_c?branch\w*\s+' + re.escape(mm.group(1)) + r'\b', body[k]):
            if best is None or k - n > best[1] - best[0]: best = (n, k)
            break
a, b = best
ins = [l.split()[0] for l in (x.strip() for x in body[a:b + 1]) if l and not l.startswith((';', '.')) and not l.endswith(':')]
c = Counter(ins)
valu = sum(v for k, v in c.items() if k.startswith('v_'))
lds = sum(v for k, v in c.items() if k.startswith('ds_'))
print(m.group(1)[:110])
print(f'loop: {len(ins)} instr, {len(ins)/steps:.1f}/step; VALU {valu/steps:.2f}/step; LDS {lds/steps:.2f}/step; SALU {sum(v for k, v in c.items() if k.startswith("s_"))/steps:.2f}/step')
for k, v in c.most_common(int(sys.argv[4]) if len(sys.argv) > 4 else 24): print(f'{v:6d} {v/steps:7.2f} {k}')
