#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S dump.  usage: isa_stats.py file.s kernel-substring [steps-per-loop]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
sub = sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
for m in re.finditer(r'^(\S*' + re.escape(sub) + r'\S*):.*$', s, re.M):
    name = m.group(1)
    if name.startswith('.'): continue
    i = m.end(); j = s.index('.Lfunc_end', i)
    body = s[i:j]
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
    c = Counter(l.split()[0] for l in lines)
    print(name[:120]); print(' total', len(lines), ' per step', round(len(lines) / steps, 1))
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print(' VALU', valu, 'per step', round(valu / steps, 1), ' SALU', sum(v for k, v in c.items() if k.startswith('s_')))
    for k, v in c.most_common(36): print(f"   {v:6d} {k}")
    open('/tmp/kernel_body.s', 'w').write(body)
    break
