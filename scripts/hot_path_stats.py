#!/usr/bin/env python3
"""Instruction mix of the HOT path of a kernel's largest loop in a hipcc -S dump: the loop body is walked as the hardware
walks it when no frame renormalises -- `s_cbranch_vccz LABEL` (the wave-uniform "nobody needs it" branch) is TAKEN,
`s_cbranch_vccnz LABEL` (its other polarity) falls through, `s_branch` is followed -- until the back edge.  usage: hot_path_stats.py file.s kernel-regex steps-per-loop"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
m = re.search(r'^(' + sys.argv[2] + r'):', s, re.M)
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
i = m.end(); j = s.index('.Lfunc_end', i); lines = s[i:j].split('\n')
labels = {}
for n, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm: labels[mm.group(1)] = n
loops = []
for n, l in enumerate(lines):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < n: loops.append((labels[mm.group(1)], n))
a, b = max(loops, key=lambda x: x[1] - x[0])
hot, n, guard = [], a, 0
while guard < 200000:
    guard += 1
    l = lines[n].strip()
    if l and not l.startswith((';', '.')) and not l.endswith(':'):
        hot.append(l)
        mm = re.match(r'(s_cbranch_vccz|s_branch)\s+(\.LBB\d+_\d+)', l)
        if mm:
            t = labels[mm.group(2)]
            if t == a: break          # the back edge
            n = t; continue
    if n == b: break
    n += 1
c = Counter(l.split()[0] for l in hot)
valu = sum(v for k, v in c.items() if k.startswith('v_'))
if __name__ == '__main__':
    print(m.group(1)[:100]); print(f'hot path: {len(hot)} instr, {len(hot)/steps:.1f}/step; VALU {valu/steps:.2f}/step; SALU {sum(v for k, v in c.items() if k.startswith("s_"))/steps:.2f}/step')
    for k, v in c.most_common(40): print(f'{v:6d} {v/steps:7.2f} {k}')
    open('/tmp/hot.s', 'w').write('\n'.join(hot))
