#!/usr/bin/env python3
"""Instruction mix of the largest loop of one kernel in a hipcc -S dump, priced with the measured gfx950 issue costs
(profiles/*valu_rate*): usage: loop_stats.py file.s kernel-regex steps-per-loop"""
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
blk = [l.strip() for l in lines[a:b + 1] if l.strip() and not l.strip().startswith((';', '.'))]
c = Counter(l.split()[0] for l in blk)
FAST = ('v_lshrrev_b32_e32', 'v_mov_b32_e32', 'v_lshlrev_b32_e32', 'v_or_b32_e32', 'v_and_b32_e32', 'v_xor_b32_e32',
        'v_add_u32_e32', 'v_sub_u32_e32', 'v_ashrrev_i32_e32', 'v_add_u16_e32', 'v_sub_u16_e32', 'v_min_u16_e32', 'v_max_i16_e32')
valu = sum(v for k, v in c.items() if k.startswith('v_'))
cyc = sum(v * (8.37 if 'permlane' in k else 2.5 if k in FAST else 4.33) for k, v in c.items() if k.startswith('v_'))
print(m.group(1)[:100]); print(f'loop lines {a}-{b}: {len(blk)} instr, {len(blk)/steps:.1f}/step; VALU {valu/steps:.1f}/step; priced {cyc/steps:.0f} cycles/step (static, includes rarely-taken paths)')
for k, v in c.most_common(30): print(f'{v:6d} {v/steps:7.2f} {k}')
open('/tmp/loop.s', 'w').write('\n'.join(lines[a:b + 1]))
