#!/usr/bin/env python3
"""Instruction mix of the fast block of lds2_update_kernel<15,0> in a hipcc -S dump of vit_hip.hip: the code between two
s_barrier that holds the 64 table reads and the eight 16-byte metric stores.  usage: k15_block_stats.py build/vit_hip.s [K [RT]]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
K = sys.argv[2] if len(sys.argv) > 2 else "15"
RT = sys.argv[3] if len(sys.argv) > 3 else ("6" if K == "15" else "0")
m = re.search(r'^(_ZN3vit\d+lds2_update_kernel(?:_c120)?ILi' + K + r'ELi0ELi' + RT + r'EEEvNS_14Lds2UpdateArgsE):', s, re.M)
body = s[m.end():s.index('.Lfunc_end', m.end())]
segs, seg = [], []
for l in body.split('\n') + ['s_barrier']:
    if 's_barrier' in l:
        segs.append(seg); seg = []
    else:
        seg.append(l)
for seg in segs:
    ins = [x.strip() for x in seg if x.strip() and not x.strip().startswith((';', '.')) and not x.strip().endswith(':')]
    if sum('ds_write_b128' in x for x in ins) == 8 and sum(x.startswith(('ds_read_b64', 'ds_read2_b64')) for x in ins) >= 30:
        c = Counter(x.split()[0] for x in ins)
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        print(f'fast block: {len(ins)} instructions, VALU {valu} ({valu / 8:.1f} per group-step), s_waitcnt {c["s_waitcnt"]}, s_nop {c["s_nop"]}, '
              f'LDS {sum(v for k, v in c.items() if k.startswith("ds_"))}, scratch {sum(v for k, v in c.items() if k.startswith("scratch_"))}')
        for k, v in c.most_common(28): print(f'{v:5d} {v / 8:7.2f} {k}')
        open('/tmp/fast.s', 'w').write('\n'.join(seg))
