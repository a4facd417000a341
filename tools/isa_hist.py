#!/usr/bin/env python3
"""Opcode histogram of the hottest loop of a kernel in a hipcc -S listing.
   tools/isa_hist.py file.s <kernel-name-substring>"""
import re, sys, collections
s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(\S*%s\S*):.*?\n(.*?)\n\s*s_endpgm' % re.escape(name), s, re.S | re.M)
lines = m.group(2).split('\n')
lab = {}
for i, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm: lab[mm.group(1)] = i
best = (0, 0)
for i, l in enumerate(lines):
    mm = re.match(r'\s+s_cbranch_\w+ (\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in lab and lab[mm.group(1)] < i and i - lab[mm.group(1)] > best[1] - best[0]:
        best = (lab[mm.group(1)], i)
c = collections.Counter()
for l in lines[best[0]:best[1] + 1]:
    t = l.strip()
    if not t or t[0] in ';.' or t.endswith(':'): continue
    op = t.split()[0]
    if 'sdwa' in t or '_sel:WORD' in t or '_sel:BYTE' in t: op += '(sdwa)'
    c[op] += 1
print(m.group(1), 'loop lines', best, 'instr', sum(c.values()), 'VALU', sum(v for k, v in c.items() if k.startswith('v_')))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30): print(f"  {k:30s}{v}")
