#!/usr/bin/env python3
"""Per-basic-block VALU counts of a kernel in a hipcc -S listing (where does the issue time go?).
   tools/isa_blocks.py file.s <kernel-name-substring>"""
import re, sys, collections
s = open(sys.argv[1]).read()
m = re.search(r'^(\S*%s\S*):.*?\n(.*?)\n\s*s_endpgm' % re.escape(sys.argv[2]), s, re.S | re.M)
lines = m.group(2).split('\n')
def flush(name, i0, i1, c):
    v = sum(n for k, n in c.items() if k.startswith('v_'))
    if sum(c.values()):
        print(f"{name:10s} {i0:5d}-{i1:5d} VALU {v:4d} ", ' '.join(f"{k.replace('_e32','').replace('_e64','')}:{n}" for k, n in c.most_common(7)))
name, i0, cnt, tot = 'entry', 0, collections.Counter(), 0
for i, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        flush(name, i0, i, cnt); name, i0, cnt = mm.group(1), i, collections.Counter(); continue
    t = l.strip()
    if not t or t[0] in ';.': continue
    cnt[t.split()[0]] += 1
flush(name, i0, len(lines), cnt)
