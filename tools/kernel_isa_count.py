#!/usr/bin/env python3
"""Instruction counts of one kernel in a --save-temps device assembly: kernel_isa_count.py <file.s> <kernel-name-substring>"""
import collections, re, sys
t = open(sys.argv[1]).read()
m = re.search(r'^(_Z\S*%s\S*):' % re.escape(sys.argv[2]), t, re.M)
j = t.index('.amdhsa_kernel ' + m.group(1))
body = t[m.end():j]
c = collections.Counter(l.split()[0] for l in body.splitlines() if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';')))
tot = sum(c.values())
cls = lambda p: sum(n for k, n in c.items() if k.startswith(p))
print('total', tot, 'valu', cls('v_'), 'salu', cls('s_'), 'lds', cls('ds_'), 'vmem', cls('global_') + cls('buffer_') + cls('flat_'))
g = re.search(re.escape(m.group(1)) + r'\.num_vgpr, (\d+)', t)
print('vgpr', g.group(1) if g else '?')
for k, n in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 20):
    print(' ', k, n)
