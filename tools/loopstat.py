#!/usr/bin/env python3
"""Print instruction-mix statistics of the innermost loops of gfx950 kernels
(from hipcc -save-temps .s files).  Usage: tools_loopstat.py file.s [substr]"""
import re, sys, collections
s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ''
funcs = re.split(r'\n(?=_Z\w+:)', s)
for f in funcs:
    name = f.split(':')[0]
    if not name.startswith('_Z') or sub not in name:
        continue
    lines = f.split('\n')
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'\s+s_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    print('==', name)
    for a, b in loops:
        c = collections.Counter()
        for l in lines[a:b + 1]:
            m = re.match(r'\s+([a-z_0-9]+)', l)
            if m and not l.strip().startswith(('.', ';')):
                c[m.group(1)] += 1
        tot = sum(c.values())
        f64 = sum(v for k, v in c.items() if 'f64' in k)
        acc = sum(v for k, v in c.items() if 'accvgpr' in k)
        print(f'  loop lines {a}-{b}: {tot} instr, f64 ops {f64}, accvgpr {acc}, top: {dict(c.most_common(12))}')
