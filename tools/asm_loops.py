#!/usr/bin/env python3
"""Loops of a kernel in a device assembly file (hipcc -S --offload-device-only): length, vector-ALU
instructions, scratch accesses inside each.   python tools/asm_loops.py file.s <mangled-name-substring>"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for f in re.split(r'\n(?=_Z\w+:)', s):
    m = re.match(r'(_Z\w+):', f)
    if not m or want not in m.group(1):
        continue
    lines = f.split('\n')
    sc = [i for i, l in enumerate(lines) if 'scratch_' in l]
    print(m.group(1)[:70], 'lines', len(lines), 'scratch instr', len(sc))
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r'(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    for i, l in enumerate(lines):
        mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            a = labels[mm.group(1)]
            n = sum(1 for k in sc if a <= k <= i)
            v = sum(1 for x in lines[a:i] if re.match(r'\s+v_', x))
            dpp = sum(1 for x in lines[a:i] if 'dpp' in x)
            print('  loop', a, i, 'len', i - a, 'valu', v, 'dpp', dpp, 'scratch in loop', n)
