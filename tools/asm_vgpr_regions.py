#!/usr/bin/env python3
"""Highest vector register named inside each loop of a kernel (device assembly from hipcc -S --offload-device-only):
   python tools/asm_vgpr_regions.py file.s <mangled-name-substring>"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for f in re.split(r'\n(?=_Z\w+:)', s):
    m = re.match(r'(_Z\w+):', f)
    if not m or want not in m.group(1):
        continue
    lines = f.split('\n')
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r'(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    print(m.group(1)[:60])
    for i, l in enumerate(lines):
        mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            a = labels[mm.group(1)]
            hi = 0
            for x in lines[a:i]:
                for r in re.findall(r'\bv(\d+)\b', x):
                    hi = max(hi, int(r))
                for r in re.findall(r'\bv\[(\d+):(\d+)\]', x):
                    hi = max(hi, int(r[1]))
            print('  loop', a, i, 'len', i - a, 'highest v', hi)
