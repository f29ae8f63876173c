#!/usr/bin/env python3
"""Instruction mix of a kernel's loops, from hipcc's assembly (-S --cuda-device-only).
usage: python tools/loop_mix.py <file.s> <substring of the mangled kernel name>
Prints, per back edge, (instructions, VALU, global loads, global stores, transcendentals, ds_bpermute) of the loop body."""
import re
import sys

src = open(sys.argv[1]).read()
for name in re.findall(r'^(_ZN\S*%s\S*):' % re.escape(sys.argv[2]), src, re.M):
    i = src.index('\n' + name + ':'); j = src.index('.end_amdhsa_kernel', i)
    lines = [re.sub(r'\s*;.*', '', l.strip()) for l in src[i:j].split('\n')]
    lines = [l for l in lines if l and (l.endswith(':') or not l.startswith(('.', ';')))]
    lab = {l[:-1]: k for k, l in enumerate(lines) if l.endswith(':')}
    out = set()
    for k, l in enumerate(lines):
        mm = re.match(r's_c?branch\w* (\S+)', l)
        if mm and mm.group(1) in lab and lab[mm.group(1)] < k:
            seg = [x for x in lines[lab[mm.group(1)]:k] if not x.endswith(':')]
            out.add((len(seg), sum(x.startswith('v_') for x in seg), sum(x.startswith('global_load') for x in seg),
                     sum(x.startswith('global_store') for x in seg), sum(bool(re.match(r'v_(exp|log|rcp|rsq|sqrt)', x)) for x in seg),
                     sum('bpermute' in x for x in seg)))
    print(name[-40:], sorted(out))
