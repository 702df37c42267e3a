#!/usr/bin/env python3
"""Instruction mix of the largest loop of a kernel in an assembly listing.  usage: isa_loop_mix.py file.s key [key...]"""
import collections, re, sys
s = open(sys.argv[1]).read()
for key in sys.argv[2:]:
    name = [m.group(1) for m in re.finditer(r"\n(_Z\w+):", s) if key in m.group(1)][0]
    i0 = s.index("\n" + name + ":"); i1 = s.index("s_endpgm", i0)
    lines = [l.split(";")[0].strip() for l in s[i0:i1].split("\n")]
    lines = [l for l in lines if l]
    pos = {l[:-1]: i for i, l in enumerate(lines) if l.startswith(".LBB") and l.endswith(":")}
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\S+)", l)
        if m and m.group(1) in pos and pos[m.group(1)] < i:
            loops.append((pos[m.group(1)], i))
    a, b = max(loops, key=lambda t: t[1] - t[0])
    body = [l for l in lines[a:b + 1] if not l.startswith(".")]
    c = collections.Counter(l.split()[0] for l in body)
    mf = sum(n for k, n in c.items() if k.startswith("v_mfma"))
    cnt = lambda pre: sum(n for k, n in c.items() if k.startswith(pre))
    print(f"{key}: {len(body)} instr; VALU {cnt('v_') - mf} MFMA {mf} SALU {cnt('s_')} DS {cnt('ds_')} VMEM {cnt('global_') + cnt('buffer_')}")
    print("   scalar:", [(k, n) for k, n in c.most_common(60) if k.startswith("s_")][:12])
