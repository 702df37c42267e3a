#!/usr/bin/env python3
"""VGPR / spill summary per kernel of a hipcc -S listing. usage: isa_regs.py file.s [substr]"""
import re, sys
s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", s, re.S):
    name, body = m.group(1), m.group(2)
    if sub not in name:
        continue
    g = lambda k: re.search(k + r":\s+(\d+)", body)
    v, sp, ag = g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"), g(r"\.agpr_count")
    print(f"{name[:90]:90s} vgpr {v.group(1) if v else '?':>4s} agpr {ag.group(1) if ag else '?':>4s} spill {sp.group(1) if sp else '?'}")
