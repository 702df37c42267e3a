#!/usr/bin/env python3
"""Instruction mix of the largest loop of one kernel in a hipcc -S listing.
usage: isa_loop.py file.s mangled_kernel_name"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
name = sys.argv[2]
i0 = s.index("\n" + name + ":")
i1 = s.index("s_endpgm", i0)
lines = [l.split(";")[0].strip() for l in s[i0:i1].split("\n")]
lines = [l for l in lines if l and (not l.startswith(".") or l.startswith(".L"))]
labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
best = None
for i, l in enumerate(lines):
    m = re.match(r"s_c?branch\w*\s+(\S+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        if best is None or i - labels[m.group(1)] > best[1] - best[0]:
            best = (labels[m.group(1)], i)
lo, hi = best
print(f"{len(lines)} lines, main loop {lo}..{hi} ({hi - lo} lines)")
c = Counter()
for l in lines[lo:hi]:
    op = l.split()[0]
    if op.endswith(":"):
        continue
    if op.startswith("v_mfma"): c["MFMA"] += 1
    elif op.startswith("ds_"): c[op] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["_".join(op.split("_")[:2])] += 1
    elif op.startswith("v_"): c["VALU"] += 1; c["  " + op] += 1
    elif op.startswith("s_waitcnt"): c["s_waitcnt"] += 1
    elif op.startswith("s_barrier"): c["s_barrier"] += 1
    elif op.startswith("s_"): c["SALU"] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{v:6d} {k}")
