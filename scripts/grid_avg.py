#!/usr/bin/env python3
"""Average duration per (kernel, grid) over a whole kernel_trace.csv. usage: grid_avg.py trace.csv [substr]"""
import csv, sys
from collections import defaultdict
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if sub not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"][:70], int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))
    agg[k][0] += 1
    agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for (n, gx, gy), (c, t) in sorted(agg.items()):
    print(f"{n:70s} {gx:>9d}x{gy:<3d} n={c:5d} avg {t / c / 1e3:8.1f} us  total {t / 1e6:8.2f} ms")
