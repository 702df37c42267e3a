#!/usr/bin/env python3
"""Per (kernel, grid) duration table of the last step in a kernel_trace.csv. usage: grid_table.py trace.csv [substr]"""
import csv, sys
from collections import defaultdict
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
# a step ends with the optimiser launch over the bulk of the arena; the carried suffix update (round 3: a
# second, small launch of the same kernel early in the next step) is not a step boundary
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
big = max(int(rows[i]["Grid_Size_X"]) for i in adam)
marks = [i for i in adam if int(rows[i]["Grid_Size_X"]) == big]
seg = rows[marks[-2] + 1: marks[-1] + 1]
agg = defaultdict(lambda: [0, 0])
for r in seg:
    if sub not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"][:64], r["Grid_Size_X"], r["Grid_Size_Y"])
    agg[k][0] += 1
    agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = 0
for (n, gx, gy), (c, t) in sorted(agg.items()):
    print(f"{n:64s} {gx:>9s}x{gy:<4s} n={c:2d} {t / 1e3:9.1f} us")
    tot += t
print(f"total {tot / 1e3:.1f} us")
