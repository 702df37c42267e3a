#!/usr/bin/env python3
"""Per-kernel HBM bandwidth table of one training step.

usage: bw_table.py TIME_kernel_trace.csv FETCH_counter_collection.csv WRITE_counter_collection.csv

The three files come from three separate runs of `bench.py` under SEGMI_SERIAL=1 (all kernels on
one stream): a `--kernel-trace` run for durations, one `--pmc FETCH_SIZE` and one `--pmc WRITE_SIZE`
pass for HBM bytes (gfx950 correction: bytes = (2*FETCH_SIZE + WRITE_SIZE) KB, see
profiles/r01_pmc_traffic.json).  Dispatches of the last step (between the last two adam_kernel
launches) are matched across files by (kernel name, occurrence index).
"""
import csv
import sys
from collections import defaultdict


def last_step(rows, name_key, order_key):
    rows = sorted(rows, key=lambda r: int(r[order_key]))
    marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[name_key]]
    lo, hi = marks[-2], marks[-1]
    out = defaultdict(list)
    for r in rows[lo + 1: hi + 1]:
        out[r[name_key]].append(r)
    return out


t = last_step(list(csv.DictReader(open(sys.argv[1]))), "Kernel_Name", "Start_Timestamp")
f = last_step(list(csv.DictReader(open(sys.argv[2]))), "Kernel_Name", "Start_Timestamp")
w = last_step(list(csv.DictReader(open(sys.argv[3]))), "Kernel_Name", "Start_Timestamp")
agg = []
tot_t = tot_b = 0
for name, lst in t.items():
    fl, wl = f.get(name, []), w.get(name, [])
    if len(fl) != len(lst) or len(wl) != len(lst):
        print(f"# skip {name[:60]}: dispatch counts differ {len(lst)} {len(fl)} {len(wl)}")
        continue
    groups = defaultdict(lambda: [0, 0.0, 0.0])
    for a, b, c in zip(lst, fl, wl):
        dur = int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
        byt = (2 * float(b["Counter_Value"]) + float(c["Counter_Value"])) * 1024
        g = groups[(a["Grid_Size_X"], a["Grid_Size_Y"], a["Grid_Size_Z"])]
        g[0] += 1
        g[1] += dur
        g[2] += byt
    for grid, (n, dur, byt) in groups.items():
        agg.append((dur, name, grid, n, byt))
        tot_t += dur
        tot_b += byt
agg.sort(reverse=True)
print(f"step kernel time {tot_t / 1e6:.3f} ms, HBM bytes {tot_b / 1e9:.2f} GB -> {tot_b / tot_t:.0f} GB/s average")
print(f"{'kernel':70s} {'grid':>14s} {'n':>3s} {'us_tot':>8s} {'MB/launch':>10s} {'GB/s':>7s}")
for dur, name, grid, n, byt in agg[:60]:
    g = "x".join(grid)
    print(f"{name[:70]:70s} {g:>14s} {n:3d} {dur / 1e3:8.1f} {byt / n / 1e6:10.1f} {byt / dur:7.0f}")
