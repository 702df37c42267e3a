#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv) as a short table."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {steps:g} steps = {tot / 1e6 / steps:.2f} ms/step")
print(f"{'kernel':100s} {'calls':>6s} {'ms/step':>9s} {'avg_us':>9s} {'%':>6s}")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6 / steps:9.3f} "
          f"{float(r['AverageNs']) / 1e3:9.1f} {float(r['Percentage']):6.2f}")
