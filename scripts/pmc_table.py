#!/usr/bin/env python3
"""Per-kernel PMC table from rocprofv3 counter_collection.csv (last dispatch of each kernel x grid)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.OrderedDict()
for r in rows:
    key = (r["Kernel_Name"][:70], r["Grid_Size"])
    d.setdefault(key, collections.OrderedDict())[r["Counter_Name"]] = float(r["Counter_Value"])
for (k, g), c in d.items():
    if filt and filt not in k:
        continue
    print(f"{k:70s} grid={g:>9s} " + " ".join(f"{n}={v:.4g}" for n, v in c.items()))
