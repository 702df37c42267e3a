#!/usr/bin/env python3
"""Top dispatches by duration with their PMC counters (rocprofv3 counter_collection.csv)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
d = collections.OrderedDict()
for r in rows:
    k = r["Dispatch_Id"]
    e = d.setdefault(k, {"name": r["Kernel_Name"][:58], "grid": r["Grid_Size"], "vgpr": r["VGPR_Count"],
                         "dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "c": {}})
    e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
top = sorted(d.values(), key=lambda e: -e["dur"])[:n]
for e in top:
    c = e["c"]
    waves = c.get("SQ_WAVES", 0) or 1
    extra = ""
    if "SQ_INSTS_VALU" in c:
        extra = (f" valu/wave={c['SQ_INSTS_VALU'] / waves:.0f} mfma/wave={c.get('SQ_INSTS_MFMA', 0) / waves:.0f}"
                 f" wait%={100 * c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.0f}"
                 f" issuestall%={100 * c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.0f}"
                 f" active%={100 * c.get('SQ_ACTIVE_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.0f}"
                 f" waves/CU={c.get('SQ_WAVE_CYCLES', 0) * 4 / (e['dur'] * 1e-6 * 2.4e9 * 256):.1f}")
    print(f"{e['name']:58s} grid={e['grid']:>8s} vgpr={e['vgpr']:>3s} {e['dur']:8.1f}us{extra}")
