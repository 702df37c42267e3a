#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 kernel trace (csv): per stream busy time, the main stream's idle
gaps, and (with -v) every launch.  usage: step_timeline.py <kernel_trace.csv> [-v]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"], r["Stream_Id"]) for r in rows)
idx = [i for i, x in enumerate(d) if "dice_fwd" in x[2]]
a, b = idx[-3], idx[-2]
seg = d[a:b]
t0 = seg[0][0]
print(f"step (loss forward to loss forward): {(seg[-1][1] - t0) / 1e3:.0f} us, {len(seg)} launches")
streams = {}
for x in seg:
    streams.setdefault(x[4], []).append(x)
for s, l in sorted(streams.items()):
    print(f"  stream {s}: {len(l)} launches, busy {sum(x[1] - x[0] for x in l) / 1e3:.0f} us, first at {(l[0][0] - t0) / 1e3:.0f}, "
          f"last ends at {(max(x[1] for x in l) - t0) / 1e3:.0f}")
main = max(streams.items(), key=lambda kv: len(kv[1]))[0]
l = streams[main]
ce, tot = l[0][1], 0
for x in l[1:]:
    g = x[0] - ce
    if g > 3000:
        tot += g
        if g > 15000:
            print(f"  main-stream gap {g / 1e3:.0f} us before {x[2][:60]} at {(x[0] - t0) / 1e3:.0f}")
    ce = max(ce, x[1])
print(f"  main-stream idle (gaps > 3 us): {tot / 1e3:.0f} us")
if "-v" in sys.argv:
    for x in seg:
        n = x[2].replace("void segmi::", "").replace("segmi::", "")[:50]
        print(f"{x[4]} {(x[0] - t0) / 1e3:7.0f} {(x[1] - x[0]) / 1e3:6.0f}  {n:50s} {x[3]}")
