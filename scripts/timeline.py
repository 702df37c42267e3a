#!/usr/bin/env python3
"""Per-queue timeline summary of a rocprofv3 kernel_trace.csv: for the last `steps` steps
(delimited by a marker kernel, default adam_kernel) print per-queue busy time, idle gaps and the
kernels on the main queue ordered by total time, with the time they ran ALONE (no other queue busy).

usage: timeline.py kernel_trace.csv [marker_substring]
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows]
ev.sort()
marks = [i for i, e in enumerate(ev) if marker in e[3]]
if len(marks) < 3:
    sys.exit("not enough marker kernels")
lo, hi = marks[-3], marks[-1]           # two full steps
seg = ev[lo + 1: hi + 1]
t0, t1 = ev[lo][1], ev[hi][1]
nsteps = 2
print(f"window {(t1 - t0) / 1e6:.3f} ms for {nsteps} steps = {(t1 - t0) / 1e6 / nsteps:.3f} ms/step, {len(seg)} dispatches")
byq = defaultdict(list)
for s, e, q, n in seg:
    byq[q].append((s, e, n))
for q, lst in byq.items():
    busy = sum(e - s for s, e, _ in lst)
    print(f"queue {q}: {len(lst) / nsteps:.0f} dispatches/step, busy {busy / 1e6 / nsteps:.3f} ms/step")
# union busy over all queues and idle time
pts = sorted((s, e) for s, e, _, _ in seg)
cur_s, cur_e = pts[0]
union = 0
for s, e in pts[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"GPU busy (union) {union / 1e6 / nsteps:.3f} ms/step, idle {(t1 - t0 - union) / 1e6 / nsteps:.3f} ms/step")
# main queue = the one with most dispatches
mainq = max(byq, key=lambda q: len(byq[q]))
others = sorted((s, e) for q, lst in byq.items() if q != mainq for s, e, _ in lst)


def overlap(s, e):
    tot = 0
    for os_, oe in others:
        if oe <= s:
            continue
        if os_ >= e:
            break
        tot += min(e, oe) - max(s, os_)
    return tot


agg = defaultdict(lambda: [0, 0, 0])
prev_end = None
gaps = 0
for s, e, n in sorted(byq[mainq]):
    a = agg[n[:90]]
    a[0] += 1
    a[1] += e - s
    a[2] += overlap(s, e)
    if prev_end is not None and s > prev_end:
        gaps += s - prev_end
    prev_end = max(prev_end or e, e)
print(f"main queue {mainq}: gaps between consecutive kernels {gaps / 1e6 / nsteps:.3f} ms/step")
print(f"{'main-queue kernel':92s} {'n/step':>6s} {'ms/step':>8s} {'avg_us':>8s} {'ovl%':>5s}")
for n, (c, t, o) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:92s} {c / nsteps:6.0f} {t / 1e6 / nsteps:8.3f} {t / 1e3 / c:8.1f} {100 * o / max(t, 1):5.0f}")
for q, lst in byq.items():
    if q == mainq:
        continue
    agg2 = defaultdict(lambda: [0, 0])
    for s, e, n in lst:
        agg2[n[:90]][0] += 1
        agg2[n[:90]][1] += e - s
    print(f"-- queue {q}")
    for n, (c, t) in sorted(agg2.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"{n:92s} {c / nsteps:6.0f} {t / 1e6 / nsteps:8.3f} {t / 1e3 / c:8.1f}")
