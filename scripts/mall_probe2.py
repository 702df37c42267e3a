"""Ping-pong traversal: a 537 MB tensor (8 x 128^3 x 16 bf16) written front-to-back by one kernel is read by
the next kernel (a) front-to-back, (b) back-to-front in quarters (4 launches, last-written quarter first).
If the 256 MB Infinity Cache keeps the most recently written part, (b) should move less from HBM.
usage: mall_probe2.py"""
import sys
import torch
sys.path.insert(0, ".")
from segmantic_amd import ops

dev = "cuda:0"
src = torch.randn((8, 128, 128, 128, 16), device=dev).to(torch.bfloat16)
buf = torch.empty_like(src)
dst = torch.empty_like(src)
sc = torch.ones(16, device=dev)
parts = {n: [torch.empty((ops.bn_stats_rows(buf[:8 // n]), 2, 16), device=dev) for _ in range(n)] for n in (1, 2, 4, 8)}


def producer():
    ops.bn_act_fwd(src, buf, sc, sc, None)        # reads src, writes buf front to back (1.07 GB of traffic)


def consumer(order, n):
    step = 8 // n
    for i in order:
        ops.bn_stats(buf[i * step:(i + 1) * step], parts[n][i])


def rw_consumer(order, n):                          # reads buf, writes dst (elementwise pass)
    step = 8 // n
    for i in order:
        ops.bn_act_fwd(buf[i * step:(i + 1) * step], dst[i * step:(i + 1) * step], sc, sc, None)


def timed(fn, reps=15):
    ts = []
    for _ in range(reps):
        producer()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


for n in (1, 2, 4, 8):
    fwd, rev = list(range(n)), list(range(n - 1, -1, -1))
    a, b = timed(lambda: consumer(fwd, n)), timed(lambda: consumer(rev, n))
    c, d = timed(lambda: rw_consumer(fwd, n)), timed(lambda: rw_consumer(rev, n))
    print(f"{n} launches: read-only  forward {a:7.1f} us  reverse {b:7.1f} us   |  read+write forward {c:7.1f} us  reverse {d:7.1f} us")
