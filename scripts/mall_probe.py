"""Does the 256 MB Infinity Cache (MALL) serve a tensor that the previous kernel has just written?
Producer = a device copy into a buffer of S MB, consumer = a read-only pass (sum) over it, timed with HIP
events; "cold" = 2 GB of unrelated traffic between the two.  usage: mall_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
from segmantic_amd import ops

dev = "cuda:0"
flush_src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
flush_dst = torch.empty_like(flush_src)


def timed(fn, reps=20):
    ts = []
    for _ in range(reps):
        pre = fn()          # returns the closure to time after its own setup
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pre()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for mb in (32, 64, 128, 192, 256, 512):
    n = mb << 20
    src = torch.randn((1, n // 2 // (128 * 128 * 16), 128, 128, 16), device=dev).to(torch.bfloat16)
    buf = torch.empty_like(src)
    part = torch.empty((ops.bn_stats_rows(buf), 2, 16), device=dev)

    def hot():
        buf.copy_(src)                       # producer: writes `buf`
        return lambda: ops.bn_stats(buf, part)

    def cold():
        buf.copy_(src)
        flush_dst.copy_(flush_src)           # 2 GB of unrelated traffic
        return lambda: ops.bn_stats(buf, part)

    def hot_rw():                            # consumer that reads buf and writes another tensor of the same size
        buf.copy_(src)
        return lambda: torch.mul(buf, 2.0, out=src2)

    src2 = torch.empty_like(src)
    th, tc = timed(hot), timed(cold)
    print(f"{mb:4d} MB: read after write  hot {th*1e3:7.1f} us = {mb/1024/th*1e3:6.2f} TB/s   cold {tc*1e3:7.1f} us = {mb/1024/tc*1e3:6.2f} TB/s")
