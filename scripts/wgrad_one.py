#!/usr/bin/env python3
"""One configuration of the weight gradient, N launches (for rocprofv3 / diag timing).
usage: wgrad_one.py [top|toptf|upT] [reps]"""
import os, sys, torch
sys.path.insert(0, ".")
from segmantic_amd import ops
DEV = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "top"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = 8
if which in ("top", "toptf"):
    xs, dys, k, s = (B, 128, 128, 128, 16), (B, 128, 128, 128, 16), 3, 1
else:
    xs, dys, k, s = (B, 128, 128, 128, 16), (B, 64, 64, 64, 32), 3, 2
x = torch.randn(xs, device=DEV).bfloat16()
dy = torch.randn(dys, device=DEV).bfloat16()
dw = torch.empty((dys[-1], xs[-1], k, k, k), device=DEV)
ws = torch.empty(ops.conv3d_wgrad_workspace(x, dy, k, s), dtype=torch.uint8, device=DEV)
in_tf = None
if which == "toptf":
    in_tf = (torch.rand(16, device=DEV) + 0.5, torch.randn(16, device=DEV) * 0.1, torch.full((1,), 0.25, device=DEV))
flush = torch.empty(256 << 20, device=DEV)
tot = 0.0
for i in range(reps + 1):
    flush.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv3d_wgrad(x, dy, dw, None, k, s, ws, in_tf=in_tf)
    e1.record(); torch.cuda.synchronize()
    if i: tot += e0.elapsed_time(e1)
print(f"{which} dbg={os.environ.get('SEGMI_WGRAD_DBG', '0')} ws={os.environ.get('SEGMI_WGRAD_WS', '1')}: {tot / reps * 1e3:7.1f} us (incl. slab reduce)")
