#!/usr/bin/env python3
"""Time the top-level 16x16 k3 weight gradient (cold caches); SEGMI_WGRAD_DBG splits its time in a diag build."""
import os, sys, torch
sys.path.insert(0, ".")
from segmantic_amd import ops
DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn((n, 128, 128, 128, 16), device=DEV).bfloat16()
dy = torch.randn((n, 128, 128, 128, 16), device=DEV).bfloat16()
dw = torch.empty((16, 16, 3, 3, 3), device=DEV)
db = torch.empty(16, device=DEV)
ws = torch.empty(ops.conv3d_wgrad_workspace(x, dy, 3, 1), dtype=torch.uint8, device=DEV)
flush = torch.empty(256 << 20, device=DEV)
for _ in range(3):
    ops.conv3d_wgrad(x, dy, dw, None, 3, 1, ws)
tot = 0.0
for _ in range(10):
    flush.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv3d_wgrad(x, dy, dw, None, 3, 1, ws)
    e1.record(); torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
print(f"wgrad dbg={os.environ.get('SEGMI_WGRAD_DBG', '0')} N={n}: {tot / 10 * 1e3:7.1f} us (incl. slab reduce)")
