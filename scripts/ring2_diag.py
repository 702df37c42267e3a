#!/usr/bin/env python3
"""Time the top 16->16 ring2 conv (identity residual) -- used with SEGMI_RING2_DBG to split its time."""
import os, sys, torch
sys.path.insert(0, ".")
from segmantic_amd import ops
DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn((n, 128, 128, 128, 16), device=DEV).bfloat16()
y = torch.empty_like(x)
w = torch.randn((16, 16, 3, 3, 3), device=DEV) * 0.05
b = torch.zeros(16, device=DEV)
pk = ops.wpack(torch.bfloat16, 0, w, 16, 16, 3)
flush = torch.empty(256 << 20, device=DEV)
for _ in range(3):
    ops.conv3d_fwd(x, y, pk, None, 0, b, 3, 1, residual=x)
tot = 0.0
for _ in range(10):
    flush.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv3d_fwd(x, y, pk, None, 0, b, 3, 1, residual=x)
    e1.record(); torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
print(f"dbg={os.environ.get('SEGMI_RING2_DBG', '0')} N={n}: {tot / 10 * 1e3:7.1f} us")
