#!/usr/bin/env python3
import sys, torch
sys.path.insert(0, ".")
from segmantic_amd import ops
DEV = "cuda:0"
flush = torch.empty(256 << 20, device=DEV)
for (n, ci, co, d) in [(8, 32, 32, 160), (8, 32, 32, 128), (4, 32, 32, 64)]:
    x = torch.randn((n, d, d, d, ci), device=DEV).bfloat16()
    y = torch.empty((n, d, d, d, co), device=DEV, dtype=torch.bfloat16)
    w = torch.randn((co, ci, 3, 3, 3), device=DEV) * 0.05
    pk = ops.wpack(torch.bfloat16, 0, w, ci, co, 3)
    ops.conv3d_fwd(x, y, pk, None, 0, None, 3, 1, residual=x)
    tot = 0.0
    for _ in range(3):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv3d_fwd(x, y, pk, None, 0, None, 3, 1, residual=x); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    us = tot / 3 * 1e3
    fl = 2.0 * n * d ** 3 * ci * co * 27
    print(f"N{n} {ci}->{co} {d}^3: {us:9.1f} us {fl / us / 1e6:7.1f} TFLOP/s  {2 * n * d**3 * 32 * 2 / us / 1e6:6.2f} TB/s(alg)")
