#!/usr/bin/env python3
"""Time segmi_conv3d_fwd on the deep-layer shapes (HIP events, 20 reps)."""
import sys, torch
sys.path.insert(0, ".")
from segmantic_amd import ops
DEV = "cuda:0"
shapes = [  # (N, Cin, Cout, D, stride)
    (8, 64, 64, 32, 1), (8, 32, 64, 64, 2), (8, 128, 128, 16, 1), (8, 64, 128, 32, 2),
    (8, 128, 256, 8, 1), (8, 256, 256, 8, 1), (8, 128, 64, 32, 1), (8, 256, 128, 16, 1),
    (4, 64, 64, 32, 1), (4, 128, 128, 16, 1), (4, 256, 256, 8, 1), (4, 128, 256, 8, 1),
    (8, 32, 32, 32, 1), (8, 128, 256, 16, 2), (4, 32, 32, 32, 1), (4, 32, 64, 64, 2), (4, 64, 128, 32, 2),
]
flush = torch.empty(256 << 20, device=DEV)
for (n, ci, co, d, s) in shapes:
    do = d // s
    x = torch.randn((n, d, d, d, ci), device=DEV).bfloat16()
    y = torch.empty((n, do, do, do, co), device=DEV, dtype=torch.bfloat16)
    w = torch.randn((co, ci, 3, 3, 3), device=DEV) * 0.05
    pk = ops.wpack(torch.bfloat16, 0, w, ci, co, 3)
    for _ in range(3):
        ops.conv3d_fwd(x, y, pk, None, 0, None, 3, s)
    tot = 0.0
    for _ in range(10):          # cold caches: 1 GiB written between the timed launches
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv3d_fwd(x, y, pk, None, 0, None, 3, s)
        e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    us = tot / 10 * 1e3
    fl = 2.0 * n * do ** 3 * ci * co * 27
    print(f"N{n} {ci:3d}->{co:3d} {d:2d}^3 s{s}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
