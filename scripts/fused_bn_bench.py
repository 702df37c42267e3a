#!/usr/bin/env python3
"""Time of one segmi_bn_act_bwd_fused launch on the deep-level shapes of the C1 training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segmantic_amd import ops

DEV = torch.device("cuda:0")
for c, sp, n in [(32, 32, 8), (64, 16, 8), (128, 8, 8), (256, 8, 8)]:
    x = torch.randn((n, sp, sp, sp, c), device=DEV).to(torch.bfloat16)
    dy = torch.randn((n, sp, sp, sp, c), device=DEV).to(torch.bfloat16)
    dx = torch.empty_like(x)
    mean = torch.zeros(c, device=DEV); invstd = torch.ones(c, device=DEV)
    gamma = torch.ones(c, device=DEV); beta = torch.zeros(c, device=DEV)
    alpha = torch.full((1,), 0.25, device=DEV)
    rows = ops.bn_act_bwd_fused_rows(x)
    part = torch.zeros((rows, 3, c), device=DEV)
    dg, db, da, coef = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty(1, device=DEV), torch.empty((2, c), device=DEV)
    fin = (n * sp ** 3, dg, db, da, coef)
    for _ in range(5):
        ops.bn_act_bwd_fused(dy, x, dx, mean, invstd, gamma, beta, alpha, part, fin)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.bn_act_bwd_fused(dy, x, dx, mean, invstd, gamma, beta, alpha, part, fin)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    mb = 3 * x.numel() * 2 / 1e6
    print(f"c={c:3d} {sp}^3 x{n}: {us:6.1f} us  ({mb:5.1f} MB algorithmic, {mb / us:5.2f} TB/s)")
