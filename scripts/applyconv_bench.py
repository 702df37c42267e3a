#!/usr/bin/env python3
"""segmi_bn_act_bwd_apply_conv against its two-launch equivalent on the two shapes of the C1 training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segmantic_amd import ops

DEV = torch.device("cuda:0")
flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)


def timed(fn, reps=5):
    fn(); fn()
    tot = 0.0
    for _ in range(reps):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


for n, sp, cout in [(8, 128, 32), (8, 64, 64)]:
    c = 16
    x = torch.randn((n, sp, sp, sp, c), device=DEV).to(torch.bfloat16)
    dy = torch.randn((n, sp, sp, sp, c), device=DEV).to(torch.bfloat16)
    dx = torch.empty_like(x)
    out = torch.empty((n, sp // 2, sp // 2, sp // 2, cout), dtype=torch.bfloat16, device=DEV)
    mean = torch.zeros(c, device=DEV); invstd = torch.ones(c, device=DEV)
    gamma = torch.ones(c, device=DEV); beta = torch.zeros(c, device=DEV)
    alpha = torch.full((1,), 0.25, device=DEV)
    coef = torch.zeros((2, c), device=DEV)
    w = torch.randn((cout, c, 3, 3, 3), device=DEV) * 0.1
    pack = ops.wpack(torch.bfloat16, 0, w, c, cout, 3)
    t_apply = timed(lambda: ops.bn_act_bwd_apply(dy, x, dx, mean, invstd, gamma, beta, alpha, coef))
    t_conv = timed(lambda: ops.conv3d_fwd(dx, out, pack, w, 0, None, 3, 2))
    t_fused = timed(lambda: ops.bn_act_bwd_apply_conv(dy, x, dx, mean, invstd, gamma, beta, alpha, coef, out, pack))
    mb = (3 * x.numel() + out.numel()) * 2 / 1e6
    print(f"{n} x {sp}^3 x16 -> {cout}: apply {t_apply:6.1f} + conv {t_conv:6.1f} = {t_apply + t_conv:6.1f} us; "
          f"fused {t_fused:6.1f} us ({mb:6.1f} MB, {mb / t_fused:4.2f} TB/s)")
