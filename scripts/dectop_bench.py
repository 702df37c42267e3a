#!/usr/bin/env python3
"""Fused full-resolution decoder (segmi_dectop_fwd) vs the two launches it replaces, cold caches.
usage: dectop_bench.py [windows]"""
import sys, torch
sys.path.insert(0, ".")
from segmantic_amd import ops
DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.randn((n, 64, 64, 64, 32), device=DEV).bfloat16()
wt = torch.randn((32, 16, 3, 3, 3), device=DEV) * 0.05
wc = torch.randn((16, 16, 3, 3, 3), device=DEV) * 0.05
scale = torch.rand(16, device=DEV) + 0.5
ub = torch.randn(16, device=DEV) * 0.1
cb = torch.randn(16, device=DEV) * 0.1
alpha = torch.full((1,), 0.25, device=DEV)
up_pack = ops.wpack(torch.bfloat16, 2, wt, 32, 16, 3, scale=scale)
cv_pack = ops.wpack(torch.bfloat16, 0, wc, 16, 16, 3)
frag = ops.dectop_up_frag(wt, scale)
h = torch.empty((n, 128, 128, 128, 16), dtype=torch.bfloat16, device=DEV)
out = torch.empty_like(h)
flush = torch.empty(256 << 20, device=DEV)

def two():
    ops.convT3d_fwd(x, h, up_pack, None, ub, prelu_alpha=alpha)
    ops.conv3d_fwd(h, out, cv_pack, None, 0, cb, 3, 1, residual=h)

def one():
    ops.dectop_fwd(x, out, frag, ub, alpha, cv_pack, cb, alpha_in_unit_range=True)

import os
only = os.environ.get("SEGMI_DECTOP_DBG")
cases = (("fused dbg=" + only, one),) if only else (("two launches", two), ("fused", one), ("two launches", two), ("fused", one))
for name, fn in cases:
    fn(); fn()
    tot = 0.0
    for _ in range(5):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print(f"{name:14s} N={n}: {tot / 5 * 1e3:8.1f} us  ({tot / 5 * 1e3 / n:6.1f} us per window)")
