#!/usr/bin/env python3
"""Time the full-resolution 16 -> 16 ring conv alone (cold caches) in its three training roles; with
SEGMI_RING3_DBG bits the time splits (conv_ring3_impl.h).  usage: ring3_diag.py [N] [role ...]"""
import os, sys, torch
sys.path.insert(0, ".")
from segmantic_amd import ops
DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
roles = sys.argv[2:] or ["fwd_tf", "dgrad", "dgrad_sums"]
S = int(os.environ.get("DIAG_SIZE", "128"))
x = torch.randn((n, S, S, S, 16), device=DEV).bfloat16()
x2 = torch.randn((n, S, S, S, 16), device=DEV).bfloat16()
y = torch.empty_like(x)
w = torch.randn((16, 16, 3, 3, 3), device=DEV) * 0.05
b = torch.zeros(16, device=DEV)
pk = ops.wpack(torch.bfloat16, 0, w, 16, 16, 3)
pkd = ops.wpack(torch.bfloat16, 1, w, 16, 16, 3)
scale, shift = torch.rand(16, device=DEV) + 0.5, torch.randn(16, device=DEV) * 0.1
alpha = torch.full((1,), 0.25, device=DEV)
mean, invstd = torch.randn(16, device=DEV) * 0.1, torch.rand(16, device=DEV) + 0.5
gamma, beta = torch.rand(16, device=DEV) + 0.5, torch.randn(16, device=DEV) * 0.1
rows = ops.conv3d_stats_rows(x, y, 3, 1)
part = torch.zeros((rows, 3, 16), device=DEV)
flush = torch.empty(256 << 20, device=DEV)


def call(role):
    if role == "fwd_tf":      # training forward of the top unit: producer's BatchNorm + PReLU while staging, identity residual
        ops.conv3d_fwd(x, y, pk, None, 0, b, 3, 1, residual=x, in_tf=(scale, shift, alpha))
    elif role == "fwd":       # the same without the transform (eval two-launch path)
        ops.conv3d_fwd(x, y, pk, None, 0, b, 3, 1, residual=x)
    elif role == "dgrad":     # input gradient with an external residual (gradient sum)
        ops.conv3d_fwd(x, y, pkd, None, 1, None, 3, 1, residual=x2)
    elif role == "dgrad_sums":
        ops.conv3d_fwd(x, y, pkd, None, 1, None, 3, 1, bn_bwd=(x2, mean, invstd, gamma, beta, alpha, part))
    elif role == "fwd_act":   # inference: folded BatchNorm in the pack, PReLU, identity residual
        ops.conv3d_fwd(x, y, pk, None, 0, b, 3, 1, prelu_alpha=alpha, residual=x)
    elif role == "plain":
        ops.conv3d_fwd(x, y, pkd, None, 1, None, 3, 1)


if "copy" in roles:
    ts = []
    for _ in range(10):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y.copy_(x); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"copy of {x.numel() * 2 / 1e6:.0f} MB: median {ts[5]:.1f} us = {2 * x.numel() * 2 / ts[5] / 1e6:.2f} TB/s (read + write)")
    roles = [r for r in roles if r != "copy"]
for role in roles:
    for _ in range(3):
        call(role)
    ts = []
    for _ in range(10):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call(role)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"ring3={os.environ.get('SEGMI_RING3', '1')} xcd={os.environ.get('SEGMI_RING3_XCD', '-')} zs={os.environ.get('SEGMI_RING_ZS', '1')} S={S} dbg={os.environ.get('SEGMI_RING3_DBG', '0'):>2} N={n} {role:10s}: "
          f"median {ts[5]:7.1f} us  min {ts[0]:7.1f}", flush=True)
