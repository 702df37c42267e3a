#!/usr/bin/env python3
"""List every separate BatchNorm / PReLU pass of one training step (shapes + flags), to see which
BN-apply / BN-backward passes are still materialised rather than fused into a conv."""
import sys, torch
sys.path.insert(0, ".")
from segmantic_amd import ops
from segmantic_amd.seg.monai_unet import Net

calls = []
def wrap(name):
    f = getattr(ops, name)
    def g(*a, **k):
        t = a[0]
        extra = ("res" if k.get("residual") is not None else "")
        calls.append((name, tuple(t.shape), extra))
        return f(*a, **k)
    setattr(ops, name, g)
for n in ("bn_act_fwd", "bn_act_bwd_reduce", "bn_act_bwd_apply", "bn_stats", "add"):
    if hasattr(ops, n):
        wrap(n)
net = Net(num_classes=16, num_channels=1, spatial_size=[128] * 3)
net.mixed_precision = True
net = net.to("cuda:0").train()
x = torch.randn(8, 1, 128, 128, 128, device="cuda:0")
y = torch.randint(0, 16, (8, 1, 128, 128, 128), device="cuda:0").float()
for it in range(2):
    calls.clear()
    net.training_step({"image": x, "label": y})
torch.cuda.synchronize()
tot = 0
for c in calls:
    n = 1
    for s in c[1]: n *= s
    print(f"{c[0]:20s} {str(c[1]):28s} {n * 2 / 1e6:8.1f} MB/tensor {c[2]}")
