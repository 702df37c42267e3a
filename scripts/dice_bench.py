#!/usr/bin/env python3
"""Dice forward / backward kernels in isolation (bf16, B=8 x 128^3, K=16), cold caches."""
import sys, torch
sys.path.insert(0, ".")
from segmantic_amd.seg.losses import _DiceState, dice_backward, dice_forward
DEV = "cuda:0"
lg = torch.randn((8, 128, 128, 128, 16), device=DEV).bfloat16()
lab = torch.randint(0, 16, (8, 1, 128, 128, 128), device=DEV).float()
st = _DiceState()
out = torch.empty_like(lg)
flush = torch.empty(256 << 20, device=DEV)
for name, fn in (("fwd", lambda: dice_forward(st, lg, lab, 1e-5, 1e-5)), ("bwd", lambda: dice_backward(st, lg, 1.0, out))):
    fn(); fn()
    tot = 0.0
    for _ in range(5):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print(f"dice {name}: {tot / 5 * 1e3:7.1f} us")
print("loss", float(dice_forward(st, lg, lab, 1e-5, 1e-5)))
