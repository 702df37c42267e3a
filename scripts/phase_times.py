#!/usr/bin/env python3
"""Un-profiled phase timing of the training step with HIP events on the main stream (rocprofv3's
per-launch overhead makes the host the bottleneck, so its timelines overstate dispatch gaps).
usage: phase_times.py [steps]"""
import os, sys, statistics, torch
sys.path.insert(0, ".")
import bench
from segmantic_amd.seg.losses import dice_backward, dice_forward

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
args = type("A", (), dict(classes=16, size=128, batch=8))()
net = bench.make_net(16, 128, "bf16", dev).train()
img, lab = bench.synthetic(8, 128, 16, 0, dev)
batch = {"image": img, "label": lab}
for _ in range(5):
    net.training_step(batch)
eng = net._engine
opt = net.optimizers()
names = ["forward", "dice fwd+bwd", "backward main chain", "wait for weight gradients", "adam"]
acc = {n: [] for n in names}
tot = []
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
recs = []
for _ in range(steps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    logits = eng.forward(img, train=True)
    ev[1].record()
    st = net.loss_function._state
    loss = dice_forward(st, logits, lab, 1e-5, 1e-5)
    dlogits = dice_backward(st, logits, 1.0, eng.dlogits_buffer(logits), bias_grad=eng.top_bias_grad())
    ev[2].record()
    # backward without the final join: replicate UNetEngine.backward
    eng._top_bias_conv = eng.levels["upru"]["units"][-1][0]
    eng._level_bwd(eng.levels, dlogits)
    eng._top_bias_conv = None
    ev[3].record()
    eng._join_side()
    ev[4].record()
    opt.step(1.0)
    eng.bump()
    ev[5].record()
    recs.append(ev)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
for ev in recs[3:]:
    for i, n in enumerate(names):
        acc[n].append(ev[i].elapsed_time(ev[i + 1]))
    tot.append(ev[0].elapsed_time(ev[5]))
print(f"wall {wall:.3f} ms/step; event total {statistics.mean(tot):.3f} ms  (SEGMI_SERIAL={os.environ.get('SEGMI_SERIAL', '0')})")
for n in names:
    print(f"  {n:28s} {statistics.mean(acc[n]):7.3f} ms")
