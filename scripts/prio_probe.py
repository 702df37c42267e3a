#!/usr/bin/env python3
"""Training step with the main chain on the default stream vs on a high-priority stream (the weight-gradient
stream stays at normal priority).  usage: prio_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net = bench.make_net(16, 128, "bf16", dev).train()
img, lab = bench.synthetic(8, 128, 16, 0, dev)
batch = {"image": img, "label": lab}


def timed(stream, steps=20):
    with torch.cuda.stream(stream):
        for _ in range(5):
            net.training_step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net.training_step(batch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3


lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
default = torch.cuda.current_stream()
high = torch.cuda.Stream(device=dev, priority=-1)
for rep in range(3):
    print(f"default stream {timed(default):.3f} ms   high-priority main stream {timed(high):.3f} ms")
