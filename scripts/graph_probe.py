#!/usr/bin/env python3
"""Does a hipGraph replay of forward + Dice + backward beat eager stream launches (dispatch gaps)?"""
import sys, time, torch
sys.path.insert(0, ".")
from bench import synthetic
from segmantic_amd.seg.monai_unet import Net
from segmantic_amd.seg.losses import dice_backward, dice_forward
dev = torch.device("cuda:0")
net = Net(num_classes=16)
net.mixed_precision = True
net.to(dev).train()
img, lab = synthetic(8, 128, 16, 0, dev)
eng = net._engine_for(img)
st = net.loss_function._state


def fb():
    eng.bump()                                   # the step's weight re-pack is part of the work
    logits = eng.forward(img, train=True)
    dice_forward(st, logits, lab, 1e-5, 1e-5)
    dl = dice_backward(st, logits, 1.0, eng.dlogits_buffer(logits), bias_grad=eng.top_bias_grad())
    eng.backward(dl, top_bias_done=True)


for _ in range(3):
    fb()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    fb()
torch.cuda.synchronize()
print(f"eager  fwd+loss+bwd: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fb()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    fb()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"graph  fwd+loss+bwd: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
