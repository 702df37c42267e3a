#!/usr/bin/env python3
"""Race / determinism soak: two identical 30-step bf16 training runs (side-stream weight gradients,
asynchronous re-pack, fused BN-apply) must end in bit-identical parameters, and two sliding-window
passes over the same volume (two lanes) in bit-identical logits and labels."""
import sys, torch
sys.path.insert(0, ".")
from bench import synthetic
from segmantic_amd.seg.monai_unet import Net
from segmantic_amd.seg.inferers import sliding_window_inference
dev = torch.device("cuda:0")


def run(steps):
    torch.manual_seed(0)
    net = Net(num_classes=16)
    net.mixed_precision = True
    net.to(dev).train()
    img, lab = synthetic(4, 128, 16, 0, dev)
    losses = []
    for _ in range(steps):
        losses.append(net.training_step({"image": img, "label": lab})["loss"])
    torch.cuda.synchronize()
    return net, net._engine.flat.clone(), [float(l.cpu()) for l in losses]


n1, f1, l1 = run(30)
n2, f2, l2 = run(30)
print("losses", l1[0], l1[-1])
assert l1 == l2, "training losses differ between identical runs"
assert torch.equal(f1, f2), "parameters differ between identical runs"
assert l1[-1] < l1[0]
n1.eval()
g = torch.Generator().manual_seed(5)
vol = torch.randn((1, 1, 256, 256, 256), generator=g).to(dev)
with torch.no_grad():
    a = sliding_window_inference(vol, (128,) * 3, 4, n1, overlap=0.5, return_labels=True)
    la, ga = a.labels.clone(), a.logits.clone()
    for _ in range(3):
        b = sliding_window_inference(vol, (128,) * 3, 4, n1, overlap=0.5, return_labels=True)
        assert torch.equal(la, b.labels) and torch.equal(ga, b.logits), "sliding-window result not reproducible"
print("soak ok")
