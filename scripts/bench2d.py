#!/usr/bin/env python3
"""2-D UNet (spatial_dims=2) training step / eval forward timing -- functional path, depth-1 volumes."""
import sys, time, torch
sys.path.insert(0, ".")
from segmantic_amd.seg.monai_unet import Net
dev = torch.device("cuda:0")
B, S, K = 16, 256, 16
net = Net(num_classes=K, num_channels=1, spatial_dims=2)
net.mixed_precision = True
net.to(dev).train()
g = torch.Generator().manual_seed(0)
img = torch.randn((B, 1, S, S), generator=g).to(dev)
lab = torch.randint(0, K, (B, 1, S, S), generator=g).float().to(dev)
batch = {"image": img, "label": lab}
for _ in range(3):
    net.training_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    net.training_step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"2-D train B={B} {S}x{S} K={K}: {dt * 1e3:.2f} ms/step = {B * S * S / dt / 1e6:.1f} Mpixel/s")
net.eval()
with torch.no_grad():
    for _ in range(3):
        net(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        net(img)
    torch.cuda.synchronize()
print(f"2-D eval forward: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
