import time, torch, sys
sys.path.insert(0, '.')
from segmantic_amd.seg.monai_unet import Net
from bench import synthetic
dev = torch.device('cuda:0')
net = Net(num_classes=16); net.mixed_precision = True; net.to(dev).train()
img, lab = synthetic(8, 128, 16, 0, dev)
b = {"image": img, "label": lab}
for _ in range(3): net.training_step(b)
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    for _ in range(5): net.training_step(b)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/5:.2f} ms/step, total {1e3*(t2-t0)/5:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): net.training_step(b)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
