#!/usr/bin/env python3
"""Benchmark step with the gradient buckets going through RCCL (one rank: identity all-reduces), to see what
the collectives' launches cost beside the backward.  usage: rccl_step_probe.py [0|1]"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, ".")
force = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", SEGMI_GRADSYNC_FORCE=force)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
sys.argv = ["bench.py"]
args = bench.parse()
net = bench.make_net(args.classes, args.size, "bf16", torch.device("cuda:0")).train()
gs = net.enable_grad_sync()
sizes = []
_launch = gs._launch
def counting(lo, hi, after=()):
    if hi > lo:
        sizes.append((hi - lo) * 4 // 1024)
    return _launch(lo, hi, after)
gs._launch = counting
img, lab = bench.synthetic(args.batch, args.size, args.classes, 0, torch.device("cuda:0"))
batch = {"image": img, "label": lab}
for _ in range(5):
    sizes.clear()
    net.training_step(batch)
print("collectives per step (KiB):", sizes, flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30):
    net.training_step(batch)
torch.cuda.synchronize()
print(f"force={force}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
dist.destroy_process_group()
