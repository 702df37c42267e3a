"""Why is a GPU leg slow right after the CPU oracle leg?  (round 3; bench.py now runs the CPU legs last)
usage: probe_cpu_then_infer.py VARIANT   with VARIANT in none | cpu | cpu_threads1 | cpu_sleep3
(OMP_WAIT_POLICY / GPU_MAX_HW_QUEUES are varied through the environment by the calling script)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import bench  # noqa: E402
from segmantic_amd.seg.inferers import sliding_window_inference  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "none"
dev = torch.device("cuda:0")
net = bench.make_net(16, 128, "bf16", dev).eval()
vol = torch.randn((1, 1, 512, 512, 512)).to(dev)


def one():
    st = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        sliding_window_inference(vol, (128,) * 3, 4, net, overlap=0.5, return_labels=True, return_logits=False, stats=st)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 1), round(st["host_enqueue_s"] * 1e3, 1)


one()                                   # warm-up (allocations, weight packs)
before = [one() for _ in range(2)]
if variant.startswith("cpu"):
    bench.cpu_baseline_train(128, 16, steps=2)
    if variant == "cpu_threads1":
        torch.set_num_threads(1)
    if variant == "cpu_sleep3":
        time.sleep(3)
after = [one() for _ in range(4)]
print(json.dumps({"variant": variant, "OMP_WAIT_POLICY": os.environ.get("OMP_WAIT_POLICY"),
                  "hwq": os.environ.get("GPU_MAX_HW_QUEUES"), "threads": torch.get_num_threads(),
                  "before_ms(total,enqueue)": before, "after_ms(total,enqueue)": after}))
