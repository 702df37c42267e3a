"""Why was the inference leg of `bench.py` host-bound when it followed the CPU-oracle leg (driver r2:
17.5 vs 21.3 volumes/s; r3 first run: 5.5 volumes/s, 168 ms to enqueue a volume instead of 8)?
usage: probe_cpu_then_infer.py VARIANT
  none          infer only
  cpu           infer, CPU oracle leg, infer
  train_cpu     train leg, CPU oracle leg, THEN build the inference network and run it (= bench.py's old order)
  train_sleep   train leg, 13 s of sleep (idle GPU, no CPU work), then inference
  train         train leg, then inference (= bench.py --no-cpu-baseline)
(OMP_WAIT_POLICY / GPU_MAX_HW_QUEUES are varied through the environment by the calling script)"""
import argparse
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
torch.cuda.set_device(dev)


def barrier():
    torch.cuda.synchronize()


def infer_times(n):
    net = bench.make_net(16, 128, "bf16", dev).eval()
    vol = torch.randn((1, 1, 512, 512, 512)).to(dev)

    def one():
        st = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            sliding_window_inference(vol, (128,) * 3, 4, net, overlap=0.5, return_labels=True, return_logits=False,
                                     stats=st)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) * 1e3, 1), round(st["host_enqueue_s"] * 1e3, 1)
    return [one() for _ in range(n)]


out = {"variant": variant, "OMP_WAIT_POLICY": os.environ.get("OMP_WAIT_POLICY"), "hwq": os.environ.get("GPU_MAX_HW_QUEUES")}
if variant in ("none", "cpu"):
    out["before_ms(total,enqueue)"] = infer_times(3)
if variant.startswith("train"):
    args = argparse.Namespace(classes=16, size=128, batch=8)
    r = bench.run_train(args, "bf16", 0, 1, dev, barrier, 5, 2)
    out["train_ms"] = round(r["dt"] / 5 * 1e3, 3)
    del r
    torch.cuda.empty_cache()
if variant in ("cpu", "train_cpu"):
    bench.cpu_baseline_train(128, 16, steps=3)
if variant == "train_sleep":
    time.sleep(13)
out["threads"] = torch.get_num_threads()
out["after_ms(total,enqueue)"] = infer_times(5)
print(json.dumps(out))
