"""bench.py order in one process: train leg, empty_cache, then 8 volumes back to back WITHOUT a device sync
between them (as the timed loop does); per volume: host time to enqueue it, and a breakdown of where the
host spends it (torch.empty calls vs kernel launches).  usage: probe_first_infer.py [notrain]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from segmantic_amd import ops
from segmantic_amd.seg.inferers import sliding_window_inference

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if "notrain" not in sys.argv:
    args = argparse.Namespace(classes=16, size=128, batch=8)
    r = bench.run_train(args, "bf16", 0, 1, dev, torch.cuda.synchronize, 10, 3)
    del r
    torch.cuda.empty_cache()
net = bench.make_net(16, 128, "bf16", dev).eval()
vol = torch.randn((1, 1, 512, 512, 512)).to(dev)

# instrument: time spent inside torch.empty and inside the C-ABI launches
acc = {"empty": 0.0, "launch": 0.0, "n_launch": 0}
_empty = torch.empty
def empty(*a, **k):
    t = time.perf_counter(); o = _empty(*a, **k); acc["empty"] += time.perf_counter() - t; return o
torch.empty = empty
from segmantic_amd import _lib
_check = ops.check
for name in ("segmi_conv3d_fwd", "segmi_convT3d_fwd", "segmi_dectop_fwd", "segmi_conv3d_fwd_pair", "segmi_conv3d_fwd_split_act", "segmi_sw_gather", "segmi_sw_blend"):
    fn = getattr(_lib.lib, name)
    def wrap(fn):
        def w(*a):
            t = time.perf_counter(); rc = fn(*a); acc["launch"] += time.perf_counter() - t; acc["n_launch"] += 1; return rc
        return w
    setattr(_lib.lib, name, wrap(fn))

torch.cuda.synchronize()
rows = []
t_all = time.perf_counter()
NV = 12
for i in range(NV):
    acc.update(empty=0.0, launch=0.0, n_launch=0)
    st = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        sliding_window_inference(vol, (128,) * 3, 4, net, overlap=0.5, return_labels=True, return_logits=False, stats=st)
    ms = torch.cuda.memory_stats()
    rows.append((round((time.perf_counter() - t0) * 1e3, 1), round(acc["empty"] * 1e3, 1), round(acc["launch"] * 1e3, 1), acc["n_launch"],
                 ms.get("num_device_alloc", -1), ms.get("num_device_free", -1), round(ms.get("reserved_bytes.all.current", 0) / 2**30, 1)))
torch.cuda.synchronize()
print("per volume (host ms total, in torch.empty, in C launches, launches, device allocs, device frees, reserved GiB):")
for r_ in rows:
    print("  ", r_)
print("wall per volume", round((time.perf_counter() - t_all) / NV * 1e3, 1), "ms")
