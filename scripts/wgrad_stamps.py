#!/usr/bin/env python3
"""Per-phase cycle stamps of workgroup 0 of the wave-specialised weight gradient (diag build only).
usage: SEGMI_LIB=.../libsegmi_diag.so python scripts/wgrad_stamps.py [top|toptf]"""
import os, sys, torch
import numpy as np
DEV = "cuda:0"
torch.zeros(1, device=DEV)
st = torch.zeros(128 * 12 * 4, dtype=torch.int64, device=DEV)
os.environ["SEGMI_WGRAD_STAMPS"] = str(st.data_ptr())
sys.path.insert(0, ".")
from segmantic_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "top"
B = 8
x = torch.randn((B, 128, 128, 128, 16), device=DEV).bfloat16()
dy = torch.randn((B, 128, 128, 128, 16), device=DEV).bfloat16()
dw = torch.empty((16, 16, 3, 3, 3), device=DEV)
ws = torch.empty(ops.conv3d_wgrad_workspace(x, dy, 3, 1), dtype=torch.uint8, device=DEV)
in_tf = None
if which == "toptf":
    in_tf = (torch.rand(16, device=DEV) + 0.5, torch.randn(16, device=DEV) * 0.1, torch.full((1,), 0.25, device=DEV))
for _ in range(3):
    ops.conv3d_wgrad(x, dy, dw, None, 3, 1, ws, in_tf=in_tf)
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(128, 12, 4).astype(np.int64)
it = slice(8, 120)
def d(a, b):
    return float(np.mean(a[it] - b[it]))
print(f"{which} dbg={os.environ.get('SEGMI_WGRAD_DBG', '0')}  (s_memtime ticks = 100 MHz REFCLK or shader clk, see ratio)")
print("iteration period (wave 0):", float(np.mean(np.diff(s[8:120, 0, 0]))))
for w in (0, 3):
    print(f"consumer wave {w}: compute {d(s[:, w, 1], s[:, w, 0]):8.1f}  barrier wait {d(s[:, w, 2], s[:, w, 1]):8.1f}")
for w in (4, 7, 8, 11):
    a = s[it, w]
    print(f"producer wave {w}: commit {np.mean(a[:, 1] - a[:, 0]):8.1f} fetch {np.mean(a[:, 2] - a[:, 1]):8.1f} "
          f"barrier {np.mean(a[:, 3] - a[:, 2]):8.1f}")
