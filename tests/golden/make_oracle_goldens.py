#!/usr/bin/env python3
"""Generates tests/golden/oracle_goldens.npz from the CPU oracle (oracle/*.py, torch-CPU fp32).

The reference's own tests hold no golden vectors for the hot path and its hot path cannot be
imported here (MONAI / Lightning / SimpleITK absent), so these vectors come from the oracle
restatement: they pin the ORACLE against drift (tests/test_oracle.py) and give the GPU tests a
second, committed target besides the live oracle (tests/test_golden_gpu.py).  Parity with the
reference itself remains "unpinned" (see oracle/__init__.py).

    python tests/golden/make_oracle_goldens.py
"""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle.metrics_ref import ref_dice_metric, ref_normalize  # noqa: E402
from oracle.resample_ref import ref_resample_grid, resample_size  # noqa: E402
from oracle.sliding_ref import ref_sliding_window_inference, window_starts  # noqa: E402
from oracle.unet_ref import RefUNet, deterministic_fill_, ref_dice_loss, synthetic_batch  # noqa: E402

torch.set_num_threads(4)
out = {}

# (i) tiny net: full logits, loss, selected gradients, post-Adam weights, eval logits
ch, st, K = (4, 8, 16), (2, 2), 3
net = deterministic_fill_(RefUNet(3, 1, K, ch, st), 0).train()
img, lab = synthetic_batch(2, 16, K, seed=1)
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
y = net(img)
opt.zero_grad()
loss = ref_dice_loss(y, lab)
loss.backward()
out["tiny_logits"] = y.detach().numpy()
out["tiny_loss"] = np.float32(loss.item())
sel = ["model.0.conv.unit0.conv.weight", "model.2.1.conv.unit0.conv.weight", "model.1.submodule.1.submodule.conv.unit1.adn.N.weight",
       "model.2.0.conv.weight", "model.0.conv.unit1.adn.A.weight"]
params = dict(net.named_parameters())
for i, k in enumerate(sel):
    out[f"tiny_grad_{i}"] = params[k].grad.numpy().copy()
opt.step()
for i, k in enumerate(sel):
    out[f"tiny_adam_{i}"] = params[k].detach().numpy().copy()
out["tiny_sel_keys"] = np.array(sel)
out["tiny_running_mean"] = net.state_dict()["model.0.conv.unit0.adn.N.running_mean"].numpy().copy()
net2 = deterministic_fill_(RefUNet(3, 1, K, ch, st), 0).eval()
with torch.no_grad():
    out["tiny_eval_logits"] = net2(img).numpy()

# (ii) full default net, K=3, 1x1x32^3 (BASELINE config 1 shape): block + checksums + loss
net = deterministic_fill_(RefUNet(3, 1, 3), 0).train()
img32, lab32 = synthetic_batch(1, 32, 3, seed=2)
y = net(img32)
out["full_logits_block"] = y[0, :, :8, :8, :8].detach().numpy()
out["full_logits_sum"] = np.float64(y.double().sum().item())
out["full_logits_abs_sum"] = np.float64(y.double().abs().sum().item())
out["full_loss"] = np.float32(ref_dice_loss(y, lab32).item())

# (iv) sliding window: 40^3 volume, roi 16^3, overlaps 0.25 / 0.5, fixed conv predictor
g = torch.Generator().manual_seed(5)
vol = torch.rand((1, 1, 40, 40, 40), generator=g)
w = torch.rand((3, 1, 3, 3, 3), generator=g) - 0.5
pred = lambda x: F.conv3d(x, w, padding=1)
for ov in (0.25, 0.5):
    o, cnt, wins = ref_sliding_window_inference(vol, (16, 16, 16), 4, pred, ov)
    tag = f"sw_{int(ov * 100)}"
    out[tag + "_starts"] = np.array(wins, dtype=np.int32)
    out[tag + "_count_hist"] = np.bincount(cnt.reshape(-1).numpy().astype(np.int64))
    out[tag + "_out_sum"] = np.float64(o.double().sum().item())
    out[tag + "_out_block"] = o[0, :, 10:14, 10:14, 10:14].numpy()
    out[tag + "_argmax_hist"] = np.bincount(torch.argmax(o, 1).reshape(-1).numpy(), minlength=3)
out["sw_vol_seed"] = np.int32(5)
# BASELINE config 3 schedule: 512^3, roi 128, overlap 0.5 -> 7 starts per dim, 343 windows
per_dim, wins512 = window_starts((512, 512, 512), (128, 128, 128), 0.5)
out["sw512_starts_dim0"] = np.array(per_dim[0], dtype=np.int32)
out["sw512_nwin"] = np.int32(len(wins512))
per_dim, wins512b = window_starts((512, 512, 512), (128, 128, 128), 0.25)
out["sw512_nwin_ov25"] = np.int32(len(wins512b))

# (v) Dice metric incl. absent class -> NaN
g = torch.Generator().manual_seed(9)
p = torch.randint(0, 5, (2, 1, 6, 7, 8), generator=g)
t = torch.randint(0, 4, (2, 1, 6, 7, 8), generator=g)
d, m = ref_dice_metric(p, t, 5, include_background=False)
out["dm_pred"], out["dm_true"] = p.numpy().astype(np.uint8), t.numpy().astype(np.uint8)
out["dm_dice"], out["dm_mean"] = d.numpy(), np.float32(m.item())

# (vi) resample: geometry asserted by the reference's tests/image/test_image.py:33-52 + values
lf = np.zeros((5, 5, 5), np.uint8)
for i in range(5):
    lf[i] = i                                 # labelfield fixture: slice k filled with k
sp = (0.5, 0.6, 0.7)
half = tuple(s / 2 for s in sp)
size2 = resample_size((5, 5, 5), sp, half)
out["rs_size_half"] = np.array(size2)
out["rs_labelfield_half_linear"] = ref_resample_grid(lf, sp, (0, 0, 0), np.eye(3), size2, half, (0, 0, 0), np.eye(3), False)
out["rs_labelfield_toref_nearest"] = ref_resample_grid(lf, sp, (0, 0, 0), np.eye(3), (12, 10, 7), half,
                                                       (1.3, -2.1, 0.75), np.eye(3), True)
ramp = np.fromfunction(lambda z, y, x: 2.0 * x - 3.0 * y + 0.5 * z + 1.0, (6, 7, 8)).astype(np.float32)
out["rs_ramp"] = ramp
out["rs_ramp_out"] = ref_resample_grid(ramp, (1, 1, 1), (0, 0, 0), np.eye(3), (13, 11, 9), (0.5, 0.5, 0.5),
                                       (0.25, 0.25, 0.25), np.eye(3), False)
x = (np.random.default_rng(4).standard_normal((2, 5, 6, 7)) * 3 + 10).astype(np.float32)
out["norm_in"], out["norm_out"] = x, ref_normalize(x)

dst = Path(__file__).resolve().parent / "oracle_goldens.npz"
np.savez_compressed(dst, **out)
print("wrote", dst, dst.stat().st_size, "bytes")
