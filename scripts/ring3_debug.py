#!/usr/bin/env python3
"""locate mismatches of the ring kernel against torch for one shape / mode (bring-up aid)"""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from segmantic_amd import ops
DEV = "cuda:0"
torch.manual_seed(0)
n, c, sp = 4, 16, (16, 64, 128)
x = torch.randn((n, c) + sp); r = torch.randn((n, c) + sp)
w = torch.randn((c, c, 3, 3, 3)) * 0.05; b = torch.randn(c) * 0.1
qb = lambda t: t.bfloat16().float()
nd = lambda t: t.permute(0, 2, 3, 4, 1).contiguous().to(DEV).bfloat16()
raw = F.conv3d(qb(x), qb(w), b, padding=1)
xd, rd = nd(x), nd(r)
pk = ops.wpack(torch.bfloat16, 0, w.to(DEV), c, c, 3)
for mode in ("plain", "res", "alpha_res", "stats", "alpha_res_stats"):
    yd = torch.full_like(xd, float("nan"))
    rows = ops.conv3d_stats_rows(xd, yd, 3, 1)
    stats = torch.zeros((rows, 2, c), device=DEV)
    al = torch.tensor([0.3], device=DEV) if "alpha" in mode else None
    ref = raw
    if al is not None: ref = F.prelu(ref, torch.tensor([0.3]))
    if "res" in mode: ref = ref + qb(r)
    ops.conv3d_fwd(xd, yd, pk, None, 0, b.to(DEV), 3, 1, prelu_alpha=al, residual=rd if "res" in mode else None,
                   stats=stats if "stats" in mode else None)
    torch.cuda.synchronize()
    y = yd.float().cpu().permute(0, 4, 1, 2, 3)
    nan = torch.isnan(y)
    bad = nan | ((y - ref).abs() > 0.05)
    print(mode, "nan", int(nan.sum()), "bad", int(bad.sum()), "of", y.numel())
    if bad.any():
        idx = bad.nonzero()
        print("  first bad (n,c,z,y,x):", idx[:5].tolist(), " last:", idx[-3:].tolist())
        for d, name in ((0, "n"), (1, "c"), (2, "z"), (3, "y"), (4, "x")):
            print("  ", name, sorted(set(idx[:, d].tolist()))[:40])
