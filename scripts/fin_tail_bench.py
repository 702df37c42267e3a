"""Cost of finalising inside the producing launch vs the separate one-workgroup launch, in isolation
(HIP events around 50 back-to-back repetitions).  usage: fin_tail_bench.py"""
import sys
import torch
sys.path.insert(0, ".")
from segmantic_amd import ops

DEV = "cuda:0"
cases = [("ring 16->16 @128^3 x8", 16, 16, (128, 128, 128), 8), ("ring 16->16 @64^3 x8", 16, 16, (64, 64, 64), 8),
         ("ks 64->64 @16^3 x8", 64, 64, (16, 16, 16), 8), ("ks 256->256 @8^3 x8", 256, 256, (8, 8, 8), 8),
         ("ks 32->32 @32^3 x8", 32, 32, (32, 32, 32), 8)]
for name, cin, cout, sp, n in cases:
    x = torch.randn((n,) + sp + (cin,), device=DEV).to(torch.bfloat16)
    y = torch.empty((n,) + sp + (cout,), device=DEV, dtype=torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3, 3), device=DEV) * 0.05
    pk = ops.wpack(torch.bfloat16, 0, w, cin, cout, 3)
    rows = ops.conv3d_stats_rows(x, y, 3, 1)
    stats = torch.zeros((rows, 2, cout), device=DEV)
    outs = [torch.empty(cout, device=DEV) for _ in range(6)]
    gamma = torch.ones(cout, device=DEV)
    count = n * sp[0] * sp[1] * sp[2]
    fin = (count, gamma, gamma, outs[4], outs[5], 0.1, 1e-5, outs[0], outs[1], outs[2], outs[3])

    def fused():
        ops.conv3d_fwd(x, y, pk, None, 0, None, 3, 1, stats=stats, stats_fin=fin)

    def separate():
        ops.conv3d_fwd(x, y, pk, None, 0, None, 3, 1, stats=stats)
        ops.bn_finalize(stats, rows, cout, count, gamma, gamma, outs[4], outs[5], 0.1, 1e-5, *outs[:4])

    def plain():
        ops.conv3d_fwd(x, y, pk, None, 0, None, 3, 1, stats=stats)

    res = {}
    for nm, fn in (("conv only", plain), ("conv + separate finalize", separate), ("conv with tail", fused)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[nm] = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{name:26s} rows {rows:5d}: " + "  ".join(f"{k} {v:7.1f} us" for k, v in res.items()))
