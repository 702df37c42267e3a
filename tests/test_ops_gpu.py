"""GPU parity of every HIP op against the CPU oracle operators (torch CPU fp32 = what MONAI's
UNet / DiceLoss reduce to, see oracle/unet_ref.py).  Tolerances are written next to each check:
f32 kernels are exact-f32 MFMA chains (only the summation order differs from oneDNN), bf16
kernels are compared against the oracle evaluated on bf16-rounded inputs.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from segmantic_amd import ops  # noqa: E402

DEV = "cuda:0"
F32_RTOL = 2e-5      # f32 path: relative to max |ref|
BF16_RTOL = 1.5e-2   # bf16 path: output rounding (2^-8) + accumulate-order noise


def to_ndhwc(x_ncdhw, dtype):
    return x_ncdhw.permute(0, 2, 3, 4, 1).contiguous().to(DEV, dtype)


def from_ndhwc(t):
    return t.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def q(x, dtype):
    return x.to(dtype).float()


def relerr(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def tol(dtype):
    return F32_RTOL if dtype == torch.float32 else BF16_RTOL


CONV_CASES = [
    # cin, cout, k, s, spatial (d,h,w), batch
    (16, 16, 3, 1, (8, 12, 20), 2),
    (16, 32, 3, 2, (10, 12, 36), 1),
    (32, 32, 3, 1, (6, 6, 8), 2),
    (32, 64, 3, 2, (8, 8, 16), 1),
    (64, 64, 3, 1, (4, 4, 8), 1),
    (128, 256, 1, 1, (4, 4, 4), 2),
    (16, 48, 3, 1, (5, 9, 17), 1),
    (16, 16, 3, 1, (16, 64, 128), 4),    # z-marching ring kernel, 256 columns, 1 segment
    (16, 16, 3, 1, (33, 60, 120), 2),    # ring kernel, 2 z-segments, ragged extents
    (32, 32, 3, 1, (33, 60, 120), 2),    # ring kernel, CK=32, two output tiles
    (128, 48, 3, 1, (5, 6, 7), 2),       # k-split kernel, odd tile count, narrow + ragged tiles
    (64, 64, 3, 1, (9, 10, 40), 2),      # k-split kernel, wide tiles, two output tiles per workgroup
    (256, 32, 3, 1, (8, 8, 8), 1),       # k-split kernel, 8 bf16 / 16 f32 chunks
    (1, 16, 3, 2, (12, 12, 12), 2),   # small-Cin MFMA kernel (first layer)
    (1, 16, 3, 2, (20, 34, 70), 1),   # small-Cin, several ragged tiles
    (2, 32, 3, 1, (5, 9, 19), 2),     # small-Cin, stride 1, two output tiles, 2 k-steps
    (4, 16, 3, 2, (9, 9, 9), 1),      # small-Cin, 4 input channels (108 k-values)
    (3, 16, 3, 1, (4, 8, 16), 1),
    (16, 3, 3, 1, (6, 7, 9), 1),      # direct kernel (K=3 head)
    (4, 8, 3, 2, (9, 9, 9), 1),       # direct kernel, odd extents
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_fwd(case, dtype):
    cin, cout, k, s, sp, n = case
    x = rnd((n, cin) + sp, 1)
    w = rnd((cout, cin, k, k, k), 2, 1.0 / math.sqrt(cin * k ** 3))
    b = rnd((cout,), 3, 0.1)
    ref = F.conv3d(q(x, dtype), q(w, dtype), b, stride=s, padding=(k - 1) // 2)
    xd = to_ndhwc(x, dtype)
    yd = torch.empty((n,) + tuple(ref.shape[2:]) + (cout,), dtype=dtype, device=DEV)
    wd, bd = w.to(DEV), b.to(DEV)
    packed = ops.wpack(dtype, 0, wd, cin, cout, k) if ops.mfma_ok(cin, cout) else None
    rows = ops.conv3d_stats_rows(xd, yd, k, s)
    stats = torch.zeros((rows, 2, cout), device=DEV)
    ops.conv3d_fwd(xd, yd, packed, wd, 0, bd, k, s, stats=stats)
    torch.cuda.synchronize()
    got = from_ndhwc(yd)
    assert relerr(got, ref) < tol(dtype)
    # fused statistics: per-channel sum / sum of squares of the (pre-rounding) conv output
    ssum = stats[:, 0].double().sum(0).cpu()
    ssq = stats[:, 1].double().sum(0).cpu()
    rs = ref.double().sum((0, 2, 3, 4))
    rq = (ref.double() ** 2).sum((0, 2, 3, 4))
    cnt = ref.numel() / cout
    assert float((ssum - rs).abs().max()) / cnt < (1e-5 if dtype == torch.float32 else 2e-2) * float(ref.abs().max())
    assert float(((ssq - rq).abs() / rq).max()) < (1e-4 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,stride,sp", [(1, 16, 2, (20, 34, 70)), (2, 32, 1, (5, 9, 19)),
                                                 (4, 16, 2, (9, 9, 9))])
def test_conv3d_fwd_pair_equals_two_convs(dtype, cin, cout, stride, sp):
    """segmi_conv3d_fwd_pair (subunit 0 + residual conv of the first ResidualUnit in one launch)
    gives the very bits of two segmi_conv3d_fwd calls, statistics included."""
    n = 2
    x = rnd((n, cin) + sp, 301)
    wa, wb = rnd((cout, cin, 3, 3, 3), 302, 0.3), rnd((cout, cin, 3, 3, 3), 303, 0.3)
    ba, bb = rnd((cout,), 304, 0.1), rnd((cout,), 305, 0.1)
    alpha = torch.tensor([0.2], device=DEV)
    xd = to_ndhwc(x, dtype)
    osp = tuple((s + 2 - 3) // stride + 1 for s in sp)
    mk = lambda: torch.empty((n,) + osp + (cout,), dtype=dtype, device=DEV)
    wad, wbd, bad, bbd = wa.to(DEV), wb.to(DEV), ba.to(DEV), bb.to(DEV)
    assert ops.conv3d_pair_ok(xd, mk(), mk())
    for act_a, with_stats in ((None, True), (alpha, False)):
        ya, yb, pa, pb = mk(), mk(), mk(), mk()
        rows = ops.conv3d_stats_rows(xd, ya, 3, stride)
        st1 = torch.zeros((rows, 2, cout), device=DEV) if with_stats else None
        st2 = torch.zeros((rows, 2, cout), device=DEV) if with_stats else None
        ops.conv3d_fwd(xd, ya, None, wad, 0, bad, 3, stride, prelu_alpha=act_a, stats=st1)
        ops.conv3d_fwd(xd, yb, None, wbd, 0, bbd, 3, stride)
        ops.conv3d_fwd_pair(xd, pa, wad, bad, pb, wbd, bbd, stride, prelu_alpha_a=act_a, stats_a=st2)
        torch.cuda.synchronize()
        assert torch.equal(ya, pa) and torch.equal(yb, pb)
        if with_stats:
            real = rows - 131                                      # rows behind are reduction scratch
            assert torch.equal(st1[:real], st2[:real])
    ref = F.conv3d(q(x, dtype), q(wb, dtype), bb, stride=stride, padding=1)
    assert relerr(from_ndhwc(pb), ref) < tol(dtype)
    # MFMA-shaped layers do not qualify
    big = torch.empty((1, 4, 4, 4, 16), dtype=dtype, device=DEV)
    assert not ops.conv3d_pair_ok(big, big.clone(), big.clone())


@pytest.mark.parametrize("c,cout,with_alpha,mode", [(16, 16, 0.15, "identity"), (16, 16, None, "stats"),
                                                   (32, 32, 0.15, "stats"), (16, 32, 1.5, "plain"),
                                                   (16, 16, 0.0, "plain")])
def test_conv3d_input_transform_equals_separate_bn_pass(c, cout, with_alpha, mode):
    """segmi_in_affine: the consumer conv (forward and weight gradient) applies the producer's
    BatchNorm-apply + PReLU while staging -- bit-identical to bn_act_fwd followed by the plain calls,
    zero padding and the identity residual included."""
    dtype = torch.bfloat16
    n, sp = 4, (18, 60, 120)          # 256 ragged columns: the ring kernel takes the layer
    u = to_ndhwc(rnd((n, c) + sp, 401, 2.0), dtype)
    scale = (1 + 0.3 * rnd((c,), 402)).to(DEV)
    shift = (0.2 * rnd((c,), 403)).to(DEV)
    # slopes in [0, 1] take the max() form of the transform, others the compare / select form
    alpha = torch.tensor([with_alpha], device=DEV) if with_alpha is not None else None
    w = rnd((cout, c, 3, 3, 3), 404, 0.05).to(DEV)
    b = (0.1 * rnd((cout,), 405)).to(DEV)
    packed = ops.wpack(dtype, 0, w, c, cout, 3)
    au = torch.empty_like(u)
    ops.bn_act_fwd(u, au, scale, shift, alpha)
    y_ref = torch.empty((n,) + sp + (cout,), dtype=dtype, device=DEV)
    y = torch.empty_like(y_ref)
    assert ops.conv3d_in_affine_ok(u, y, 3, 1)
    rows = ops.conv3d_stats_rows(u, y, 3, 1)
    st_ref = torch.zeros((rows, 2, cout), device=DEV) if mode == "stats" else None
    st = torch.zeros((rows, 2, cout), device=DEV) if mode == "stats" else None
    tf = (scale, shift, alpha)
    ops.conv3d_fwd(au, y_ref, packed, None, 0, b, 3, 1, residual=au if mode == "identity" else None, stats=st_ref)
    ops.conv3d_fwd(u, y, packed, None, 0, b, 3, 1, residual=u if mode == "identity" else None, stats=st, in_tf=tf)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    if st is not None:
        assert torch.equal(st[:rows - 131], st_ref[:rows - 131])
    # weight gradient with the transform on X
    dy = to_ndhwc(rnd((n, cout) + sp, 406), dtype)
    ws = torch.empty(ops.conv3d_wgrad_workspace(au, dy, 3, 1), dtype=torch.uint8, device=DEV)
    dw_ref, dw = torch.empty_like(w), torch.empty_like(w)
    ops.conv3d_wgrad(au, dy, dw_ref, None, 3, 1, ws)
    torch.cuda.synchronize()
    ops.conv3d_wgrad(u, dy, dw, None, 3, 1, ws, in_tf=tf)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw_ref)
    # layers whose kernels cannot apply the transform say so (stride 2; 16-channel layers off the ring)
    small = torch.empty((1, 4, 4, 4, 16), dtype=dtype, device=DEV)
    assert not ops.conv3d_in_affine_ok(small, small.clone(), 3, 1)
    half = torch.empty((n,) + tuple((e + 1) // 2 for e in sp) + (cout,), dtype=dtype, device=DEV)
    assert not ops.conv3d_in_affine_ok(u, half, 3, 2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3d_epilogue_prelu_residual_and_views(dtype):
    """PReLU + residual epilogue, reading from / writing into channel slices (concat by offset)."""
    n, cin, cout, sp = 1, 16, 16, (6, 10, 18)
    x = rnd((n, cin) + sp, 5)
    w = rnd((cout, cin, 3, 3, 3), 6, 0.05)
    b = rnd((cout,), 7, 0.1)
    r = rnd((n, cout) + sp, 8)
    alpha = 0.2
    ref = F.prelu(F.conv3d(q(x, dtype), q(w, dtype), b, padding=1), torch.tensor([alpha])) + q(r, dtype)
    big_in = torch.zeros((n,) + sp + (48,), dtype=dtype, device=DEV)
    big_in[..., 16:32] = to_ndhwc(x, dtype)
    big_out = torch.zeros((n,) + sp + (32,), dtype=dtype, device=DEV)
    rd = to_ndhwc(r, dtype)
    wd, bd = w.to(DEV), b.to(DEV)
    ad = torch.tensor([alpha], device=DEV)
    packed = ops.wpack(dtype, 0, wd, cin, cout, 3)
    ops.conv3d_fwd(big_in[..., 16:32], big_out[..., 16:32], packed, None, 0, bd, 3, 1,
                   prelu_alpha=ad, residual=rd)
    torch.cuda.synchronize()
    assert relerr(from_ndhwc(big_out[..., 16:32]), ref) < tol(dtype)
    assert float(big_out[..., :16].float().abs().max()) == 0.0


@pytest.mark.parametrize("n,sp,mode", [(4, (18, 64, 120), "tf_identity"), (1, (64, 64, 128), "tf_identity"),
                                       (4, (18, 64, 120), "alpha_res_stats"), (1, (62, 60, 128), "plain"),
                                       (2, (33, 61, 125), "sums")])
def test_conv_ring3_dma_ring_ragged_and_segmented_shapes(n, sp, mode):
    """conv_ring3_kernel (LDS-DMA ring, round 4) away from the benchmark shape: extents that are not multiples of the
    (4, 8, 16) step / tile (the zero fill of partial planes, rows and columns comes from the buffer range check, the
    stores of lanes outside the tensor are dropped by it), z-split columns (several segments per column: the halo
    planes of a segment are real data of its neighbour), the in-place input transform on border workgroups (padding
    chunks must stay zero), every epilogue.  Against torch on the bf16-rounded operands."""
    dtype = torch.bfloat16
    c = 16
    x, r = rnd((n, c) + sp, 521), rnd((n, c) + sp, 522)
    w, b = rnd((c, c, 3, 3, 3), 523, 0.05), rnd((c,), 524, 0.1)
    xd, rd = to_ndhwc(x, dtype), to_ndhwc(r, dtype)
    yd = torch.full_like(xd, float("nan"))
    assert "ring3" in ops.conv3d_fwd_kernel_name(xd, yd, 3, 1), ops.conv3d_fwd_kernel_name(xd, yd, 3, 1)
    if mode == "tf_identity":
        scale, shift = (rnd((c,), 525) * 0.3 + 1.0).to(DEV), (rnd((c,), 526) * 0.2).to(DEV)
        alpha = torch.tensor([0.2], device=DEV)
        t = q(x, dtype) * scale.cpu().view(1, -1, 1, 1, 1) + shift.cpu().view(1, -1, 1, 1, 1)
        t = q(torch.where(t > 0, t, 0.2 * t), dtype)
        ref = F.conv3d(t, q(w, dtype), b, padding=1) + t
        ops.conv3d_fwd(xd, yd, ops.wpack(dtype, 0, w.to(DEV), c, c, 3), None, 0, b.to(DEV), 3, 1, residual=xd,
                       in_tf=(scale, shift, alpha))
    elif mode == "alpha_res_stats":
        raw = F.conv3d(q(x, dtype), q(w, dtype), b, padding=1)
        ref = F.prelu(raw, torch.tensor([0.3])) + q(r, dtype)
        rows = ops.conv3d_stats_rows(xd, yd, 3, 1)
        stats = torch.zeros((rows, 2, c), device=DEV)
        ops.conv3d_fwd(xd, yd, ops.wpack(dtype, 0, w.to(DEV), c, c, 3), None, 0, b.to(DEV), 3, 1,
                       prelu_alpha=torch.tensor([0.3], device=DEV), residual=rd, stats=stats)
        torch.cuda.synchronize()
        ssum, ssq = stats[:, 0].double().sum(0).cpu(), stats[:, 1].double().sum(0).cpu()
        cnt = raw.numel() / c
        assert float((ssum - raw.double().sum((0, 2, 3, 4))).abs().max()) / cnt < 2e-2 * float(raw.abs().max())
        assert float((ssq - (raw.double() ** 2).sum((0, 2, 3, 4))).abs().max()) / cnt < 2e-2 * float(raw.abs().max()) ** 2
    elif mode == "plain":
        ref = F.conv3d(q(x, dtype), q(w, dtype), None, padding=1)
        ops.conv3d_fwd(xd, yd, ops.wpack(dtype, 0, w.to(DEV), c, c, 3), None, 0, None, 3, 1)
    else:       # input gradient + the BatchNorm-backward sums of the layer its output flows into
        ref = F.conv_transpose3d(q(x, dtype), q(w, dtype), None, padding=1) + q(x, dtype)      # dgrad + identity residual
        mean, invstd = (rnd((c,), 527) * 0.3).to(DEV), (rnd((c,), 528).abs() + 0.5).to(DEV)
        gamma, beta = (rnd((c,), 529) + 1.5).to(DEV), (rnd((c,), 530) * 0.3).to(DEV)
        rows = ops.conv3d_stats_rows(xd, yd, 3, 1)
        part = torch.full((rows, 3, c), float("nan"), device=DEV)
        dg, db, coef = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty((2, c), device=DEV)
        ops.conv3d_fwd(xd, yd, ops.wpack(dtype, 1, w.to(DEV), c, c, 3), None, 1, None, 3, 1, residual=xd,
                       bn_bwd=(rd, mean, invstd, gamma, beta, None, part),
                       bn_bwd_fin=(n * sp[0] * sp[1] * sp[2], dg, db, None, coef))
        torch.cuda.synchronize()
        gq = from_ndhwc(yd).double()                               # the stored gradient the sums are taken of
        xhat = (q(r, dtype).double() - mean.cpu().double().view(1, -1, 1, 1, 1)) * invstd.cpu().double().view(1, -1, 1, 1, 1)
        want_dg, want_db = (gq * xhat).sum((0, 2, 3, 4)), gq.sum((0, 2, 3, 4))
        assert float((dg.cpu().double() - want_dg).abs().max()) < 2e-4 * float(want_dg.abs().max()) + 1e-3 * float(want_dg.abs().mean())
        assert float((db.cpu().double() - want_db).abs().max()) < 2e-4 * float(want_db.abs().max()) + 1e-3 * float(want_db.abs().mean())
    torch.cuda.synchronize()
    got = from_ndhwc(yd)
    assert bool(torch.isfinite(got).all())                          # every voxel written (the output started as NaN)
    assert relerr(got, ref) < BF16_RTOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3d_ring_kernel_epilogue(dtype):
    """ring kernel (full-resolution shapes): PReLU + prefetched residual + fused statistics"""
    n, c, sp = 4, 16, (16, 64, 128)
    x, r = rnd((n, c) + sp, 15), rnd((n, c) + sp, 16)
    w, b = rnd((c, c, 3, 3, 3), 17, 0.05), rnd((c,), 18, 0.1)
    raw = F.conv3d(q(x, dtype), q(w, dtype), b, padding=1)
    ref = F.prelu(raw, torch.tensor([0.3])) + q(r, dtype)
    xd, rd = to_ndhwc(x, dtype), to_ndhwc(r, dtype)
    yd = torch.empty_like(xd)
    wd = w.to(DEV)
    packed = ops.wpack(dtype, 0, wd, c, c, 3)
    rows = ops.conv3d_stats_rows(xd, yd, 3, 1)
    stats = torch.zeros((rows, 2, c), device=DEV)
    ops.conv3d_fwd(xd, yd, packed, None, 0, b.to(DEV), 3, 1, prelu_alpha=torch.tensor([0.3], device=DEV),
                   residual=rd, stats=stats)
    torch.cuda.synchronize()
    assert relerr(from_ndhwc(yd), ref) < tol(dtype)
    ssum = stats[:, 0].double().sum(0).cpu()
    assert float((ssum - raw.double().sum((0, 2, 3, 4))).abs().max()) / (raw.numel() / c) < \
        (1e-5 if dtype == torch.float32 else 2e-2) * float(raw.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3d_ksplit_kernel_epilogue(dtype):
    """k-split kernel (deep layers): PReLU + residual + fused statistics, channel-slice views"""
    n, cin, cout, sp = 2, 128, 64, (6, 9, 11)
    x, r = rnd((n, cin) + sp, 35), rnd((n, cout) + sp, 36)
    w, b = rnd((cout, cin, 3, 3, 3), 37, 0.02), rnd((cout,), 38, 0.1)
    raw = F.conv3d(q(x, dtype), q(w, dtype), b, padding=1)
    ref = F.prelu(raw, torch.tensor([0.3])) + q(r, dtype)
    big_in = torch.zeros((n,) + sp + (192,), dtype=dtype, device=DEV)
    big_in[..., 64:192] = to_ndhwc(x, dtype)
    yd = torch.empty((n,) + sp + (cout,), dtype=dtype, device=DEV)
    rd = to_ndhwc(r, dtype)
    wd = w.to(DEV)
    packed = ops.wpack(dtype, 0, wd, cin, cout, 3)
    rows = ops.conv3d_stats_rows(big_in[..., 64:192], yd, 3, 1)
    stats = torch.zeros((rows, 2, cout), device=DEV)
    ops.conv3d_fwd(big_in[..., 64:192], yd, packed, None, 0, b.to(DEV), 3, 1,
                   prelu_alpha=torch.tensor([0.3], device=DEV), residual=rd, stats=stats)
    torch.cuda.synchronize()
    assert relerr(from_ndhwc(yd), ref) < tol(dtype)
    ssum = stats[:, 0].double().sum(0).cpu()
    assert float((ssum - raw.double().sum((0, 2, 3, 4))).abs().max()) / (raw.numel() / cout) < \
        (1e-5 if dtype == torch.float32 else 2e-2) * float(raw.abs().max())
    ssq = stats[:, 1].double().sum(0).cpu()
    assert float(((ssq - (raw.double() ** 2).sum((0, 2, 3, 4))).abs() / (raw.double() ** 2).sum((0, 2, 3, 4))).max()) < \
        (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c,sp", [(16, (16, 64, 128)), (32, (20, 30, 70))])
def test_conv3d_ring_kernel_identity_residual(dtype, c, sp):
    """out = conv(x) + x with residual == input view: the ring kernel takes the residual rows
    from its LDS planes; must equal the separately-loaded residual bit for bit."""
    n = 2
    x = rnd((n, c) + sp, 19)
    w, b = rnd((c, c, 3, 3, 3), 20, 0.05), rnd((c,), 21, 0.1)
    ref = F.conv3d(q(x, dtype), q(w, dtype), b, padding=1) + q(x, dtype)
    xd = to_ndhwc(x, dtype)
    x2 = xd.clone()
    y1, y2 = torch.empty_like(xd), torch.empty_like(xd)
    wd = w.to(DEV)
    packed = ops.wpack(dtype, 0, wd, c, c, 3)
    ops.conv3d_fwd(xd, y1, packed, None, 0, b.to(DEV), 3, 1, residual=xd)    # LDS residual
    ops.conv3d_fwd(xd, y2, packed, None, 0, b.to(DEV), 3, 1, residual=x2)    # HBM residual
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    assert relerr(from_ndhwc(y1), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(16, 16, (6, 8, 20), 2), (32, 16, (4, 6, 8), 1), (4, 8, (5, 5, 5), 1)])
def test_conv3d_dgrad_s1(case, dtype):
    """stride-1 dgrad = conv with flipped / transposed weights (pack kind 1)."""
    cin, cout, sp, n = case
    w = rnd((cout, cin, 3, 3, 3), 11, 0.05)
    dy = rnd((n, cout) + sp, 12)
    ref = F.conv_transpose3d(q(dy, dtype), q(w, dtype), stride=1, padding=1)
    dyd = to_ndhwc(dy, dtype)
    dxd = torch.empty((n,) + sp + (cin,), dtype=dtype, device=DEV)
    wd = w.to(DEV)
    packed = ops.wpack(dtype, 1, wd, cout, cin, 3) if ops.mfma_ok(cin, cout) else None
    ops.conv3d_fwd(dyd, dxd, packed, wd, 1, None, 3, 1)
    torch.cuda.synchronize()
    assert relerr(from_ndhwc(dxd), ref) < tol(dtype)


CONVT_CASES = [
    (32, 16, (6, 6, 20), 2, 0),
    (64, 16, (4, 6, 8), 1, 0),
    (16, 32, (5, 4, 9), 1, 0),
    (16, 16, (3, 5, 17), 1, 1),   # odd output extent (2*in - 1): dgrad of a conv on odd input
    (384, 64, (4, 4, 4), 1, 0),
    (8, 4, (4, 5, 6), 2, 0),      # direct kernel
    (64, 16, (5, 3, 33), 1, 0),   # persistent kernel, two bf16 chunks, ragged x tiles
    (32, 16, (3, 4, 16), 2, 1),   # persistent kernel, odd output extent
    (32, 16, (24, 40, 48), 3, 0), # persistent kernel, several tiles per workgroup + ragged tail
    (32, 32, (5, 6, 36), 2, 0),   # persistent kernel, two output tiles (K = 32 decoder top)
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_wpack_batch_is_bit_identical_to_single_packs(dtype):
    """One batched launch (every layer, every kind) == the per-layer packs, byte for byte."""
    g = torch.Generator(device="cpu").manual_seed(3)
    entries, singles = [], []
    for kind, cin, cout, k, scaled in [(0, 16, 32, 3, False), (1, 32, 16, 3, False), (0, 64, 16, 1, True),
                                       (2, 32, 16, 3, False), (2, 48, 32, 3, True), (1, 16, 16, 1, False),
                                       (0, 256, 128, 3, False)]:
        if kind == 0:
            w = torch.randn(cout, cin, k, k, k, generator=g).to(DEV)
        else:   # kinds 1/2 read [cin][cout][taps]
            w = torch.randn(cin, cout, k, k, k, generator=g).to(DEV)
        sc = (torch.rand(cout, generator=g) + 0.5).to(DEV) if scaled else None
        entries.append((kind, w, sc, cin, cout, k))
        singles.append(ops.wpack(dtype, kind, w, cin, cout, k, scale=sc))
    batch = ops.WpackBatch(dtype, entries)
    for rep in range(2):          # second run re-uses the uploaded table
        for buf in batch.packed:
            buf.fill_(0xAB)
        batch.run()
        torch.cuda.synchronize()
        for got, want in zip(batch.packed, singles):
            assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONVT_CASES)
def test_convT3d_fwd(case, dtype):
    cin, cout, sp, n, odd = case
    x = rnd((n, cin) + sp, 21)
    w = rnd((cin, cout, 3, 3, 3), 22, 1.0 / math.sqrt(cin * 27 / 8))
    b = rnd((cout,), 23, 0.1)
    ref = F.conv_transpose3d(q(x, dtype), q(w, dtype), b, stride=2, padding=1,
                             output_padding=0 if odd else 1)
    xd = to_ndhwc(x, dtype)
    yd = torch.empty((n,) + tuple(ref.shape[2:]) + (cout,), dtype=dtype, device=DEV)
    wd, bd = w.to(DEV), b.to(DEV)
    packed = ops.wpack(dtype, 2, wd, cin, cout, 3) if ops.mfma_ok(cin, cout) else None
    rows = ops.convT3d_stats_rows(xd, yd)
    stats = torch.zeros((rows, 2, cout), device=DEV)
    ops.convT3d_fwd(xd, yd, packed, wd, bd, stats=stats)
    torch.cuda.synchronize()
    assert relerr(from_ndhwc(yd), ref) < tol(dtype)
    ssum = stats[:, 0].double().sum(0).cpu()
    rs = ref.double().sum((0, 2, 3, 4))
    assert float((ssum - rs).abs().max()) / (ref.numel() / cout) < (1e-5 if dtype == torch.float32 else 2e-2) * float(ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(32, 16, (4, 6, 20), 2), (64, 16, (3, 3, 16), 1), (128, 32, (4, 4, 8), 1),
                                  (32, 32, (3, 4, 18), 1)])
def test_convT3d_epilogue_prelu_residual(case, dtype):
    """PReLU + residual epilogue (the form the dgrad of a stride-2 conv uses) on both kernels."""
    cin, cout, sp, n = case
    x = rnd((n, cin) + sp, 24)
    w = rnd((cin, cout, 3, 3, 3), 25, 1.0 / math.sqrt(cin * 27 / 8))
    b = rnd((cout,), 26, 0.1)
    osp = tuple(2 * d for d in sp)
    res = rnd((n, cout) + osp, 27)
    alpha = torch.tensor([0.2])
    ref = F.conv_transpose3d(q(x, dtype), q(w, dtype), b, stride=2, padding=1, output_padding=1)
    ref = F.prelu(ref, alpha) + q(res, dtype)
    xd = to_ndhwc(x, dtype)
    rd = to_ndhwc(res, dtype)
    yd = torch.empty((n,) + osp + (cout,), dtype=dtype, device=DEV)
    wd = w.to(DEV)
    packed = ops.wpack(dtype, 2, wd, cin, cout, 3)
    ops.convT3d_fwd(xd, yd, packed, wd, b.to(DEV), prelu_alpha=alpha.to(DEV), residual=rd)
    torch.cuda.synchronize()
    assert relerr(from_ndhwc(yd), ref) < tol(dtype)


WGRAD_CASES = [
    (16, 16, 3, 1, (4, 16, 32), 2),
    (16, 32, 3, 2, (8, 8, 32), 1),
    (32, 32, 3, 1, (4, 8, 8), 2),
    (64, 32, 3, 2, (8, 16, 16), 1),
    (32, 64, 3, 1, (5, 7, 9), 1),
    (16, 64, 3, 2, (8, 8, 32), 1),    # bf16: 4 output tiles share one X staging
    (16, 64, 3, 1, (4, 8, 16), 1),
    (16, 32, 1, 1, (4, 8, 16), 2),    # k1, 2x1 blocking
    (16, 96, 3, 1, (3, 5, 7), 1),     # 2x1 blocking with 3 output chunks, narrow tiles
    (1, 16, 3, 2, (12, 12, 12), 2),   # small-Cin MFMA kernel
    (1, 16, 3, 2, (20, 36, 70), 1),   # small-Cin, many ragged tiles
    (2, 16, 3, 1, (5, 9, 19), 2),     # small-Cin, stride 1
    (4, 32, 3, 2, (8, 10, 34), 1),    # small-Cin, two output-channel tiles, 7 combo tiles
    (3, 16, 3, 1, (4, 8, 16), 1),
    (128, 256, 1, 1, (4, 4, 4), 2),   # direct (k1)
    (16, 3, 3, 1, (6, 6, 6), 1),      # direct
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv3d_wgrad(case, dtype):
    cin, cout, k, s, sp, n = case
    x = rnd((n, cin) + sp, 31)
    osp = tuple((d + 2 * ((k - 1) // 2) - k) // s + 1 for d in sp)
    dy = rnd((n, cout) + osp, 32)
    xq = q(x, dtype).requires_grad_(False)
    w0 = torch.zeros((cout, cin, k, k, k), requires_grad=True)
    b0 = torch.zeros((cout,), requires_grad=True)
    F.conv3d(xq, w0, b0, stride=s, padding=(k - 1) // 2).backward(q(dy, dtype))
    xd, dyd = to_ndhwc(x, dtype), to_ndhwc(dy, dtype)
    dw = torch.empty_like(w0, device=DEV)
    db = torch.empty_like(b0, device=DEV)
    ws = torch.empty(ops.conv3d_wgrad_workspace(xd, dyd, k, s), dtype=torch.uint8, device=DEV)
    ops.conv3d_wgrad(xd, dyd, dw, db, k, s, ws)
    torch.cuda.synchronize()
    # bf16 inputs are exact products in f32; only summation order differs
    assert relerr(dw.cpu(), w0.grad) < 5e-5
    assert relerr(db.cpu(), b0.grad) < 5e-5


@pytest.mark.parametrize("case", [(16, 16, 3, 1, (32, 32, 32), 2), (32, 64, 3, 2, (16, 16, 16), 2)])
def test_wgrad_grid_budget_changes_the_partition_not_the_gradient(case):
    """The `cus` argument of segmi_conv3d_wgrad: the weight-gradient kernels size their grids (and their partial
    slabs) for that many compute units -- the engine gives them half the chip beside the main chain.  Any budget
    yields the gradient of the torch reference (another partition of the same f32 sums); the budget is a per-call
    argument (no process-wide state), clamped to multiples of 8 in [8, 256], <= 0 = the whole chip."""
    cin, cout, k, s, sp, n = case
    dtype = torch.bfloat16
    x = rnd((n, cin) + sp, 131)
    osp = tuple((d + 2 * ((k - 1) // 2) - k) // s + 1 for d in sp)
    dy = rnd((n, cout) + osp, 132)
    w0 = torch.zeros((cout, cin, k, k, k), requires_grad=True)
    F.conv3d(q(x, dtype), w0, None, stride=s, padding=(k - 1) // 2).backward(q(dy, dtype))
    xd, dyd = to_ndhwc(x, dtype), to_ndhwc(dy, dtype)
    if "SEGMI_WGRAD_CUS" not in os.environ:
        assert [ops.wgrad_cus(c) for c in (0, -5, 3, 64, 100, 1000)] == [256, 256, 8, 64, 96, 256]
    got = {}
    sizes = {}
    for cus in (0, 128, 64, 8):
        dw = torch.full_like(w0, float("nan"), device=DEV)
        sizes[cus] = ops.conv3d_wgrad_workspace(xd, dyd, k, s, cus)
        ws = torch.empty(sizes[cus], dtype=torch.uint8, device=DEV)
        ops.conv3d_wgrad(xd, dyd, dw, None, k, s, ws, cus=cus)
        torch.cuda.synchronize()
        assert relerr(dw.cpu(), w0.grad) < 5e-5, cus
        got[cus] = dw.cpu()
    assert float((got[0] - got[64]).abs().max()) <= 1e-4 * float(got[0].abs().max())
    if "SEGMI_WGRAD_CUS" not in os.environ:
        assert sizes[8] < sizes[0]          # fewer slabs for a smaller budget


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_convT3d_wgrad_via_conv_wgrad(dtype):
    """ConvTranspose3d weight grad == stride-2 conv wgrad with x := dy_T, dy := x_T."""
    n, cin, cout, sp = 1, 32, 16, (4, 6, 10)
    x = rnd((n, cin) + sp, 41)
    w0 = torch.zeros((cin, cout, 3, 3, 3), requires_grad=True)
    y = F.conv_transpose3d(q(x, dtype), w0, stride=2, padding=1, output_padding=1)
    dy = rnd(tuple(y.shape), 42)
    y.backward(q(dy, dtype))
    xd, dyd = to_ndhwc(x, dtype), to_ndhwc(dy, dtype)
    dw = torch.empty_like(w0, device=DEV)
    ws = torch.empty(ops.conv3d_wgrad_workspace(dyd, xd, 3, 2), dtype=torch.uint8, device=DEV)
    ops.conv3d_wgrad(dyd, xd, dw, None, 3, 2, ws)
    torch.cuda.synchronize()
    assert relerr(dw.cpu(), w0.grad) < 5e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c", [16, 32, 3])
def test_bn_prelu_fwd_bwd(dtype, c):
    n, sp = 2, (6, 10, 14)
    x = rnd((n, c) + sp, 51, 2.0)
    gamma = 1 + 0.2 * rnd((c,), 52)
    beta = 0.1 * rnd((c,), 53)
    alpha = torch.tensor([0.25])
    res = rnd((n, c) + sp, 54)
    dy = rnd((n, c) + sp, 55)
    xq = q(x, dtype).requires_grad_(True)
    g0, b0, a0 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    z = F.batch_norm(xq, rm, rv, g0, b0, training=True, momentum=0.1, eps=1e-5)
    y = F.prelu(z, a0) + q(res, dtype)
    y.backward(q(dy, dtype))

    xd = to_ndhwc(x, dtype)
    rows = ops.bn_stats_rows(xd)
    part = torch.empty((rows, 2, c), device=DEV)
    ops.bn_stats(xd, part)
    mean, invstd, scale, shift = (torch.empty(c, device=DEV) for _ in range(4))
    rmd, rvd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    gd, bd, ad = gamma.to(DEV), beta.to(DEV), alpha.to(DEV)
    count = n * sp[0] * sp[1] * sp[2]
    ops.bn_finalize(part, rows, c, count, gd, bd, rmd, rvd, 0.1, 1e-5, mean, invstd, scale, shift)
    yd = torch.empty_like(xd)
    ops.bn_act_fwd(xd, yd, scale, shift, ad, to_ndhwc(res, dtype))
    torch.cuda.synchronize()
    t = tol(dtype)
    assert relerr(from_ndhwc(yd), y.detach()) < t
    assert relerr(rmd.cpu(), rm) < 1e-5 and relerr(rvd.cpu(), rv) < 1e-5
    # backward
    dyd = to_ndhwc(dy, dtype)
    rrows = ops.bn_act_bwd_rows(xd)
    rp = torch.empty((rrows, 3, c), device=DEV)
    ops.bn_act_bwd_reduce(dyd, xd, mean, invstd, gd, bd, ad, rp)
    dg, dbt, coef = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty((2, c), device=DEV)
    da = torch.empty(1, device=DEV)
    ops.bn_act_bwd_finalize(rp, rrows, c, count, gd, invstd, dg, dbt, da, coef)
    dxd = torch.empty_like(xd)
    ops.bn_act_bwd_apply(dyd, xd, dxd, mean, invstd, gd, bd, ad, coef)
    torch.cuda.synchronize()
    assert relerr(dg.cpu(), g0.grad) < 1e-4
    assert relerr(dbt.cpu(), b0.grad) < 1e-4
    assert relerr(da.cpu(), a0.grad) < 1e-4
    assert relerr(from_ndhwc(dxd), xq.grad) < (1e-4 if dtype == torch.float32 else 1.5e-2)


def _drop_mask(numel, p, seed):
    """numpy restatement of norm_act.hip drop_mult (keep -> 1/(1-p), drop -> 0) over element indices."""
    e = np.arange(numel, dtype=np.uint64)
    h = ((e & 0xFFFFFFFF) * 0x9E3779B1) & 0xFFFFFFFF
    h ^= np.uint64(seed)
    h ^= ((e >> np.uint64(32)) * 0x85EBCA77) & 0xFFFFFFFF
    h ^= h >> np.uint64(16); h = (h * 0x7FEB352D) & 0xFFFFFFFF
    h ^= h >> np.uint64(15); h = (h * 0x846CA68B) & 0xFFFFFFFF
    h ^= h >> np.uint64(16)
    keep = (h >> np.uint64(8)) >= np.uint64(int(np.float32(p) * np.float32(16777216.0)))
    return keep.astype(np.float32) / (1.0 - p)


@pytest.mark.parametrize("c", [3, 16, 64])
def test_bn_dropout_prelu_fwd_bwd(c):
    """ADN ordering "NDA" (MONAI blocks/adn.py as used by monai_unet.py:83-92): norm -> dropout -> act.
    The mask is a counter hash of the logical element index, recomputed in backward."""
    dtype = torch.float32
    n, sp = 2, (6, 10, 14)
    pdrop, seed = 0.3, 0xC0FFEE
    x = rnd((n, c) + sp, 151, 2.0)
    gamma = 1 + 0.2 * rnd((c,), 152)
    beta = 0.1 * rnd((c,), 153)
    alpha = torch.tensor([0.25])
    dy = rnd((n, c) + sp, 155)
    count = n * sp[0] * sp[1] * sp[2]
    mask_ndhwc = torch.from_numpy(_drop_mask(count * c, pdrop, seed)).reshape((n,) + sp + (c,))
    mask = mask_ndhwc.permute(0, 4, 1, 2, 3)
    assert abs(float((mask > 0).float().mean()) - (1 - pdrop)) < 0.02
    xq = x.clone().requires_grad_(True)
    g0, b0, a0 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    z = F.batch_norm(xq, None, None, g0, b0, training=True, eps=1e-5)
    y = F.prelu(z * mask, a0)
    y.backward(dy)

    xd = to_ndhwc(x, dtype)
    rows = ops.bn_stats_rows(xd)
    part = torch.empty((rows, 2, c), device=DEV)
    ops.bn_stats(xd, part)
    mean, invstd, scale, shift = (torch.empty(c, device=DEV) for _ in range(4))
    rmd, rvd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    gd, bd, ad = gamma.to(DEV), beta.to(DEV), alpha.to(DEV)
    ops.bn_finalize(part, rows, c, count, gd, bd, rmd, rvd, 0.1, 1e-5, mean, invstd, scale, shift)
    yd = torch.empty_like(xd)
    ops.bn_act_fwd(xd, yd, scale, shift, ad, None, dropout=(pdrop, seed))
    torch.cuda.synchronize()
    got = from_ndhwc(yd)
    assert torch.equal(got == 0, (mask == 0) | (y.detach() == 0))     # the very same voxels are dropped
    assert relerr(got, y.detach()) < 1e-5
    dyd = to_ndhwc(dy, dtype)
    rrows = ops.bn_act_bwd_rows(xd)
    rp = torch.empty((rrows, 3, c), device=DEV)
    ops.bn_act_bwd_reduce(dyd, xd, mean, invstd, gd, bd, ad, rp, dropout=(pdrop, seed))
    dg, dbt, coef = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty((2, c), device=DEV)
    da = torch.empty(1, device=DEV)
    ops.bn_act_bwd_finalize(rp, rrows, c, count, gd, invstd, dg, dbt, da, coef)
    dxd = torch.empty_like(xd)
    ops.bn_act_bwd_apply(dyd, xd, dxd, mean, invstd, gd, bd, ad, coef, dropout=(pdrop, seed))
    torch.cuda.synchronize()
    assert relerr(dg.cpu(), g0.grad) < 1e-4
    assert relerr(dbt.cpu(), b0.grad) < 1e-4
    assert relerr(da.cpu(), a0.grad) < 1e-4
    assert relerr(from_ndhwc(dxd), xq.grad) < 1e-4
    # p = 0 is the identity (no mask evaluated)
    y0 = torch.empty_like(xd)
    ops.bn_act_fwd(xd, y0, scale, shift, ad, None, dropout=(0.0, seed))
    y1 = torch.empty_like(xd)
    ops.bn_act_fwd(xd, y1, scale, shift, ad, None)
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k", [2, 3, 16, 32])
def test_softmax_dice(dtype, k):
    from oracle.unet_ref import ref_dice_loss
    n, sp = 2, (12, 20, 24)
    lg = rnd((n, k) + sp, 61, 3.0)
    g = torch.Generator().manual_seed(62)
    lab = torch.randint(0, k, (n, 1) + sp, generator=g).float()
    lq = q(lg, dtype).requires_grad_(True)
    loss = ref_dice_loss(lq, lab)
    loss.backward()
    ld = to_ndhwc(lg, dtype)
    labd = lab.to(DEV).reshape(-1).contiguous()
    chunks = ops.dice_chunks(ld)
    part = torch.empty((n, chunks, 3, k), device=DEV)
    coef = torch.empty((n, 2, k), device=DEV)
    out = torch.empty(1, device=DEV)
    ops.softmax_dice_fwd(ld, labd, part, coef, out)
    dl = torch.empty_like(ld)
    ops.softmax_dice_bwd(ld, labd, coef, 1.0, dl)
    torch.cuda.synchronize()
    assert abs(float(out.cpu()) - float(loss)) < 1e-6 * max(1.0, abs(float(loss)))  # loss within 1e-4 rel gate
    assert relerr(from_ndhwc(dl), lq.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
    # the same pass can also deliver sum_voxels dlogits (bias gradient of the logits' producer)
    dl2 = torch.empty_like(ld)
    db = torch.empty(k, device=DEV)
    ops.softmax_dice_bwd(ld, labd, coef, 1.0, dl2, scratch=part, bias_grad=db)
    torch.cuda.synchronize()
    assert torch.equal(dl2, dl)
    ref_db = lq.grad.double().sum((0, 2, 3, 4))
    assert float((db.cpu().double() - ref_db).abs().max()) < 1e-5 + (1e-4 if dtype == torch.float32 else 1e-2) * float(lq.grad.abs().sum() / k)


def test_adam_sgd_match_torch():
    n = 100003
    p0, g = rnd((n,), 71), rnd((n,), 72, 0.1)
    for amsgrad in (False, True):
        pr = p0.clone().requires_grad_(True)
        opt = torch.optim.Adam([pr], lr=1e-3, amsgrad=amsgrad)
        pd, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        vm = torch.zeros(n, device=DEV) if amsgrad else None
        for step in range(1, 4):
            pr.grad = g * step
            opt.step()
            ops.adam_step(pd, (g * step).to(DEV), m, v, vm, 1e-3, 0.9, 0.999, 1e-8, 0.0, step)
        torch.cuda.synchronize()
        assert float((pd.cpu() - pr.detach()).abs().max()) < 2e-7
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=1e-2, momentum=0.9)
    pd, buf = p0.to(DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        pr.grad = g * step
        opt.step()
        ops.sgd_step(pd, (g * step).to(DEV), buf, 1e-2, 0.9, 0.0, step == 1)
    torch.cuda.synchronize()
    assert float((pd.cpu() - pr.detach()).abs().max()) < 2e-7


@pytest.mark.parametrize("decouple", [False, True])
@pytest.mark.parametrize("eps", [1e-8, 1e-16])
def test_adabelief_matches_adabelief_pytorch_rule(decouple, eps):
    """reference monai_unet.py:305-314: AdaBelief(lr, eps, betas=(0.9, 0.999), weight_decouple,
    fixed_decay=False, rectify=False) -- known-answer test against oracle/optim_ref.py, including
    the in-place ``exp_avg_var.add_(eps)`` that lets eps accumulate in the stored second moment."""
    from oracle.optim_ref import RefAdaBelief
    n = 50021
    p0, g = rnd((n,), 75).numpy(), rnd((n,), 76, 0.1).numpy()
    for wd in (0.0, 1e-2):
        ref = RefAdaBelief(n, lr=1e-3, eps=eps, weight_decay=wd, weight_decouple=decouple)
        pr = p0.copy()
        pd = torch.from_numpy(p0).to(DEV)
        m, s = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        for step in range(1, 4):
            gs = (g * step).astype(np.float32)
            pr = ref.step(pr, gs)
            ops.adabelief_step(pd, torch.from_numpy(gs).to(DEV), m, s, 1e-3, 0.9, 0.999, eps, wd,
                               decouple, step)
        torch.cuda.synchronize()
        # f32 update of O(1) parameters: 1 ulp of the parameter + fma-vs-separate rounding
        assert float(np.abs(pd.cpu().numpy() - pr).max()) < 3e-7, (decouple, eps, wd)
        assert float(np.abs(m.cpu().numpy() - ref.m).max()) < 1e-7
        rel = np.abs(s.cpu().numpy() - ref.s) / np.maximum(np.abs(ref.s), 1e-30)
        assert float(rel.max()) < 1e-5
        # the stored second moment carries the accumulated eps (3 steps: ~eps*(1+b2+b2^2))
        assert float(s.min()) >= eps * 2.9


def test_flat_adabelief_from_reference_optimizer_dict():
    """make_optimizer() with the reference's ``optimizer`` dict (monai_unet.py:83-90) drives the
    same kernel over the arena."""
    from oracle.optim_ref import RefAdaBelief
    from segmantic_amd.seg.optim import make_optimizer
    n = 4099
    p0, g = rnd((n,), 77).numpy(), rnd((n,), 78, 0.1).numpy()
    flat, grad = torch.from_numpy(p0).to(DEV), torch.from_numpy(g).to(DEV)
    opt = make_optimizer({"optimizer": "AdaBelief", "lr": 2e-3, "epsilon": 1e-16,
                          "weight_decouple": True}, flat, grad)
    ref = RefAdaBelief(n, lr=2e-3, eps=1e-16, weight_decouple=True)
    pr = p0.copy()
    for _ in range(2):
        pr = ref.step(pr, g)
        opt.step()
    torch.cuda.synchronize()
    assert float(np.abs(flat.cpu().numpy() - pr).max()) < 3e-7


def test_argmax_bit_exact_with_ties():
    g = torch.Generator().manual_seed(81)
    for K in (7, 4, 16, 32):     # scalar kernel (7) and the K/4-lanes-per-voxel kernel
        lg = torch.randint(-3, 4, (2, K, 5, 6, 9), generator=g).float()  # many exact ties
        lg[0, K // 2, 1, 2, 3] = float("nan")                             # NaN counts as maximal
        lg[1, 1, 0, 0, 0] = float("nan"); lg[1, K - 1, 0, 0, 0] = float("nan")   # first NaN wins
        ref = torch.argmax(lg, dim=1)
        for dt in (torch.float32, torch.bfloat16):
            ld = to_ndhwc(lg, dt)
            for ldt in (torch.uint8, torch.int16, torch.int32):
                lab = torch.empty((2, 5, 6, 9), dtype=ldt, device=DEV)
                ops.argmax(ld, lab)
                torch.cuda.synchronize()
                assert torch.equal(lab.cpu().long(), ref), (K, dt, ldt)


def test_sliding_window_ops_match_oracle():
    from oracle.sliding_ref import ref_sliding_window_inference, window_starts
    img = rnd((1, 1, 20, 27, 33), 91)
    K, roi = 4, (16, 16, 16)
    wts = rnd((K, 1, 3, 3, 3), 92)

    def predictor(x):
        return F.conv3d(x, wts, padding=1)

    for overlap in (0.25, 0.5):
        ref, cnt_ref, wins = ref_sliding_window_inference(img, roi, 4, predictor, overlap)
        imd = img.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
        acc = torch.zeros((1, 20, 27, 33, K), device=DEV)
        cnt = torch.zeros((20, 27, 33), device=DEV)
        for g0 in range(0, len(wins), 4):
            grp = wins[g0:g0 + 4]
            wd = torch.empty((len(grp),) + roi + (1,), device=DEV)
            ops.sw_gather(imd, 0, grp, wd)
            pred = predictor(wd.cpu().permute(0, 4, 1, 2, 3))  # oracle predictor: isolates the data movement
            pd = pred.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
            ops.sw_scatter_add(pd, grp, acc, cnt)
        lab = torch.empty((20, 27, 33), dtype=torch.uint8, device=DEV)
        ops.sw_finalize(acc, cnt, lab, write_logits=True)
        torch.cuda.synchronize()
        assert torch.equal(cnt.cpu(), cnt_ref[0, 0])
        got = acc.cpu().permute(0, 4, 1, 2, 3)
        assert torch.equal(got, ref)  # same f32 accumulation order -> bit exact
        assert torch.equal(lab.cpu().long(), torch.argmax(ref, 1)[0])


def test_sw_gather_more_than_16_windows_per_group():
    """a window group larger than the kernel's 16 origins per launch is gathered in several launches"""
    img = rnd((1, 1, 20, 27, 33), 95)
    imd = img.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    roi = (8, 8, 8)
    wins = [(z, y, x) for z in (0, 6, 12) for y in (0, 9, 19) for x in (0, 12, 25)]   # 27 windows
    wd = torch.empty((len(wins),) + roi + (1,), device=DEV)
    ops.sw_gather(imd, 0, wins, wd)
    torch.cuda.synchronize()
    for i, (z, y, x) in enumerate(wins):
        assert torch.equal(wd[i, ..., 0].cpu(), img[0, 0, z:z + 8, y:y + 8, x:x + 8]), i


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("K", [4, 16, 3])
def test_sliding_window_deferred_blend_matches_oracle(K, dtype):
    """segmi_sw_blend: every window kept, one ordered blend pass == the sequential reference
    (bit-exact in f32), including the argmax and a two-shard partial blend; both kernel variants."""
    from oracle.sliding_ref import ref_sliding_window_inference
    from segmantic_amd.seg.inferers import dense_starts
    img = rnd((1, 1, 20, 27, 33), 93)
    roi = (16, 16, 16)
    wts = rnd((K, 1, 3, 3, 3), 94)

    def predictor(x):   # window predictions rounded to the cache dtype, as the network emits them
        return q(F.conv3d(x, wts, padding=1), dtype)

    # overlap <= 0.5: at most two windows cover a coordinate per dimension -> sw_blend2_kernel (all covering loads
    # issued up front); 0.75: up to four -> the generic sw_blend_kernel.  Both must reproduce the reference's sums.
    for overlap in (0.25, 0.5, 0.75):
        ref, cnt_ref, wins = ref_sliding_window_inference(img, roi, 4, predictor, overlap)
        per_dim = dense_starts((20, 27, 33), roi, overlap)
        cache = torch.empty((len(wins),) + roi + (K,), dtype=dtype, device=DEV)
        for g0 in range(0, len(wins), 4):   # same window batches as the oracle (CPU conv kernels
            grp = wins[g0:g0 + 4]           # may round differently for another batch size)
            pred = predictor(torch.cat([img[:, :, z:z + 16, y:y + 16, x:x + 16] for z, y, x in grp]))
            cache[g0:g0 + len(grp)] = pred.permute(0, 2, 3, 4, 1).to(DEV).to(dtype)
        out = torch.empty((1, 20, 27, 33, K), device=DEV)
        cnt = torch.empty((20, 27, 33), device=DEV)
        lab = torch.empty((20, 27, 33), dtype=torch.uint8, device=DEV)
        ops.sw_blend(cache, per_dim, 0, len(wins), roi, 20, 27, 33, out_logits=out, out_count=cnt, labels=lab)
        torch.cuda.synchronize()
        assert torch.equal(cnt.cpu(), cnt_ref[0, 0])
        got = out.cpu().permute(0, 4, 1, 2, 3)
        assert torch.equal(got, ref)
        assert torch.equal(lab.cpu().long(), torch.argmax(ref, 1)[0])
        # labels only (no f32 logits volume)
        lab2 = torch.empty_like(lab)
        if K % (8 if dtype == torch.bfloat16 else 4) == 0:
            ops.sw_blend(cache, per_dim, 0, len(wins), roi, 20, 27, 33, labels=lab2)
            torch.cuda.synchronize()
            assert torch.equal(lab2, lab)
        # two window shards, un-normalised partial sums: their sum / count reproduces the blend
        # up to the association of the f32 additions across the shard boundary
        mid = len(wins) // 2
        parts = []
        for lo, hi in ((0, mid), (mid, len(wins))):
            a = torch.empty((1, 20, 27, 33, K), device=DEV)
            c = torch.empty((20, 27, 33), device=DEV)
            ops.sw_blend(cache[lo:hi].contiguous(), per_dim, lo, hi, roi, 20, 27, 33, out_logits=a,
                         out_count=c, normalize=False)
            parts.append((a, c))
        torch.cuda.synchronize()
        csum = parts[0][1] + parts[1][1]
        assert torch.equal(csum.cpu(), cnt_ref[0, 0])
        tot = ((parts[0][0] + parts[1][0]) / csum[None, ..., None]).cpu().permute(0, 4, 1, 2, 3)
        assert float((tot - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_label_counts_and_dice_metric():
    from oracle.metrics_ref import ref_dice_metric
    g = torch.Generator().manual_seed(95)
    K = 5
    pred = torch.randint(0, K, (1, 1, 9, 10, 11), generator=g)
    true = torch.randint(0, K - 1, (1, 1, 9, 10, 11), generator=g)  # class K-1 absent -> NaN
    counts = torch.zeros((K, 3), dtype=torch.int64, device=DEV)
    ops.label_counts(pred.int().to(DEV).reshape(-1), true.int().to(DEV).reshape(-1), K, counts)
    torch.cuda.synchronize()
    c = counts.cpu().double()
    dice = torch.where(c[:, 2] > 0, 2 * c[:, 0] / (c[:, 1] + c[:, 2]), torch.full((K,), float("nan"), dtype=torch.double))
    ref, _ = ref_dice_metric(pred, true, K, include_background=True)
    assert torch.allclose(dice.float(), ref[0], equal_nan=True, atol=1e-6)


@pytest.mark.parametrize("nearest", [False, True])
def test_resample_matches_itk_oracle(nearest):
    from oracle.resample_ref import ref_resample_grid, resample_size
    rng = np.random.default_rng(7)
    arr = rng.standard_normal((9, 11, 13)).astype(np.float32)
    sp_in, sp_out = (0.5, 0.6, 0.7), (0.3, 0.45, 0.4)
    size = resample_size(arr.shape[::-1], sp_in, sp_out)
    ref = ref_resample_grid(arr, sp_in, (1, 2, 3), np.eye(3), size, sp_out, (1, 2, 3), np.eye(3), nearest)
    m = np.zeros((3, 4))
    for d in range(3):
        m[d, d] = sp_out[d] / sp_in[d]
    got = ops.resample3d(torch.from_numpy(arr).to(DEV), size[::-1], m, nearest=nearest)
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=1e-6)
    # integer pixel type: labels
    lab = rng.integers(0, 200, (9, 11, 13)).astype(np.uint8)
    ref_l = ref_resample_grid(lab, sp_in, (0, 0, 0), np.eye(3), size, sp_out, (0, 0, 0), np.eye(3), nearest)
    got_l = ops.resample3d(torch.from_numpy(lab).to(DEV), size[::-1], m, nearest=nearest)
    torch.cuda.synchronize()
    got_n = got_l.cpu().numpy()
    if nearest:
        assert np.array_equal(got_n, ref_l)
    else:
        # integer pixels are truncated (ITK C-cast): bit-exact except where the real value sits
        # within 1e-9 of an integer, where the last f64 bit of the index map decides
        real = ref_resample_grid(lab, sp_in, (0, 0, 0), np.eye(3), size, sp_out, (0, 0, 0),
                                 np.eye(3), nearest, return_real=True)
        diff = got_n.astype(np.int64) != ref_l.astype(np.int64)
        assert np.all(np.abs(real[diff] - np.round(real[diff])) < 1e-9)
        assert diff.mean() < 0.02


def test_normalize_intensity():
    from oracle.metrics_ref import ref_normalize
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2, 17, 19, 23)) * 37 + 100).astype(np.float32)
    x[1] = 5.0  # constant channel: std 0 -> divide by 1
    ref = ref_normalize(x)
    got = ops.normalize_intensity_(torch.from_numpy(x.copy()).to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=2e-6)


def _np_trilinear_border(vol, cz, cy, cx):
    D, H, W = vol.shape
    cz, cy, cx = np.clip(cz, 0, D - 1), np.clip(cy, 0, H - 1), np.clip(cx, 0, W - 1)
    z0, y0, x0 = np.floor(cz).astype(int), np.floor(cy).astype(int), np.floor(cx).astype(int)
    z1, y1, x1 = np.minimum(z0 + 1, D - 1), np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    fz, fy, fx = cz - z0, cy - y0, cx - x0
    a0 = vol[z0, y0, x0] * (1 - fx) + vol[z0, y0, x1] * fx
    a1 = vol[z0, y1, x0] * (1 - fx) + vol[z0, y1, x1] * fx
    a2 = vol[z1, y0, x0] * (1 - fx) + vol[z1, y0, x1] * fx
    a3 = vol[z1, y1, x0] * (1 - fx) + vol[z1, y1, x1] * fx
    b0, b1 = a0 * (1 - fy) + a1 * fy, a2 * (1 - fy) + a3 * fy
    return b0 * (1 - fz) + b1 * fz


def test_warp_crop_patches_identity_and_rotation_zoom():
    """Spatial augmentation composed into the patch gather (monai_unet.py:181-217)."""
    from segmantic_amd.seg.augment import _rot, to_index_map_xyz
    g = torch.Generator().manual_seed(7)
    D, H, W = 20, 24, 28
    img = torch.randn((1, D, H, W, 1), generator=g)
    lab = torch.randint(0, 4, (D, H, W), generator=g).float()
    imd, lad = img.to(DEV), lab.to(DEV)
    roi = (8, 12, 16)
    starts = [[0, 3, 5, 7], [0, -2, 15, 20]]      # the second one leaves the volume (SpatialPad = 0)
    flips = [0, 5]
    o1 = torch.empty((2,) + roi + (1,), device=DEV); l1 = torch.empty((2,) + roi, device=DEV)
    o2 = torch.empty_like(o1); l2 = torch.empty_like(l1)
    ops.crop_patches(imd, lad, starts, flips, o1, l1)
    ops.warp_crop_patches(imd, lad, starts, flips, np.eye(4)[:3], o2, l2)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    # rotation about d0 by 0.3 rad and zoom 1.2 about the centre
    ctr = (np.array([D, H, W]) - 1) / 2.0
    to_c, from_c = np.eye(4), np.eye(4)
    to_c[:3, 3], from_c[:3, 3] = -ctr, ctr
    zm = np.diag([1 / 1.2, 1 / 1.2, 1 / 1.2, 1.0])
    m = from_c @ _rot(0, -0.3) @ zm @ to_c
    ops.warp_crop_patches(imd, lad, starts, flips, to_index_map_xyz(m), o2, l2)
    torch.cuda.synchronize()
    got, gl = o2.cpu().numpy()[..., 0], l2.cpu().numpy()
    vol, lv = img.numpy()[0, ..., 0], lab.numpy()
    for w, (st, fl) in enumerate(zip(starts, flips)):
        zz, yy, xx = np.meshgrid(np.arange(roi[0]), np.arange(roi[1]), np.arange(roi[2]), indexing="ij")
        az = st[1] + (roi[0] - 1 - zz if fl & 1 else zz)
        ay = st[2] + (roi[1] - 1 - yy if fl & 2 else yy)
        ax = st[3] + (roi[2] - 1 - xx if fl & 4 else xx)
        inside = (az >= 0) & (az < D) & (ay >= 0) & (ay < H) & (ax >= 0) & (ax < W)
        src = np.einsum("ij,j...->i...", m[:3, :3], np.stack([az, ay, ax]).astype(np.float64)) + m[:3, 3][:, None, None, None]
        ref = np.where(inside, _np_trilinear_border(vol.astype(np.float64), src[0], src[1], src[2]), 0.0)
        assert np.abs(got[w] - ref).max() < 2e-4
        cz, cy, cx = (np.clip(src[i], 0, s - 1) for i, s in enumerate((D, H, W)))
        nz, ny, nx = (np.minimum(np.floor(c + 0.5).astype(int), s - 1) for c, s in zip((cz, cy, cx), (D, H, W)))
        rl = np.where(inside, lv[nz, ny, nx], 0.0)
        # nearest-neighbour picks may differ where the f32 coordinate sits on a .5 boundary
        assert (gl[w] != rl).mean() < 2e-3


def test_intensity_augment_matches_numpy():
    """RandAdjustContrast / RandHistogramShift / RandBiasField arithmetic (monai_unet.py:205-208)."""
    from numpy.polynomial.legendre import leggrid3d
    g = torch.Generator().manual_seed(9)
    n, rd, rh, rw = 3, 6, 10, 12
    x = torch.randn((n, rd, rh, rw, 1), generator=g)
    xd = x.to(DEV).contiguous()
    rng = np.random.RandomState(3)
    con = (np.array([1, 0, 1], np.uint8), np.array([0.7, 2.0, 3.1], np.float32))
    ctrl = np.tile(np.linspace(0, 1, 10), (n, 1))
    for i in range(n):
        for k in range(1, 9):
            ctrl[i, k] = rng.uniform(ctrl[i, k - 1], ctrl[i, k + 1])
    hist = (np.array([1, 1, 0], np.uint8), ctrl.astype(np.float32))
    coef = rng.uniform(0, 0.1, (n, 20)).astype(np.float32)
    bias = (np.array([0, 1, 1], np.uint8), coef)
    ops.intensity_augment(xd, con, hist, bias)
    torch.cuda.synchronize()
    got = xd.cpu().numpy()[..., 0]
    for i in range(n):
        v = x.numpy()[i, ..., 0].astype(np.float64)
        if con[0][i]:
            mn, rgn = v.min(), v.max() - v.min()
            v = ((v - mn) / (rgn + 1e-7)) ** float(con[1][i]) * rgn + mn
        if hist[0][i]:
            mn, mx = v.min(), v.max()
            xp = np.linspace(0, 1, 10) * (mx - mn) + mn
            v = np.interp(v, xp, hist[1][i].astype(np.float64) * (mx - mn) + mn)
        if bias[0][i]:
            cm = np.zeros((4, 4, 4))
            k = 0
            for a in range(4):
                for b in range(4 - a):
                    for c in range(4 - a - b):
                        cm[a, b, c] = coef[i, k]; k += 1
            coords = [np.linspace(-1, 1, d) for d in (rd, rh, rw)]
            v = v * np.exp(leggrid3d(coords[0], coords[1], coords[2], cm))
        assert np.abs(got[i] - v).max() < 2e-4 * max(1.0, np.abs(v).max()), i


def test_kspace_augment_matches_numpy_fft():
    """RandGibbsNoise / RandKSpaceSpikeNoise (monai_unet.py:209-210) against numpy.fft with the
    reference's shift conventions; odd and non-power-of-two extents exercise the direct DFT."""
    g = torch.Generator().manual_seed(13)
    n, shp = 4, (6, 9, 10)
    x = torch.randn((n,) + shp + (1,), generator=g)
    xd = x.to(DEV).contiguous()
    gon = np.array([1, 0, 1, 0], np.uint8); alpha = np.array([0.33, 0.52, 0.71, 0.1], np.float32)   # radii away from any bin distance
    son = np.array([0, 1, 1, 0], np.uint8)
    loc = np.array([[1, 2, 3], [5, 0, 9], [3, 4, 5], [0, 0, 0]], np.int32)
    u = np.array([0.1, 0.6, 0.9, 0.5], np.float32)
    ops.kspace_augment(xd, (gon, alpha), (son, loc, u))
    torch.cuda.synchronize()
    got = xd.cpu().numpy()[..., 0]
    ax = (0, 1, 2)
    for i in range(n):
        v = x.numpy()[i, ..., 0].astype(np.float64)
        if gon[i]:
            k = np.fft.fftshift(np.fft.fftn(np.fft.ifftshift(v, axes=ax), axes=ax), axes=ax)
            r = (1 - float(alpha[i])) * max(shp) * np.sqrt(2) / 2.0
            ctr = (np.array(shp) - 1) / 2
            zz, yy, xx = np.ogrid[0:shp[0], 0:shp[1], 0:shp[2]]
            dist = np.sqrt((zz - ctr[0]) ** 2 + (yy - ctr[1]) ** 2 + (xx - ctr[2]) ** 2)
            k = k * (dist <= r)
            v = np.fft.fftshift(np.fft.ifftn(np.fft.ifftshift(k, axes=ax), axes=ax), axes=ax).real
        if son[i]:
            k = np.fft.fftshift(np.fft.fftn(np.fft.ifftshift(v, axes=ax), axes=ax), axes=ax)
            log_abs = np.log(np.abs(k) + 1e-10)
            phase = np.angle(k)
            inten = log_abs.mean() * 2.5 * (0.95 + 0.15 * float(u[i]))
            log_abs[tuple(loc[i])] = inten
            k = np.exp(log_abs) * np.exp(1j * phase)
            v = np.fft.fftshift(np.fft.ifftn(np.fft.ifftshift(k, axes=ax), axes=ax), axes=ax).real
        assert np.abs(got[i] - v).max() < 5e-4 * max(1.0, np.abs(v).max()), i
    assert np.array_equal(got[3], x.numpy()[3, ..., 0])          # untouched patch


def test_ensemble_kernels_match_monai_semantics():
    """MeanEnsemble with weights, VoteEnsemble (ties -> smallest label), SelectBestEnsemble."""
    g = torch.Generator().manual_seed(21)
    E, K, n = 3, 5, 4097
    logits = [torch.randn((1, K, n), generator=g) for _ in range(E)]
    w = [0.81, 0.9, 0.42]
    out = torch.empty((1, K, n), device=DEV)
    ops.ensemble_mean([t.to(DEV) for t in logits], w, out)
    st = torch.stack(logits)
    wt = torch.tensor(w).view(E, 1, 1, 1)
    ref = (st * wt / wt.mean(0, keepdim=True)).mean(0)
    torch.cuda.synchronize()
    assert float((out.cpu() - ref).abs().max()) < 1e-5
    labs = [torch.randint(0, K, (n,), generator=g, dtype=torch.int32) for _ in range(E)]
    lo = torch.empty((n,), dtype=torch.int32, device=DEV)
    ops.ensemble_vote([t.to(DEV) for t in labs], lo)
    oh = torch.stack([torch.nn.functional.one_hot(t.long(), K).float() for t in labs]).mean(0)
    torch.cuda.synchronize()
    assert torch.equal(lo.cpu().long(), oh.argmax(1))
    sel = {1: 2, 3: 0, 4: 1, 2: 0}                       # tissue -> model, applied in this order
    ops.ensemble_select([t.to(DEV) for t in labs], sel, lo)
    ref = torch.zeros((n,), dtype=torch.int32)
    for tissue, model in sel.items():
        ref[labs[model] == tissue] = tissue
    torch.cuda.synchronize()
    assert torch.equal(lo.cpu(), ref)


def test_ensemble_select_matches_the_reference_test_vector(golden_dir):
    """The reference's own SelectBestEnsembled vector (tests/seg/test_transforms.py:9-43, committed as
    tests/golden/reference_select_best.json): label form through ops.ensemble_select, one-hot form
    through the product's argmax -> select -> one-hot chain (ensemble_creator, monai_unet.py)."""
    import json

    from oracle.ensemble_ref import ref_select_best
    from segmantic_amd.seg.monai_unet import _argmax_labels, _one_hot_logits
    g = json.loads((golden_dir / "reference_select_best.json").read_text())
    lmd = {int(t): int(m) for t, m in g["label_model_dict"]}
    want = torch.tensor(g["expected"], dtype=torch.int32)
    labs = [torch.tensor(p, dtype=torch.int32).to(DEV) for p in g["preds"]]
    out = torch.full((3,), -1, dtype=torch.int32, device=DEV)
    ops.ensemble_select(labs, lmd, out)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), want)
    # one-hot form: [K, 3, 1, 1] score maps per model -> arg-max labels -> select -> one-hot
    k = g["num_classes"]
    onehots = [torch.nn.functional.one_hot(torch.tensor(p), k).T.reshape(1, k, 3, 1, 1).float().to(DEV)
               for p in g["preds"]]
    labs2 = [_argmax_labels(o)[0, 0].contiguous() for o in onehots]
    out2 = torch.empty_like(labs2[0])
    ops.ensemble_select(labs2, lmd, out2)
    got = _one_hot_logits(out2, k)
    torch.cuda.synchronize()
    ref = ref_select_best([o[0].cpu() for o in onehots], lmd)
    assert torch.equal(got.cpu(), ref)
    assert torch.equal(got.cpu().argmax(0).reshape(-1).int(), want)


def test_confusion_matrix_metric_and_empty_dice_aggregate_match_monai_rule():
    """reference monai_unet.py:645-646, 705-725: ConfusionMatrixMetric(sensitivity, specificity,
    precision, accuracy) -- tp/fp/tn/fn averaged over (volume, class) first, then the ratios."""
    from oracle.metrics_ref import ref_confusion_metrics, ref_dice_metric
    from segmantic_amd.seg.losses import ConfusionMatrixMetric, DiceMetric
    g = torch.Generator().manual_seed(123)
    K = 5
    pred = torch.randint(0, K, (3, 1, 9, 10, 11), generator=g)
    true = torch.randint(0, K - 1, (3, 1, 9, 10, 11), generator=g)      # class 4 absent in truth
    cm = ConfusionMatrixMetric(K)
    for b in range(3):                                                  # accumulated over calls
        cm(pred[b:b + 1].to(DEV), true[b:b + 1].to(DEV))
    got = [float(v) for v in cm.aggregate()]
    want = ref_confusion_metrics(pred, true, K)
    assert np.allclose(got, want, rtol=1e-6, atol=0), (got, want)
    # Dice metric: all classes absent in the truth -> MONAI's mean reduction yields 0, not NaN
    dm = DiceMetric(K, include_background=False)
    zeros = torch.zeros((1, 1, 4, 4, 4), dtype=torch.long)
    d = dm(pred[:1, :, :4, :4, :4].to(DEV), zeros.to(DEV))
    assert bool(torch.isnan(d).all())
    assert float(dm.aggregate()) == 0.0
    assert float(ref_dice_metric(pred[:1, :, :4, :4, :4], zeros, K)[1]) == 0.0


def test_bn_bwd_sums_stay_accurate_when_the_mean_dwarfs_the_spread():
    """the fused epilogue accumulates sum dz*(x - mean); the round-2 form sum dz*x - mean*sum dz
    cancelled when |mean| >> std.  x = 50 + N(0, 1): dgamma from the fused sums must match an f64 reference."""
    n, d, h, w, c = 2, 32, 64, 128, 16
    g = torch.Generator().manual_seed(5)
    dy = rnd((n, c, d, h, w), 411, 0.5)
    xr = torch.randn((n, c, d, h, w), generator=g) + 50.0
    dyd, xd = to_ndhwc(dy, torch.bfloat16), to_ndhwc(xr, torch.bfloat16)
    pk = ops.wpack(torch.bfloat16, 1, rnd((c, c, 3, 3, 3), 413, 0.08).to(DEV), c, c, 3)
    xq = from_ndhwc(xd).double()
    mean = xq.mean((0, 2, 3, 4))
    invstd = 1.0 / torch.sqrt(xq.var((0, 2, 3, 4), unbiased=False) + 1e-5)
    gamma, beta = (rnd((c,), 416) + 1.5), (rnd((c,), 417) * 0.3)
    dx = torch.empty_like(dyd)
    rows = ops.conv3d_stats_rows(dyd, dx, 3, 1)
    part = torch.zeros((rows, 3, c), device=DEV)
    dg, db, coef = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty((2, c), device=DEV)
    ops.conv3d_fwd(dyd, dx, pk, None, 1, None, 3, 1,
                   bn_bwd=(xd, mean.float().to(DEV), invstd.float().to(DEV), gamma.to(DEV), beta.to(DEV), None, part),
                   bn_bwd_fin=(n * d * h * w, dg, db, None, coef))
    torch.cuda.synchronize()
    gq = from_ndhwc(dx).double()                                   # the stored gradient the sums are taken of
    xhat = (xq - mean.view(1, -1, 1, 1, 1)) * invstd.view(1, -1, 1, 1, 1)
    want_dg, want_db = (gq * xhat).sum((0, 2, 3, 4)), gq.sum((0, 2, 3, 4))
    assert float((dg.cpu().double() - want_dg).abs().max()) < 2e-4 * float(want_dg.abs().max()) + 1e-3 * float(want_dg.abs().mean())
    assert float((db.cpu().double() - want_db).abs().max()) < 2e-4 * float(want_db.abs().max()) + 1e-3 * float(want_db.abs().mean())


@pytest.mark.parametrize("c", [16, 32])
@pytest.mark.parametrize("shape,residual", [((2, 32, 64, 128), "in"), ((2, 33, 60, 120), "other"), ((4, 16, 64, 128), None)])
def test_bn_bwd_sums_in_the_input_gradient_epilogue_match_the_separate_pass(shape, residual, c):
    """segmi_bn_bwd_sums: the ring kernel's input-gradient launch also writes the partial rows of the
    BatchNorm-backward reduction over (its own stored output, x_raw).  Same dx bits; dgamma / dbeta /
    dalpha / coef equal to the separate two-tensor pass up to f32 summation order.  c = 32: the 32 -> 32
    full-resolution layers of BASELINE config 4 (conv_ring2<bf16, 32, 2>, round 4)."""
    n, d, h, w = shape
    dy = rnd((n, c, d, h, w), 401, 0.5)
    xr = rnd((n, c, d, h, w), 402, 2.0) + 0.3
    wt = rnd((c, c, 3, 3, 3), 403, 0.08)
    dyd, xd = to_ndhwc(dy, torch.bfloat16), to_ndhwc(xr, torch.bfloat16)
    pk = ops.wpack(torch.bfloat16, 1, wt.to(DEV), c, c, 3)
    assert ops.conv3d_bn_bwd_sums_ok(dyd, dyd, 3, 1) and ops.conv3d_in_affine_ok(dyd, dyd, 3, 1)
    mean = (rnd((c,), 404) * 0.5).to(DEV)
    invstd = (rnd((c,), 405).abs() + 0.5).to(DEV)
    gamma = (rnd((c,), 406) + 1.5).to(DEV)
    beta = (rnd((c,), 407) * 0.3).to(DEV)
    alpha = torch.full((1,), 0.25, device=DEV)
    res = {"in": dyd, "other": to_ndhwc(rnd((n, c, d, h, w), 408), torch.bfloat16), None: None}[residual]
    count = n * d * h * w

    def finalize(part, rows):
        dg, db, da = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty(1, device=DEV)
        coef = torch.empty((2, c), device=DEV)
        ops.bn_act_bwd_finalize(part, rows, c, count, gamma, invstd, dg, db, da, coef)
        torch.cuda.synchronize()
        return dg.cpu(), db.cpu(), da.cpu(), coef.cpu()

    dx_a = torch.empty_like(dyd)
    ops.conv3d_fwd(dyd, dx_a, pk, None, 1, None, 3, 1, residual=res)
    rows_a = ops.bn_act_bwd_rows(xd)
    part_a = torch.empty((rows_a, 3, c), device=DEV)
    ops.bn_act_bwd_reduce(dx_a, xd, mean, invstd, gamma, beta, alpha, part_a)
    ref = finalize(part_a, rows_a)

    dx_b = torch.empty_like(dyd)
    rows_b = ops.conv3d_stats_rows(dyd, dx_b, 3, 1)
    part_b = torch.full((rows_b, 3, c), float("nan"), device=DEV)
    ops.conv3d_fwd(dyd, dx_b, pk, None, 1, None, 3, 1, residual=res,
                   bn_bwd=(xd, mean, invstd, gamma, beta, alpha, part_b))
    got = finalize(part_b, rows_b)
    assert torch.equal(dx_a, dx_b)
    for a, b, name in zip(ref, got, ("dgamma", "dbeta", "dalpha", "coef")):
        scale = float(a.abs().max()) + 1e-6
        assert float((a - b).abs().max()) < 2e-4 * scale + 1e-3 * float(a.abs().mean()), name
    # no PReLU (alpha = None): the third sum is unused, dz = g
    part_c = torch.empty((rows_b, 3, c), device=DEV)
    ops.conv3d_fwd(dyd, dx_b, pk, None, 1, None, 3, 1, residual=res,
                   bn_bwd=(xd, mean, invstd, gamma, beta, None, part_c))
    ops.bn_act_bwd_reduce(dx_a, xd, mean, invstd, gamma, beta, None, part_a)
    dg = torch.empty(c, device=DEV); db = torch.empty(c, device=DEV); coef = torch.empty((2, c), device=DEV)
    ops.bn_act_bwd_finalize(part_a, rows_a, c, count, gamma, invstd, dg, db, None, coef)
    dg2 = torch.empty(c, device=DEV); db2 = torch.empty(c, device=DEV); coef2 = torch.empty((2, c), device=DEV)
    ops.bn_act_bwd_finalize(part_c, rows_b, c, count, gamma, invstd, dg2, db2, None, coef2)
    torch.cuda.synchronize()
    assert float((dg - dg2).abs().max()) < 2e-4 * float(dg.abs().max()) + 1e-3 * float(dg.abs().mean())
    assert float((db - db2).abs().max()) < 2e-4 * float(db.abs().max()) + 1e-3 * float(db.abs().mean())


@pytest.mark.parametrize("slope", [0.25, 1.7, -0.3])
@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (1, 16, 24, 32), (3, 6, 8, 40), (1, 2, 8, 8)])
def test_fused_full_resolution_decoder_is_bit_identical_to_the_two_launches(shape, slope):
    """segmi_dectop_fwd: ConvTranspose3d(32 -> 16) + folded BN + PReLU -> conv(16 -> 16) + identity
    residual in one launch; same tap order and k-slot layout as the separate kernels => same bits."""
    n, d, h, w = shape
    x = to_ndhwc(rnd((n, 32, d, h, w), 501), torch.bfloat16)
    wt = (rnd((32, 16, 3, 3, 3), 502, 0.06)).to(DEV)          # ConvTranspose3d layout [Cin, Cout, 3,3,3]
    wc = (rnd((16, 16, 3, 3, 3), 503, 0.08)).to(DEV)
    scale = (rnd((16,), 504).abs() + 0.5).to(DEV)
    ub = (rnd((16,), 505) * 0.2).to(DEV)
    cb = (rnd((16,), 506) * 0.2).to(DEV)
    alpha = torch.full((1,), slope, device=DEV)
    fine = (n, 2 * d, 2 * h, 2 * w, 16)
    # two launches
    up_pack = ops.wpack(torch.bfloat16, 2, wt, 32, 16, 3, scale=scale)
    cv_pack = ops.wpack(torch.bfloat16, 0, wc, 16, 16, 3)
    hmid = torch.empty(fine, dtype=torch.bfloat16, device=DEV)
    ops.convT3d_fwd(x, hmid, up_pack, None, ub, prelu_alpha=alpha)
    ref = torch.empty_like(hmid)
    ops.conv3d_fwd(hmid, ref, cv_pack, None, 0, cb, 3, 1, residual=hmid)
    # one launch
    out = torch.full(fine, float("nan"), dtype=torch.bfloat16, device=DEV)
    assert ops.dectop_ok(x, out)
    ops.dectop_fwd(x, out, ops.dectop_up_frag(wt, scale), ub, alpha, cv_pack, cb, alpha_in_unit_range=0.0 <= slope <= 1.0)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())
    assert torch.equal(out, ref), f"max diff {float((out.float() - ref.float()).abs().max())}"
    if 0.0 <= slope <= 1.0:        # the generic PReLU path gives the same bits
        out2 = torch.full(fine, float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.dectop_fwd(x, out2, ops.dectop_up_frag(wt, scale), ub, alpha, cv_pack, cb, alpha_in_unit_range=False)
        torch.cuda.synchronize()
        assert torch.equal(out2, ref)
    # independent check of the pair against torch (oracle semantics), bf16 tolerance
    xr = from_ndhwc(x)
    hq = F.prelu(F.conv_transpose3d(xr, q(wt.cpu() * scale.cpu().view(1, -1, 1, 1, 1), torch.bfloat16), ub.cpu(),
                                    stride=2, padding=1, output_padding=1), alpha.cpu())
    hq = q(hq, torch.bfloat16)
    want = F.conv3d(hq, q(wc.cpu(), torch.bfloat16), cb.cpu(), padding=1) + hq
    assert relerr(from_ndhwc(out), want) < BF16_RTOL
    # shapes the kernel does not take are refused by the query
    bad = torch.empty((n, 2 * d, 2 * h + 2, 2 * w, 16), dtype=torch.bfloat16, device=DEV)
    assert not ops.dectop_ok(x, bad)


WS_WGRAD_CASES = [
    # cin, cout, k, s, spatial of x (d,h,w), batch -- enough tiles (>= 4 per workgroup of the 256 / 128-wide grids)
    # that the wave-specialised kernel takes the launch; ragged extents, volume borders in every dimension
    (16, 16, 3, 1, (33, 60, 120), 2),      # row-split consumers, ragged z / y / x
    (16, 32, 3, 1, (18, 64, 128), 4),      # 2 output-channel tiles share the X staging
    (16, 32, 3, 2, (64, 64, 128), 2),      # stride 2 (the transposed-conv gradient of the decoder top)
    (16, 64, 3, 2, (63, 66, 126), 2),      # stride 2, two output-channel chunks (grid.y = 2), ragged
    (16, 16, 3, 2, (48, 80, 160), 2),      # stride 2, 16 x 16
    (16, 16, 3, 1, (40, 64, 8), 8),        # narrow volumes: the 8-wide tile, tap-split consumers
    (32, 16, 3, 1, (18, 64, 128), 2),      # 32 input channels: X staged in two 16-channel chunks (ci0 = 0, 16)
    (32, 16, 3, 2, (64, 64, 128), 2),      # the same with stride 2
    (32, 32, 3, 1, (18, 64, 128), 4),      # >= 32 channels on both sides: the 2 x 1 channel tile (round 4), grid.y = 2
    (64, 32, 3, 1, (18, 64, 128), 2),      # four input-channel chunks
    (32, 64, 3, 2, (64, 64, 128), 2),      # stride 2 with two output-channel chunks of 32
]


@pytest.mark.parametrize("case", WS_WGRAD_CASES)
def test_wave_specialised_wgrad_matches_torch(case):
    cin, cout, k, s, sp, n = case
    x = rnd((n, cin) + sp, 331)
    osp = tuple((d + 2 - k) // s + 1 for d in sp)
    dy = rnd((n, cout) + osp, 332)
    w0 = torch.zeros((cout, cin, k, k, k), requires_grad=True)
    b0 = torch.zeros((cout,), requires_grad=True)
    F.conv3d(q(x, torch.bfloat16), w0, b0, stride=s, padding=1).backward(q(dy, torch.bfloat16))
    xd, dyd = to_ndhwc(x, torch.bfloat16), to_ndhwc(dy, torch.bfloat16)
    dw = torch.empty_like(w0, device=DEV)
    db = torch.empty_like(b0, device=DEV)
    ws = torch.empty(ops.conv3d_wgrad_workspace(xd, dyd, k, s), dtype=torch.uint8, device=DEV)
    ops.conv3d_wgrad(xd, dyd, dw, db, k, s, ws)
    torch.cuda.synchronize()
    assert relerr(dw.cpu(), w0.grad) < 5e-5
    assert relerr(db.cpu(), b0.grad) < 5e-5
    # a channel-slice view (ld > c) of a wider buffer as X: the buffer descriptor covers the view
    wide = torch.zeros(xd.shape[:4] + (2 * cin,), dtype=torch.bfloat16, device=DEV)
    wide[..., cin:] = xd
    ops.conv3d_wgrad(wide[..., cin:], dyd, dw, None, k, s, ws)
    torch.cuda.synchronize()
    assert relerr(dw.cpu(), w0.grad) < 5e-5
    # the fused input transform (segmi_in_affine) on the wave-specialised path: same bits as the
    # weight gradient of the separately normalised tensor
    scale = (rnd((cin,), 333).abs() + 0.5).to(DEV)
    shift = (rnd((cin,), 334) * 0.3).to(DEV)
    alpha = torch.full((1,), 0.25, device=DEV)
    xn = torch.empty_like(xd)
    ops.bn_act_fwd(xd, xn, scale, shift, alpha)
    dw_ref, dw_tf = torch.empty_like(dw), torch.empty_like(dw)
    ops.conv3d_wgrad(xn, dyd, dw_ref, None, k, s, ws)
    ops.conv3d_wgrad(xd, dyd, dw_tf, None, k, s, ws, in_tf=(scale, shift, alpha))
    torch.cuda.synchronize()
    assert torch.equal(dw_ref, dw_tf)


def test_wgrad_of_an_operand_past_the_4GiB_descriptor_is_cut_along_the_batch():
    """A channel slice of a wide buffer at BASELINE config 4's extent -- 8 x 160^3 voxels x 96 channels x 2 B =
    6.3 GB -- is more than one buffer descriptor of the wave-specialised kernel covers (32-bit offsets, < 4 GiB).
    The call is cut into batch parts that fit (one launch each, one reduce over all slabs) instead of falling back
    to the tile-at-a-time kernel.  Checked through linearity in the batch: the gradient of the whole batch is the sum
    of the gradients of its halves (each of which the kernel takes directly), in f32 round-off."""
    n, S, c = 8, 160, 32
    g = torch.Generator(device=DEV).manual_seed(77)
    wide = torch.empty((n, S, S, S, 3 * c), dtype=torch.bfloat16, device=DEV)
    for i in range(n):
        wide[i] = torch.randn((S, S, S, 3 * c), device=DEV, generator=g).bfloat16()
    dy = torch.empty((n, S, S, S, c), dtype=torch.bfloat16, device=DEV)
    for i in range(n):
        dy[i] = torch.randn((S, S, S, c), device=DEV, generator=g).bfloat16()
    x = wide[..., c:2 * c]
    assert x.numel() // c * 3 * c * 2 > 0xfff00000
    dw = [torch.full((c, c, 3, 3, 3), float("nan"), device=DEV) for _ in range(3)]
    ws = torch.empty(ops.conv3d_wgrad_workspace(x, dy, 3, 1), dtype=torch.uint8, device=DEV)
    ops.conv3d_wgrad(x, dy, dw[0], None, 3, 1, ws)
    h = n // 2
    ops.conv3d_wgrad(x[:h], dy[:h], dw[1], None, 3, 1, ws)
    ops.conv3d_wgrad(x[h:], dy[h:], dw[2], None, 3, 1, ws)
    torch.cuda.synchronize()
    want = dw[1].double() + dw[2].double()
    assert torch.isfinite(dw[0]).all()
    assert float((dw[0].double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    # and it is the wave-specialised kernel's partition: twice the slabs of one half
    half = ops.conv3d_wgrad_workspace(x[:h], dy[:h], 3, 1)
    assert ops.conv3d_wgrad_workspace(x, dy, 3, 1) > half


# ------------------------------------------------------------------ finalisation inside the producing launch
def _fin_bufs(c, seed=0):
    rm = (rnd((c,), 900 + seed) * 0.2).to(DEV)
    rv = (rnd((c,), 901 + seed).abs() + 0.5).to(DEV)
    outs = [torch.full((c,), float("nan"), device=DEV) for _ in range(4)]
    return rm, rv, outs


FIN_CONV_CASES = [
    # cin, cout, k, s, spatial, batch, transposed       kernel family
    (16, 16, 3, 1, (16, 64, 128), 4, False),           # z-marching ring
    (32, 32, 3, 1, (33, 60, 120), 2, False),           # ring, CK = 32, two tiles
    (64, 64, 3, 1, (9, 10, 40), 2, False),             # k-split
    (256, 32, 3, 1, (8, 8, 8), 1, False),              # k-split, deep
    (16, 32, 3, 2, (10, 12, 36), 1, False),            # tile kernel, stride 2
    (128, 256, 1, 1, (4, 4, 4), 2, False),             # tile kernel, k1
    (1, 16, 3, 2, (20, 34, 70), 1, False),             # small-Cin
    (16, 3, 3, 1, (6, 7, 9), 1, False),                # direct kernel + bn_stats (separate finalisation inside the call)
    (32, 16, 3, 2, (8, 16, 32), 2, True),              # transposed, persistent parity-class kernel
    (128, 32, 3, 2, (4, 4, 8), 1, True),               # transposed, tile kernel
    (384, 64, 3, 2, (2, 2, 2), 2, True),               # transposed, deep
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", FIN_CONV_CASES)
def test_bn_statistics_finalised_by_the_producing_launch(case, dtype):
    """segmi_bn_fin (csrc/fin_tail.h): the convolution that writes the statistics rows also finalises
    them -- mean / invstd / scale / shift / running statistics equal the separate segmi_bn_finalize
    launch on the same rows (f64 sums of the same f32 rows in another fixed order: <= 1e-6 relative),
    the output tensor is bit-identical, and the result is bitwise reproducible run to run."""
    cin, cout, k, s, sp, n, transposed = case
    x = rnd((n, cin) + sp, 11)
    xd = to_ndhwc(x, dtype)
    if transposed:
        w = rnd((cin, cout, 3, 3, 3), 12, 1.0 / math.sqrt(cin * 27 / 8))
        osp = tuple(2 * v for v in sp)
        kind = 2
    else:
        w = rnd((cout, cin, k, k, k), 12, 1.0 / math.sqrt(cin * k ** 3))
        osp = tuple((v + 2 * ((k - 1) // 2) - k) // s + 1 for v in sp)
        kind = 0
    b = rnd((cout,), 13, 0.1).to(DEV)
    wd = w.to(DEV)
    packed = ops.wpack(dtype, kind, wd, cin, cout, 3 if transposed else k) if ops.mfma_ok(cin, cout) else None
    gamma, beta = (1 + 0.2 * rnd((cout,), 14)).to(DEV), (0.1 * rnd((cout,), 15)).to(DEV)
    count = n * osp[0] * osp[1] * osp[2]

    def run(fused):
        y = torch.empty((n,) + osp + (cout,), dtype=dtype, device=DEV)
        rows = ops.convT3d_stats_rows(xd, y) if transposed else ops.conv3d_stats_rows(xd, y, k, s)
        stats = torch.zeros((rows, 2, cout), device=DEV)
        rm, rv, (mean, invstd, scale, shift) = _fin_bufs(cout)
        fin = (count, gamma, beta, rm, rv, 0.1, 1e-5, mean, invstd, scale, shift)
        if transposed:
            ops.convT3d_fwd(xd, y, packed, wd, b, stats=stats, stats_fin=fin if fused else None)
        else:
            ops.conv3d_fwd(xd, y, packed, wd, 0, b, k, s, stats=stats, stats_fin=fin if fused else None)
        if not fused:
            ops.bn_finalize(stats, rows, cout, count, gamma, beta, rm, rv, 0.1, 1e-5, mean, invstd, scale, shift)
        torch.cuda.synchronize()
        return y, [t.clone() for t in (mean, invstd, scale, shift, rm, rv)]

    y0, ref = run(False)
    y1, got = run(True)
    y2, again = run(True)
    assert torch.equal(y0, y1)
    for a, g_, g2, name in zip(ref, got, again, ("mean", "invstd", "scale", "shift", "running_mean", "running_var")):
        assert bool(torch.isfinite(g_).all()), name
        assert float((a - g_).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-9, name
        assert torch.equal(g_, g2), name + " not reproducible"


@pytest.mark.parametrize("case", [(16, 32, (32, 34, 66), 2), (32, 64, (16, 32, 32), 2), (64, 128, (8, 16, 16), 3),
                                  (16, 16, (16, 32, 64), 1)])
def test_training_pair_of_a_residual_units_stride2_convolutions_is_the_two_launches(case):
    """segmi_conv3d_fwd_split_act in training: the first subunit's convolution (with its BatchNorm statistics and
    their finalisation) and the residual convolution of one input as ONE launch over a pack built from the two
    parameter tensors (segmi_wpack_desc.w_src2) -- the output halves are the two launches' tensors bit for bit, the
    statistics are those of the first half alone (same rows, same finalisation)."""
    cin, c, sp, n = case
    dtype = torch.bfloat16
    xd = to_ndhwc(rnd((n, cin) + sp, 41), dtype)
    wa = rnd((c, cin, 3, 3, 3), 42, 1.0 / math.sqrt(cin * 27)).to(DEV)
    wb = rnd((c, cin, 3, 3, 3), 43, 1.0 / math.sqrt(cin * 27)).to(DEV)
    ba, bb = rnd((c,), 44, 0.1).to(DEV), rnd((c,), 45, 0.1).to(DEV)
    osp = tuple((v - 1) // 2 + 1 for v in sp)
    count = n * osp[0] * osp[1] * osp[2]
    gamma, beta = (1 + 0.2 * rnd((c,), 46)).to(DEV), (0.1 * rnd((c,), 47)).to(DEV)
    m = torch.full((n,) + osp + (2 * c,), float("nan"), dtype=dtype, device=DEV)
    if not ops.conv3d_split_act_ok(xd, m, 3, 2):
        pytest.skip("another kernel family takes this layer")
    # the two launches
    ya = torch.empty((n,) + osp + (c,), dtype=dtype, device=DEV)
    yb = torch.empty_like(ya)
    rows = ops.conv3d_stats_rows(xd, ya, 3, 2)
    stats = torch.zeros((rows, 2, c), device=DEV)
    rm, rv, (mean, invstd, scale, shift) = _fin_bufs(c)
    ops.conv3d_fwd(xd, ya, ops.wpack(dtype, 0, wa, cin, c, 3), wa, 0, ba, 3, 2, stats=stats,
                   stats_fin=(count, gamma, beta, rm, rv, 0.1, 1e-5, mean, invstd, scale, shift))
    ops.conv3d_fwd(xd, yb, ops.wpack(dtype, 0, wb, cin, c, 3), wb, 0, bb, 3, 2)
    # one launch
    batch = ops.WpackBatch(dtype, [(0, wa, None, cin, 2 * c, 3, wb, c)])
    batch.run()
    rows_m = ops.conv3d_stats_rows(xd, m, 3, 2)
    stats_m = torch.zeros((rows_m, 2, c), device=DEV)
    rm2, rv2, (mean2, invstd2, scale2, shift2) = _fin_bufs(c)
    ops.conv3d_fwd_split_act(xd, m, batch.packed[0], ba, None, c, 3, 2, bias_b=bb, stats=stats_m,
                             stats_fin=(count, gamma, beta, rm2, rv2, 0.1, 1e-5, mean2, invstd2, scale2, shift2))
    torch.cuda.synchronize()
    assert torch.equal(m[..., :c], ya) and torch.equal(m[..., c:], yb)
    for a, b_, name in zip((mean, invstd, scale, shift, rm, rv), (mean2, invstd2, scale2, shift2, rm2, rv2),
                           ("mean", "invstd", "scale", "shift", "running_mean", "running_var")):
        assert bool(torch.isfinite(b_).all()), name
        assert float((a - b_).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-9, name
    # the pack of two sources is the pack of the concatenated weight
    cat = ops.wpack(dtype, 0, torch.cat([wa, wb], 0).contiguous(), cin, 2 * c, 3)
    assert torch.equal(cat, batch.packed[0])
    # the paired INPUT gradient: one transposed convolution over [dy_a | dy_b] with the kind-2 pack of the two
    # sources = the sum of the two input gradients (f32 accumulation over 2c channels instead of two bf16-rounded
    # launches: tolerance), and that pack is the pack of the concatenated weight
    bd = ops.WpackBatch(dtype, [(2, wa, None, 2 * c, cin, 3, wb, c)])
    bd.run()
    assert torch.equal(ops.wpack(dtype, 2, torch.cat([wa, wb], 0).contiguous(), 2 * c, cin, 3), bd.packed[0])
    dy = to_ndhwc(rnd((n, 2 * c) + osp, 48), dtype)
    dx_pair = torch.full_like(xd, float("nan"))
    ops.convT3d_fwd(dy, dx_pair, bd.packed[0], None, None)
    dx_two = torch.empty_like(xd)
    ops.convT3d_fwd(dy[..., c:], dx_two, ops.wpack(dtype, 2, wb, c, cin, 3), None, None)
    ops.convT3d_fwd(dy[..., :c], dx_two, ops.wpack(dtype, 2, wa, c, cin, 3), None, None, residual=dx_two)
    torch.cuda.synchronize()
    assert relerr(from_ndhwc(dx_pair), from_ndhwc(dx_two)) < BF16_RTOL
    # a bias table for both halves in one tensor (bias_b = None) gives the same tensor
    m2 = torch.empty_like(m)
    ops.conv3d_fwd_split_act(xd, m2, cat, torch.cat([ba, bb]).contiguous(), None, c, 3, 2)
    torch.cuda.synchronize()
    assert torch.equal(m2, m)


def test_bn_statistics_finalised_by_the_pair_launch():
    cin, cout, sp, n, s = 1, 16, (20, 34, 70), 2, 2
    dtype = torch.bfloat16
    xd = to_ndhwc(rnd((n, cin) + sp, 21), dtype)
    wa, wb = rnd((cout, cin, 3, 3, 3), 22, 0.2).to(DEV), rnd((cout, cin, 3, 3, 3), 23, 0.2).to(DEV)
    ba, bb = rnd((cout,), 24, 0.1).to(DEV), rnd((cout,), 25, 0.1).to(DEV)
    osp = tuple((v - 1) // 2 + 1 for v in sp)
    count = n * osp[0] * osp[1] * osp[2]
    gamma, beta = (1 + 0.2 * rnd((cout,), 26)).to(DEV), (0.1 * rnd((cout,), 27)).to(DEV)

    def run(fused):
        ya = torch.empty((n,) + osp + (cout,), dtype=dtype, device=DEV)
        yb = torch.empty_like(ya)
        assert ops.conv3d_pair_ok(xd, ya, yb)
        rows = ops.conv3d_stats_rows(xd, ya, 3, s)
        stats = torch.zeros((rows, 2, cout), device=DEV)
        rm, rv, (mean, invstd, scale, shift) = _fin_bufs(cout)
        fin = (count, gamma, beta, rm, rv, 0.1, 1e-5, mean, invstd, scale, shift)
        ops.conv3d_fwd_pair(xd, ya, wa, ba, yb, wb, bb, s, stats_a=stats, stats_fin_a=fin if fused else None)
        if not fused:
            ops.bn_finalize(stats, rows, cout, count, gamma, beta, rm, rv, 0.1, 1e-5, mean, invstd, scale, shift)
        torch.cuda.synchronize()
        return ya, yb, [t.clone() for t in (mean, invstd, scale, shift, rm, rv)]

    a0, b0, ref = run(False)
    a1, b1, got = run(True)
    assert torch.equal(a0, a1) and torch.equal(b0, b1)
    for a, g_ in zip(ref, got):
        assert float((a - g_).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-9


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c,sp,n", [(16, (16, 64, 64), 2), (64, (9, 10, 12), 2), (256, (8, 8, 8), 1), (3, (5, 6, 7), 2)])
def test_bn_backward_sums_finalised_by_the_reduce_launch(c, sp, n, dtype):
    x = rnd((n, c) + sp, 31, 2.0)
    dy = rnd((n, c) + sp, 32)
    xd, dyd = to_ndhwc(x, dtype), to_ndhwc(dy, dtype)
    mean = (rnd((c,), 33) * 0.5).to(DEV)
    invstd = (rnd((c,), 34).abs() + 0.5).to(DEV)
    gamma, beta = (rnd((c,), 35) + 1.5).to(DEV), (rnd((c,), 36) * 0.3).to(DEV)
    alpha = torch.full((1,), 0.25, device=DEV)
    count = n * sp[0] * sp[1] * sp[2]
    rows = ops.bn_act_bwd_rows(xd)

    def run(fused):
        part = torch.zeros((rows, 3, c), device=DEV)
        dg, db, da = (torch.full((m,), float("nan"), device=DEV) for m in (c, c, 1))
        coef = torch.full((2, c), float("nan"), device=DEV)
        ops.bn_act_bwd_reduce(dyd, xd, mean, invstd, gamma, beta, alpha, part,
                              fin=(count, dg, db, da, coef) if fused else None)
        if not fused:
            ops.bn_act_bwd_finalize(part, rows, c, count, gamma, invstd, dg, db, da, coef)
        torch.cuda.synchronize()
        return [t.clone() for t in (dg, db, da, coef)]

    ref, got, again = run(False), run(True), run(True)
    for a, g_, g2, name in zip(ref, got, again, ("dgamma", "dbeta", "dalpha", "coef")):
        assert bool(torch.isfinite(g_).all()), name
        assert float((a - g_).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-9, name
        assert torch.equal(g_, g2), name


def test_bn_backward_sums_finalised_by_the_input_gradient_launch():
    """ring kernel MODE 4 (segmi_bn_bwd_sums) with its `fin`: the launch that produces the gradient and the
    partial rows also writes dgamma / dbeta / dalpha / coef"""
    n, d, h, w, c = 2, 32, 64, 128, 16
    dyd = to_ndhwc(rnd((n, c, d, h, w), 41, 0.5), torch.bfloat16)
    xd = to_ndhwc(rnd((n, c, d, h, w), 42, 2.0) + 0.3, torch.bfloat16)
    pk = ops.wpack(torch.bfloat16, 1, rnd((c, c, 3, 3, 3), 43, 0.08).to(DEV), c, c, 3)
    mean, invstd = (rnd((c,), 44) * 0.5).to(DEV), (rnd((c,), 45).abs() + 0.5).to(DEV)
    gamma, beta = (rnd((c,), 46) + 1.5).to(DEV), (rnd((c,), 47) * 0.3).to(DEV)
    alpha = torch.full((1,), 0.25, device=DEV)
    count = n * d * h * w

    def run(fused):
        dx = torch.empty_like(dyd)
        rows = ops.conv3d_stats_rows(dyd, dx, 3, 1)
        part = torch.zeros((rows, 3, c), device=DEV)
        dg, db, da = (torch.full((m,), float("nan"), device=DEV) for m in (c, c, 1))
        coef = torch.full((2, c), float("nan"), device=DEV)
        ops.conv3d_fwd(dyd, dx, pk, None, 1, None, 3, 1, bn_bwd=(xd, mean, invstd, gamma, beta, alpha, part),
                       bn_bwd_fin=(count, dg, db, da, coef) if fused else None)
        if not fused:
            ops.bn_act_bwd_finalize(part, rows, c, count, gamma, invstd, dg, db, da, coef)
        torch.cuda.synchronize()
        return dx, [t.clone() for t in (dg, db, da, coef)]

    dx0, ref = run(False)
    dx1, got = run(True)
    assert torch.equal(dx0, dx1)
    for a, g_, name in zip(ref, got, ("dgamma", "dbeta", "dalpha", "coef")):
        assert bool(torch.isfinite(g_).all()), name
        assert float((a - g_).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-9, name


def test_ring_kernel_predicates_refuse_samples_beyond_its_32_bit_addressing():
    """the z-marching ring kernel addresses one sample with 32-bit offsets; samples beyond
    that must be routed to the 64-bit kernels by EVERY predicate (not fail in the launcher)."""
    ok = torch.empty((1, 128, 128, 128, 16), dtype=torch.bfloat16, device=DEV)
    assert ops.conv3d_in_affine_ok(ok, ok, 3, 1) and ops.conv3d_bn_bwd_sums_ok(ok, ok, 3, 1)
    assert "ring" in ops.conv3d_fwd_kernel_name(ok, ok, 3, 1)
    big = torch.empty((1, 416, 416, 416, 16), dtype=torch.bfloat16, device=DEV)     # 2.3 GB, one sample
    assert not ops.conv3d_in_affine_ok(big, big, 3, 1) and not ops.conv3d_bn_bwd_sums_ok(big, big, 3, 1)
    assert "ring" not in ops.conv3d_fwd_kernel_name(big, big, 3, 1)
    # and the layer still runs (a thin slab of it: the tile kernel), with statistics rows that match its kernel
    del big
    x = to_ndhwc(rnd((1, 16, 4, 416, 416), 5), torch.bfloat16)
    w = rnd((16, 16, 3, 3, 3), 6, 0.05)
    y = torch.empty_like(x)
    rows = ops.conv3d_stats_rows(x, y, 3, 1)
    stats = torch.zeros((rows, 2, 16), device=DEV)
    ops.conv3d_fwd(x, y, ops.wpack(torch.bfloat16, 0, w.to(DEV), 16, 16, 3), None, 0, None, 3, 1, stats=stats)
    torch.cuda.synchronize()
    ref = F.conv3d(from_ndhwc(x), q(w, torch.bfloat16), padding=1)
    assert relerr(from_ndhwc(y), ref) < BF16_RTOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c,sp,n,with_alpha", [(32, (32, 32, 32), 8, True), (64, (16, 16, 16), 8, True),
                                               (256, (8, 8, 8), 8, True), (128, (8, 8, 8), 2, False),
                                               (48, (5, 6, 7), 2, True), (16, (9, 10, 12), 1, True)])
def test_bn_backward_as_one_launch_matches_the_three_calls(c, sp, n, with_alpha, dtype):
    """segmi_bn_act_bwd_fused (small tensors): reduce + finalise + apply behind a grid-wide hand-off.  Same
    dgamma / dbeta / dalpha / coef as the separate calls up to the summation order of the partial rows; dx
    equal up to the rounding flips those last bits cause; bitwise reproducible."""
    x = rnd((n, c) + sp, 61, 2.0)
    dy = rnd((n, c) + sp, 62)
    xd, dyd = to_ndhwc(x, dtype), to_ndhwc(dy, dtype)
    mean = (rnd((c,), 63) * 0.5).to(DEV)
    invstd = (rnd((c,), 64).abs() + 0.5).to(DEV)
    gamma, beta = (rnd((c,), 65) + 1.5).to(DEV), (rnd((c,), 66) * 0.3).to(DEV)
    alpha = torch.full((1,), 0.25, device=DEV) if with_alpha else None
    count = n * sp[0] * sp[1] * sp[2]
    assert ops.bn_act_bwd_fused_ok(dyd, xd, xd)

    def outs():
        return (torch.full((c,), float("nan"), device=DEV), torch.full((c,), float("nan"), device=DEV),
                torch.full((1,), float("nan"), device=DEV) if with_alpha else None, torch.full((2, c), float("nan"), device=DEV))

    def three():
        dg, db, da, coef = outs()
        rows = ops.bn_act_bwd_rows(xd)
        part = torch.zeros((rows, 3, c), device=DEV)
        ops.bn_act_bwd_reduce(dyd, xd, mean, invstd, gamma, beta, alpha, part)
        ops.bn_act_bwd_finalize(part, rows, c, count, gamma, invstd, dg, db, da, coef)
        dx = torch.empty_like(xd)
        ops.bn_act_bwd_apply(dyd, xd, dx, mean, invstd, gamma, beta, alpha, coef)
        torch.cuda.synchronize()
        return dx, [dg, db, coef] + ([da] if with_alpha else [])

    def one():
        dg, db, da, coef = outs()
        rows = ops.bn_act_bwd_fused_rows(xd)
        part = torch.zeros((rows, 3, c), device=DEV)
        dx = torch.full_like(xd, float("nan"))
        ops.bn_act_bwd_fused(dyd, xd, dx, mean, invstd, gamma, beta, alpha, part, (count, dg, db, da, coef))
        torch.cuda.synchronize()
        return dx, [dg, db, coef] + ([da] if with_alpha else [])

    dx0, ref = three()
    dx1, got = one()
    dx2, again = one()
    for a, g_, g2 in zip(ref, got, again):
        assert bool(torch.isfinite(g_).all())
        assert float((a - g_).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-7
        assert torch.equal(g_, g2)
    assert torch.equal(dx1, dx2)
    assert bool(torch.isfinite(dx1.float()).all())
    tol_dx = 1e-5 if dtype == torch.float32 else 1e-2
    assert float((dx0.float() - dx1.float()).abs().max()) <= tol_dx * float(dx0.float().abs().max())
    # tensors beyond 32 MB are refused by the query
    big = torch.empty((8, 64, 64, 64, 16), dtype=torch.bfloat16, device=DEV)
    assert not ops.bn_act_bwd_fused_ok(big, big, big)


@pytest.mark.parametrize("max_wgs", [128, 40, 1])
def test_bn_backward_one_launch_under_a_workgroup_cap(max_wgs):
    """`max_wgs` of segmi_bn_act_bwd_fused: the engine caps the launch at the CUs the weight-gradient stream leaves
    free, so that every workgroup of the grid-wide hand-off is resident.  Any cap gives the sums of the three calls
    (another partition of the rows) and a finite dx; the launch never exceeds the cap or the device's capacity."""
    c, sp, n, dtype = 32, (32, 32, 32), 4, torch.bfloat16
    x, dy = rnd((n, c) + sp, 71, 2.0), rnd((n, c) + sp, 72)
    xd, dyd = to_ndhwc(x, dtype), to_ndhwc(dy, dtype)
    mean, invstd = (rnd((c,), 73) * 0.5).to(DEV), (rnd((c,), 74).abs() + 0.5).to(DEV)
    gamma, beta = (rnd((c,), 75) + 1.5).to(DEV), (rnd((c,), 76) * 0.3).to(DEV)
    alpha = torch.full((1,), 0.25, device=DEV)
    count = n * sp[0] * sp[1] * sp[2]
    cap = ops.bn_act_bwd_fused_wgs(xd, 0)
    assert 1 <= cap <= 256 and 1 <= ops.bn_act_bwd_fused_wgs(xd, max_wgs) <= min(max_wgs, cap)

    def run(mw):
        dg, db = torch.full((c,), float("nan"), device=DEV), torch.full((c,), float("nan"), device=DEV)
        da, coef = torch.full((1,), float("nan"), device=DEV), torch.full((2, c), float("nan"), device=DEV)
        part = torch.zeros((ops.bn_act_bwd_fused_rows(xd), 3, c), device=DEV)
        dx = torch.full_like(xd, float("nan"))
        ops.bn_act_bwd_fused(dyd, xd, dx, mean, invstd, gamma, beta, alpha, part, (count, dg, db, da, coef), max_wgs=mw)
        torch.cuda.synchronize()
        return dx, [dg, db, da, coef]

    dx0, ref = run(0)
    dx1, got = run(max_wgs)
    for a, g_ in zip(ref, got):
        assert bool(torch.isfinite(g_).all())
        assert float((a - g_).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-7
    assert bool(torch.isfinite(dx1.float()).all())
    assert float((dx0.float() - dx1.float()).abs().max()) <= 1e-2 * float(dx0.float().abs().max())
    assert ops.fused_timeouts() == 0


def test_bn_backward_one_launch_expiry_is_counted_and_raised():
    """The bounded wait of the grid-wide hand-off: with the test hook withholding the flag and a
    tiny poll bound every workgroup gives up -- the launch ENDS (no hang), dx is NaN (never a plausible wrong
    gradient), the host-visible counter says how many workgroups expired, and `check_fused_timeouts` turns it into
    an exception and clears it.  The next launch, hook off, is healthy again."""
    c, sp, n, dtype = 32, (16, 16, 16), 2, torch.bfloat16
    x, dy = rnd((n, c) + sp, 81, 2.0), rnd((n, c) + sp, 82)
    xd, dyd = to_ndhwc(x, dtype), to_ndhwc(dy, dtype)
    mean, invstd = (rnd((c,), 83) * 0.5).to(DEV), (rnd((c,), 84).abs() + 0.5).to(DEV)
    gamma, beta = (rnd((c,), 85) + 1.5).to(DEV), (rnd((c,), 86) * 0.3).to(DEV)
    count = n * sp[0] * sp[1] * sp[2]

    def run():
        dg, db, coef = torch.zeros((c,), device=DEV), torch.zeros((c,), device=DEV), torch.zeros((2, c), device=DEV)
        part = torch.zeros((ops.bn_act_bwd_fused_rows(xd), 3, c), device=DEV)
        dx = torch.zeros_like(xd)
        ops.bn_act_bwd_fused(dyd, xd, dx, mean, invstd, gamma, beta, None, part, (count, dg, db, None, coef))
        torch.cuda.synchronize()
        return dx

    ops.fused_timeouts(reset=True)
    wgs = ops.bn_act_bwd_fused_wgs(xd, 0)
    ops.fused_test_hook(poll_limit=64, no_publish=True)
    try:
        dx = run()
    finally:
        ops.fused_test_hook()
    assert bool(torch.isnan(dx.float()).all())
    assert ops.fused_timeouts() == wgs
    with pytest.raises(RuntimeError, match="gave up waiting"):
        ops.check_fused_timeouts("test")
    assert ops.fused_timeouts() == 0
    assert bool(torch.isfinite(run().float()).all())
    ops.check_fused_timeouts("test")


@pytest.mark.parametrize("n,sp,cout,with_alpha,with_affine", [(2, (32, 32, 32), 32, True, True),
                                                             (1, (18, 20, 34), 64, True, True),
                                                             (1, (31, 33, 37), 16, False, True),
                                                             (2, (16, 24, 40), 32, True, False)])
def test_bn_backward_apply_inside_the_stride2_convolution_equals_the_two_calls(n, sp, cout, with_alpha, with_affine):
    """segmi_bn_act_bwd_apply_conv: the BatchNorm / PReLU backward apply computed while the stride-2
    convolution (a transposed convolution's input gradient) stages its halo tile.  dx and the convolution
    result must equal segmi_bn_act_bwd_apply followed by segmi_conv3d_fwd bit for bit -- odd extents cover
    partial tiles and the ownership rule (every dx voxel written exactly once)."""
    c = 16
    dtype = torch.bfloat16
    x = rnd((n, c) + sp, 71, 2.0)
    dy = rnd((n, c) + sp, 72)
    xd, dyd = to_ndhwc(x, dtype), to_ndhwc(dy, dtype)
    mean = (rnd((c,), 73) * 0.5).to(DEV)
    invstd = (rnd((c,), 74).abs() + 0.5).to(DEV)
    gamma = (rnd((c,), 75) + 1.5).to(DEV) if with_affine else None
    beta = (rnd((c,), 76) * 0.3).to(DEV) if with_affine else None
    alpha = torch.full((1,), 0.25, device=DEV) if with_alpha else None
    coef = (rnd((2, c), 77) * 0.1).to(DEV).contiguous()
    w = (rnd((cout, c, 3, 3, 3), 78) * 0.1).to(DEV)
    pack = ops.wpack(dtype, 0, w, c, cout, 3)
    osp = tuple((v - 1) // 2 + 1 for v in sp)
    out0 = torch.full((n,) + osp + (cout,), float("nan"), dtype=dtype, device=DEV)
    out1 = torch.full_like(out0, float("nan"))
    dx0 = torch.full_like(xd, float("nan"))
    dx1 = torch.full_like(xd, float("nan"))
    assert ops.bn_act_bwd_apply_conv_ok(dyd, xd, dx1, out1)
    ops.bn_act_bwd_apply(dyd, xd, dx0, mean, invstd, gamma, beta, alpha, coef)
    ops.conv3d_fwd(dx0, out0, pack, w, 0, None, 3, 2)
    ops.bn_act_bwd_apply_conv(dyd, xd, dx1, mean, invstd, gamma, beta, alpha, coef, out1, pack)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dx1.float()).all()) and bool(torch.isfinite(out1.float()).all())
    bad = (dx0.view(torch.int16) != dx1.view(torch.int16))
    assert not bool(bad.any()), (int(bad.sum()), bad.nonzero()[:4].tolist(),
                                 float((dx0.float() - dx1.float()).abs().max()))
    assert torch.equal(out0, out1), float((out0.float() - out1.float()).abs().max())
    # not eligible: other channel counts, f32, aliased output
    assert not ops.bn_act_bwd_apply_conv_ok(dyd.float(), xd.float(), dx1.float(), out1.float())
    with pytest.raises(Exception):
        ops.bn_act_bwd_apply_conv(dyd, xd, xd, mean, invstd, gamma, beta, alpha, coef, out1, pack)


def test_cu_masked_stream_runs_kernels():
    """segmi_stream_create_cumask (hipExtStreamCreateWithCUMask): a stream restricted to 64 CUs computes the
    same bits as the default stream (measured as a scheduling tool in round 3 and not used: DESIGN section 6)."""
    s = ops.cu_masked_stream(64)
    x = to_ndhwc(rnd((1, 16, 8, 16, 32), 71), torch.bfloat16)
    sc, sh = (rnd((16,), 72) + 1.5).to(DEV), rnd((16,), 73).to(DEV)
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    ops.bn_act_fwd(x, y0, sc, sh, None)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        ops.bn_act_fwd(x, y1, sc, sh, None)
    s.synchronize()
    assert torch.equal(y0, y1)
    with pytest.raises(RuntimeError):
        ops.cu_masked_stream(13)
