"""End-to-end GPU parity of the UNet engine against the CPU oracle (oracle/unet_ref.py):
logits, Dice loss, every parameter gradient, BatchNorm running statistics and the post-Adam
parameters after the reference's training_step order (monai_unet.py:339-348).

Gates: f32 path logits <= 1e-3 relative (we assert 2e-4), loss within 1e-4
relative; bf16 path reported with looser, stated tolerances and argmax agreement.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.unet_ref import RefUNet, deterministic_fill_, ref_dice_loss, synthetic_batch  # noqa: E402
from segmantic_amd.seg.monai_unet import Net  # noqa: E402

DEV = "cuda:0"


def build_pair(k, cin, channels, strides, seed=0):
    ref = RefUNet(3, cin, k, channels, strides)
    deterministic_fill_(ref, seed)
    net = Net(num_classes=k, num_channels=cin, channels=channels, strides=strides)
    net.load_state_dict({"_model." + kk: v.clone() for kk, v in ref.state_dict().items()})
    return ref, net


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-20))


CONFIGS = [
    # name, K, cin, channels, strides, size, batch
    ("tiny-direct", 3, 1, (4, 8, 16), (2, 2), 16, 2),
    ("mfma-3level", 16, 1, (16, 32, 64), (2, 2), 32, 2),
    ("default-K3-32", 3, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2), 32, 2),   # BASELINE config 1 shape
    ("default-K16-48x32", 16, 2, (16, 32, 64, 128, 256), (2, 2, 2, 2), 32, 1),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_train_step_parity_f32(cfg):
    _, k, cin, channels, strides, size, batch = cfg
    ref, net = build_pair(k, cin, channels, strides)
    img, lab = synthetic_batch(batch, size, k, seed=1)
    if cin > 1:
        img = torch.cat([img * (1 + 0.1 * i) for i in range(cin)], 1)
    # ---- oracle: reference step order
    ref.train()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-4)
    out_ref = ref(img)
    opt.zero_grad()
    loss_ref = ref_dice_loss(out_ref, lab)
    loss_ref.backward()
    grads_ref = {n: p.grad.clone() for n, p in ref.named_parameters()}
    opt.step()
    # ---- HIP path
    net.to(DEV).train()
    net.mixed_precision = False
    eng = net._engine_for()
    res = net.training_step({"image": img.to(DEV), "label": lab.to(DEV)})
    torch.cuda.synchronize()
    got = eng._bufs["logits.t"][..., :eng.net.out_channels].float().cpu().permute(0, 4, 1, 2, 3)
    assert rel(got, out_ref.detach()) < 2e-4           # gate: 1e-3 relative
    loss = float(res["loss"].cpu())
    assert abs(loss - float(loss_ref.detach())) < 1e-4 * abs(float(loss_ref.detach()))
    # gradients (flat arena views).  Conv biases in front of a BatchNorm have an exactly-zero
    # gradient in exact arithmetic (pure rounding noise in both implementations), hence the
    # absolute floor relative to the largest gradient of the network.
    gmax = max(float(g.abs().max()) for g in grads_ref.values())
    bad = []
    for n, p in net._model.named_parameters():
        g = p.grad.cpu()
        gr = grads_ref[n]
        err = float((g - gr).abs().max())
        # PReLU alpha / bias gradients are single global sums with heavy cancellation
        rtol = 2e-2 if (n.endswith(".A.weight") or n.endswith(".bias")) else 2e-3
        lim = rtol * float(gr.abs().max()) + 2e-6 * gmax
        if err > lim:
            bad.append((n, err, float(gr.abs().max())))
    assert not bad, bad[:8]
    # post-Adam parameters and BN running statistics
    sd_ref = ref.state_dict()
    for kk, v in net._model.state_dict().items():
        r = sd_ref[kk]
        if not v.dtype.is_floating_point:
            assert int(v) == int(r)
            continue
        if "running" in kk:
            assert rel(v.cpu(), r) < 1e-4, kk
    # Adam's first step moves every weight by ~lr * sign(g): compare the update direction where
    # the gradient is well above rounding noise
    for n, p in net._model.named_parameters():
        gr = grads_ref[n]
        mask = gr.abs() > 1e-3 * gmax   # Adam's first step is lr*sign(g): skip rounding-noise gradients
        if mask.any():
            d_ref = (dict(ref.named_parameters())[n].detach() - p.detach().cpu())[mask].abs().max()
            assert float(d_ref) < 5e-6, (n, float(d_ref))


def test_eval_forward_folded_bn_f32():
    ref, net = build_pair(16, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2))
    img, _ = synthetic_batch(2, 32, 16, seed=3)
    ref.eval()
    with torch.no_grad():
        out_ref = ref(img)
    net.to(DEV).eval()
    with torch.no_grad():
        out = net(img.to(DEV))
    torch.cuda.synchronize()
    assert out.shape == out_ref.shape
    assert rel(out.float().cpu(), out_ref) < 2e-4
    lab = torch.argmax(out.float().cpu(), 1)
    lab_ref = torch.argmax(out_ref, 1)
    mism = lab != lab_ref
    if mism.any():   # only exact near-ties may flip
        top2 = torch.topk(out_ref, 2, dim=1).values
        gap = (top2[:, 0] - top2[:, 1])[mism]
        assert float(gap.max()) < 1e-4 * float(out_ref.abs().max())
    assert float(mism.float().mean()) < 1e-4


@pytest.mark.parametrize("train", [True, False])
def test_bf16_path_close_to_f32_oracle(train):
    k = 16
    ref, net = build_pair(k, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2))
    img, lab = synthetic_batch(2, 32, k, seed=5)
    ref.train(train)
    out_ref = ref(img).detach()
    net.to(DEV).train(train)
    net.mixed_precision = True
    with torch.no_grad():
        out = net(img.to(DEV))
    torch.cuda.synchronize()
    assert out.dtype == torch.bfloat16
    err = rel(out.float().cpu(), out_ref)
    # bf16 storage through 17 normalised layers (reported, not the parity gate).  Measured in round 4 over three
    # seeds: training mode 7.4e-3 .. 9.0e-3 / 99.2 % / loss within 5.5e-6, eval 5.6e-3 .. 6.1e-3 / 99.4 %; the gates
    # (6e-2 / 97 % / 2e-2 until then) are those numbers + margin, as at 64^3 / 128^3 below
    assert err < 2e-2, err
    agree = float((torch.argmax(out.float().cpu(), 1) == torch.argmax(out_ref, 1)).float().mean())
    assert agree > 0.985, agree
    if train:
        res = net.training_step({"image": img.to(DEV), "label": lab.to(DEV)})
        torch.cuda.synchronize()
        assert abs(float(res["loss"].cpu()) - float(ref_dice_loss(out_ref, lab))) < 1e-4


@pytest.mark.parametrize("size", [64, 128])
def test_bf16_training_forward_vs_oracle_at_benchmark_scale(size, record_property):
    """the benchmarked bf16 path against the CPU oracle at BASELINE config 2's scale (one patch
    of 64^3 / 128^3, 16 labels, training-mode BatchNorm): relative logit error, argmax agreement and loss deviation
    are RECORDED (junit properties + stdout) and gated at the measured values + margin.  The oracle forward takes a
    few seconds on the test box's host cores; the exact-f32 path is checked on the same input against the north_star
    gate (1e-3 relative; asserted 1e-5)."""
    k = 16
    ref, net = build_pair(k, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2))
    img, lab = synthetic_batch(1, size, k, seed=11)
    ref.train()
    with torch.no_grad():
        out_ref = ref(img)
    loss_ref = float(ref_dice_loss(out_ref, lab))
    net.to(DEV).train()
    # f32 storage, exact-f32 MFMA chains
    with torch.no_grad():
        out32 = net(img.to(DEV)).float().cpu()
    e32 = rel(out32, out_ref)
    # bf16 storage
    net.mixed_precision = True
    with torch.no_grad():
        out16 = net(img.to(DEV)).float().cpu()
    torch.cuda.synchronize()
    e16 = rel(out16, out_ref)
    agree = float((torch.argmax(out16, 1) == torch.argmax(out_ref, 1)).float().mean())
    agree32 = float((torch.argmax(out32, 1) == torch.argmax(out_ref, 1)).float().mean())
    res = net.training_step({"image": img.to(DEV), "label": lab.to(DEV)})
    dl = abs(float(res["loss"].cpu()) - loss_ref) / abs(loss_ref)
    print(f"\nbf16 vs oracle @ {size}^3: rel. logit error {e16:.3e}, argmax agreement {agree:.5f}, loss rel. dev. {dl:.2e}; "
          f"f32: rel. error {e32:.2e}, argmax agreement {agree32:.6f}")
    for name, v in (("bf16_rel_err", e16), ("bf16_argmax_agreement", agree), ("bf16_loss_rel_dev", dl),
                    ("f32_rel_err", e32), ("f32_argmax_agreement", agree32)):
        record_property(f"{name}_{size}", v)
    # measured on MI355X (round 4): bf16 8.9e-3 / 99.24 % / 1.3e-6 at 64^3, 7.2e-3 / 99.27 % / 6.9e-7 at 128^3;
    # f32 9.5e-7 / 100 % and 9.2e-7 / 99.9997 %.  Gates = measured + margin (north_star: 1e-3 on the f32 logits).
    assert e32 < 1e-5, e32
    assert agree32 > 0.9999, agree32
    assert e16 < 2e-2, e16
    assert agree > 0.985, agree
    assert dl < 1e-4, dl


def test_autograd_bridge_matches_fused_step():
    ref, net = build_pair(3, 1, (16, 32, 64), (2, 2))
    img, lab = synthetic_batch(2, 32, 3, seed=7)
    net.to(DEV).train()
    out = net(img.to(DEV))
    assert out.requires_grad
    loss = net.loss_function(out, lab.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    # parameter-shaped gradient views in state-dict order (the arena itself pads the class axis)
    g1 = torch.cat([net._engine._galias[n].reshape(-1) for n, _ in net._model.named_parameters()])
    ref.train()
    l2 = ref_dice_loss(ref(img), lab)
    l2.backward()
    assert abs(float(loss.cpu()) - float(l2)) < 1e-5
    gref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    big = gref.abs() > 1e-3 * gref.abs().max()
    assert float((g1.cpu() - gref)[big].abs().max() / gref.abs().max()) < 2e-3


def test_checkpoint_roundtrip_and_reference_key_layout(tmp_path):
    ref, net = build_pair(2, 1, (16, 32, 64, 128, 256), (2, 2, 2, 2))
    assert sum(p.numel() for p in net.parameters()) == 4808917
    net.to(DEV)
    p = tmp_path / "epoch=3-val_loss=0.12-val_dice=0.8765.ckpt"
    net.save_checkpoint(p, epoch=3)
    ck = torch.load(p, map_location="cpu", weights_only=False)
    assert set(ck["state_dict"]) == {"_model." + k for k in ref.state_dict()}
    assert ck["hyper_parameters"]["num_classes"] == 2
    net2 = Net.load_from_checkpoint(p)
    for (k1, v1), (k2, v2) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1.cpu(), v2.cpu())
    # scripts/extract_unet.py layout: bare inner state dict
    torch.save({k[len("_model."):]: v for k, v in ck["state_dict"].items()}, tmp_path / "unet.pth")
    net3 = Net.load_from_checkpoint(tmp_path / "unet.pth")
    assert net3.num_classes == 2


def test_net_refuses_cpu_compute():
    net = Net(num_classes=2, channels=(4, 8), strides=(2,))
    with pytest.raises(RuntimeError, match="MI355X"):
        net(torch.zeros(1, 1, 8, 8, 8))
