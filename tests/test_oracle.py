"""CPU tests of the oracle itself: facts pinned by the reference / its dependencies' published
figures, known-answer cases, and the committed golden vectors (drift guard)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.metrics_ref import ref_argmax, ref_dice_metric, ref_normalize
from oracle.resample_ref import ref_resample, ref_resample_grid, resample_size
from oracle.sliding_ref import ref_sliding_window_inference, scan_intervals, window_starts
from oracle.unet_ref import RefUNet, deterministic_fill_, ref_dice_loss, synthetic_batch

torch.set_num_threads(4)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(golden_dir / "oracle_goldens.npz", allow_pickle=False)


# ------------------------------------------------------------------ pinned by published facts
@pytest.mark.parametrize("k,count", [(2, 4808917), (3, 4809920), (16, 4827873), (32, 4862497)])
def test_unet_parameter_count(k, count):
    """4,808,917 for (in=1, out=2) is the figure MONAI's spleen tutorial prints for exactly
    UNet(channels 16-256, strides 2, num_res_units=2, norm=BATCH); SURVEY.md 8a lists the rest."""
    assert sum(p.numel() for p in RefUNet(3, 1, k).parameters()) == count


def test_unet_state_dict_layout():
    sd = RefUNet(3, 1, 2).state_dict()
    assert len(sd) == 148
    assert sd["model.0.conv.unit0.conv.weight"].shape == (16, 1, 3, 3, 3)
    assert sd["model.0.residual.weight"].shape == (16, 1, 3, 3, 3)
    assert sd["model.0.conv.unit0.adn.A.weight"].shape == (1,)
    assert sd["model.2.0.conv.weight"].shape == (32, 2, 3, 3, 3)          # ConvT: [Cin, Cout, k..]
    assert sd["model.2.1.conv.unit0.conv.weight"].shape == (2, 2, 3, 3, 3)
    assert "model.2.1.conv.unit0.adn.N.weight" not in sd                     # top unit is conv only
    bottom = "model.1.submodule.1.submodule.1.submodule.1.submodule."
    assert sd[bottom + "conv.unit0.conv.weight"].shape == (256, 128, 3, 3, 3)
    assert sd[bottom + "residual.weight"].shape == (256, 128, 1, 1, 1)      # k1 when only channels change
    assert sd["model.1.submodule.1.submodule.1.submodule.2.0.conv.weight"].shape == (384, 64, 3, 3, 3)


def test_unet_output_shape_and_skip_order():
    net = RefUNet(3, 2, 5, (4, 8, 16), (2, 2)).eval()
    with torch.no_grad():
        y = net(torch.zeros(1, 2, 16, 16, 8))
    assert y.shape == (1, 5, 16, 16, 8)


def test_sliding_window_schedule_of_baseline_configs():
    """512^3 / roi 128 / overlap 0.5 -> interval 64, starts [0..384], 343 windows;
    reference default overlap 0.25 -> interval 96, 125 windows."""
    assert scan_intervals((512,) * 3, (128,) * 3, 0.5) == [64, 64, 64]
    per, wins = window_starts((512,) * 3, (128,) * 3, 0.5)
    assert per[0] == [0, 64, 128, 192, 256, 320, 384] and len(wins) == 343
    assert wins[1] == (0, 0, 64)                      # last spatial dim fastest
    per, wins = window_starts((512,) * 3, (128,) * 3, 0.25)
    assert per[0] == [0, 96, 192, 288, 384] and len(wins) == 125
    assert scan_intervals((128, 200, 128), (128, 128, 128), 0.25) == [128, 96, 128]


def test_resample_geometry_asserted_by_reference_tests():
    """tests/image/test_image.py:33-52: half the spacing -> double the size; resample_to_ref ->
    size and spacing of the reference grid."""
    lf = np.zeros((5, 5, 5), np.uint8)
    out, sp = ref_resample(lf, (0.5, 0.6, 0.7), (0.25, 0.3, 0.35))
    assert out.shape == (10, 10, 10) and sp == (0.25, 0.3, 0.35)
    out = ref_resample_grid(lf, (0.5, 0.6, 0.7), (0, 0, 0), np.eye(3), (12, 10, 7), (0.25, 0.3, 0.35),
                            (1.3, -2.1, 0.75), np.eye(3), True)
    assert out.shape == (7, 10, 12) and out.dtype == np.uint8
    assert resample_size((5, 5, 5), (0.5, 0.6, 0.7), (0.3, 0.3, 0.3)) == (9, 10, 12)


# ------------------------------------------------------------------ known answers
def test_dice_loss_known_answers():
    n, k, sp = 2, 4, (6, 5, 4)
    g = torch.Generator().manual_seed(0)
    lab = torch.randint(0, k, (n, 1) + sp, generator=g).float()
    # uniform logits: p = 1/K everywhere
    loss = ref_dice_loss(torch.zeros((n, k) + sp), lab)
    N = sp[0] * sp[1] * sp[2]
    exp = 0.0
    for b in range(n):
        for c in range(k):
            nc = float((lab[b] == c).sum())
            exp += 1 - (2 * nc / k + 1e-5) / (nc + N / k + 1e-5)
    assert abs(float(loss) - exp / (n * k)) < 1e-6
    # perfect, saturated prediction -> loss ~ 0
    onehot = F.one_hot(lab[:, 0].long(), k).movedim(-1, 1).float()
    assert float(ref_dice_loss(onehot * 100.0, lab)) < 1e-5


def test_dice_metric_absent_class_is_nan_and_ignored():
    p = torch.tensor([0, 1, 1, 2, 2, 2]).reshape(1, 1, 1, 2, 3)
    t = torch.tensor([0, 1, 2, 2, 2, 0]).reshape(1, 1, 1, 2, 3)
    d, m = ref_dice_metric(p, t, 4, include_background=False)
    assert torch.isnan(d[0, 2])                                # class 3 absent in truth
    assert abs(float(d[0, 0]) - 2 * 1 / (2 + 1)) < 1e-6         # class 1
    assert abs(float(d[0, 1]) - 2 * 2 / (3 + 3)) < 1e-6         # class 2
    assert abs(float(m) - (2 / 3 + 2 / 3) / 2) < 1e-6


def test_argmax_first_index_on_ties():
    lg = torch.tensor([[1.0, 3.0, 3.0, 2.0]]).reshape(1, 4, 1, 1, 1)
    assert int(ref_argmax(lg)) == 1


def test_trilinear_reproduces_a_linear_ramp_and_border_rules():
    ramp = np.fromfunction(lambda z, y, x: 2.0 * x - 3.0 * y + 0.5 * z + 1.0, (6, 7, 8)).astype(np.float32)
    out = ref_resample_grid(ramp, (1, 1, 1), (0, 0, 0), np.eye(3), (13, 11, 9), (0.5, 0.5, 0.5),
                            (0.25, 0.25, 0.25), np.eye(3), False)
    zz, yy, xx = np.meshgrid(np.arange(9), np.arange(11), np.arange(13), indexing="ij")
    exp = 2.0 * (0.25 + 0.5 * xx) - 3.0 * (0.25 + 0.5 * yy) + 0.5 * (0.25 + 0.5 * zz) + 1.0
    np.testing.assert_allclose(out, exp, atol=1e-5)
    a = np.arange(4, dtype=np.float32).reshape(1, 1, 4) + 1
    # continuous index -0.5 (inside, clamps to the edge value), -0.6 (outside -> 0), 3.49, 3.5 (outside)
    o = ref_resample_grid(a, (1, 1, 1), (0, 0, 0), np.eye(3), (1, 1, 1), (1, 1, 1), (-0.5, 0, 0), np.eye(3), False)
    assert o[0, 0, 0] == 1.0
    o = ref_resample_grid(a, (1, 1, 1), (0, 0, 0), np.eye(3), (1, 1, 1), (1, 1, 1), (-0.6, 0, 0), np.eye(3), False)
    assert o[0, 0, 0] == 0.0
    o = ref_resample_grid(a, (1, 1, 1), (0, 0, 0), np.eye(3), (1, 1, 1), (1, 1, 1), (3.49, 0, 0), np.eye(3), False)
    assert o[0, 0, 0] == 4.0
    o = ref_resample_grid(a, (1, 1, 1), (0, 0, 0), np.eye(3), (1, 1, 1), (1, 1, 1), (3.5, 0, 0), np.eye(3), True)
    assert o[0, 0, 0] == 0.0


def test_sliding_window_identity_predictor_returns_the_image():
    img, _ = synthetic_batch(1, 24, 2, seed=3)
    out, cnt, wins = ref_sliding_window_inference(img, (16, 16, 16), 4, lambda x: x, 0.5)
    assert torch.allclose(out, img, atol=1e-6)
    assert cnt.min() >= 1 and cnt.max() == 8 and len(wins) == 8
    # image smaller than the roi: symmetric zero padding, output cropped back
    small = img[..., :10, :12, :16]
    out, cnt, wins = ref_sliding_window_inference(small, (16, 16, 16), 4, lambda x: x, 0.25)
    assert out.shape == small.shape and torch.allclose(out, small)


def test_normalize_zero_std_channel():
    x = np.stack([np.full((3, 3, 3), 5.0, np.float32), np.arange(27, dtype=np.float32).reshape(3, 3, 3)])
    y = ref_normalize(x)
    assert np.all(y[0] == 0.0)
    assert abs(y[1].mean()) < 1e-6 and abs(y[1].std() - 1.0) < 1e-6


# ------------------------------------------------------------------ golden vectors (drift guard)
def test_oracle_matches_committed_goldens(gold):
    ch, st, K = (4, 8, 16), (2, 2), 3
    net = deterministic_fill_(RefUNet(3, 1, K, ch, st), 0).train()
    img, lab = synthetic_batch(2, 16, K, seed=1)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    y = net(img)
    opt.zero_grad()
    loss = ref_dice_loss(y, lab)
    loss.backward()
    np.testing.assert_allclose(y.detach().numpy(), gold["tiny_logits"], rtol=0, atol=2e-5)
    assert abs(float(loss) - float(gold["tiny_loss"])) < 1e-6
    params = dict(net.named_parameters())
    for i, k in enumerate(gold["tiny_sel_keys"]):
        g = gold[f"tiny_grad_{i}"]
        np.testing.assert_allclose(params[str(k)].grad.numpy(), g, rtol=0, atol=2e-3 * np.abs(g).max() + 1e-9)
    opt.step()
    np.testing.assert_allclose(net.state_dict()["model.0.conv.unit0.adn.N.running_mean"].numpy(),
                               gold["tiny_running_mean"], atol=1e-6)


def test_sliding_window_goldens(gold):
    g = torch.Generator().manual_seed(int(gold["sw_vol_seed"]))
    vol = torch.rand((1, 1, 40, 40, 40), generator=g)
    w = torch.rand((3, 1, 3, 3, 3), generator=g) - 0.5
    for ov, tag in ((0.25, "sw_25"), (0.5, "sw_50")):
        o, cnt, wins = ref_sliding_window_inference(vol, (16, 16, 16), 4, lambda x: F.conv3d(x, w, padding=1), ov)
        assert np.array_equal(np.array(wins, dtype=np.int32), gold[tag + "_starts"])
        assert np.array_equal(np.bincount(cnt.reshape(-1).numpy().astype(np.int64)), gold[tag + "_count_hist"])
        np.testing.assert_allclose(o[0, :, 10:14, 10:14, 10:14].numpy(), gold[tag + "_out_block"], atol=1e-5)
    assert int(gold["sw512_nwin"]) == 343 and int(gold["sw512_nwin_ov25"]) == 125


# ------------------------------------------------------------------ pinned by the reference's own vector
def test_select_best_matches_the_reference_test_vector(golden_dir):
    """/root/reference/tests/seg/test_transforms.py:9-43: preds [1,1,1] / [2,0,2] / [2,1,0], dict
    {1: 0, 2: 1, 0: 2} -> [2,1,0]; then the same through one_hot(num_classes=3, dim=0)."""
    import json

    from oracle.ensemble_ref import ref_select_best
    g = json.loads((golden_dir / "reference_select_best.json").read_text())
    shape = tuple(g["shape"])
    preds = [torch.tensor(p, dtype=torch.float32).reshape(shape) for p in g["preds"]]
    lmd = {int(t): int(m) for t, m in g["label_model_dict"]}
    assert list(lmd.items()) == [(1, 0), (2, 1), (0, 2)]                    # insertion order matters
    want = torch.tensor(g["expected"], dtype=torch.float32).reshape(shape)
    assert torch.equal(ref_select_best(preds, lmd), want)
    k = g["num_classes"]
    oh = lambda t: F.one_hot(t.long()[0], k).movedim(-1, 0).float()         # monai one_hot(dim=0) of [1, ...]
    got = ref_select_best([oh(p) for p in preds], lmd)
    assert got.shape == (k,) + shape[1:] and torch.equal(got, oh(want))
