"""GPU end-to-end tests: committed golden vectors, sliding-window inference through the real
network vs the oracle, image ops (drop-in ``processing`` API), and the
``segmantic-unet train-config`` / ``predict`` surface on a tiny synthetic NIfTI dataset."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle.sliding_ref import ref_sliding_window_inference  # noqa: E402
from oracle.unet_ref import RefUNet, deterministic_fill_, ref_dice_loss, synthetic_batch  # noqa: E402
from segmantic_amd.seg.inferers import SlidingWindowInferer, sliding_window_inference  # noqa: E402
from segmantic_amd.seg.monai_unet import Net  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(golden_dir / "oracle_goldens.npz", allow_pickle=False)


def pair(k, channels, strides, cin=1):
    ref = deterministic_fill_(RefUNet(3, cin, k, channels, strides), 0)
    net = Net(num_classes=k, num_channels=cin, channels=channels, strides=strides)
    net.load_state_dict({"_model." + kk: v.clone() for kk, v in ref.state_dict().items()})
    return ref, net.to(DEV)


def test_hip_path_reproduces_committed_goldens(gold):
    ref, net = pair(3, (4, 8, 16), (2, 2))
    img, lab = synthetic_batch(2, 16, 3, seed=1)
    net.train()
    res = net.training_step({"image": img.to(DEV), "label": lab.to(DEV)})
    torch.cuda.synchronize()
    logits = net._engine._bufs["logits.t"][..., :net.num_classes].float().cpu().permute(0, 4, 1, 2, 3).numpy()
    g = gold["tiny_logits"]
    assert np.abs(logits - g).max() / np.abs(g).max() < 2e-4            # 1e-3 gate
    assert abs(float(res["loss"].cpu()) - float(gold["tiny_loss"])) < 1e-4 * float(gold["tiny_loss"])
    sd = net._model.state_dict()
    for i, k in enumerate(gold["tiny_sel_keys"]):
        a = gold[f"tiny_adam_{i}"]
        # Adam's first step is lr * sign(g): equal except where g is rounding noise (sign may flip)
        close = np.abs(sd[str(k)].cpu().numpy() - a) < 3e-6
        assert close.mean() > 0.99, (str(k), close.mean())
    np.testing.assert_allclose(sd["model.0.conv.unit0.adn.N.running_mean"].cpu().numpy(), gold["tiny_running_mean"], atol=1e-5)
    # eval mode (folded BN) against the golden eval logits
    _, net2 = pair(3, (4, 8, 16), (2, 2))
    net2.eval()
    with torch.no_grad():
        y = net2(img.to(DEV)).float().cpu().numpy()
    ge = gold["tiny_eval_logits"]
    assert np.abs(y - ge).max() / np.abs(ge).max() < 2e-4
    # full default net, K=3, 32^3 (BASELINE config 1 shape)
    _, net3 = pair(3, (16, 32, 64, 128, 256), (2, 2, 2, 2))
    img32, lab32 = synthetic_batch(1, 32, 3, seed=2)
    net3.train()
    with torch.no_grad():
        y = net3(img32.to(DEV)).float().cpu()
    blk = gold["full_logits_block"]
    assert np.abs(y[0, :, :8, :8, :8].numpy() - blk).max() / np.abs(blk).max() < 2e-4
    assert abs(float(y.double().abs().sum()) - float(gold["full_logits_abs_sum"])) < 1e-4 * float(gold["full_logits_abs_sum"])


@pytest.mark.parametrize("overlap,mode", [(0.25, "constant"), (0.5, "constant"), (0.5, "gaussian")])
def test_sliding_window_through_the_network_vs_oracle(overlap, mode):
    ref, net = pair(4, (16, 32, 64), (2, 2))
    ref.eval(); net.eval()
    img, _ = synthetic_batch(1, 40, 4, seed=4)
    img = img[..., :40, :36, :44]
    with torch.no_grad():
        o_ref, _, wins = ref_sliding_window_inference(img, (16, 16, 16), 4, ref, overlap, mode)
        res = sliding_window_inference(img.to(DEV), (16, 16, 16), 4, net, overlap, mode, return_labels=True)
    torch.cuda.synchronize()
    got = res.logits.cpu()
    assert got.shape == o_ref.shape
    assert float((got - o_ref).abs().max() / o_ref.abs().max()) < 2e-4
    lab_ref = torch.argmax(o_ref, 1)
    mism = res.labels.cpu().long()[:, 0] != lab_ref
    if mism.any():    # only near-ties may flip
        top2 = torch.topk(o_ref, 2, dim=1).values
        assert float((top2[:, 0] - top2[:, 1])[mism].max()) < 1e-4 * float(o_ref.abs().max())
    assert float(mism.float().mean()) < 1e-3
    # the label volume is bit-exact w.r.t. OUR logits (fused argmax == torch.argmax)
    assert torch.equal(res.labels.cpu().long()[:, 0], torch.argmax(got, 1))


@pytest.mark.parametrize("mode", ["constant", "gaussian"])
@pytest.mark.parametrize("mixed", [False, True])
def test_sliding_window_deferred_and_streaming_blends_are_bit_identical(mode, mixed):
    """Same windows, same f32 addition order: keeping all predictions and blending once gives
    the very bits the per-group accumulator gives (logits, counts and labels)."""
    _, net = pair(16, (16, 32, 64), (2, 2))
    net.eval()
    net.mixed_precision = mixed
    img, _ = synthetic_batch(1, 40, 4, seed=5)
    img = img[..., :40, :36, :44].to(DEV)
    with torch.no_grad():
        a = sliding_window_inference(img, (16, 16, 16), 4, net, 0.5, mode, return_labels=True,
                                     blend="stream")
        la, ca, ba = a.logits.clone(), a.count.clone(), a.labels.clone()
        b = sliding_window_inference(img, (16, 16, 16), 4, net, 0.5, mode, return_labels=True,
                                     blend="deferred")
        c = sliding_window_inference(img, (16, 16, 16), 4, net, 0.5, mode, return_labels=True,
                                     blend="deferred", return_logits=False)
    torch.cuda.synchronize()
    assert torch.equal(la, b.logits) and torch.equal(ca, b.count) and torch.equal(ba, b.labels)
    assert c.logits is None and torch.equal(c.labels, ba)


@pytest.mark.parametrize("world", [2, 3])
def test_sliding_window_z_slabs_reproduce_the_full_volume_bit_for_bit(world):
    """Sharded inference of one volume: every rank's slab (all windows that touch it, ordered
    blend) equals the same planes of the unsharded result -- logits, counts and labels."""
    from segmantic_amd.seg.inferers import z_slabs
    _, net = pair(4, (16, 32, 64), (2, 2))
    net.eval()
    img, _ = synthetic_batch(1, 40, 4, seed=8)
    img = img[..., :37, :36, :40].to(DEV)
    with torch.no_grad():
        full = sliding_window_inference(img, (16, 16, 16), 4, net, 0.5, "gaussian", return_labels=True,
                                        blend="deferred")
        fl, fc, fb = full.logits.clone(), full.count.clone(), full.labels.clone()
        for a, b in z_slabs(37, world):
            part = sliding_window_inference(img, (16, 16, 16), 4, net, 0.5, "gaussian",
                                            return_labels=True, z_slab=(a, b))
            assert torch.equal(part.logits, fl[:, :, a:b]) and torch.equal(part.count, fc[:, a:b])
            assert torch.equal(part.labels, fb[:, :, a:b])


def test_sliding_window_image_smaller_than_roi_and_inferer_class():
    ref, net = pair(2, (16, 32), (2,))
    ref.eval(); net.eval()
    img, _ = synthetic_batch(1, 16, 2, seed=6)
    img = img[..., :12, :16, :10]
    with torch.no_grad():
        o_ref, _, _ = ref_sliding_window_inference(img, (16, 16, 16), 4, ref, 0.25)
        got = SlidingWindowInferer(roi_size=(16, 16, 16), sw_batch_size=4)(img.to(DEV), net)
    assert got.shape == o_ref.shape
    assert float((got.cpu() - o_ref).abs().max() / o_ref.abs().max()) < 2e-4


def test_full_size_properties_128_bf16():
    """BASELINE-size checks through size-independent properties: batch items are independent in
    eval mode, sliding-window of a single-window volume equals the plain forward, logits finite."""
    torch.manual_seed(128)          # (own seed for the initialisation: see test_full_size_c4_160_k32_training_step_bf16)
    net = Net(num_classes=16).to(DEV).eval()
    net.mixed_precision = True
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 1, 128, 128, 128), generator=g).to(DEV)
    with torch.no_grad():
        y = net(x).float().clone()
        y0 = net(x[:1].contiguous()).float().clone()
        sw = sliding_window_inference(x[:1].contiguous(), (128, 128, 128), 4, net)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all())
    assert torch.equal(y[:1], y0)                                  # deterministic, batch independent
    assert float((sw - y0).abs().max()) < 1e-6 * float(y0.abs().max()) + 1e-6
    net.train()
    img, lab = x, torch.randint(0, 16, (2, 1, 128, 128, 128), generator=g).float().to(DEV)
    l1 = float(net.training_step({"image": img, "label": lab})["loss"].cpu())
    for _ in range(5):
        l2 = float(net.training_step({"image": img, "label": lab})["loss"].cpu())
    assert np.isfinite(l1) and l2 < l1                              # the optimiser descends


def test_processing_dropin_geometry_and_values():
    """reference tests/image/test_image.py:33-52 restated on the HIP resample path"""
    from oracle.resample_ref import ref_resample_grid
    from segmantic_amd.image import processing
    lf = processing.make_image(shape=(5, 5, 5), spacing=(0.5, 0.6, 0.7))
    for i in range(5):
        lf.data[i] = i
    spacing = [s / 2.0 for s in lf.GetSpacing()]
    res = processing.resample(lf, target_spacing=spacing)
    assert list(res.GetSize()) == [2 * s for s in lf.GetSize()]
    ref_img = processing.make_image((12, 10, 7), spacing, pixel_type=processing.sitkUInt16)
    ref_img.SetOrigin([1.3, -2.1, 0.75])
    out = processing.resample_to_ref(lf, ref_img, nearest=True)
    assert list(out.GetSize()) == list(ref_img.GetSize()) and list(out.GetSpacing()) == list(ref_img.GetSpacing())
    exp = ref_resample_grid(lf.numpy(), lf.GetSpacing(), (0, 0, 0), np.eye(3), (12, 10, 7), spacing, (1.3, -2.1, 0.75), np.eye(3), True)
    assert np.array_equal(out.numpy(), exp)
    # pad o crop_center identity, extract_slices (test_image.py:7-30)
    sl = processing.extract_slices(lf, axis=2)
    assert len(sl) == 5 and sl[3].GetSize() == (5, 5) and int(sl[3].data[0, 0]) == 3
    c = processing.crop_center(lf, target_size=(5, 5, 1))
    assert c.GetSize()[2] == 1
    # rotation about z by 90 degrees through apply_transform
    f = processing.Image(np.random.default_rng(0).standard_normal((6, 8, 8)).astype(np.float32))
    T = np.eye(4); T[:2, :2] = [[0, -1], [1, 0]]; T[:3, 3] = [7, 0, 0]
    rot = processing.apply_transform(f, f, T, nearest=False)
    exp = ref_resample_grid(f.numpy(), (1, 1, 1), (0, 0, 0), np.eye(3), (8, 8, 6), (1, 1, 1), (0, 0, 0), np.eye(3), False, transform=T)
    np.testing.assert_allclose(rot.numpy(), exp, atol=1e-6)


def _write_dataset(root: Path, n=4, size=24, classes=3):
    from segmantic_amd.data.nifti import write_nifti
    (root / "image").mkdir(parents=True)
    (root / "label").mkdir()
    A = np.diag([1.0, 1.0, 1.0, 1.0])
    for i in range(n):
        img, lab = synthetic_batch(1, size, classes, seed=20 + i)
        write_nifti(root / "image" / f"c{i}.nii.gz", (img[0, 0].numpy() * 100 + 300).astype(np.float32).transpose(2, 1, 0), A)
        write_nifti(root / "label" / f"c{i}.nii.gz", lab[0, 0].numpy().astype(np.uint8).transpose(2, 1, 0), A)
    dl = {"labels": {"1": "a", "2": "b"},
          "training": [{"image": f"image/c{i}.nii.gz", "label": f"label/c{i}.nii.gz"} for i in range(n - 1)],
          "validation": [{"image": f"image/c{n - 1}.nii.gz", "label": f"label/c{n - 1}.nii.gz"}],
          "test": [f"image/c{n - 1}.nii.gz"]}
    (root / "dataset.json").write_text(json.dumps(dl))
    # predict reads the "test" section through the decathlon loader, which takes image/label dicts
    dl["test"] = [{"image": f"image/c{n - 1}.nii.gz", "label": f"label/c{n - 1}.nii.gz"}]
    (root / "predict.json").write_text(json.dumps(dl))
    return root / "dataset.json"


def test_cli_train_config_then_predict(tmp_path):
    """BASELINE config 0 surface (segmantic-unet train-config / predict) on the GPU path."""
    import yaml
    from typer.testing import CliRunner

    from segmantic_amd.commands.monai_unet_cli import app
    datalist = _write_dataset(tmp_path / "data")
    out = tmp_path / "results"
    cfg = {"datalist": str(datalist), "output_dir": str(out), "spatial_size": [16, 16, 16],
           "channels": [16, 32, 64], "strides": [2, 2], "max_epochs": 3, "mixed_precision": False,
           "num_samples": 2, "gpu_ids": [0], "optimizer": {"optimizer": "Adam", "lr": 1e-3, "amsgrad": False}}
    (tmp_path / "cfg.yml").write_text(yaml.safe_dump(cfg))
    runner = CliRunner()
    res = runner.invoke(app, ["train-config", "-c", str(tmp_path / "cfg.yml")])
    assert res.exit_code == 0, (res.output, res.exception)
    ckpts = sorted(out.glob("epoch=*-val_loss=*-val_dice=*.ckpt"))
    assert 1 <= len(ckpts) <= 3
    assert (out / "Dataset.json").exists() and (out / "logs" / "metrics.csv").exists()
    ck = torch.load(ckpts[-1], map_location="cpu", weights_only=False)
    assert ck["hyper_parameters"]["num_classes"] == 3 and "_model.model.0.conv.unit0.conv.weight" in ck["state_dict"]
    res = runner.invoke(app, ["predict", "-d", str(datalist.parent / "predict.json"), "-m", str(ckpts[-1]), "-r", str(tmp_path / "pred"), "--gpu-ids", "0"])
    assert res.exit_code == 0, (res.output, res.exception)
    from segmantic_amd.data.nifti import read_nifti
    pred, _ = read_nifti(tmp_path / "pred" / "c3.nii.gz")
    assert pred.shape == (24, 24, 24) and pred.max() <= 2
    assert (tmp_path / "pred" / f"mean_dice_{ckpts[-1].stem}_generalized_score.txt").exists()


def test_training_with_spatial_and_intensity_augmentation(tmp_path):
    """`augment_spatial` / `augment_intensity` of the config schema (monai_unet.py:181-212) run on the
    GPU sampler: patches keep their shape, labels stay integral, the loss is finite and decreasing
    over a few epochs is not required (random data) -- the run completes and checkpoints."""
    import warnings

    from segmantic_amd.seg.monai_unet import train
    datalist = _write_dataset(tmp_path / "data")
    out = tmp_path / "results"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        train(datalist=datalist, output_dir=out, spatial_size=[16, 16, 16], channels=(16, 32, 64),
              strides=(2, 2), max_epochs=2, mixed_precision=False, num_samples=2, gpu_ids=[0],
              augment_spatial=True, augment_intensity=True)
    assert len(list(out.glob("epoch=*-val_loss=*-val_dice=*.ckpt"))) >= 1
    rows = (out / "logs" / "metrics.csv").read_text().strip().splitlines()
    assert len(rows) == 3 and all(np.isfinite(float(r.split(",")[1])) for r in rows[1:])
    # the sampler itself: augmented patches are finite and the warped labels are still class ids
    from segmantic_amd.seg import trainer as tr

    class _N:  # the attributes make_batch reads
        spatial_size, num_samples, num_classes = [16, 16, 16], 4, 3
        augment_spatial, augment_intensity = True, True
        device = torch.device(DEV)
    from segmantic_amd.seg.dataset import PairedDataSet
    ds = PairedDataSet.load_from_json(datalist)
    cache = tr.CachedVolumes(ds.training_files(), _N.device, 3)
    rng = np.random.RandomState(11)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(12):
            b = tr.make_batch(_N, cache, [0, 1], rng)
            assert b["image"].shape == (8, 1, 16, 16, 16) and b["label"].shape == (8, 1, 16, 16, 16)
            assert bool(torch.isfinite(b["image"]).all())
            lab = b["label"].cpu()
            assert bool((lab == lab.round()).all()) and 0 <= float(lab.min()) and float(lab.max()) <= 2


def test_fit_with_batch_prefetch_is_bit_identical_to_the_serial_sampler(tmp_path, monkeypatch):
    """the next step's batch is built on a side stream while the current step runs (BatchPrefetcher):
    same draws in the same order, so the trained weights equal those of the serial loop bit for bit"""
    import warnings

    from segmantic_amd.seg.monai_unet import Net, train
    datalist = _write_dataset(tmp_path / "data")
    weights = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SEGMI_PREFETCH", flag)
        torch.manual_seed(1234)
        np.random.seed(1234)
        out = tmp_path / f"results{flag}"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            train(datalist=datalist, output_dir=out, spatial_size=[16, 16, 16], channels=(16, 32, 64),
                  strides=(2, 2), max_epochs=2, mixed_precision=True, num_samples=2, gpu_ids=[0],
                  augment_intensity=True)
        ck = sorted(out.glob("epoch=*.ckpt"))[-1]
        sd = torch.load(ck, map_location="cpu", weights_only=False)["state_dict"]
        weights.append(sd)
    assert weights[0].keys() == weights[1].keys()
    for k in weights[0]:
        assert torch.equal(weights[0][k], weights[1][k]), k


def test_cross_validate_and_ensemble_predict(tmp_path):
    """`cross-validate` (k-fold data lists, one training process per fold) and `ensemble-predict`
    (mean / vote / select_best) of the reference CLI on the GPU path."""
    import yaml
    from typer.testing import CliRunner

    from segmantic_amd.commands.monai_unet_cli import app
    from segmantic_amd.data.nifti import read_nifti
    datalist = _write_dataset(tmp_path / "data")
    img_dir, lab_dir = tmp_path / "data" / "image", tmp_path / "data" / "label"
    tissue = tmp_path / "labels.txt"
    tissue.write_text("V7\nN2\nC1.00 0.00 0.00 0.50 A\nC0.00 1.00 0.00 0.50 B\n")
    cfg_dir = tmp_path / "config_files"
    cfg_dir.mkdir()
    (cfg_dir / "plain.yml").write_text(yaml.safe_dump({
        "spatial_size": [16, 16, 16], "channels": [16, 32, 64], "strides": [2, 2], "max_epochs": 1,
        "mixed_precision": False, "num_samples": 2, "gpu_ids": [0], "tissue_list": str(tissue)}))
    out = tmp_path / "cv"
    cv = {"image_dir": str(img_dir), "labels_dir": str(lab_dir), "tissue_list": str(tissue),
          "output_dir": str(out), "config_files_dir": str(cfg_dir), "num_splits": 2, "gpu_ids": [0]}
    (tmp_path / "cv.yml").write_text(yaml.safe_dump(cv))
    runner = CliRunner()
    res = runner.invoke(app, ["cross-validate", "-c", str(tmp_path / "cv.yml")])
    assert res.exit_code == 0, (res.output, res.exception)
    folds = sorted((out / "datafolds").glob("fold_*.json"))
    assert len(folds) == 2
    ckpts = sorted((out / "plain").glob("*/epoch=*-val_loss=*-val_dice=*.ckpt"))
    assert len(ckpts) >= 2                                      # one per fold at least
    models = tmp_path / "models"
    models.mkdir()
    for c in ckpts[:2]:
        (models / c.name.replace("epoch=0", f"epoch={c.parent.name}")).write_bytes(c.read_bytes())
    cand = tmp_path / "best.yml"
    cand.write_text(yaml.safe_dump({"A": 0, "B": 1}))
    for mode, extra in (("mean", []), ("vote", []), ("select_best", ["-cy", str(cand)])):
        rd = tmp_path / f"ens_{mode}"
        res = runner.invoke(app, ["ensemble-predict", "-d", str(datalist.parent / "predict.json"), "-m", str(models),
                                  "-t", str(tissue), "-r", str(rd), "-cm", mode, "--gpu-ids", "0"] + extra)
        assert res.exit_code == 0, (mode, res.output, res.exception)
        pred, _ = read_nifti(rd / "c3.nii.gz")
        assert pred.shape == (24, 24, 24) and pred.max() <= 2


def test_spatial_dims_2_matches_oracle_and_trains():
    """2-D UNet (reference tests/seg/test_unet.py:15-20 builds one): the 2-D kernels are the 3-D
    ones on a depth-1 volume with the [k,k] weights in the centre plane of [k,k,k]."""
    ref = deterministic_fill_(RefUNet(2, 1, 3, (16, 32, 64), (2, 2)), 0)
    net = Net(num_classes=3, num_channels=1, spatial_dims=2, channels=(16, 32, 64), strides=(2, 2))
    net.load_state_dict({"_model." + k: v.clone() for k, v in ref.state_dict().items()})
    net.to(DEV)
    assert net.spatial_dims == 2
    assert tuple(dict(net._model.named_parameters())["model.0.conv.unit0.conv.weight"].shape) == (16, 1, 3, 3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((4, 1, 32, 48), generator=g)
    lab = torch.randint(0, 3, (4, 1, 32, 48), generator=g).float()
    ref.train(); net.train()
    y_ref = ref(x)
    loss_ref = ref_dice_loss(y_ref, lab)
    loss_ref.backward()
    res = net.training_step({"image": x.to(DEV), "label": lab.to(DEV)})
    torch.cuda.synchronize()
    y = net._engine._bufs["logits.t"][..., :net.num_classes].float().cpu().permute(0, 4, 1, 2, 3).squeeze(2)
    assert float((y - y_ref.detach()).abs().max() / y_ref.detach().abs().max()) < 2e-4
    assert abs(float(res["loss"].cpu()) - float(loss_ref)) < 1e-4 * float(loss_ref)
    # gradients landed in the 2-D views (centre plane of the embedded kernels); off-centre taps stay 0
    sd_ref = dict(ref.named_parameters())
    for key in ("model.0.conv.unit0.conv.weight", "model.2.0.conv.weight", "model.1.submodule.1.submodule.residual.weight"):
        gw = net._engine._gviews[key]
        g2 = net._engine._galias[key]
        gr = sd_ref[key].grad
        assert float((g2.cpu() - gr).abs().max()) < 2e-3 * float(gr.abs().max()) + 1e-7, key
        if gw.dim() == 5 and gw.shape[2] == 3:
            assert float(gw[:, :, 0].abs().max()) == 0.0 and float(gw[:, :, 2].abs().max()) == 0.0
    # eval + 2-D sliding window + checkpoint shapes
    net.eval(); ref.eval()
    with torch.no_grad():
        ye = net(x.to(DEV)).float().cpu()
        assert ye.shape == (4, 3, 32, 48)
        sw = sliding_window_inference(x[:1].to(DEV), (16, 16), 4, net, 0.5, return_labels=True)
        assert sw.logits.shape == (1, 3, 32, 48) and sw.labels.shape == (1, 1, 32, 48)
    assert tuple(net.state_dict()["_model.model.0.conv.unit0.conv.weight"].shape) == (16, 1, 3, 3)


@pytest.mark.parametrize("K", [3, 20])
def test_class_count_not_a_multiple_of_16_runs_padded_on_mfma(K):
    """num_classes = 3 / 20: the K-channel layers are stored with 16 / 32 channels (zeros) so they
    take the MFMA kernels; logits, loss and every gradient still match the oracle, parameters and
    checkpoints keep MONAI's shapes, and the padding stays exactly zero through an optimiser step."""
    ref, net = pair(K, (16, 32, 64), (2, 2))
    img, lab = synthetic_batch(2, 32, K, seed=9)
    ref.train(); net.train()
    y_ref = ref(img)
    loss_ref = ref_dice_loss(y_ref, lab)
    loss_ref.backward()
    res = net.training_step({"image": img.to(DEV), "label": lab.to(DEV)})
    torch.cuda.synchronize()
    eng = net._engine
    assert eng.kpad == (16 if K == 3 else 32)
    full = eng._bufs["logits.t"].float().cpu()
    assert full.shape[-1] == eng.kpad and float(full[..., K:].abs().max()) == 0.0     # padded classes are 0
    y = full[..., :K].permute(0, 4, 1, 2, 3)
    assert float((y - y_ref.detach()).abs().max() / y_ref.detach().abs().max()) < 2e-4
    assert abs(float(res["loss"].detach().cpu()) - float(loss_ref.detach())) < 1e-4 * float(loss_ref.detach())
    sd = net._model.state_dict()
    assert tuple(sd["model.2.1.conv.unit0.conv.weight"].shape) == (K, K, 3, 3, 3)
    assert tuple(sd["model.2.0.conv.weight"].shape)[1] == K and tuple(sd["model.2.0.adn.N.running_var"].shape) == (K,)
    # gradients of the padded layers (compared before the step overwrote nothing: grads persist)
    rp = dict(ref.named_parameters())
    for key in ("model.2.1.conv.unit0.conv.weight", "model.2.1.conv.unit0.conv.bias", "model.2.0.conv.weight",
                "model.2.0.adn.N.weight", "model.0.conv.unit1.conv.weight"):
        g = eng._galias[key].cpu()
        gr = rp[key].grad
        assert g.shape == gr.shape
        assert float((g - gr).abs().max()) < 3e-3 * float(gr.abs().max()) + 2e-6 * float(
            max(t.grad.abs().max() for t in rp.values())), key
    # padding of weights and gradients is exactly zero, also after the Adam step that just ran
    for key in ("model.2.1.conv.unit0.conv.weight", "model.2.0.conv.weight", "model.2.0.conv.bias"):
        wv, gv = eng._pviews[key], eng._gviews[key]
        mask = torch.ones_like(wv, dtype=torch.bool)
        mask[eng._arena_layout(key, rp[key])[1]] = False
        assert float(wv[mask].abs().max()) == 0.0 and float(gv[mask].abs().max()) == 0.0, key
    # eval + sliding window + a reloaded checkpoint give the same labels
    net.eval()
    with torch.no_grad():
        a = sliding_window_inference(img[:1].to(DEV), (16, 16, 16), 4, net, 0.5, return_labels=True)
    assert a.logits.shape == (1, K, 32, 32, 32) and int(a.labels.max()) < K
    net2 = Net(num_classes=K, num_channels=1, channels=(16, 32, 64), strides=(2, 2))
    net2.load_state_dict(net.state_dict())
    net2.to(DEV).eval()
    with torch.no_grad():
        b = sliding_window_inference(img[:1].to(DEV), (16, 16, 16), 4, net2, 0.5, return_labels=True)
    assert torch.equal(a.labels, b.labels) and torch.equal(a.logits, b.logits)


@pytest.mark.parametrize("act", ["RELU", "LEAKYRELU"])
def test_fixed_slope_activations_match_oracle(act):
    """`act` of the config schema: ReLU / LeakyReLU(0.01) run on the PReLU-shaped kernels with a fixed
    slope; the ADN then has no `A.weight` parameter (MONAI key layout)."""
    ref = deterministic_fill_(RefUNet(3, 1, 16, (16, 32, 64), (2, 2), act=act), 0)
    net = Net(num_classes=16, num_channels=1, channels=(16, 32, 64), strides=(2, 2), act=act)
    assert not any(".A." in k for k in net.state_dict())
    net.load_state_dict({"_model." + k: v.clone() for k, v in ref.state_dict().items()})
    net.to(DEV)
    img, lab = synthetic_batch(2, 32, 16, seed=12)
    net.eval(); ref.eval()            # eval first: the training step below changes the weights
    with torch.no_grad():
        ye, yr = net(img.to(DEV)).float().cpu(), ref(img)
    assert float((ye - yr).abs().max() / yr.abs().max()) < 2e-4
    ref.train(); net.train()
    y_ref = ref(img)
    loss_ref = ref_dice_loss(y_ref, lab)
    loss_ref.backward()
    res = net.training_step({"image": img.to(DEV), "label": lab.to(DEV)})
    torch.cuda.synchronize()
    y = net._engine._bufs["logits.t"][..., :16].float().cpu().permute(0, 4, 1, 2, 3)
    assert float((y - y_ref.detach()).abs().max() / y_ref.detach().abs().max()) < 2e-4
    assert abs(float(res["loss"].detach().cpu()) - float(loss_ref.detach())) < 1e-4 * float(loss_ref.detach())
    rp = dict(ref.named_parameters())
    gmax = max(float(t.grad.abs().max()) for t in rp.values())
    for key in ("model.0.conv.unit0.conv.weight", "model.1.submodule.0.conv.unit1.conv.weight",
                "model.2.0.adn.N.weight", "model.2.1.conv.unit0.conv.weight"):
        g, gr = net._engine._galias[key].cpu(), rp[key].grad
        # (kinked activations: pre-activations within rounding of 0 may take the other branch)
        assert float((g - gr).abs().max()) < 1e-2 * float(gr.abs().max()) + 2e-6 * gmax, key


class _HashDropout(torch.nn.Module):
    """Test stand-in for the oracle's nn.Dropout: the mask the HIP kernels derive from (seed, element
    index) (norm_act.hip drop_mult), so the whole step can be compared value for value."""

    def __init__(self, p, seed):
        super().__init__()
        self.p, self.seed = p, seed

    def forward(self, x):
        if not self.training:
            return x
        from test_ops_gpu import _drop_mask
        n, c = x.shape[:2]
        sp = tuple(x.shape[2:])
        m = torch.from_numpy(_drop_mask(x.numel(), self.p, self.seed)).reshape((n,) + sp + (c,))
        return x * m.permute(0, 4, 1, 2, 3)


def test_dropout_training_step_matches_oracle_with_the_same_masks():
    """`dropout` of the config schema (monai_unet.py:83-92 -> UNet(dropout=...)): ADN order norm ->
    dropout -> act; eval ignores it; a training step reproduces the oracle given the same masks."""
    pd = 0.2
    ref = deterministic_fill_(RefUNet(3, 1, 16, (16, 32, 64), (2, 2), dropout=pd), 0)
    net = Net(num_classes=16, num_channels=1, channels=(16, 32, 64), strides=(2, 2), dropout=pd)
    net0 = Net(num_classes=16, num_channels=1, channels=(16, 32, 64), strides=(2, 2))
    sd = {"_model." + k: v.clone() for k, v in ref.state_dict().items()}
    net.load_state_dict(sd); net0.load_state_dict(sd)
    net.to(DEV); net0.to(DEV)
    net.mixed_precision = net0.mixed_precision = False
    img, lab = synthetic_batch(2, 32, 16, seed=21)
    net.eval(); net0.eval()
    with torch.no_grad():
        assert torch.equal(net(img.to(DEV)), net0(img.to(DEV)))        # eval: dropout is the identity
    net.train(); ref.train()
    eng = net._engine_for()
    eng.dropout_seed = 1234
    # masks of the first training forward (_drop_step 0 -> 1)
    eng._drop_step = 0
    seeds = {}
    eng._drop_step += 1
    for bn in eng._bns:
        seeds[bn.prefix] = bn.drop()[1]
    eng._drop_step -= 1
    assert len(set(seeds.values())) == len(seeds)
    for prefix, seed in seeds.items():
        mod = ref.model.get_submodule(prefix)       # engine prefixes are relative to `model.`
        mod.D = _HashDropout(pd, seed)
    y_ref = ref(img)
    loss_ref = ref_dice_loss(y_ref, lab)
    loss_ref.backward()
    res = net.training_step({"image": img.to(DEV), "label": lab.to(DEV)})
    torch.cuda.synchronize()
    y = eng._bufs["logits.t"][..., :16].float().cpu().permute(0, 4, 1, 2, 3).clone()
    assert float((y - y_ref.detach()).abs().max() / y_ref.detach().abs().max()) < 2e-4
    assert abs(float(res["loss"].detach().cpu()) - float(loss_ref.detach())) < 1e-4 * float(loss_ref.detach())
    rp = dict(ref.named_parameters())
    gmax = max(float(t.grad.abs().max()) for t in rp.values())
    for key, gr in ((k, t.grad) for k, t in rp.items() if k.endswith("conv.weight") or ".N.weight" in k):
        g = eng._galias[key].cpu()
        assert float((g - gr).abs().max()) < 5e-3 * float(gr.abs().max()) + 2e-6 * gmax, key
    # the next step draws different masks
    net.training_step({"image": img.to(DEV), "label": lab.to(DEV)})
    torch.cuda.synchronize()
    y2 = eng._bufs["logits.t"][..., :16].float().cpu().permute(0, 4, 1, 2, 3)
    assert float((y2 - y).abs().max()) > 1e-2 * float(y.abs().max())


def test_fused_bn_apply_training_step_is_bit_identical_to_the_separate_pass():
    """segmi_in_affine in the engine: with the BatchNorm-apply folded into the consumer conv / weight
    gradient (top of the decoder, first-level unit) a bf16 training step gives the very same
    logits, loss, gradients and post-Adam weights as with the separate bn_act_fwd pass."""
    ref = deterministic_fill_(RefUNet(3, 1, 16, (16, 32, 64), (2, 2)), 0)
    sd = {"_model." + k: v.clone() for k, v in ref.state_dict().items()}
    img, lab = synthetic_batch(4, 64, 16, seed=31)       # 64^3: the ring kernel takes the 16-channel layers
    results = []
    for fuse in (True, False):
        net = Net(num_classes=16, num_channels=1, channels=(16, 32, 64), strides=(2, 2))
        net.load_state_dict(sd)
        net.to(DEV).train()
        net.mixed_precision = True
        eng = net._engine_for()
        eng.fuse_bn_apply = fuse
        res = net.training_step({"image": img.to(DEV), "label": lab.to(DEV)})
        torch.cuda.synchronize()
        fused_layers = sum(1 for sv in eng._saved.values() for k, v in sv.items()
                           if k.startswith("tf") and v is not None)
        results.append((eng._bufs["logits.t"].clone(), float(res["loss"].cpu()), eng.flat_grad.clone(),
                        eng.flat.clone(), fused_layers))
    (la, lossa, ga, wa, na), (lb, lossb, gb, wb, nb) = results
    assert na >= 1 and nb == 0            # the fused run really took the fused path
    assert torch.equal(la, lb) and lossa == lossb
    assert torch.equal(ga, gb) and torch.equal(wa, wb)


@pytest.mark.parametrize("mixed", [False, True])
def test_merged_pair_convolutions_in_eval_are_bit_identical(mixed):
    """Inference: subunit 0 + residual convolution of the deeper units as one split-activation launch
    (segmi_conv3d_fwd_split_act) gives the very bits of the two separate launches."""
    _, net = pair(16, (16, 32, 64, 128), (2, 2, 2))
    net.eval()
    net.mixed_precision = mixed
    img, _ = synthetic_batch(2, 64, 4, seed=9)
    eng = net._engine_for()
    outs = []
    with torch.no_grad():
        for merge in (True, False):
            eng.merge_eval_pairs = merge
            outs.append(net(img.to(DEV)).clone())
    torch.cuda.synchronize()
    used = sum(1 for k in eng._bufs if k.endswith(".merged"))
    assert used >= 2, sorted(eng._bufs)
    assert torch.equal(outs[0], outs[1])


# ------------------------------------------------------------------------------------------
# BASELINE full-size configurations (C3: 512^3 sliding window; C4: 160^3 / K = 32 training on
# one GPU): size-independent properties, so that the kernels these sizes select (deferred blend
# of 343 windows over two lanes; ``conv_ring2<32,.>`` one workgroup per CU, ``convt_ps<.,2>``)
# are part of the driver's GPU record.
# ------------------------------------------------------------------------------------------
def test_full_size_c3_512_sliding_window_bf16():
    from segmantic_amd.seg.inferers import window_starts
    net = Net(num_classes=16).to(DEV).eval()
    net.mixed_precision = True
    net.spatial_size = [128, 128, 128]
    wins = window_starts((512, 512, 512), (128, 128, 128), 0.5)
    assert len(wins) == 343 and wins[0] == (0, 0, 0) and wins[-1] == (384, 384, 384)
    g = torch.Generator().manual_seed(3)
    # smooth field + noise: neighbouring windows see different content, logits are not constant
    low = torch.randn((1, 1, 16, 16, 16), generator=g)
    vol = F.interpolate(low, size=(512, 512, 512), mode="trilinear", align_corners=False)
    vol = (vol + 0.25 * torch.randn(vol.shape, generator=g)).to(DEV)
    with torch.no_grad():
        d = sliding_window_inference(vol, (128, 128, 128), 4, net, 0.5, "constant",
                                     return_labels=True, blend="deferred", return_logits=False)
        lab_d, cnt_d = d.labels.clone(), d.count.clone()
        del d
        s = sliding_window_inference(vol, (128, 128, 128), 4, net, 0.5, "constant",
                                     return_labels=True, blend="stream")
    torch.cuda.synchronize()
    assert lab_d.shape == (1, 1, 512, 512, 512) and lab_d.dtype == torch.uint8
    assert bool(torch.isfinite(s.logits).all())
    # count map of the dense schedule: 1 at the corners, 8 where two windows overlap per axis
    assert float(cnt_d.min()) == 1.0 and float(cnt_d.max()) == 8.0
    assert torch.equal(cnt_d, s.count)
    assert torch.equal(lab_d, s.labels)                      # deferred == streaming, bit for bit
    assert torch.equal(lab_d[0, 0].long(), torch.argmax(s.logits[0], 0))
    assert int(lab_d.long().unique().numel()) > 1
    # where only window 0 contributes, the volume result is the plain forward of that block.  Not
    # bit for bit: a 16-window group and a single patch pick different kernels for some layers
    # (the z-marching ring needs >= 256 columns), i.e. another f32 summation order before the bf16
    # roundings -- the values agree to bf16 resolution
    with torch.no_grad():
        y = net(vol[:, :, :128, :128, :128].contiguous()).float()
    a, b = s.logits[0, :, :64, :64, :64], y[0, :, :64, :64, :64]
    assert float((a - b).abs().max()) <= 2.0 ** -6 * float(b.abs().max())
    assert float((a - b).abs().mean()) <= 2.0 ** -9 * float(b.abs().mean())


def test_full_size_c4_160_k32_training_step_bf16():
    # (its own seed: "the loss falls within five steps" is a property of a sane initialisation, not of whatever state
    # the tests before it left in the global generator -- the test failed once in six full-suite runs in round 4)
    torch.manual_seed(160)
    net = Net(num_classes=32).to(DEV).train()
    net.mixed_precision = True
    g = torch.Generator().manual_seed(4)
    x = torch.randn((2, 1, 160, 160, 160), generator=g).to(DEV)
    lab = torch.randint(0, 32, (2, 1, 20, 20, 20), generator=g).float()
    lab = F.interpolate(lab, size=(160, 160, 160), mode="nearest").to(DEV)      # 8^3 blobs
    eng = net._engine_for(x)
    eng.timed = {"2.1.conv.unit0.conv:fwd", "2.1.conv.unit0.conv:wgrad", "2.1.conv.unit0.conv:dgrad"}
    l1 = float(net.training_step({"image": x, "label": lab})["loss"].cpu())
    logits = eng._bufs["logits.t"]
    assert tuple(logits.shape) == (2, 160, 160, 160, 32) and logits.dtype == torch.bfloat16
    assert bool(torch.isfinite(logits.float()).all())
    torch.cuda.synchronize()        # the carried weight gradients of the step are still being written otherwise
    assert bool(torch.isfinite(eng.flat_grad).all()) and float(eng.flat_grad.abs().max()) > 0
    # the 32 -> 32 full-resolution layer takes the ring kernel (its live timing exists)
    assert len(eng.timing_ms("2.1.conv.unit0.conv:fwd")) == 1
    top = eng.levels["upru"]["units"][-1][0]
    assert (top.cin, top.cout) == (32, 32)
    from segmantic_amd import ops
    # ... i.e. the z-marching ring kernel <bf16, 32 channels> (this query == conv_ring_ok in conv.hip)
    assert ops.conv3d_in_affine_ok(logits, logits, 3, 1)
    for _ in range(4):
        l2 = float(net.training_step({"image": x, "label": lab})["loss"].cpu())
    assert np.isfinite(l1) and np.isfinite(l2) and l2 < l1
    eng.timed = None


def test_sliding_window_more_than_64_origins_falls_back_to_streaming():
    """``segmi_sw_blend`` takes at most 64 origins per dimension (segmi.h); beyond that the driver
    must take the streaming blend -- for any predictor callable (boundary B4) -- and still equal
    the oracle."""
    img = torch.randn((1, 1, 8, 8, 140), generator=torch.Generator().manual_seed(9))

    def predictor(x):                       # K = 2, position independent
        x = x.float()
        return torch.cat([x, 0.5 - 2.0 * x], 1)

    with torch.no_grad():
        o_ref, _, wins = ref_sliding_window_inference(img, (4, 4, 4), 4, predictor, 0.5, "constant")
        for blend in ("auto", "deferred"):
            res = sliding_window_inference(img.to(DEV), (4, 4, 4), 4, predictor, 0.5, "constant",
                                           return_labels=True, blend=blend)
            torch.cuda.synchronize()
            assert float((res.logits.cpu() - o_ref).abs().max()) < 1e-6
            assert torch.equal(res.labels.cpu().long()[:, 0], torch.argmax(o_ref, 1))
    assert len(wins) == 3 * 3 * 69


def test_dice_loss_refuses_more_than_64_classes_loudly():
    from segmantic_amd.seg.losses import DiceLoss
    lg = torch.randn((1, 65, 4, 4, 4), device=DEV)
    lab = torch.zeros((1, 1, 4, 4, 4), device=DEV)
    with pytest.raises(RuntimeError, match="64 classes"):
        DiceLoss(to_onehot_y=True, softmax=True)(lg, lab)
    ok = DiceLoss(to_onehot_y=True, softmax=True)(lg[:, :64].contiguous(), lab)
    assert bool(torch.isfinite(ok))


def test_bn_backward_sums_fused_into_input_gradient_launches_match_the_separate_pass():
    """SEGMI_FUSE_BN_BWD (segmi_bn_bwd_sums): logits, loss and stored activation gradients are
    bit-identical, parameter gradients agree to f32 summation order."""
    from segmantic_amd.seg.unet import UNetEngine
    img, lab = synthetic_batch(2, 64, 16, seed=11)
    grads = {}
    for flag in (False, True):
        UNetEngine.fuse_bn_bwd = flag
        try:
            _, net = pair(16, (16, 32, 64), (2, 2))
            net.mixed_precision = True
            net.train()
            res = net.training_step({"image": img.to(DEV), "label": lab.to(DEV)})
            torch.cuda.synchronize()
            eng = net._engine
            grads[flag] = (eng.flat_grad.clone(), float(res["loss"].cpu()), eng._bufs["du"].clone())
        finally:
            UNetEngine.fuse_bn_bwd = True
    (ga, la, dua), (gb, lb, dub) = grads[False], grads[True]
    assert la == lb
    # du = BatchNorm-backward apply of the top level: depends on the fused sums through coef only
    assert float((dua.float() - dub.float()).abs().max()) <= 2e-2 * float(dua.float().abs().max())
    assert float((ga - gb).abs().max()) < 2e-3 * float(ga.abs().max())
    # (a coef that differs in its last bits flips bf16 roundings of du, which every later gradient sees)
    assert float((ga - gb).abs().mean()) < 2e-3 * float(ga.abs().mean()) + 1e-9


def test_fused_eval_decoder_top_is_bit_identical_and_used_by_the_sliding_window():
    from segmantic_amd.seg.unet import UNetEngine
    _, net = pair(16, (16, 32, 64), (2, 2))
    net.eval()
    net.mixed_precision = True
    img, _ = synthetic_batch(2, 64, 16, seed=13)
    img = img.to(DEV)
    outs = {}
    default = UNetEngine.fuse_eval_top
    assert default is True                       # round 3: the fused launch wins and is the default
    for flag in (False, True):
        UNetEngine.fuse_eval_top = flag
        try:
            with torch.no_grad():
                outs[flag] = net(img).float().clone()
                sw = sliding_window_inference(img[:1], (32, 32, 32), 4, net, 0.5, return_labels=True)
                outs[(flag, "sw")] = (sw.logits.clone(), sw.labels.clone())
        finally:
            UNetEngine.fuse_eval_top = default
    torch.cuda.synchronize()
    assert torch.equal(outs[False], outs[True])
    assert torch.equal(outs[(False, "sw")][0], outs[(True, "sw")][0])
    assert torch.equal(outs[(False, "sw")][1], outs[(True, "sw")][1])


def test_train_config_with_bundle_dictionaries_adabelief_and_mixed_image_formats(tmp_path):
    """The `preprocessing` / `augmentation` keys of the config schema as MONAI-bundle dictionaries
    (reference monai_unet.py:232-262), AdaBelief + cosine schedule (:305-337), MetaImage / NRRD inputs,
    `predict --spacing`: the run trains, checkpoints, predicts on the source grid."""
    import yaml
    from typer.testing import CliRunner

    from segmantic_amd.commands.monai_unet_cli import app
    from segmantic_amd.data.imageio import read_image, write_image
    datalist = _write_dataset(tmp_path / "data", n=4, size=24)
    root = datalist.parent
    # re-write two volumes as MetaImage / NRRD with an LPS-flipped, anisotropic geometry
    dl = json.loads(datalist.read_text())
    A = np.array([[-1.0, 0, 0, 10.0], [0, -1.0, 0, 20.0], [0, 0, 1.5, -5.0], [0, 0, 0, 1.0]])
    for i, ext in ((0, ".mha"), (1, ".nrrd")):
        for kind in ("image", "label"):
            arr, _ = read_image(root / kind / f"c{i}.nii.gz")
            write_image(root / kind / f"c{i}{ext}", arr, A)
            (root / kind / f"c{i}.nii.gz").unlink()
        dl["training"][i] = {"image": f"image/c{i}{ext}", "label": f"label/c{i}{ext}"}
    datalist.write_text(json.dumps(dl))
    keys = ["@image_key", "@label_key"]
    cfg = {"datalist": str(datalist), "output_dir": str(tmp_path / "results"), "spatial_size": [16, 16, 16],
           "channels": [16, 32, 64], "strides": [2, 2], "max_epochs": 2, "mixed_precision": True, "gpu_ids": [0],
           "optimizer": {"optimizer": "AdaBelief", "lr": 1e-3, "epsilon": 1e-8, "weight_decouple": True},
           "lr_scheduling": {"scheduler": "Cosine", "T_0": 2, "T_multi": 1},
           "preprocessing": {"_target_": "Compose", "transforms": [
               {"_target_": "LoadImaged", "keys": keys, "reader": "ITKReader", "ensure_channel_first": True},
               {"_target_": "Orientationd", "keys": keys, "axcodes": "RAS"},
               {"_target_": "NormalizeIntensityd", "keys": "@image_key", "channel_wise": True},
               {"_target_": "CropForegroundd", "keys": keys, "source_key": "@label_key"},
               {"_target_": "DataStatsd", "keys": "@image_key", "_disabled_": True},
               {"_target_": "EnsureTyped", "keys": keys}]},
           "augmentation": {"_target_": "Compose", "transforms": [
               {"_target_": "SpatialPadd", "keys": keys, "spatial_size": [16, 16, 16]},
               {"_target_": "RandCropByLabelClassesd", "keys": keys, "label_key": "@label_key",
                "spatial_size": [16, 16, 16], "num_classes": 3, "num_samples": 3, "ratios": [0, 1, 1]},
               {"_target_": "RandFlipd", "keys": keys, "prob": 0.5, "spatial_axis": 0}]}}
    runner = CliRunner()
    # a flip configured for ONE axis cannot be expressed by the device sampler (it flips every axis
    # independently with one probability): refused by name instead of silently flipping all three
    (tmp_path / "cfg.yml").write_text(yaml.safe_dump(cfg))
    res = runner.invoke(app, ["train-config", "-c", str(tmp_path / "cfg.yml")])
    assert res.exit_code != 0 and "RandFlipd" in str(res.exception)
    cfg["augmentation"]["transforms"][2:] = [{"_target_": "RandFlipd", "keys": keys, "prob": 0.5, "spatial_axis": a}
                                             for a in range(3)]
    (tmp_path / "cfg.yml").write_text(yaml.safe_dump(cfg))
    res = runner.invoke(app, ["train-config", "-c", str(tmp_path / "cfg.yml")])
    assert res.exit_code == 0, (res.output, res.exception)
    ckpts = sorted((tmp_path / "results").glob("epoch=*-val_loss=*-val_dice=*.ckpt"))
    assert ckpts
    res = runner.invoke(app, ["predict", "-d", str(root / "predict.json"), "-m", str(ckpts[-1]), "-r",
                              str(tmp_path / "pred"), "--spacing", "1.5", "--spacing", "1.5", "--spacing", "1.5",
                              "--gpu-ids", "0"])      # typer list option: repeated flag, as in the reference CLI
    assert res.exit_code == 0, (res.output, res.exception)
    assert "Total Conf. Matrix Metrics:" in res.output and "sensitivity" in res.output
    pred, _ = read_image(tmp_path / "pred" / "c3.nii.gz")
    assert pred.shape == (24, 24, 24) and pred.max() <= 2
    # an unsupported transform in the dictionary is refused by name
    cfg["augmentation"]["transforms"].append({"_target_": "RandGaussianNoised", "keys": "@image_key"})
    (tmp_path / "bad.yml").write_text(yaml.safe_dump(cfg))
    res = runner.invoke(app, ["train-config", "-c", str(tmp_path / "bad.yml")])
    assert res.exit_code != 0 and "RandGaussianNoised" in str(res.exception)


def _hip_trajectory(cfg, mixed_precision):
    """the protocol of tests/golden/make_convergence_golden.py on the HIP path"""
    K, S, B = cfg["labels"], cfg["patch"], cfg["batch"]
    _, net = pair(K, (16, 32, 64, 128, 256), (2, 2, 2, 2))
    net.mixed_precision = mixed_precision
    net.optimizer = dict(Net.optimizer, lr=cfg["lr"])
    net.train()
    val = [synthetic_batch(1, S, K, seed=900 + i) for i in range(cfg["val_volumes"])]
    out = {"train_loss": [], "val_dice": [], "val_loss": []}
    for step in range(cfg["steps"]):
        img, lab = synthetic_batch(B, S, K, seed=100 + step)
        res = net.training_step({"image": img.to(DEV), "label": lab.to(DEV)})
        out["train_loss"].append(float(res["loss"].cpu()))
        if (step + 1) % cfg["validate_every"] == 0:
            for img_v, lab_v in val:                           # reference validation_step, :350-363
                net.validation_step({"image": img_v.to(DEV), "label": lab_v.to(DEV)})
            losses = [float(o["val_loss"].sum().item()) for o in net.validation_step_outputs]
            out["val_loss"].append(sum(losses) / len(losses))
            out["val_dice"].append(float(net.dice_metric.aggregate().item()))
            net.dice_metric.reset()
            net.validation_step_outputs.clear()
            net.train()
    return out


def test_convergence_matches_cpu_reference(golden_dir, record_property):
    """north_star's last gate: "Dice within 1e-4 of the CPU reference on synthetic data" (reference
    path monai_unet.py:339-397).  30 training steps of BASELINE config 1 (32^3, 3 labels, batch 8, fp32,
    Adam) from the same weights on the same batches, validation (sliding window roi 160, argmax,
    DiceMetric) every 10 steps; the CPU oracle's trajectory is committed
    (tests/golden/convergence_c1.json, two runs with different thread counts: the validation Dice is a
    discrete function of near-tied logits, and the oracle's own spread reaches 1.1e-4 at step 30).
    f32 path: every validation Dice within 1e-4 of the nearest oracle run, every training loss within
    1e-4 relative.  bf16 path: deviation recorded, loosely bounded."""
    g = json.loads((golden_dir / "convergence_c1.json").read_text())
    cfg, runs = g["config"], g["runs"]
    hip = _hip_trajectory(cfg, mixed_precision=False)
    dd = [min(abs(h - r["val_dice"][i]) for r in runs) for i, h in enumerate(hip["val_dice"])]
    dl = [min(abs(h - r["val_loss"][i]) for r in runs) for i, h in enumerate(hip["val_loss"])]
    dt = max(min(abs(h - r["train_loss"][i]) / r["train_loss"][i] for r in runs)
             for i, h in enumerate(hip["train_loss"]))
    print(f"\nconvergence f32: |val_dice - oracle| = {dd}, |val_loss - oracle| = {dl}, train-loss rel. {dt:.2e}; "
          f"oracle self-spread {g['oracle_self_spread']['val_dice']}")
    record_property("f32_val_dice_dev", dd)
    assert len(hip["val_dice"]) == len(runs[0]["val_dice"]) == 3
    assert max(dd) <= 1e-4, (hip["val_dice"], [r["val_dice"] for r in runs])
    assert max(dl) <= 1e-4, (hip["val_loss"], [r["val_loss"] for r in runs])
    assert dt <= 1e-4
    assert hip["train_loss"][-1] < hip["train_loss"][0] - 0.02           # it does train
    # bf16 (the benchmarked precision): reported, with a loose sanity bound
    hb = _hip_trajectory(cfg, mixed_precision=True)
    bd = [min(abs(h - r["val_dice"][i]) for r in runs) for i, h in enumerate(hb["val_dice"])]
    bt = max(abs(h - runs[0]["train_loss"][i]) / runs[0]["train_loss"][i] for i, h in enumerate(hb["train_loss"]))
    print(f"convergence bf16: |val_dice - oracle| = {bd}, train-loss rel. {bt:.2e}")
    record_property("bf16_val_dice_dev", bd)
    assert max(bd) < 5e-2 and bt < 2e-2


@pytest.mark.parametrize("size,batch", [(64, 2), (128, 1)])
def test_deferred_and_fused_schedules_leave_bit_identical_gradients(size, batch):
    """the default training path re-orders side-stream launches (weight gradients of the upper
    decoder levels issued late, SEGMI_DEFER_TOP_WGRAD) and holds operand buffers across levels; round 3
    carries the weight gradients of the two full-resolution decoder convolutions (and the optimiser update
    + re-pack of those layers) over the END of the step, beside the next forward.  None of it may change a
    single bit of the gradient arena, of the updated weights, of a checkpoint taken right after a step or
    of an eval forward: immediate issue, every flush depth and carry on / off give the same."""
    K = 16
    img, lab = synthetic_batch(batch, size, K, seed=5)
    batch_d = {"image": img.to(DEV), "label": lab.to(DEV)}

    def run(defer, depth=None, carry=False, steps=3, apply_conv=True, carry_levels=1):
        from segmantic_amd.seg.unet import UNetEngine
        _, net = pair(K, (16, 32, 64, 128, 256), (2, 2, 2, 2))
        net.mixed_precision = True
        net.train()
        keep = UNetEngine.carry_levels
        UNetEngine.carry_levels = carry_levels          # read when the engine lays out its plan
        try:
            eng = net._engine_for(batch_d["image"])
        finally:
            UNetEngine.carry_levels = keep
        assert len(eng._carry_lvls) == carry_levels
        eng.defer_top_wgrad = defer
        eng.carry_top_wgrad = carry
        eng.fuse_apply_conv = apply_conv
        if depth is not None:
            eng._defer_depth_env = str(depth)
        for _ in range(steps):
            net.training_step(batch_d)
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}    # reader of the arena: syncs by itself
        torch.cuda.synchronize()
        # an eval forward right after training sees the final weights of the carried layers
        net.eval()
        with torch.no_grad():
            y = net(batch_d["image"]).float().cpu()
        return eng.flat_grad.clone(), eng.flat.clone(), sd, y

    g0, w0, sd0, y0 = run(False)
    # (defer, flush depth, weight gradients of the full-resolution decoder carried into the next step)
    for defer, depth, carry in ((True, None, False), (True, 0, False), (True, 2, False), (True, None, True),
                                (False, None, True)):
        g, w, sd, y = run(defer, depth, carry)
        assert torch.equal(g0, g), (defer, depth, carry, float((g0 - g).abs().max()))
        assert torch.equal(w0, w), (defer, depth, carry)
        assert all(torch.equal(sd0[k], sd[k]) for k in sd0), (defer, depth, carry)
        assert torch.equal(y0, y), (defer, depth, carry)
    # the up paths of the two / three upper levels carried over the step boundary
    for lv in (2, 3):
        g, w, sd, y = run(False, None, True, carry_levels=lv)
        assert torch.equal(g0, g), ("carry levels", lv, float((g0 - g).abs().max()))
        assert torch.equal(w0, w) and torch.equal(y0, y), ("carry levels", lv)
        assert all(torch.equal(sd0[k], sd[k]) for k in sd0), ("carry levels", lv)
    # the up layers' BatchNorm-backward apply as its own launch instead of inside the stride-2 convolution
    # that consumes it (segmi_bn_act_bwd_apply_conv): same bits
    g, w, sd, y = run(True, None, True, apply_conv=False)
    assert torch.equal(g0, g), ("apply_conv off", float((g0 - g).abs().max()))
    assert torch.equal(w0, w) and torch.equal(y0, y)


def test_parameters_readers_wait_for_the_carried_update():
    """the carried top-level update of a `training_step` runs on the weight-gradient stream AFTER the
    call returns.  A reader of `net.parameters()` / `named_parameters()` on the caller's stream (gradient clipping,
    EMA, logging) must be ordered behind it: a copy taken right after the step, with no synchronisation of the
    caller's own, equals the weights after a full device sync -- also from a stream that is not the training one."""
    K, size = 16, 128
    img, lab = synthetic_batch(1, size, K, seed=7)
    batch_d = {"image": img.to(DEV), "label": lab.to(DEV)}
    _, net = pair(K, (16, 32, 64, 128, 256), (2, 2, 2, 2))
    net.mixed_precision = True
    net.train()
    eng = net._engine_for(batch_d["image"])
    assert eng.carry_top_wgrad and len(eng._carry_lvls) >= 1
    net.training_step(batch_d)
    torch.cuda.synchronize()
    other = torch.cuda.Stream()
    for reader_stream in (torch.cuda.current_stream(), other):
        net.training_step(batch_d)
        with torch.cuda.stream(reader_stream):
            early = torch.cat([p.detach().flatten() for p in net.parameters()]).clone()
            early_named = {k: v.detach().clone() for k, v in net.named_parameters()}
        torch.cuda.synchronize()
        late = torch.cat([p.detach().flatten() for p in net.parameters()])
        assert torch.equal(early, late)
        assert all(torch.equal(v, dict(net.named_parameters())[k].detach()) for k, v in early_named.items())


def test_training_step_raises_after_a_fused_bn_expiry():
    """a NaN never silently enters a run.  With the test hook forcing the bounded wait of the
    one-launch BatchNorm backward to expire, the step that follows raises (`check_fused_timeouts`) instead of
    training on; with the hook off the same network steps normally again."""
    from segmantic_amd import ops
    K, size = 16, 32
    img, lab = synthetic_batch(2, size, K, seed=8)
    batch_d = {"image": img.to(DEV), "label": lab.to(DEV)}
    _, net = pair(K, (16, 32, 64, 128, 256), (2, 2, 2, 2))
    net.mixed_precision = True
    net.train()
    eng = net._engine_for(batch_d["image"])
    if not eng.fuse_bn_bwd_small:
        pytest.skip("one-launch BatchNorm backward switched off (SEGMI_FUSE_BN_BWD_SMALL=0)")
    net.training_step(batch_d)
    torch.cuda.synchronize()
    ops.check_fused_timeouts("healthy step")
    ops.fused_test_hook(poll_limit=64, no_publish=True)
    try:
        net.training_step(batch_d)
        torch.cuda.synchronize()
    finally:
        ops.fused_test_hook()
    assert ops.fused_timeouts() > 0
    with pytest.raises(RuntimeError, match="gave up waiting"):
        net.training_step(batch_d)
    assert ops.fused_timeouts() == 0


@pytest.mark.parametrize("vol,roi,overlap", [((96, 96, 96), 32, 0.5), ((72, 80, 88), 32, 0.25), ((64, 64, 70), 32, 0.5)])
def test_windows_read_in_place_equal_gathered_windows(vol, roi, overlap, monkeypatch):
    """sliding-window driver, own network: the first-layer kernel reads the windows as views of the volume
    (ops.WindowBatch / segmi_windows) instead of a gathered batch.  Same bits; volumes whose rows are not
    4-element aligned (W = 70) silently keep the gather."""
    from segmantic_amd import ops
    _, net = pair(16, (16, 32, 64), (2, 2))
    net.eval()
    net.mixed_precision = True
    g = torch.Generator().manual_seed(21)
    img = torch.randn((1, 1) + vol, generator=g).to(DEV)
    calls = {"views": 0}
    real = ops.WindowBatch.windows

    def counting(self):
        calls["views"] += 1
        return real(self)
    monkeypatch.setattr(ops.WindowBatch, "windows", counting)
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SEGMI_SW_VIEWS", flag)
        with torch.no_grad():
            r = sliding_window_inference(img, (roi,) * 3, 4, net, overlap, return_labels=True)
        torch.cuda.synchronize()
        outs[flag] = (r.logits.clone(), r.labels.clone())
        if flag == "0":
            assert calls["views"] == 0
    assert (calls["views"] > 0) == (vol[2] % 4 == 0)
    assert torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1])


@pytest.mark.parametrize("vol,roi,overlap,mode", [((96, 96, 96), 32, 0.5, "constant"), ((72, 80, 88), 32, 0.25, "gaussian"),
                                                   ((40, 64, 64), 32, 0.5, "constant"), ((32, 48, 48), 32, 0.5, "constant")])
def test_pipelined_slab_blend_is_bit_identical(vol, roi, overlap, mode, monkeypatch):
    """The deferred blend runs slab by slab on a second stream as the z-levels of windows complete (round 3);
    logits, labels and the count map are those of the one-shot blend, bit for bit (same kernel, same ordered
    sums), also when the volume has a single z-level (nothing to pipeline)."""
    _, net = pair(16, (16, 32, 64), (2, 2))
    net.eval()
    net.mixed_precision = True
    g = torch.Generator().manual_seed(33)
    img = torch.randn((2, 1) + vol, generator=g).to(DEV)
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SEGMI_SW_PIPE_BLEND", flag)
        with torch.no_grad():
            r = sliding_window_inference(img, (roi,) * 3, 4, net, overlap, mode=mode, return_labels=True)
            l = sliding_window_inference(img, (roi,) * 3, 4, net, overlap, mode=mode, return_labels=True,
                                         return_logits=False)
        torch.cuda.synchronize()
        outs[flag] = (r.logits.clone(), r.labels.clone(), r.count.clone(), l.labels.clone())
        assert l.logits is None
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)
    assert torch.equal(outs["1"][1], outs["1"][3])


def test_prediction_cache_is_kept_across_calls_and_changes_nothing(monkeypatch):
    """The deferred blend's prediction cache is a workspace of the driver (kept per device, replaced when the
    shape changes, SEGMI_SW_KEEP_CACHE=0: allocated per call): same labels and logits either way, for volumes
    enqueued back to back without a device sync and on a caller-chosen stream, and `release_workspaces()`
    gives the memory back."""
    from segmantic_amd.seg import inferers
    _, net = pair(16, (16, 32, 64), (2, 2))
    net.eval()
    net.mixed_precision = True
    g = torch.Generator().manual_seed(44)
    vols = [torch.randn((1, 1, 96, 96, 96), generator=g).to(DEV) for _ in range(3)]
    small = torch.randn((1, 1, 64, 64, 64), generator=g).to(DEV)

    def run(v, **kw):
        with torch.no_grad():
            return sliding_window_inference(v, (32,) * 3, 4, net, 0.5, return_labels=True, **kw)

    monkeypatch.setenv("SEGMI_SW_KEEP_CACHE", "0")
    inferers.release_workspaces()
    ref = [run(v) for v in vols]
    ref_small = run(small)
    torch.cuda.synchronize()
    assert not inferers._CACHE_WS
    monkeypatch.setenv("SEGMI_SW_KEEP_CACHE", "1")
    got = [run(v) for v in vols]                      # back to back, no sync in between
    ptr = inferers._CACHE_WS[torch.device(DEV).index][0].data_ptr()
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                     # another stream: must wait for the last call's blend
        got.append(run(vols[0]))
    torch.cuda.current_stream().wait_stream(side)
    assert inferers._CACHE_WS[torch.device(DEV).index][0].data_ptr() == ptr      # one buffer, reused
    got_small = run(small)                            # another shape replaces it
    assert inferers._CACHE_WS[torch.device(DEV).index][0].shape[0] != len(ref[0].labels) and len(inferers._CACHE_WS) == 1
    torch.cuda.synchronize()
    for a, b in zip(ref + [ref[0]], got):
        assert torch.equal(a.labels, b.labels) and torch.equal(a.logits, b.logits)
    assert torch.equal(ref_small.labels, got_small.labels) and torch.equal(ref_small.logits, got_small.logits)
    inferers.release_workspaces()
    assert not inferers._CACHE_WS
