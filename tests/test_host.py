"""CPU tests of the host-side drop-in surface: C-ABI export check, config / CLI schema, labels,
datasets, sliding-window scheduling, LR schedules, NIfTI IO.  Expected values in
tests/golden/reference_host.json were captured from the reference's importable modules
(tests/golden/make_reference_goldens.py)."""
import inspect
import json
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch
import yaml

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def ref(golden_dir):
    return json.loads((golden_dir / "reference_host.json").read_text())


# ------------------------------------------------------------------ C ABI
def _header_functions():
    txt = (ROOT / "include" / "segmi.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(segmi_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from segmantic_amd import _lib
    declared = _header_functions()
    assert len(declared) >= 35
    assert sorted(_lib.SIGNATURES) == declared            # binding mirrors the header one to one
    nm = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\sT\s+(segmi_\w+)", nm))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert _lib.lib.segmi_version() == 1
    # no torch / C++ types leak through the boundary: only C symbols with the segmi_ prefix are public API
    assert _lib.lib.segmi_wpack_bytes(1, 0, 16, 16, 3) == 14 * 1 * 64 * 8 * 2
    assert _lib.lib.segmi_wpack_bytes(0, 0, 16, 16, 3) == 27 * 1 * 64 * 4 * 4
    assert _lib.lib.segmi_wpack_bytes(1, 0, 3, 16, 3) == 0          # not an MFMA shape


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib

    import segmantic_amd._lib as L
    monkeypatch.setenv("SEGMI_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        importlib.reload(L)
    monkeypatch.delenv("SEGMI_LIB")
    importlib.reload(L)


def test_ops_refuse_cpu_tensors():
    from segmantic_amd import ops
    with pytest.raises(RuntimeError, match="MI355X"):
        ops.act(torch.zeros(1, 2, 2, 2, 4))


# ------------------------------------------------------------------ config / CLI schema
def function1(path: Path, arg_int: int, arg_float: float = -1.5):
    pass


def function2(arg_int: int, path: Path = None):
    pass


def test_cli_helpers_match_reference(ref):
    from segmantic_amd.utils.cli import get_default_args, is_path, validate_args
    for f in (function1, function2):
        assert get_default_args(inspect.signature(f)) == ref["default_args"][f.__name__]
    va = validate_args({"path": "/path/file.txt", "arg_int": 10}, inspect.signature(function1))
    assert {k: (str(v), type(v).__name__) for k, v in va.items()} == {k: tuple(v) for k, v in ref["validate_args"].items()}
    with pytest.raises(ValueError) as e:
        validate_args({"path": "/p", "arg_int": 1, "foo": 42}, inspect.signature(function2))
    assert [type(e.value).__name__, str(e.value)] == ref["validate_args_error"]
    # reference tests/utils/test_cli.py:19-30
    for f in (function1, function2):
        sig = inspect.signature(f)
        valid = validate_args(get_default_args(sig), sig)
        for k in sig.parameters:
            if valid[k] is not None:
                assert is_path(sig.parameters[k]) == isinstance(valid[k], Path)


def test_config_dumps_match_reference(ref, tmp_path):
    from segmantic_amd.utils import config
    assert config.dumps(ref["config_sample"], False) == ref["config_yaml"]
    assert config.dumps(ref["config_sample"], True) == ref["config_json"]
    for name in ("c.yml", "c.json"):
        config.dump(ref["config_sample"], tmp_path / name)
        assert config.load(tmp_path / name) == ref["config_sample"]


def test_train_signature_is_the_reference_schema():
    """keys and defaults of monai_unet.train (reference :400-428)"""
    from segmantic_amd.seg.monai_unet import train
    from segmantic_amd.utils.cli import get_default_args
    d = get_default_args(inspect.signature(train))
    assert list(d) == ["datalist", "image_dir", "labels_dir", "output_dir", "checkpoint_file", "num_classes",
                       "num_channels", "spatial_dims", "spatial_size", "preprocessing", "augmentation",
                       "augment_intensity", "augment_spatial", "channels", "strides", "dropout", "act",
                       "num_samples", "optimizer", "lr_scheduling", "max_epochs", "early_stop_patience",
                       "mixed_precision", "cache_rate", "gpu_ids", "tissue_list"]
    assert d["datalist"] == "<required option: Path>" and d["output_dir"] == "<required option: Path>"
    assert d["channels"] == [16, 32, 64, 128, 256] and d["strides"] == [2, 2, 2, 2]
    assert d["num_classes"] == 0 and d["num_channels"] == 1 and d["spatial_dims"] == 3 and d["spatial_size"] == []
    assert d["max_epochs"] == 600 and d["early_stop_patience"] == 50 and d["mixed_precision"] is True
    assert d["gpu_ids"] == [0] and d["num_samples"] == 4 and d["act"] == "PRELU" and d["dropout"] == 0.0
    assert all(p.kind == inspect.Parameter.KEYWORD_ONLY for p in inspect.signature(train).parameters.values())


def test_cli_print_defaults_and_unknown_key(tmp_path):
    """reference tests/seg/test_unet.py:23-27 + unknown-key rejection (utils/cli.py:34-44)"""
    from typer.testing import CliRunner

    from segmantic_amd.commands.monai_unet_cli import app
    runner = CliRunner()
    for name in ("foo.json", "foo.yml"):
        out = tmp_path / name
        res = runner.invoke(app, ["train-config", "-c", str(out), "--print-defaults"])
        assert res.exit_code == 0 and out.exists()
        cfg = json.loads(out.read_text()) if name.endswith("json") else yaml.safe_load(out.read_text())
        assert cfg["max_epochs"] == 600 and cfg["channels"] == [16, 32, 64, 128, 256]
    bad = tmp_path / "bad.yml"
    bad.write_text("datalist: x.json\noutput_dir: out\nnot_an_option: 1\n")
    res = runner.invoke(app, ["train-config", "-c", str(bad)])
    assert res.exit_code != 0 and isinstance(res.exception, ValueError)
    assert "Unexpected argument not_an_option" in str(res.exception)
    res = runner.invoke(app, ["train-config"])
    assert res.exit_code != 0 and "Invalid '--config-file' argument" in str(res.exception)


def test_train_argument_validation(tmp_path):
    from segmantic_amd.seg.monai_unet import train
    (tmp_path / "d.json").write_text(json.dumps({"labels": {"1": "a", "3": "b"}, "training": [], "validation": []}))
    with pytest.raises(ValueError, match="redundant"):
        train(datalist=tmp_path / "d.json", output_dir=tmp_path, num_classes=3, tissue_list=tmp_path / "t.txt")
    with pytest.raises(ValueError, match="contiguous"):
        train(datalist=tmp_path / "d.json", output_dir=tmp_path)
    (tmp_path / "e.json").write_text(json.dumps({"labels": {}, "training": [], "validation": []}))
    with pytest.raises(ValueError, match="expected to be > 1"):
        train(datalist=tmp_path / "e.json", output_dir=tmp_path)


# ------------------------------------------------------------------ labels / datasets
def test_tissue_lists(ref, tmp_path):
    from segmantic_amd.image.labels import load_decathlon_tissuelist, load_tissue_list, save_tissue_list
    (tmp_path / "labels.txt").write_text(ref["tissue_txt"])
    assert load_tissue_list(tmp_path / "labels.txt") == ref["tissue_list"]
    (tmp_path / "dataset.json").write_text(json.dumps(ref["decathlon_json"]))
    assert load_decathlon_tissuelist(tmp_path / "dataset.json") == ref["decathlon_tissuelist"]
    save_tissue_list(ref["tissue_list"], tmp_path / "rt.txt")
    assert load_tissue_list(tmp_path / "rt.txt") == ref["tissue_list"]
    (tmp_path / "dup.txt").write_text("C0 0 0 0.5 A\nC0 0 0 0.5 A\n")
    with pytest.raises(KeyError):
        load_tissue_list(tmp_path / "dup.txt")


def _mock(tmp_path):
    for d in ("image", "label"):
        (tmp_path / d).mkdir()
        for n in "abcde":
            (tmp_path / d / f"{n}.nii.gz").touch()


def test_paired_dataset(ref, tmp_path):
    from segmantic_amd.seg.dataset import PairedDataSet
    from segmantic_amd.utils.file_iterators import find_matching_files
    _mock(tmp_path)
    (tmp_path / "label" / "zzz.nii.gz").touch()
    ds = PairedDataSet(image_dir=tmp_path / "image", labels_dir=tmp_path / "label", valid_split=0.2, shuffle=True, random_seed=42)
    assert len(ds.training_files()) == len(ref["split_seed42"]["train"]) == 4
    assert len(ds.validation_files()) == len(ref["split_seed42"]["val"]) == 1
    ds3 = PairedDataSet(image_dir=tmp_path / "image", labels_dir=tmp_path / "label", valid_split=0.2, shuffle=False, max_files=3)
    assert [len(ds3.training_files()), len(ds3.validation_files())] == ref["split_3files"] == [2, 1]
    m = find_matching_files([tmp_path / "image" / "*.nii.gz", tmp_path / "label" / "*.nii.gz"], verbose=False)
    assert sorted([[p.name for p in t] for t in m]) == ref["matching"]
    (tmp_path / "label" / "zzz.nii.gz").unlink()
    (tmp_path / "dataset.json").write_text(json.dumps(ref["decathlon_json"]))
    dj = PairedDataSet.load_from_json(tmp_path / "dataset.json")
    assert [p["image"].name for p in dj.training_files()] == ref["load_from_json"]["train"]
    assert [p["image"].name for p in dj.validation_files()] == ref["load_from_json"]["val"]
    assert [str(p["image"]) for p in dj.test_files()] == ref["load_from_json"]["test"]
    dumped = json.loads(dj.dump_dataset())
    assert sorted(dumped) == ref["dump_keys"]
    assert {k: Path(v).name for k, v in dumped["training"][0].items()} == ref["dump_training_first"]
    folds = PairedDataSet.kfold_crossval(7, [{"image": Path(f"i{k}"), "label": Path(f"l{k}")} for k in range(9)],
                                         tmp_path / "folds", shuffle=False)
    assert len(folds) == 7 and all(f.exists() for f in folds)
    sizes = [len(json.loads(f.read_text())["validation"]) for f in folds]
    assert sizes == [2, 2, 1, 1, 1, 1, 1]


def test_make_device(ref):
    from segmantic_amd.seg.utils import make_device
    if not torch.cuda.is_available():
        assert str(make_device([])) == ref["make_device"]["[]"]
    assert str(make_device([-1])) == ref["make_device"]["[-1]"]
    assert str(make_device([1])) == ref["make_device"]["[1]"]


# ------------------------------------------------------------------ Net surface (no compute)
def test_net_hyperparameters_as_reference_test():
    """reference tests/seg/test_unet.py:15-20"""
    from segmantic_amd.seg.monai_unet import Net
    net = Net(num_classes=3, num_channels=4, spatial_dims=2, spatial_size=[64] * 2)
    assert net.hparams.num_classes == 3 and net.hparams.num_channels == 4
    assert net.hparams.spatial_dims == 2 and net.hparams.spatial_size == [64] * 2
    assert net.num_classes == 3 and net.spatial_dims == 2
    assert Net(num_classes=2).spatial_size == [96, 96, 96]


def test_net_state_dict_is_key_compatible_with_the_oracle_layout():
    from oracle.unet_ref import RefUNet
    from segmantic_amd.seg.monai_unet import Net
    for k, ch, st in ((2, (16, 32, 64, 128, 256), (2, 2, 2, 2)), (3, (4, 8, 16), (2, 2))):
        a = Net(num_classes=k, channels=ch, strides=st).state_dict()
        b = RefUNet(3, 1, k, ch, st).state_dict()
        assert list(a) == ["_model." + x for x in b]
        assert all(a["_model." + x].shape == v.shape and a["_model." + x].dtype == v.dtype for x, v in b.items())


# ------------------------------------------------------------------ sliding-window schedule, LR schedules
def test_window_schedule_matches_oracle():
    from oracle.sliding_ref import window_starts as ref_ws
    from segmantic_amd.seg.inferers import window_starts
    for img, roi, ov in (((512,) * 3, (128,) * 3, 0.5), ((40, 37, 51), (16, 16, 16), 0.25), ((160, 200, 96), (96,) * 3, 0.5),
                         ((16, 16, 16), (16, 16, 16), 0.25), ((20, 16, 33), (16, 16, 16), 0.9)):
        assert window_starts(img, roi, ov) == ref_ws(img, roi, ov)[1]


def test_lr_schedulers_match_torch():
    from segmantic_amd.seg.optim import CosineAnnealingWarmRestarts, FlatOptimizer, ReduceLROnPlateau
    p = torch.nn.Parameter(torch.zeros(1))
    for T0, Tm in ((5, 1), (3, 2)):
        topt = torch.optim.SGD([p], lr=0.1)
        ts = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(topt, T_0=T0, T_mult=Tm, eta_min=0)
        fo = FlatOptimizer(torch.zeros(1), torch.zeros(1), 0.1)
        fs = CosineAnnealingWarmRestarts(fo, T0, Tm)
        for _ in range(20):
            topt.step(); ts.step(); fs.step()
            assert abs(topt.param_groups[0]["lr"] - fo.lr) < 1e-12
    topt = torch.optim.SGD([p], lr=0.1)
    ts = torch.optim.lr_scheduler.ReduceLROnPlateau(topt, mode="min", factor=0.5, patience=2)
    fo = FlatOptimizer(torch.zeros(1), torch.zeros(1), 0.1)
    fs = ReduceLROnPlateau(fo, factor=0.5, patience=2)
    for m in [1.0, 0.9, 0.95, 0.94, 0.93, 0.92, 0.5, 0.6, 0.6, 0.6, 0.6, 0.6]:
        topt.step(); ts.step(m); fs.step(m)
        assert abs(topt.param_groups[0]["lr"] - fo.lr) < 1e-12


def test_nifti_roundtrip_and_orientation(tmp_path):
    from segmantic_amd.data.nifti import read_nifti, write_nifti
    from segmantic_amd.seg.pipeline import from_ras, to_ras
    a = np.arange(3 * 4 * 5, dtype=np.int16).reshape(3, 4, 5)
    A = np.array([[0, -0.6, 0, 10], [0.5, 0, 0, 20], [0, 0, -0.7, 30], [0, 0, 0, 1.0]])   # PLI-ish
    for ext in (".nii", ".nii.gz"):
        write_nifti(tmp_path / ("t" + ext), a, A)
        b, B = read_nifti(tmp_path / ("t" + ext))
        assert np.array_equal(a, b) and np.allclose(A, B)
    vol = torch.from_numpy(a.transpose(2, 1, 0).copy())[None].float()          # [1, x, y, z]
    ras, A2, rec = to_ras(vol, A)
    R = A2[:3, :3]
    assert np.all(np.diag(R) > 0) and np.allclose(R - np.diag(np.diag(R)), 0)  # axis aligned, positive
    # the physical position of every voxel is preserved
    i = np.array([1, 2, 0, 1.0])
    v = vol[0, 1, 2, 0]
    j = np.linalg.solve(A2, A @ i)
    assert ras[0][tuple(int(round(x)) for x in j[:3])] == v
    assert torch.equal(from_ras(ras, rec), vol)


def test_augmentation_draws_follow_the_reference_distributions():
    """monai_unet.py:181-212: four spatial transforms at prob 0.2 each, invertible composed map,
    monotone histogram control points, 20 bias coefficients in [0, 0.1)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "segmi_augment", Path(__file__).resolve().parents[1] / "segmantic_amd" / "seg" / "augment.py")
    aug = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(aug)
    rng = np.random.RandomState(0)
    fired = 0
    for _ in range(2000):
        m = aug.draw_spatial(rng, (40, 50, 60))
        if m is None:
            continue
        fired += 1
        p = aug.forward_point(m, (10, 20, 30))
        assert np.allclose((m @ np.array([*p, 1.0]))[:3], (10, 20, 30))
        ctr = np.array([19.5, 24.5, 29.5, 1.0])
        assert np.allclose(m @ ctr, ctr)                       # rotations and zoom keep the centre
        s = np.linalg.det(m[:3, :3]) ** (1 / 3)
        assert 1 / 1.3 - 1e-9 <= s <= 1 / 0.8 + 1e-9            # pull-back scale = 1 / zoom
    assert abs(fired / 2000 - (1 - 0.8 ** 4)) < 0.04
    xyz = aug.to_index_map_xyz(np.arange(16, dtype=float).reshape(4, 4))
    assert xyz.shape == (3, 4) and xyz[0, 0] == 10 and xyz[2, 3] == 3 and xyz[0, 3] == 11
    con, hist, bias, gibbs, spike = aug.draw_intensity(rng, 64, (16, 24, 32))
    assert gibbs[1].min() >= 0 and gibbs[1].max() <= 1 and spike[1].shape == (64, 3)
    assert spike[1][:, 0].max() < 16 and spike[1][:, 1].max() < 24 and spike[1][:, 2].max() < 32
    assert hist[1].shape == (64, 10) and np.all(np.diff(hist[1], axis=1) >= 0)
    assert np.all(hist[1][:, 0] == 0) and np.all(hist[1][:, -1] == 1)
    assert bias[1].shape == (64, 20) and bias[1].min() >= 0 and bias[1].max() < 0.1
    assert con[1].min() >= 0.5 and con[1].max() <= 4.5


def test_hot_kernels_do_not_spill_registers(tmp_path):
    """Reads the code-object metadata of the built library: no kernel may spill VGPRs to scratch
    (a spilling MFMA kernel runs several times slower and nothing else would notice)."""
    import shutil
    llvm = "/opt/rocm/lib/llvm/bin"
    so = ROOT / "segmantic_amd" / "csrc" / "libsegmi.so"
    if not (so.exists() and Path(f"{llvm}/llvm-objdump").exists()):
        pytest.skip("library or llvm tools not present")
    shutil.copy(so, tmp_path / "lib.so")
    subprocess.run([f"{llvm}/llvm-objdump", "--offloading", "lib.so"], cwd=tmp_path, check=True,
                   capture_output=True)
    spilled, seen = [], 0
    for f in sorted(tmp_path.glob("lib.so.*gfx950")):
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", str(f)], capture_output=True,
                               text=True).stdout
        for name, cnt in re.findall(r"\.name:\s+(\S+).*?\.vgpr_spill_count:\s+(\d+)", notes, re.S):
            seen += 1
            if int(cnt):
                spilled.append((name, int(cnt)))
    assert seen > 100, seen
    assert not spilled, spilled


def test_load_from_checkpoint_accepts_a_lightning_file_with_extra_entries(tmp_path):
    """A checkpoint written by the reference's Lightning ``Net`` carries more than the network:
    loss / metric buffers in ``state_dict`` (e.g. ``loss_function.class_weight``), optimizer and
    callback state, extra hyper-parameters.  Only ``_model.*`` and the constructor kwargs matter
    (reference ``monai_unet.py:564-574``)."""
    import torch
    from segmantic_amd.seg.monai_unet import Net
    src = Net(num_classes=3, channels=(4, 8), strides=(2,))
    sd = {k: v.detach().clone() for k, v in src.state_dict().items()}
    assert all(k.startswith("_model.") for k in sd)
    sd["loss_function.class_weight"] = torch.ones(3)
    sd["dice_metric._buffers"] = torch.zeros(1)
    ckpt = {"state_dict": sd, "epoch": 12, "global_step": 345,
            "pytorch-lightning_version": "2.1.0",
            "hyper_parameters": {"num_classes": 3, "num_channels": 1, "spatial_dims": 3,
                                 "spatial_size": [32, 32, 32], "channels": (4, 8), "strides": (2,),
                                 "dropout": 0.0, "act": "PRELU", "some_future_flag": True},
            "optimizer_states": [{"state": {}, "param_groups": []}],
            "lr_schedulers": [{}], "callbacks": {"EarlyStopping": {"wait_count": 3}}}
    path = tmp_path / "epoch=12-val_loss=0.31-val_dice=0.8123.ckpt"
    torch.save(ckpt, str(path))
    net = Net.load_from_checkpoint(path)
    assert net.num_classes == 3 and net.spatial_size == [32, 32, 32]
    for k, v in src.state_dict().items():
        assert torch.equal(net.state_dict()[k], v), k


def test_image_io_nifti_metaimage_nrrd_round_trip_and_lps_to_ras(tmp_path):
    """LoadImaged(reader="ITKReader") reads any ITK format and hands MONAI a RAS affine (reference
    monai_unet.py:157-162): MetaImage and NRRD store LPS geometry, NIfTI stores RAS."""
    import numpy as np
    from segmantic_amd.data.imageio import read_image, write_image
    g = np.random.RandomState(0)
    arr = (g.rand(5, 6, 7) * 100).astype(np.float32)
    A = np.array([[0.0, -0.8, 0.0, 12.5], [1.1, 0.0, 0.0, -3.0], [0.0, 0.0, 2.5, 40.0], [0, 0, 0, 1.0]])
    for name in ("v.nii.gz", "v.mha", "v.nrrd"):
        write_image(tmp_path / name, arr, A)
        back, A2 = read_image(tmp_path / name)
        assert back.dtype == np.float32 and np.array_equal(back, arr), name
        assert np.allclose(A2, A, atol=1e-5), name
    lab = (g.rand(5, 6, 7) * 4).astype(np.uint8)
    write_image(tmp_path / "l.nrrd", lab, A)
    assert np.array_equal(read_image(tmp_path / "l.nrrd")[0], lab)
    # a hand-written MetaImage header as ITK writes it (LPS: identity direction, offset (1, 2, 3), spacing .5 .6 .7)
    data = np.arange(2 * 3 * 4, dtype=np.int16).reshape(4, 3, 2)
    hdr = ("ObjectType = Image\nNDims = 3\nBinaryData = True\nBinaryDataByteOrderMSB = False\nCompressedData = False\n"
           "TransformMatrix = 1 0 0 0 1 0 0 0 1\nOffset = 1 2 3\nCenterOfRotation = 0 0 0\nAnatomicalOrientation = RAI\n"
           "ElementSpacing = 0.5 0.6 0.7\nDimSize = 2 3 4\nElementType = MET_SHORT\nElementDataFile = LOCAL\n")
    (tmp_path / "itk.mha").write_bytes(hdr.encode() + data.tobytes())
    arr2, A3 = read_image(tmp_path / "itk.mha")
    assert np.array_equal(arr2, data)
    assert np.allclose(A3, np.array([[-0.5, 0, 0, -1.0], [0, -0.6, 0, -2.0], [0, 0, 0.7, 3.0], [0, 0, 0, 1.0]]))
    # external data file + zlib compression (.mhd)
    import zlib
    (tmp_path / "e.zraw").write_bytes(zlib.compress(data.tobytes()))
    (tmp_path / "e.mhd").write_text(hdr.replace("CompressedData = False", "CompressedData = True")
                                    .replace("ElementDataFile = LOCAL", "ElementDataFile = e.zraw"))
    assert np.array_equal(read_image(tmp_path / "e.mhd")[0], data)
    # NRRD header in RAS space, raw encoding
    nh = ("NRRD0004\ntype: short\ndimension: 3\nspace: right-anterior-superior\nsizes: 2 3 4\n"
          "space directions: (0.5,0,0) (0,0.6,0) (0,0,0.7)\nendian: little\nencoding: raw\nspace origin: (1,2,3)\n\n")
    (tmp_path / "r.nrrd").write_bytes(nh.encode() + data.tobytes())
    arr3, A4 = read_image(tmp_path / "r.nrrd")
    assert np.array_equal(arr3, data) and np.allclose(A4, np.array([[0.5, 0, 0, 1], [0, 0.6, 0, 2], [0, 0, 0.7, 3], [0, 0, 0, 1.0]]))
    # the same header with CRLF line ends; a detached header is refused by name
    (tmp_path / "c.nrrd").write_bytes(nh.replace("\n", "\r\n").encode() + data.tobytes())
    arr4, A5 = read_image(tmp_path / "c.nrrd")
    assert np.array_equal(arr4, data) and np.allclose(A5, A4)
    import pytest
    (tmp_path / "d.nrrd").write_bytes(nh.replace("encoding: raw", "encoding: raw\ndata file: d.raw").encode())
    with pytest.raises(ValueError, match="detached"):
        read_image(tmp_path / "d.nrrd")
    with pytest.raises(ValueError, match="unsupported image format"):
        read_image(tmp_path / "x.png")
