#!/usr/bin/env python3
"""Generates tests/golden/reference_host.json by IMPORTING the importable, pure-Python pieces of
the reference (``/root/reference/src``: utils.cli, utils.config, image.labels, seg.dataset,
seg.utils, utils.file_iterators) and recording their outputs on small inputs.  The fixture is
data only (inputs + expected outputs); it pins the host-side drop-in surface (SURVEY.md 8b B1/B2,
8c "importable pieces").  Run in the build container only -- the reference is absent on the GPU box.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_goldens.py
"""
import inspect
import json
import os
import sys
import tempfile
from pathlib import Path

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")

from segmantic.image.labels import load_decathlon_tissuelist, load_tissue_list  # noqa: E402
from segmantic.seg.dataset import PairedDataSet  # noqa: E402
from segmantic.seg.utils import make_device  # noqa: E402
from segmantic.utils import config  # noqa: E402
from segmantic.utils.cli import get_default_args, validate_args  # noqa: E402
from segmantic.utils.file_iterators import find_matching_files  # noqa: E402


def function1(path: Path, arg_int: int, arg_float: float = -1.5):
    pass


def function2(arg_int: int, path: Path = None):
    pass


out = {}
out["default_args"] = {f.__name__: get_default_args(inspect.signature(f)) for f in (function1, function2)}
va = validate_args({"path": "/path/file.txt", "arg_int": 10}, inspect.signature(function1))
out["validate_args"] = {k: (str(v), type(v).__name__) for k, v in va.items()}
try:
    validate_args({"path": "/p", "arg_int": 1, "foo": 42}, inspect.signature(function2))
except Exception as e:
    out["validate_args_error"] = [type(e).__name__, str(e)]
sample = {"datalist": "/d/dataset.json", "channels": [16, 32], "optimizer": {"lr": 1e-4, "amsgrad": False},
          "spatial_size": [], "tissue_list": None, "mixed_precision": True}
out["config_sample"] = sample
out["config_yaml"] = config.dumps(sample, False)
out["config_json"] = config.dumps(sample, True)

tissue_txt = "V7\nN3\nC0.00 0.00 1.00 0.50 Bone\nC0.00 1.00 0.00 0.50 Fat\nC1.00 0.00 0.00 0.50 Skin Tissue\n"
out["tissue_txt"] = tissue_txt
with tempfile.TemporaryDirectory() as td:
    td = Path(td)
    (td / "labels.txt").write_text(tissue_txt)
    out["tissue_list"] = load_tissue_list(td / "labels.txt")
    dl = {"labels": {"1": "liver", "2": "tumor"}, "training": [{"image": "image/*.nii.gz", "label": "label/*.nii.gz"}],
          "validation": [{"image": "image/c.nii.gz", "label": "label/c.nii.gz"}], "test": ["image/x.nii.gz"]}
    (td / "dataset.json").write_text(json.dumps(dl))
    out["decathlon_json"] = dl
    out["decathlon_tissuelist"] = load_decathlon_tissuelist(td / "dataset.json")
    names = ["a", "b", "c", "d", "e"]
    (td / "image").mkdir()
    (td / "label").mkdir()
    for n in names:
        (td / "image" / f"{n}.nii.gz").touch()
        (td / "label" / f"{n}.nii.gz").touch()
    (td / "label" / "zzz.nii.gz").touch()
    ds = PairedDataSet(image_dir=td / "image", labels_dir=td / "label", valid_split=0.2, shuffle=True, random_seed=42)
    out["split_seed42"] = {"train": sorted(p["image"].name for p in ds.training_files()),
                           "val": sorted(p["image"].name for p in ds.validation_files()),
                           "train_order": [p["image"].name for p in ds.training_files()],
                           "val_order": [p["image"].name for p in ds.validation_files()]}
    ds3 = PairedDataSet(image_dir=td / "image", labels_dir=td / "label", valid_split=0.2, shuffle=False, max_files=3)
    out["split_3files"] = [len(ds3.training_files()), len(ds3.validation_files())]
    m = find_matching_files([td / "image" / "*.nii.gz", td / "label" / "*.nii.gz"], verbose=False)
    out["matching"] = sorted([[p.name for p in t] for t in m])
    (td / "label" / "zzz.nii.gz").unlink()      # load_from_json asserts equal glob counts
    dj = PairedDataSet.load_from_json(td / "dataset.json")
    out["load_from_json"] = {"train": [p["image"].name for p in dj.training_files()],
                             "val": [p["image"].name for p in dj.validation_files()],
                             "test": [str(p["image"]) for p in dj.test_files()]}
    dumped = json.loads(dj.dump_dataset())
    out["dump_keys"] = sorted(dumped.keys())
    out["dump_training_first"] = {k: Path(v).name for k, v in dumped["training"][0].items()}
out["make_device"] = {"[]": str(make_device([])), "[-1]": str(make_device([-1])), "[1]": str(make_device([1]))}
out["cuda_available_when_generated"] = False

dst = Path(__file__).resolve().parent / "reference_host.json"
dst.write_text(json.dumps(out, indent=1))      # key order of config_sample matters: not sorted
print("wrote", dst)
