"""MONAI-bundle ``preprocessing`` / ``augmentation`` dictionaries (reference
``src/segmantic/seg/monai_unet.py:232-262``): the resolver semantics pinned by the reference's own
tests ``tests/seg/test_unet.py:30-96``, on the reference's own fixture
(``tests/testing_data/config.json`` -> ``tests/golden/reference_bundle_config.json``)."""
import json

import pytest

from segmantic_amd.utils.bundle import (Compose, ConfigParser, TransformSpec, plan_augmentation,
                                        plan_preprocessing)


@pytest.fixture()
def options(golden_dir):
    return json.loads((golden_dir / "reference_bundle_config.json").read_text())


def test_load_preprocessing(options):          # reference test_load_preprocessing
    parser = ConfigParser({"image_key": "image", "preprocessing": options["preprocessing"]})
    parser.parse(True)
    transforms = parser.get_parsed_content("preprocessing")
    assert isinstance(transforms, Compose) and len(transforms) == 5
    assert [t.target for t in transforms] == ["LoadImaged", "EnsureChannelFirstd", "NormalizeIntensityd",
                                              "RandRotated", "EnsureTyped"]
    assert all(t.kwargs["keys"] == "image" for t in transforms)        # "@image_key" resolved
    assert transforms.transforms[2].kwargs["nonzero"] is True


def test_load_empty_preprocessing():            # reference test_load_empty_preprocessing
    parser = ConfigParser({"image_key": "image", "preprocessing": {}})
    parser.parse(True)
    transforms = parser.get_parsed_content("preprocessing")
    assert isinstance(transforms, dict) and len(transforms) == 0


def test_load_not_existing_preprocessing():     # reference test_load_not_existing_preprocessing
    parser = ConfigParser({"image_key": "image"})
    parser.parse(True)
    with pytest.raises(KeyError):
        parser.get_parsed_content("preprocessing")
    assert "preprocessing" not in parser


def test_load_disabled_preprocessing():         # reference test_load_disabled_preprocessing
    parser = ConfigParser({"image_key": "image",
                           "preprocessing": {"_target_": "DataStatsD", "_disabled_": True, "keys": "image"}})
    parser.parse(True)
    assert parser.get_parsed_content("preprocessing") is None


def test_load_postprocessing_reference_to_preprocessing(options):    # reference test_load_postprocessing
    parser = ConfigParser({k: options[k] for k in ("image_key", "preprocessing", "postprocessing")})
    parser.parse(True)
    post = parser.get_parsed_content("postprocessing")
    assert isinstance(post, Compose) and [t.name for t in post] == ["EnsureType", "Invert", "AsDiscrete"]
    inv = post.transforms[1]
    assert isinstance(inv.kwargs["transform"], Compose)              # "@preprocessing" -> the resolved Compose
    assert inv.kwargs["transform"] is parser.get_parsed_content("preprocessing")
    assert inv.kwargs["nearest_interp"] is False and post.transforms[2].kwargs["argmax"] is True


def test_trainer_entry_resolves_to_a_spec(options):                  # reference test_load_trainer (no Lightning here)
    parser = ConfigParser({"trainer": options["trainer"]})
    parser.parse(True)
    tr = parser.get_parsed_content("trainer")
    assert isinstance(tr, TransformSpec) and tr.target == "pytorch_lightning.Trainer"
    assert tr.kwargs["max_epochs"] == options["trainer"]["max_epochs"]


def test_expressions_and_cycles_are_refused():
    p = ConfigParser({"a": "$__import__('os').system('true')", "b": "@c", "c": "@b", "d": {"x": "@e#1"}, "e": [1, 2]})
    with pytest.raises(NotImplementedError):
        p.get_parsed_content("a")
    with pytest.raises(ValueError, match="circular"):
        p.get_parsed_content("b")
    assert p.get_parsed_content("d") == {"x": 2}


def _comp(transforms):
    return ConfigParser({"c": {"_target_": "Compose", "transforms": transforms}}).get_parsed_content("c")


DEFAULT_PRE = [
    {"_target_": "LoadImaged", "keys": ["image", "label"], "reader": "ITKReader", "ensure_channel_first": True},
    {"_target_": "Orientationd", "keys": ["image", "label"], "axcodes": "RAS"},
    {"_target_": "NormalizeIntensityd", "keys": "image", "nonzero": False, "channel_wise": True},
    {"_target_": "CropForegroundd", "keys": ["image", "label"], "source_key": "label"},
    {"_target_": "EnsureTyped", "keys": ["image", "label"]},
    {"_target_": "Spacingd", "keys": ["image", "label"], "pixdim": [1.0, 1.0, 2.5]}]
KEYS = ["image", "label"]
CROP = {"_target_": "RandCropByLabelClassesd", "keys": KEYS, "label_key": "label", "spatial_size": [96, 96, 96],
        "num_classes": 3, "num_samples": 2, "ratios": [0, 1, 1]}
FLIPS = [{"_target_": "RandFlipd", "keys": KEYS, "prob": 0.2, "spatial_axis": a} for a in range(3)]
SPATIAL = [{"_target_": "RandRotated", "keys": KEYS, "prob": 0.2, "range_z": 0.4, "mode": ["bilinear", "nearest"]},
           {"_target_": "RandRotated", "keys": KEYS, "prob": 0.2, "range_x": 0.4, "mode": ["bilinear", "nearest"]},
           {"_target_": "RandRotated", "keys": KEYS, "prob": 0.2, "range_y": 0.4, "mode": ["bilinear", "nearest"]},
           {"_target_": "RandZoomd", "keys": KEYS, "prob": 0.2, "min_zoom": 0.8, "max_zoom": 1.3,
            "mode": ["area", "nearest"]}]
INTENSITY = [{"_target_": "RandAdjustContrastd", "keys": "image", "prob": 0.2, "gamma": [0.5, 4.5]},
             {"_target_": "RandHistogramShiftd", "keys": "image", "prob": 0.2, "num_control_points": 10},
             {"_target_": "RandBiasFieldd", "keys": "image", "prob": 0.2},
             {"_target_": "RandGibbsNoised", "keys": "image", "prob": 0.2, "alpha": [0.0, 1.0]},
             {"_target_": "RandKSpaceSpikeNoised", "keys": "image", "prob": 0.2}]


def test_plans_map_onto_the_device_pipeline(options):
    """the reference's default_preprocessing / default_augmentation (monai_unet.py:151-219) spelled out as
    bundle dictionaries map onto the device pipeline"""
    plan = plan_preprocessing(_comp(DEFAULT_PRE))
    assert plan == {"orientation": True, "normalize": True, "crop_foreground": True, "spacing": [1.0, 1.0, 2.5],
                    "spacing_label_nearest": False}
    assert plan_preprocessing({}) is None and plan_preprocessing(None) is None
    # a per-key Spacingd mode is honoured: nearest for the label
    pre = DEFAULT_PRE[:-1] + [dict(DEFAULT_PRE[-1], mode=["bilinear", "nearest"])]
    assert plan_preprocessing(_comp(pre))["spacing_label_nearest"] is True
    # the reference's fixture asks for nonzero=True and a random rotation inside pre-processing: refused by name
    ref = ConfigParser({"image_key": "image", "preprocessing": options["preprocessing"]})
    with pytest.raises(ValueError, match="RandRotated"):
        plan_preprocessing(ref.get_parsed_content("preprocessing"))
    sp = {"_target_": "SpatialPadd", "keys": KEYS, "spatial_size": [96, 96, 96]}
    plan = plan_augmentation(_comp([sp, CROP, {"_target_": "RandRotated", "keys": KEYS, "prob": 0.2, "_disabled_": True}]
                                   + FLIPS))
    assert plan["num_samples"] == 2 and plan["flip_prob"] == 0.2 and plan["flip_axes"] == [0, 1, 2]
    assert not plan["augment_spatial"] and not plan["augment_intensity"] and plan["num_classes"] == 3
    full = plan_augmentation(_comp(SPATIAL + [sp, CROP] + INTENSITY + FLIPS))
    assert full["augment_spatial"] and full["augment_intensity"] and full["spatial_size"] == [96, 96, 96]
    with pytest.raises(ValueError, match="RandGaussianNoised"):
        plan_augmentation(_comp([CROP, {"_target_": "RandGaussianNoised", "keys": "image"}]))


@pytest.mark.parametrize("bad,match", [
    (DEFAULT_PRE[:-1] + [dict(DEFAULT_PRE[-1], padding_mode="zeros")], "padding_mode"),
    (DEFAULT_PRE[:-1] + [dict(DEFAULT_PRE[-1], diagonal=True)], "diagonal"),
    (DEFAULT_PRE[:-1] + [dict(DEFAULT_PRE[-1], mode="nearest")], "bilinear"),
    (DEFAULT_PRE[:2] + [dict(DEFAULT_PRE[2], nonzero=True)] + DEFAULT_PRE[3:], "nonzero"),
    (DEFAULT_PRE[:2] + [dict(DEFAULT_PRE[2], keys=["image", "label"])] + DEFAULT_PRE[3:], "keys"),
    (DEFAULT_PRE[:3] + [dict(DEFAULT_PRE[3], margin=4)] + DEFAULT_PRE[4:], "margin"),
    (DEFAULT_PRE[:1] + [dict(DEFAULT_PRE[1], axcodes="LPS")] + DEFAULT_PRE[2:], "axcodes"),
    (DEFAULT_PRE[:1] + [dict(DEFAULT_PRE[1], frobnicate=1)] + DEFAULT_PRE[2:], "frobnicate"),
])
def test_preprocessing_arguments_the_device_does_not_implement_are_refused(bad, match):
    """a configured value must never be dropped silently"""
    with pytest.raises(ValueError, match=match):
        plan_preprocessing(_comp(bad))


@pytest.mark.parametrize("bad,match", [
    ([CROP, FLIPS[0]], "once per spatial axis"),                                   # one axis only
    ([CROP, FLIPS[0], FLIPS[1], dict(FLIPS[2], prob=0.5)], "one probability"),
    ([CROP, {"_target_": "RandFlipd", "keys": KEYS, "prob": 0.2}], "spatial_axis"),   # all axes at once
    ([dict(CROP, ratios=[1, 1, 1])], "ratios"),
    ([{k: v for k, v in CROP.items() if k != "ratios"}], "ratios"),
    ([dict(CROP, num_classes=4)], "num_classes"),
    ([CROP, INTENSITY[0]], "as a whole"),                                          # partial intensity block
    ([CROP] + INTENSITY[:4] + [dict(INTENSITY[4], prob=0.5)], "prob"),
    ([CROP, dict(INTENSITY[0], gamma=[0.7, 1.5])] + INTENSITY[1:], "gamma"),
    ([CROP] + SPATIAL[:3], "as a whole"),                                          # rotations without the zoom
    ([CROP, dict(SPATIAL[0], range_z=0.8)] + SPATIAL[1:], "range_z"),
    ([CROP] + SPATIAL[:3] + [dict(SPATIAL[3], max_zoom=2.0)], "max_zoom"),
    ([CROP, {"_target_": "SpatialPadd", "keys": KEYS, "spatial_size": [96] * 3, "mode": "reflect"}], "mode"),
])
def test_augmentation_arguments_the_device_does_not_implement_are_refused(bad, match):
    with pytest.raises(ValueError, match=match):
        plan_augmentation(_comp(bad))
